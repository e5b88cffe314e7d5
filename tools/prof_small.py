#!/usr/bin/env python3
"""BASELINE configs[1] / configs[2] as rocprofv3 targets: one 1 GiB segment (256 x 4 MiB, device resident), CRC32C only or AES-256-GCM +
CRC32C, a few batches (tools/pmc_small.sh collects FETCH_SIZE / WRITE_SIZE per launch of crc32c_partial_kernel / gcm_ctr_ghash_kernel).
  python tools/prof_small.py crc|gcm_crc [--data /tmp/k256.npy] [--reps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["crc", "gcm_crc"])
    ap.add_argument("--data", default="")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    N = nat.Native(); N.init(1, [0])
    n, CH = 256, synth.CHUNK
    dev = torch.device("cuda", 0)
    src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
    if args.data and os.path.exists(args.data):
        src[:] = torch.from_numpy(np.load(args.data)[:n * CH]).to(dev)
    else:
        for i in range(n):
            src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
        if args.data:
            np.save(args.data, src.cpu().numpy())
    flags = nat.CRC if args.workload == "crc" else nat.ENCRYPT | nat.CRC
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    dst = torch.empty(n * slot if flags & nat.ENCRYPT else 64, dtype=torch.uint8, device=dev)
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for i in range(n):
        d["iv"][i] = np.frombuffer(synth.iv_for(0, i), np.uint8)
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    ctx = N.ctx_create(0, n, CH)
    out = {"workload": args.workload, "chunks": n}
    for it in range(args.reps):
        t0 = time.perf_counter()
        if args.workload == "crc":
            N.crc32c_batch(d, src.data_ptr(), nat.MEM_DEVICE, ctx=ctx)
        else:
            N.transform_batch(params, d, src.data_ptr(), dst.data_ptr(), dst.numel(), nat.MEM_DEVICE, ctx=ctx)
        torch.cuda.synchronize()
        tm = N.ctx_timing(ctx)
        out["wall_ms_%d" % it] = round((time.perf_counter() - t0) * 1e3, 3); out["kernel_ms_%d" % it] = round(tm.crc_ms if args.workload == "crc" else tm.gcm_ms, 4)
    assert (d["status"] == 0).all()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
