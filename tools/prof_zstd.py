#!/usr/bin/env python3
"""Phase profile of zstd_compress_kernel (libtsxform_prof.so, `make -C csrc prof`): s_memtime lap timers per chunk.
Usage (GPU box): python tools/prof_zstd.py [--chunks 2048] [--dist K] -> JSON with mean cycles per bucket."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "setup+table zero", 1: "split_block", 2: "match: search steps", 3: "match: extend+store", 4: "match_block (parse of the block, whole)",
         5: "lit: histogram/sample", 6: "lit: lane0 huffman build", 7: "lit: huffman encode", 8: "seq: codes+hist", 9: "seq: lane0 FSE tables",
         10: "seq: lane0 encode", 11: "emit/copy + misc", 12: "#search steps", 13: "#sequences", 14: "total cycles", 15: "sum K (positions evaluated)",
         16: "CRC32C head", 17: "GCM tail / copy to the slot", 18: "gather literals", 19: "#steps with hash collision (slow path)", 20: "#blocks", 21: "#extension passes from global"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=2048)
    ap.add_argument("--dist", default="K")
    ap.add_argument("--out", default="")
    ap.add_argument("--data", default="", help=".npy cache of the 256 distinct chunks (made on first use): keeps generator kernels out of rocprofv3 runs")
    ap.add_argument("--chain", action="store_true", help="full chain (CRC head + compress + GCM tail in the compressor wave) instead of compress only")
    ap.add_argument("--uniq", type=int, default=256, help="distinct chunks (replicated to --chunks); content B is generated on the host, ~5 s per chunk")
    ap.add_argument("--profile", default="1_5_7", choices=["1_5_7", "1_5_6"], help="Zstd profile (the 1.5.7 pre-splitter on / off)")
    ap.add_argument("--config", default="", help="key=value,... for tsx_debug_config before the batches (e.g. fetch_quiet_ms=2000)")
    ap.add_argument("--where", action="store_true", help="per-chunk wall time by where the chunk ran: guest waves (reserved CUs) against the others")
    ap.add_argument("--lib", default="libtsxform_prof.so", help="libtsxform_prof.so (lap timers) or libtsxform.so (plain, for rocprofv3 runs)")
    args = ap.parse_args()
    import torch
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    N = nat.Native(os.path.join(os.path.dirname(nat.LIB_PATH), args.lib) if args.lib == "libtsxform.so" else os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", args.lib))
    has_prof = "_prof" in args.lib
    for kv in (args.config or "").split(","):
        if kv:
            N.debug_config(kv.split("=")[0], int(kv.split("=")[1]))
    N.init(1, [0])
    n, CH = args.chunks, synth.CHUNK
    dev = torch.device("cuda", 0)
    src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
    uniq = min(n, args.uniq)                                   # one distinct segment, replicated (profiling only)
    if args.data and os.path.exists(args.data):
        src[:uniq * CH] = torch.from_numpy(np.load(args.data)[:uniq * CH]).to(dev)
    else:
        for i in range(uniq):
            src[i * CH:(i + 1) * CH] = synth.gen_chunk(args.dist, 1000, 0, i, CH, device=dev)
        if args.data:
            np.save(args.data, src[:uniq * CH].cpu().numpy())
    for i in range(uniq, n, uniq):
        m = min(uniq, n - i)
        src[i * CH:(i + m) * CH] = src[:m * CH]
    flags = (nat.COMPRESS | nat.ENCRYPT | nat.CRC) if args.chain else nat.COMPRESS
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    dst = torch.empty(n * slot, dtype=torch.uint8, device=dev)
    prof = torch.zeros(n * 24, dtype=torch.int64, device=dev)
    if has_prof:
        N.lib.tsx_debug_set_prof.restype = None
        N.lib.tsx_debug_set_prof.argtypes = [C.c_void_p]
        N.lib.tsx_debug_set_prof(prof.data_ptr())
    d = np.zeros(n, nat.DESC_DTYPE)
    d["src_off"] = np.arange(n, dtype=np.uint64) * CH
    d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot
    d["dst_cap"] = slot
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=nat.ZSTD_PROFILE_1_5_7 if args.profile == "1_5_7" else nat.ZSTD_PROFILE_1_5_6)
    ctx = N.ctx_create(0, n, CH)
    res = {}
    for it in range(2):
        t0 = time.perf_counter()
        N.transform_batch(params, d, src.data_ptr(), dst.data_ptr(), dst.numel(), nat.MEM_DEVICE, ctx=ctx)
        torch.cuda.synchronize()
        res["wall_ms_%d" % it] = (time.perf_counter() - t0) * 1e3
        res["zstd_ms_%d" % it] = N.ctx_timing(ctx).zstd_ms
    p = prof.cpu().numpy().reshape(n, 24)
    if args.where:
        # the LAST batch: wall time of every chunk (100 MHz clock), begin relative to the batch's first begin, by kind of wave
        t0 = p[:, 2].min()
        dur = (p[:, 3] - p[:, 2]) / 1e5; beg = (p[:, 2] - t0) / 1e5; guest = (p[:, 19] >> 16) & 1; key = p[:, 19] & 0xFFF
        def q(a):
            return None if a.size == 0 else [round(float(x), 1) for x in np.percentile(a, [0, 50, 90, 99, 100])]
        res["where"] = {"guest_chunks": int(guest.sum()), "other_chunks": int((1 - guest).sum()),
                        "chunk_ms_p0_50_90_99_100": {"guests": q(dur[guest == 1]), "others": q(dur[guest == 0])},
                        "begin_ms_p0_50_90_99_100": {"guests": q(beg[guest == 1]), "others": q(beg[guest == 0])},
                        "wave_start_ms_p0_50_90_99_100": q((p[:, 21] - t0) / 1e5), "ticket_taken_ms_p0_50_90_99_100": q((p[:, 22] - t0) / 1e5),
                        "ticket_to_begin_us_p0_50_90_99_100": q((p[:, 2] - p[:, 22]) / 1e2), "distinct_waves": int(np.unique(p[:, 21]).size),
                        "batch_ms_first_begin_to_last_end": round(float((p[:, 3].max() - t0) / 1e5), 1),
                        "slowest_20": sorted([(round(float(dur[i]), 1), round(float(beg[i]), 1), int(guest[i]), int(key[i])) for i in np.argsort(-dur)[:20]], reverse=True),
                        "chunks_that_ended_on_another_cu": int(((p[:, 19] >> 20) & 0xFFF != key).sum()), "chunks_per_cu_max_where_taken": int(np.bincount(((p[:, 19] >> 20) & 0xFFF).astype(np.int64)).max()),
                        "chunks_per_cu_max": int(np.bincount(key.astype(np.int64)).max()), "cus_used": int((np.bincount(key.astype(np.int64)) > 0).sum())}
    mean = p.mean(axis=0)
    res["chunks"] = n
    res["mean_out"] = float(d["dst_len"].mean())
    tot = max(mean[14], 1)
    res["buckets"] = {NAMES[k]: {"mean": float(mean[k]), "frac_of_total": float(mean[k] / tot) if (k < 12 or 15 < k < 19) else None} for k in NAMES}
    res["total_min_max"] = [int(p[:, 14].min()), int(p[:, 14].max())]
    res["cycles_per_seq"] = float(tot / max(mean[13], 1))
    s = json.dumps(res, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s)


if __name__ == "__main__":
    main()
