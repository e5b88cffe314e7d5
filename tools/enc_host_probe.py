#!/usr/bin/env python3
"""Encrypt-only (AES-256-GCM + CRC32C, no compression) host -> host: where do the milliseconds of a 1 GiB segment go?  (the
driver line says 25 GiB/s, INTEGRATION.md promised 38 from round 2.)  One 256 x 4 MiB batch between registered host buffers, an explicit
context, best of N; variants: the GCM waves write into the caller's buffer (zero-copy output, the default) or the output travels through
copy engines; pieces of 16 / 32 / 64 (default) / 128 / 256 MiB; the whole batch in one piece.  Run it without torch (the system's HIP runtime,
what a JVM gets) and with --with-torch (torch's bundled runtime: bench.py's situation).  One JSON line per variant.
  python tools/broker_leg.py --gen /dev/shm/s.npy /dev/shm/i.npy 1 256 4194304 K     (once)
  python tools/enc_host_probe.py --src /dev/shm/s.npy --ivs /dev/shm/i.npy [--with-torch]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", required=True); ap.add_argument("--ivs", required=True)
    ap.add_argument("--with-torch", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    if a.with_torch:
        import torch  # noqa: F401
    else:
        assert "torch" not in sys.modules
    import tsxform
    from tsxform import synth
    from numa_bind import bind_to_gpu_numa_node
    nat = tsxform._native
    N = nat.Native(); N.init(1, [0])
    aff = bind_to_gpu_numa_node(0)
    CH, n = 4 << 20, 256
    flags = nat.ENCRYPT | nat.CRC
    src = np.load(a.src)[:n * CH]; ivs = np.load(a.ivs)[:n]
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    dst = np.zeros(n * slot, np.uint8); back = np.zeros(n * CH, np.uint8)
    for b in (src, dst, back):
        N.host_register(b)
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    ctx = N.ctx_create(0, n, CH)
    d0 = np.zeros(n, nat.DESC_DTYPE); d0["src_off"] = np.arange(n, dtype=np.uint64) * CH; d0["src_len"] = CH
    d0["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d0["dst_cap"] = slot; d0["iv"] = ivs
    ref = None
    for zc in (1, 0):
        for sub_mib in (0, 16, 32, 128, 256, -1):
            N.debug_config("no_zero_copy_out", 0 if zc else 1)
            N.debug_config("no_pipeline", 1 if sub_mib < 0 else 0)
            N.debug_config("sub_bytes", max(sub_mib, 0) << 20)
            ts = []; tm = None
            for _ in range(a.reps):
                d = d0.copy()
                t0 = time.perf_counter()
                N.transform_batch(params, d, src, dst, dst.size, nat.MEM_HOST, ctx=ctx)
                ts.append(time.perf_counter() - t0)
                tm = N.ctx_timing(ctx)
            took_zc = bool(N.lib.tsx_debug_last_zero_copy(ctx))          # (the inverse call below resets the context's flag)
            assert (d["status"] == 0).all()
            dig = int(np.bitwise_xor.reduce(d["crc32c"])) ^ int(dst[::4099].astype(np.uint64).sum() & 0xFFFFFFFF)
            ref = dig if ref is None else ref
            e = np.zeros(n, nat.DESC_DTYPE); e["src_off"] = d["dst_off"]; e["src_len"] = d["dst_len"]; e["dst_off"] = np.arange(n, dtype=np.uint64) * CH; e["dst_cap"] = CH
            ti = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                N.detransform_batch(params, e, dst, back, back.size, nat.MEM_HOST, ctx=ctx)
                ti.append(time.perf_counter() - t0)
            print(json.dumps({"tag": a.tag, "runtime": "torch's bundled HIP" if a.with_torch else "system HIP (no torch)", "zero_copy_output": bool(zc),
                              "piece": "one piece" if sub_mib < 0 else "%d MiB" % (sub_mib or 64), "zero_copy_taken": took_zc,
                              "ms_best": round(min(ts) * 1e3, 2), "ms_median": round(float(np.median(ts)) * 1e3, 2), "gibs_best": round(1.0 / min(ts), 2),
                              "h2d_span_ms": round(tm.h2d_ms, 2), "d2h_span_ms": round(tm.d2h_ms, 2), "gcm_ms": round(tm.gcm_ms, 2), "crc_ms": round(tm.crc_ms, 2),
                              "inverse_ms_best": round(min(ti) * 1e3, 2), "inverse_gibs": round(1.0 / min(ti), 2), "same_bytes": dig == ref,
                              "round_trip_exact": bool((e["status"] == 0).all() and np.array_equal(back, src)), "cpu_affinity": aff}), flush=True)
    N.debug_config("no_zero_copy_out", 0); N.debug_config("no_pipeline", 0); N.debug_config("sub_bytes", 0)


if __name__ == "__main__":
    main()
