#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary as the JNI shim uses it (TSX_MEM_HOST: caller's host buffers in, host buffers out).
Reported in DESIGN.md next to the device-resident `value` of bench.py - never as `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime)
import tsxform
from tsxform import synth
nat = tsxform._native
N = nat.Native(nat.LIB_PATH); N.init(1, [0])
n, CH = int(sys.argv[1]) if len(sys.argv) > 1 else 256, synth.CHUNK
if os.path.exists("/tmp/k256.npy"):
    seg = np.load("/tmp/k256.npy")
else:
    seg = np.concatenate([synth.gen_chunk("K", 1000, 0, i, CH) for i in range(256)])
src = np.concatenate([seg] * ((n + 255) // 256))[:n * CH]
for flags, name in [(nat.ENCRYPT | nat.CRC, "gcm+crc"), (nat.COMPRESS | nat.ENCRYPT | nat.CRC, "zstd+gcm+crc")]:
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    dst = np.zeros(n * slot, np.uint8)
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    ctx = N.ctx_create(0, n, CH)
    for it in range(3):
        t0 = time.perf_counter()
        N.transform_batch(params, d, src, dst, dst.size, nat.MEM_HOST, ctx=ctx)
        el = time.perf_counter() - t0
    t = N.ctx_timing(ctx)
    print("%-14s %4d chunks host->host: %8.1f ms = %6.2f GiB/s of original bytes   (h2d %.1f ms, kernels %.1f ms, d2h %.1f ms; pageable host memory)"
          % (name, n, el * 1e3, n * CH / 2**30 / el, t.h2d_ms, t.crc_ms + t.zstd_ms + t.gcm_ms, t.d2h_ms))
    N.ctx_destroy(ctx)
