#!/usr/bin/env python3
"""PCIe-inclusive rates of the boundary as the JNI shim uses it (TSX_MEM_HOST / TSX_MEM_HOST_PACKED: caller's host buffers in,
host buffers out), forward and inverse, with pageable and with registered (tsx_host_register) buffers, staged pipeline on / off.
Reported in DESIGN.md and in bench.py's `end_to_end` object - never as `value`.   usage: hostmem_bench.py [chunks] [--json]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime)
import tsxform
from tsxform import synth
nat = tsxform._native
PCIE_GBS = 64.0       # PCIe 5.0 x16, one direction


def measure(N, n=256, reps=3, modes=("gcm+crc", "zstd+gcm+crc"), verbose=True):
    CH = synth.CHUNK
    if os.path.exists("/tmp/k256.npy"):
        seg = np.load("/tmp/k256.npy")
    else:                                              # generated on the device (seconds), not with numpy (minutes)
        dev = torch.device("cuda", 0)
        seg = torch.cat([synth.gen_chunk("K", 1000, 0, i, CH, device=dev) for i in range(256)]).cpu().numpy()
        np.save("/tmp/k256.npy", seg)
    src = np.concatenate([seg] * ((n + 255) // 256))[:n * CH].copy()
    rows = []
    for name in modes:
        flags = {"gcm+crc": nat.ENCRYPT | nat.CRC, "zstd+gcm+crc": nat.COMPRESS | nat.ENCRYPT | nat.CRC}[name]
        slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
        dst = np.zeros(n * slot, np.uint8)
        back = np.zeros(n * CH, np.uint8)
        params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
        ctx = N.ctx_create(0, n, CH)
        for pinned in (False, True):
            if pinned:
                for a in (src, dst, back):
                    N.host_register(a)
            for pipe in ((True, False) if name == "gcm+crc" else (True,)):
                N.debug_config("no_pipeline", 0 if pipe else 1)
                d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
                d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
                for i in range(n):
                    d["iv"][i] = np.frombuffer(synth.iv_for(0, i), np.uint8)
                best = 1e9
                for _ in range(reps):
                    t0 = time.perf_counter()
                    N.transform_batch(params, d, src, dst, dst.size, nat.MEM_HOST, ctx=ctx)
                    best = min(best, time.perf_counter() - t0)
                t = N.ctx_timing(ctx)
                assert (d["status"] == 0).all()
                e = np.zeros(n, nat.DESC_DTYPE); e["src_off"] = d["dst_off"]; e["src_len"] = d["dst_len"]
                e["dst_off"] = np.arange(n, dtype=np.uint64) * CH; e["dst_cap"] = CH
                ibest = 1e9
                for _ in range(reps):
                    t0 = time.perf_counter()
                    N.detransform_batch(params, e, dst, back, back.size, nat.MEM_HOST, ctx=ctx)
                    ibest = min(ibest, time.perf_counter() - t0)
                ti = N.ctx_timing(ctx)
                ok = bool((e["status"] == 0).all() and np.array_equal(back, src))
                gib = n * CH / 2**30
                moved = (n * CH + int(d["dst_len"].sum())) / 1e9          # bytes over PCIe, both directions
                row = {"chain": name, "chunks": n, "host_memory": "registered" if pinned else "pageable", "pipelined": pipe,
                       "transform_ms": round(best * 1e3, 2), "transform_gibs": round(gib / best, 3),
                       "transform_pcie_frac": round(moved / best / (2 * PCIE_GBS), 3),
                       "detransform_ms": round(ibest * 1e3, 2), "detransform_gibs": round(gib / ibest, 3),
                       "kernels_ms": round(t.crc_ms + t.zstd_ms + t.gcm_ms, 2), "h2d_span_ms": round(t.h2d_ms, 2), "d2h_span_ms": round(t.d2h_ms, 2),
                       "inverse_kernels_ms": round(ti.crc_ms + ti.unzstd_ms + ti.gcm_ms, 2), "round_trip_exact": ok}
                rows.append(row)
                if verbose:
                    print("%-13s %4d chunks %-10s %-9s  host->host %8.1f ms = %6.2f GiB/s | inverse %8.1f ms = %6.2f GiB/s | kernels %.1f / %.1f ms  exact=%s"
                          % (name, n, row["host_memory"], "pipelined" if pipe else "one-shot", best * 1e3, gib / best, ibest * 1e3, gib / ibest,
                             row["kernels_ms"], row["inverse_kernels_ms"], ok), flush=True)
            if pinned:
                for a in (src, dst, back):
                    N.host_unregister(a)
        N.debug_config("no_pipeline", 0)
        N.ctx_destroy(ctx)
    return rows


if __name__ == "__main__":
    N = nat.Native(nat.LIB_PATH); N.init(1, [0])
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = measure(N, int(args[0]) if args else 256)
    if "--json" in sys.argv:
        print(json.dumps(rows))
