#!/usr/bin/env python3
"""One 2048-chunk batch at a time (device resident, explicit context, torch-free): how long does a batch take, and what do the service's
waves do meanwhile?  (Round 6: with guest waves on the reserved CUs a lone batch took 6 s instead of 1.1 s.)  For every configuration:
`--batches` batches with `--pause-ms` between them, wall time of each, the service's counters around each, and device-side progress
(chunks done, live waves) sampled every 50 ms through the pinned mirrors.  One JSON line per configuration.
  python tools/broker_leg.py --gen /dev/shm/s.npy /dev/shm/i.npy 1 256 4194304 K     (once)
  python tools/lone_batch_probe.py --src /dev/shm/s.npy --ivs /dev/shm/i.npy --configs fetch_quiet_ms=0 fetch_quiet_ms=2000"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", required=True); ap.add_argument("--ivs", required=True)
    ap.add_argument("--configs", nargs="+", default=["fetch_quiet_ms=0", "fetch_quiet_ms=2000"])
    ap.add_argument("--batches", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=2048)
    ap.add_argument("--pause-ms", type=float, default=20.0)
    ap.add_argument("--prof-lib", default="", help="tools/_libs/libtsxform_prof.so: per-chunk begin / end / CU of every batch (guest waves against the others)")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--pre-config", default="", help="key=value,... set BEFORE tsx_init (e.g. svc_normal_priority=1)")
    ap.add_argument("--bind", action="store_true", help="bind the process to the GPU's NUMA node first (as the other probes do)")
    a = ap.parse_args()
    assert "torch" not in sys.modules
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    N = nat.Native(a.prof_lib) if a.prof_lib else nat.Native()
    for kv in (a.pre_config or "").split(","):
        if kv:
            N.debug_config(kv.split("=")[0], int(kv.split("=")[1]))
    N.init(1, [0])
    aff = None
    if a.bind:
        from numa_bind import bind_to_gpu_numa_node
        aff = bind_to_gpu_numa_node(0)
    CH, B, n = 4 << 20, 256, a.chunks
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    hsrc = np.load(a.src)[:B * CH]; ivs = np.load(a.ivs)[:B]
    dsrc = N.device_malloc(n * CH)
    for k in range(n // B):
        N.h2d(dsrc + k * B * CH, hsrc)
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    ddst = N.device_malloc(n * slot)
    d0 = np.zeros(n, nat.DESC_DTYPE); d0["src_off"] = np.arange(n, dtype=np.uint64) * CH; d0["src_len"] = CH
    d0["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d0["dst_cap"] = slot; d0["iv"] = np.tile(ivs, (n // B, 1))
    ctx = N.ctx_create(0, n, CH)
    dprof = None
    if a.prof_lib:
        import ctypes as C
        dprof = N.device_malloc(n * 24 * 8)
        N.lib.tsx_debug_set_prof.restype = None; N.lib.tsx_debug_set_prof.argtypes = [C.c_void_p]
        N.lib.tsx_debug_set_prof(dprof)
    for cfg in a.configs:
        for kv in cfg.split(","):
            N.debug_config(kv.split("=")[0], int(kv.split("=")[1]))
        time.sleep(2.5)                                                  # whatever ran before: quiet again
        rows = []
        for b in range(a.batches):
            d = d0.copy()
            s0 = N.service_stats(0)
            samples = []; going = [True]

            def sampler():
                t0 = time.perf_counter()
                while going[0]:
                    s = N.service_stats(0)
                    samples.append((round((time.perf_counter() - t0) * 1e3), int(s["device_chunks"]) - int(s0["device_chunks"]), int(s["live_waves"]), int(s["running"])))
                    time.sleep(0.05)
            th = threading.Thread(target=sampler if not a.no_sampler else (lambda: None)); th.start()
            t0 = time.perf_counter()
            N.transform_batch(params, d, dsrc, ddst, n * slot, nat.MEM_DEVICE, ctx=ctx, src_size=n * CH)
            ms = (time.perf_counter() - t0) * 1e3
            going[0] = False; th.join()
            s1 = N.service_stats(0)
            assert (d["status"] == 0).all()
            where = None
            if dprof:
                hp = np.zeros(n * 24, np.uint64); N.d2h(hp, dprof)
                p = hp.reshape(n, 24).astype(np.int64)
                t0_ = p[:, 2].min(); dur = (p[:, 3] - p[:, 2]) / 1e5; beg = (p[:, 2] - t0_) / 1e5; guest = (p[:, 19] >> 16) & 1

                def q(x):
                    return None if x.size == 0 else [round(float(v), 1) for v in np.percentile(x, [0, 50, 90, 99, 100])]
                where = {"guest_chunks": int(guest.sum()), "chunk_ms_p0_50_90_99_100": {"guests": q(dur[guest == 1]), "others": q(dur[guest == 0])},
                         "begin_ms_p0_50_90_99_100": {"guests": q(beg[guest == 1]), "others": q(beg[guest == 0])},
                         "cycles_M_p50": {"guests": None if guest.sum() == 0 else round(float(np.median(p[guest == 1, 14])) / 1e6), "others": round(float(np.median(p[guest == 0, 14])) / 1e6)}}
            rows.append({"ms": round(ms, 1), "where": where, "launches": int(s1["launches"] - s0["launches"]), "guest_launches": int(s1["guest_launches"] - s0["guest_launches"]),
                         "wave_starts": int(s1["wave_starts"] - s0["wave_starts"]), "reserved_exits": int(s1["reserved_exits"] - s0["reserved_exits"]),
                         "relocated": int(s1["relocated_waves"] - s0["relocated_waves"]),
                         "progress_ms_chunks_live_running": samples[::max(1, len(samples) // 12)]})
            time.sleep(a.pause_ms / 1e3)
        print(json.dumps({"config": cfg, "pre_config": a.pre_config, "cpu_affinity": aff, "sampler": not a.no_sampler, "chunks": n, "pause_ms": a.pause_ms, "waves": int(N.service_stats(0)["waves"]), "batches": rows}), flush=True)


if __name__ == "__main__":
    main()
