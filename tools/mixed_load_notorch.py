#!/usr/bin/env python3
"""tools/mixed_load_probe.py for a process WITHOUT torch: the system's HIP runtime, as a broker's JVM loads it through libtsxform.so (a
process has one HIP runtime, the first one loaded; every python process that imports torch runs torch's bundled one).  Upload load in the
broker's shape (--callers context-less calls of ONE 256-chunk segment each, registered host buffers, slot layout) or as device-resident
2048-chunk batches on explicit contexts (--shape batches: buffers from tsx_device_malloc); meanwhile this thread restores 1 and 4 chunks
host -> host through a context of its own.  One JSON line.
  python tools/broker_leg.py --gen /dev/shm/s.npy /dev/shm/i.npy 1 256 4194304 K     (once, with torch, in a process of its own)
  python tools/mixed_load_notorch.py --src /dev/shm/s.npy --ivs /dev/shm/i.npy [--shape broker|batches] [--callers 32] [--seconds 12] [--reserved-cus n] [--no-fetch]"""
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
GiB = float(1 << 30)


def run(args):
    assert "torch" not in sys.modules
    import tsxform
    from tsxform import synth
    from numa_bind import bind_to_gpu_numa_node
    nat = tsxform._native
    N = nat.Native()
    for kv in (args.config or "").split(","):
        if kv:
            N.debug_config(kv.split("=")[0], int(kv.split("=")[1]))
    N.init(1, [0], fetch_reserved_cus=None if args.reserved_cus < 0 else args.reserved_cus, service_max_launch_ms=None if args.max_launch_ms < 0 else args.max_launch_ms)
    affinity = bind_to_gpu_numa_node(0)
    CH, B = args.chunk, 256
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    hsrc = np.load(args.src)[:B * CH]; ivs = np.load(args.ivs)[:B]
    N.host_register(hsrc)
    slot = (N.transformed_bound(CH, flags) + 15) // 16 * 16 + 16
    T = args.callers
    broker = args.shape == "broker"
    n = B if broker else 2048
    d = np.zeros(B, nat.DESC_DTYPE); d["src_off"] = np.arange(B, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(B, dtype=np.uint64) * slot; d["dst_cap"] = N.transformed_bound(CH, flags); d["iv"] = ivs
    # the chunks the fetches restore: one transform of the first 4 into a registered host buffer
    hfr = np.zeros(4 * slot, np.uint8); hbk = np.zeros(4 * CH, np.uint8); N.host_register(hfr); N.host_register(hbk)
    d4 = d[:4].copy()
    N.transform_batch(params, d4, hsrc, hfr, hfr.size, nat.MEM_HOST, ctx=None)
    assert (d4["status"] == 0).all()
    if broker:
        hdsts = []
        for t in range(T):
            hb = np.zeros(B * slot, np.uint8); N.host_register(hb); hdsts.append(hb)
        ds = [d.copy() for _ in range(T)]
    else:
        dsrc = N.device_malloc(n * CH)
        for k in range(n // B):
            N.h2d(dsrc + k * B * CH, hsrc)
        dslot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
        dd = np.zeros(n, nat.DESC_DTYPE); dd["src_off"] = np.arange(n, dtype=np.uint64) * CH; dd["src_len"] = CH
        dd["dst_off"] = np.arange(n, dtype=np.uint64) * dslot; dd["dst_cap"] = dslot; dd["iv"] = np.tile(ivs, (n // B, 1))
        ctxs = [N.ctx_create(0, n, CH) for _ in range(T)]
        ddsts = [N.device_malloc(n * dslot) for _ in range(T)]
        ds = [dd.copy() for _ in range(T)]
    fctx = N.ctx_create(0, 4, CH)

    def fetch(k):
        e = np.zeros(k, nat.DESC_DTYPE); e["src_off"] = d4["dst_off"][:k]; e["src_len"] = d4["dst_len"][:k]; e["iv"] = d4["iv"][:k]
        e["dst_off"] = np.arange(k, dtype=np.uint64) * CH; e["dst_cap"] = CH
        t0 = time.perf_counter()
        N.detransform_batch(params, e, hfr, hbk, hbk.size, nat.MEM_HOST, ctx=fctx)
        dt = time.perf_counter() - t0
        assert (e["status"] == 0).all()
        return dt * 1e3

    for k in (1, 4):
        fetch(k)
    idle = {k: round(float(np.median([fetch(k) for _ in range(7)])), 3) for k in (1, 4)}
    assert np.array_equal(hbk, hsrc[:4 * CH])
    stop = [False]; stamps = []; lock = threading.Lock(); done = [0] * T

    def worker(t):
        while not stop[0]:
            if broker:
                N.transform_batch(params, ds[t], hsrc, hdsts[t], hdsts[t].size, nat.MEM_HOST, ctx=None)
            else:
                N.transform_batch(params, ds[t], dsrc, ddsts[t], n * dslot, nat.MEM_DEVICE, ctx=ctxs[t], src_size=n * CH)
            done[t] += 1
            with lock:
                stamps.append(time.perf_counter())

    if args.phases:
        # what makes a fetch wait for the END of a service launch?  A: the first fetch after uploads began; B: fetches every 50 ms; C: one after a
        # pause of the fetch side; D: uploads stop, the service kernel goes, fetches go on, uploads start again - the first fetch 2 s later
        rows = []
        def note(phase, v):
            rows.append({"phase": phase, "at_s": round(time.perf_counter() - t0, 2), "ms": round(v, 2), "service_launches": N.service_stats(0)["launches"]}); print(json.dumps(rows[-1]), flush=True)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        sampling = [True]

        def sampler():
            while sampling[0]:
                st_ = N.service_stats(0)
                print(json.dumps({"sample_at_s": round(time.perf_counter() - t0, 2), **{k: st_[k] for k in ("launches", "watchdog_launches", "running", "device_chunks", "wave_starts", "reserved_exits", "live_waves", "live_waves_max")}}), flush=True)
                time.sleep(1.0)

        sth = threading.Thread(target=sampler); sth.start()
        [x.start() for x in th]
        time.sleep(2.0)
        note("A first fetch after uploads began", fetch(1))
        tb = time.perf_counter()
        vals = []
        while time.perf_counter() - tb < 3.0:
            vals.append(fetch(1)); time.sleep(0.05)
        note("B every 50 ms: max of %d" % len(vals), max(vals))
        time.sleep(4.0)
        note("C after a 4 s pause of the fetch side", fetch(1))
        note("C again", fetch(1))
        stop[0] = True
        [x.join() for x in th]
        sampling[0] = False; sth.join()
        if args.phases_short:
            return rows
        time.sleep(1.0)
        for _ in range(3):
            note("D uploads stopped, fetch side stays warm", fetch(1)); time.sleep(0.3)
        stop[0] = False
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        tb = time.perf_counter()
        vals = []
        while time.perf_counter() - tb < 2.0:
            vals.append(fetch(1)); time.sleep(0.05)
        note("D uploads began again, fetches every 50 ms throughout: max of %d" % len(vals), max(vals))
        stop[0] = True
        [x.join() for x in th]
        return rows
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    quiet_ms = [int(kv.split("=")[1]) for kv in (args.config or "").split(",") if kv.startswith("fetch_quiet_ms=")]
    if quiet_ms and quiet_ms[0] > 0:
        time.sleep(quiet_ms[0] / 1e3 + 0.2)                              # the warm-up fetches are forgotten: the uploads begin with guest waves on the reserved CUs
    sv0 = N.service_stats(0)
    t0 = time.perf_counter()
    samples = []; sampling = [args.sample_service]

    def sampler():                                                       # where does a window lose its seconds?  device-side progress every 100 ms
        while sampling[0]:
            st_ = N.service_stats(0)
            samples.append([round(time.perf_counter() - t0, 2)] + [int(st_[k]) for k in ("device_chunks", "live_waves", "running", "launches", "watchdog_launches")])
            time.sleep(0.1)

    sth = threading.Thread(target=sampler); sth.start()
    [x.start() for x in th]
    time.sleep(2.0)
    lat = {1: [], 4: []}; first = {}
    while time.perf_counter() - t0 < args.seconds:
        if args.no_fetch:
            time.sleep(0.2); continue
        for k in (1, 4):
            v = fetch(k); lat[k].append(v); first.setdefault(k, round(v, 2))
        time.sleep(0.05)
    stop[0] = True
    [x.join() for x in th]
    el = time.perf_counter() - t0
    sampling[0] = False; sth.join()
    assert args.no_fetch or np.array_equal(hbk, hsrc[:4 * CH])
    assert all((x["status"] == 0).all() for x in ds)
    st = N.service_stats(0)
    out = {"tag": args.tag, "process": "no torch: the system's HIP runtime", "upload_shape": args.shape, "compress_callers": T, "chunks_offered": T * n, "reserved_cus": st["reserved_cus"],
           "service_launches": st["launches"], "watchdog_launches": st["watchdog_launches"], "cpu_affinity": affinity, "fetching": not args.no_fetch,
           "compress_gibs_whole_window": round(sum(done) * n * CH / GiB / el, 3), "fetch_idle_ms": idle}
    out["service"] = {k: st[k] - sv0[k] for k in ("launches", "guest_launches", "yielded_waves", "returned_chunks", "readmissions", "rotations", "wave_starts", "reserved_exits", "relocated_waves")}
    out["completions_at_s"] = [round(float(x - t0), 2) for x in sorted(stamps)]; out["window_s"] = round(el, 2)
    if samples:
        # chunks the device finished per 100 ms sample (a hole shows as a run of zeros), and the samples around the slowest second
        a = np.asarray(samples); dc = np.diff(a[:, 1]); out["service_samples"] = {"fields": ["at_s", "device_chunks", "live_waves", "running", "launches", "watchdog_launches"],
            "chunks_per_sample": dc.tolist(), "live_waves": a[:, 2].tolist(), "running": a[:, 3].tolist(), "launches": a[:, 4].tolist(), "at_s": a[:, 0].tolist()}
    da = np.sort(np.asarray(stamps)) - t0
    if da.size >= 8:
        k0, k1 = int(da.size * 0.2), int(da.size * 0.8)
        out["compress_gibs_slope"] = round(float(np.polyfit(da[k0:k1], np.arange(k0, k1), 1)[0]) * n * CH / GiB, 3)
    for k in (() if args.no_fetch else (1, 4)):
        a = np.asarray(lat[k])
        out["fetch_%d_under_load_ms" % k] = {"n": int(a.size), "first": first[k], "p50": round(float(np.median(a)), 2), "p95": round(float(np.percentile(a, 95)), 2), "max": round(float(a.max()), 2)}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", required=True); ap.add_argument("--ivs", required=True)
    ap.add_argument("--shape", default="broker", choices=["broker", "batches"])
    ap.add_argument("--callers", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--chunk", type=int, default=4 << 20)
    ap.add_argument("--reserved-cus", type=int, default=-1)
    ap.add_argument("--max-launch-ms", type=int, default=-1)
    ap.add_argument("--no-fetch", action="store_true")
    ap.add_argument("--sample-service", action="store_true", help="record tsx_service_stats every 100 ms (device-side progress over the window)")
    ap.add_argument("--config", default="", help="key=value,... for tsx_debug_config before tsx_init (measurement variants)")
    ap.add_argument("--phases", action="store_true")
    ap.add_argument("--phases-short", action="store_true", help="with --phases: stop after phase C")
    ap.add_argument("--tag", default="")
    run(ap.parse_args())
