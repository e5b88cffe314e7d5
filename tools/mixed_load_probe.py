#!/usr/bin/env python3
"""A broker uploads and serves fetches at the same time.  How long does a fetch (tsx_detransform_batch of 1 / 4 chunks, host -> host,
registered buffers, its own context) take while T caller threads keep the chip full of compressor waves (device-resident 2048-chunk
batches, as bench.py's timed region)?  One JSON line: latencies on the idle device and under load.
  python tools/mixed_load_probe.py [--callers 5] [--seconds 12] [--reserved-cus n]     (n: tsx_config.fetch_reserved_cus; default: the library's)"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402
import tsxform  # noqa: E402
from tsxform import synth  # noqa: E402

nat = tsxform._native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--callers", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--tag", default="")
    ap.add_argument("--shape", default="batches", choices=["batches", "broker"],
                    help="the upload load: `batches` = --callers explicit contexts x 2048-chunk device-resident batches (more chunks queued than the chip has "
                         "slots); `broker` = --callers context-less calls of ONE 256-chunk segment each, registered host buffers, slot layout (tools/broker_leg.py)")
    ap.add_argument("--trace", action="store_true", help="the library's phase trace (stderr) for the fetches under load")
    ap.add_argument("--max-launch-ms", type=int, default=-1)
    ap.add_argument("--no-fetch", action="store_true", help="the same upload load without any fetch: the rate the fetches (and the reservation) are set against")
    ap.add_argument("--reserved-cus", type=int, default=-1, help="tsx_config.fetch_reserved_cus (-1: the library's default)")
    args = ap.parse_args()
    N = nat.Native(); N.init(1, [0], fetch_reserved_cus=None if args.reserved_cus < 0 else args.reserved_cus, service_max_launch_ms=None if args.max_launch_ms < 0 else args.max_launch_ms)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from numa_bind import bind_to_gpu_numa_node
    AFFINITY = bind_to_gpu_numa_node(0)                              # before any host buffer is allocated (profiles/r04_broker_numa.txt)
    dev = torch.device("cuda", 0)
    CH, n = synth.CHUNK, 2048
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
    for i in range(256):
        src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
    for i in range(256, n, 256):
        src[i * CH:(i + 256) * CH] = src[:256 * CH]
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for i in range(n):
        d["iv"][i] = np.frombuffer(synth.iv_for(0, i % 256), np.uint8)
    T = args.callers
    broker = args.shape == "broker"
    if not broker:
        ctxs = [N.ctx_create(0, n, CH) for _ in range(T)]
        dsts = [torch.empty(n * slot, dtype=torch.uint8, device=dev) for _ in range(T)]
        ds = [d.copy() for _ in range(T)]
        for t in range(T):
            N.transform_batch(params, ds[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
    else:
        B = 256
        hslot = (N.transformed_bound(CH, flags) + 15) // 16 * 16 + 16
        hsrc = src[:B * CH].cpu().numpy(); N.host_register(hsrc)
        hdsts = []
        for t in range(T):
            hb = np.zeros(B * hslot, np.uint8); N.host_register(hb); hdsts.append(hb)
        ds = []
        for t in range(T):
            dd = d[:B].copy(); dd["dst_off"] = np.arange(B, dtype=np.uint64) * hslot; dd["dst_cap"] = N.transformed_bound(CH, flags); ds.append(dd)
        dsts = [torch.empty(4 * slot, dtype=torch.uint8, device=dev)]
        d4 = d[:4].copy()
        c0 = N.ctx_create(0, 4, CH)
        N.transform_batch(params, d4, src.data_ptr(), dsts[0].data_ptr(), dsts[0].numel(), nat.MEM_DEVICE, ctx=c0)      # the chunks the fetches restore
        N.ctx_destroy(c0)
        ds_fetch = d4
    torch.cuda.synchronize()
    # the fetch side: 4 transformed chunks in a registered host buffer, restored into a registered host buffer
    hfr = dsts[0][:4 * slot].cpu().numpy(); hbk = np.zeros(4 * CH, np.uint8)
    N.host_register(hfr); N.host_register(hbk)
    fctx = N.ctx_create(0, 4, CH)
    want = src[:4 * CH].cpu().numpy()

    if not broker:
        ds_fetch = ds[0]

    def fetch(k):
        e = np.zeros(k, nat.DESC_DTYPE); e["src_off"] = ds_fetch["dst_off"][:k]; e["src_len"] = ds_fetch["dst_len"][:k]; e["iv"] = ds_fetch["iv"][:k]
        e["dst_off"] = np.arange(k, dtype=np.uint64) * CH; e["dst_cap"] = CH
        t0 = time.perf_counter()
        N.detransform_batch(params, e, hfr, hbk, hbk.size, nat.MEM_HOST, ctx=fctx)
        dt = time.perf_counter() - t0
        assert (e["status"] == 0).all()
        return dt

    for k in (1, 4):
        fetch(k)
    idle = {k: float(np.median([fetch(k) for _ in range(7)])) * 1e3 for k in (1, 4)}
    assert np.array_equal(hbk, want)
    stop = [False]
    done = [0] * T
    stamps = []
    slock = threading.Lock()

    def worker(t):
        while not stop[0]:
            if broker:
                N.transform_batch(params, ds[t], hsrc, hdsts[t], hdsts[t].size, nat.MEM_HOST, ctx=None)
            else:
                N.transform_batch(params, ds[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
            done[t] += 1
            with slock:
                stamps.append(time.perf_counter())

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    time.sleep(2.0)                                                     # the chip is full
    if args.trace:
        N.debug_config("trace", 1)
    lat = {1: [], 4: []}
    while time.perf_counter() - t0 < args.seconds:
        if args.no_fetch:
            time.sleep(0.2)
            continue
        for k in (1, 4):
            lat[k].append(fetch(k) * 1e3)
        time.sleep(0.05)
    stop[0] = True
    [x.join() for x in th]
    el = time.perf_counter() - t0
    assert np.array_equal(hbk, want)
    st = N.service_stats(0)
    out = {"tag": args.tag, "reserved_cus": st["reserved_cus"], "cu_keys_seen": st["cu_keys_seen"], "service_launches": st["launches"], "watchdog_launches": st["watchdog_launches"], "rotations": st["rotations"],
           "compress_callers": T,
           "upload_shape": args.shape, "chunks_offered": T * (256 if broker else n),
           "compress_gibs_while_fetching": round(sum(done) * (256 if broker else n) * CH / float(1 << 30) / el, 3), "fetching": not args.no_fetch,
           "fetch_idle_ms": {k: round(v, 3) for k, v in idle.items()}}
    da = np.sort(np.asarray(stamps)) - t0
    if da.size >= 8:                                                  # rate without ramp and drain: slope of completions over the middle 60 %
        k0, k1 = int(da.size * 0.2), int(da.size * 0.8)
        out["compress_gibs_slope"] = round(float(np.polyfit(da[k0:k1], np.arange(k0, k1), 1)[0]) * (256 if broker else n) * CH / float(1 << 30), 3)
    for k in (() if args.no_fetch else (1, 4)):
        a = np.asarray(lat[k])
        out["fetch_%d_under_load_ms" % k] = {"n": int(a.size), "median": round(float(np.median(a)), 2), "p95": round(float(np.percentile(a, 95)), 2), "max": round(float(a.max()), 2)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
