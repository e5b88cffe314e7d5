#!/usr/bin/env python3
"""Where does a fetch wait while the compressor service's kernel is alive?  (gpurun r05a / r05b: the first fetch after uploads began came back
when the service kernel ended - 18.6 s / 58 s, its age limit - every later one in milliseconds.)  Upload load as tools/mixed_load_probe.py
(--callers x 2048-chunk device-resident batches); then, every 2 s, one operation each of: a CRC of one chunk (device memory, a context made
BEFORE the load), the same on a context made DURING the load, a one-chunk fetch device -> device, a one-chunk fetch host -> host -
with the library's phase trace on stderr.  One JSON line per operation.
  python tools/fetch_block_probe.py [--callers 5] [--seconds 20] [--svc-normal-priority] [--max-launch-ms 15000]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tsxform  # noqa: E402
from tsxform import synth  # noqa: E402

nat = tsxform._native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--callers", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--svc-normal-priority", action="store_true")
    ap.add_argument("--max-launch-ms", type=int, default=15000)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    N = nat.Native()
    if args.svc_normal_priority:
        N.debug_config("svc_normal_priority", 1)
    N.init(1, [0], service_max_launch_ms=args.max_launch_ms)
    dev = torch.device("cuda", 0)
    CH, n, T = synth.CHUNK, 2048, args.callers
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
    for i in range(64):
        src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
    for i in range(64, n, 64):
        src[i * CH:(i + 64) * CH] = src[:64 * CH]
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for i in range(n):
        d["iv"][i] = np.frombuffer(synth.iv_for(0, i % 64), np.uint8)
    ctxs = [N.ctx_create(0, n, CH) for _ in range(T)]
    dsts = [torch.empty(n * slot, dtype=torch.uint8, device=dev) for _ in range(T)]
    ds = [d.copy() for _ in range(T)]
    N.transform_batch(params, ds[0], src.data_ptr(), dsts[0].data_ptr(), dsts[0].numel(), nat.MEM_DEVICE, ctx=ctxs[0])
    torch.cuda.synchronize()
    one = dsts[0][:slot].clone()                                       # one transformed chunk, device and host copies
    hfr = one.cpu().numpy(); hbk = np.zeros(CH, np.uint8); dbk = torch.empty(CH, dtype=torch.uint8, device=dev)
    N.host_register(hfr); N.host_register(hbk)
    c_crc, c_dev, c_host = N.ctx_create(0, 4, CH), N.ctx_create(0, 4, CH), N.ctx_create(0, 4, CH)
    e1 = np.zeros(1, nat.DESC_DTYPE); e1["src_len"] = ds[0]["dst_len"][:1]; e1["iv"] = ds[0]["iv"][:1]; e1["dst_cap"] = CH

    def op_crc(ctx):
        dd = d[:1].copy(); N.crc32c_batch(dd, src.data_ptr(), nat.MEM_DEVICE, ctx=ctx); return int(dd["status"][0])

    def op_dev():
        e = e1.copy(); N.detransform_batch(params, e, one.data_ptr(), dbk.data_ptr(), CH, nat.MEM_DEVICE, ctx=c_dev); return int(e["status"][0])

    def op_host():
        e = e1.copy(); N.detransform_batch(params, e, hfr, hbk, hbk.size, nat.MEM_HOST, ctx=c_host); return int(e["status"][0])

    for f in (lambda: op_crc(c_crc), op_dev, op_host):
        f(); f()
    stop = [False]

    def loader(t):
        while not stop[0]:
            N.transform_batch(params, ds[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])

    th = [threading.Thread(target=loader, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    time.sleep(2.0)
    if args.trace:
        N.debug_config("trace", 1)
    rows = []
    late = None
    while time.perf_counter() - t0 < args.seconds:
        if late is None:
            late = N.ctx_create(0, 4, CH)                              # streams made while the service kernel is alive
        for name, f in (("crc, context made before the load", lambda: op_crc(c_crc)), ("crc, context made during the load", lambda: op_crc(late)),
                        ("fetch 1 chunk, device -> device", op_dev), ("fetch 1 chunk, host -> host", op_host)):
            a = time.perf_counter()
            st = f()
            b = time.perf_counter()
            row = {"tag": args.tag, "at_s": round(a - t0, 2), "op": name, "ms": round((b - a) * 1e3, 2), "status": st, "service": {k: N.service_stats(0)[k] for k in ("launches", "watchdog_launches", "running")}}
            rows.append(row)
            print(json.dumps(row), flush=True)
        time.sleep(1.0)
    N.debug_config("trace", 0)
    stop[0] = True
    [x.join() for x in th]
    print(json.dumps({"tag": args.tag, "svc_normal_priority": bool(args.svc_normal_priority), "summary_ms": {nm: [r["ms"] for r in rows if r["op"] == nm] for nm in sorted(set(r["op"] for r in rows))}}), flush=True)


if __name__ == "__main__":
    main()
