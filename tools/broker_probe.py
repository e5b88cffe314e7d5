#!/usr/bin/env python3
"""The shape a broker drives the forward chain in: T caller threads (RLM upload threads, reference README.md:218-222,
RemoteStorageManager.java:400-432), each submitting B-chunk batches back to back.  Prints one JSON line per configuration.

  --mem device   resident buffers (what bench.py's `value` is quoted on)
  --mem host     TSX_MEM_HOST_PACKED from / to tsx_host_register'ed host buffers (what the JNI shim passes)
  --ctxless      pooled contexts (tsx_transform_batch(ctx = NULL)), as GpuTransformChunkEnumeration calls
Each configuration is "threads x batch": e.g.  --configs 10x256,20x256,3x2048
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GiB = float(1 << 30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="10x256,20x256,3x2048")
    ap.add_argument("--mem", default="device", choices=["device", "host"])
    ap.add_argument("--ctxless", action="store_true")
    ap.add_argument("--seconds", type=float, default=6.0, help="approximate run time per configuration")
    ap.add_argument("--pool-chunks", type=int, default=2048, help="distinct source chunks generated (threads take slices)")
    ap.add_argument("--tag", default="")
    ap.add_argument("--src-file", default="", help=".npy of the source chunks: written if missing (needs torch), loaded otherwise - with --mem host the process then "
                    "runs WITHOUT torch, i.e. on the system's HIP runtime as a broker's JVM does (torch bundles its own, older one: profiles/r03_copy_engine_probe.txt)")
    ap.add_argument("--gen-only", action="store_true")
    ap.add_argument("--layout", default="slots", choices=["slots", "packed"], help="host output layout: bound-sized slots (TSX_MEM_HOST, what "
                    "GpuTransformChunkEnumeration.java:167-201 issues) or packed (TSX_MEM_HOST_PACKED into a 2 MiB-per-chunk buffer, the round-3 shape)")
    args = ap.parse_args()
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    CH = synth.CHUNK
    P = args.pool_chunks
    torch = None
    hsrc = None
    if args.mem == "host" and args.src_file and os.path.exists(args.src_file):
        hsrc = np.load(args.src_file)
        assert hsrc.size == P * CH
    if hsrc is None:
        import torch                                   # before libtsxform: a process has ONE HIP runtime, the first one loaded (torch bundles its own)
    N = nat.Native()
    N.init(1, [0])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from numa_bind import bind_to_gpu_numa_node
    AFFINITY = bind_to_gpu_numa_node(0)                              # before any host buffer is allocated (profiles/r04_broker_numa.txt)
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    if hsrc is None:
        dev = torch.device("cuda", 0)
        src = torch.empty(P * CH, dtype=torch.uint8, device=dev)
        for i in range(P):
            src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000 + i // 256, i // 256, i % 256, CH, device=dev)
        torch.cuda.synchronize()
        if args.mem == "host" or args.gen_only:
            hsrc = src.cpu().numpy()
            if args.src_file:
                np.save(args.src_file, hsrc)
        if args.gen_only:
            return
    if args.mem == "host":
        N.host_register(hsrc)
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    for cfg in args.configs.split(","):
        T, B = (int(x) for x in cfg.split("x"))
        descs, dsts, ctxs, offs = [], [], [], []
        for t in range(T):
            lo = (t * B) % max(P - B + 1, 1)
            d = np.zeros(B, nat.DESC_DTYPE)
            d["src_off"] = (np.arange(B, dtype=np.uint64) + np.uint64(lo)) * np.uint64(CH)
            d["src_len"] = CH
            d["dst_off"] = np.arange(B, dtype=np.uint64) * np.uint64(slot)
            d["dst_cap"] = slot
            for i in range(B):
                d["iv"][i] = np.frombuffer(synth.iv_for(t, i), np.uint8)
            descs.append(d)
            if args.mem == "device":
                dsts.append(torch.empty(B * slot, dtype=torch.uint8, device=dev))
            else:
                h = np.zeros(B * (slot if args.layout == "slots" else (2 << 20)), np.uint8)   # (packed: 0.31 x 4 MiB per chunk, 2 MiB of room)
                N.host_register(h)
                dsts.append(h)
            ctxs.append(None if args.ctxless else N.ctx_create(0, B, CH))

        def call(t):
            if args.mem == "device":
                N.transform_batch(params, descs[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
            else:
                N.transform_batch(params, descs[t], hsrc, dsts[t], dsts[t].size, nat.MEM_HOST if args.layout == "slots" else nat.MEM_HOST_PACKED, ctx=ctxs[t])

        for t in range(min(T, 4)):                       # workspaces / pools exist before the clock starts
            call(t)
        if torch is not None:
            torch.cuda.synchronize()
        done = [0] * T
        lat = [[] for _ in range(T)]
        stop_at = [0.0]

        def worker(t):
            while time.perf_counter() < stop_at[0]:
                a = time.perf_counter()
                call(t)
                lat[t].append(time.perf_counter() - a)
                done[t] += 1

        t0 = time.perf_counter()
        stop_at[0] = t0 + args.seconds
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
        if torch is not None:
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ok = all(bool((d["status"] == 0).all()) for d in descs)
        allat = np.concatenate([np.asarray(x) for x in lat]) if sum(done) else np.zeros(1)
        phases = None                                                    # (round 4's per-phase hook went with the launch combiner; tsx_ctx_timing has h2d / zstd / d2h per call)
        print(json.dumps({"tag": args.tag, "phases_mean_per_call": phases, "torch_in_process": "torch" in sys.modules, "threads": T, "batch_chunks": B, "mem": args.mem, "ctxless": args.ctxless, "layout": args.layout if args.mem == "host" else None,
                          "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "batches": int(sum(done)), "seconds": round(el, 3),
                          "gibs": round(sum(done) * B * CH / GiB / el, 3), "ms_per_call_median": round(float(np.median(allat)) * 1e3, 1),
                          "ms_per_call_p95": round(float(np.percentile(allat, 95)) * 1e3, 1), "ok": ok}), flush=True)
        for c in ctxs:
            if c is not None:
                N.ctx_destroy(c)
        if args.mem == "host":
            for h in dsts:
                N.host_unregister(h)
        del dsts
        if torch is not None:
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
