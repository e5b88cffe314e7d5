#!/usr/bin/env python3
"""The broker-shaped leg of bench.py's `end_to_end` object, runnable in a process of its own WITHOUT torch.

Shape (reference README.md:218-222, RemoteStorageManager.java:400-432: >= 10 RLM upload threads, one segment each): `callers`
threads, every call ONE B-chunk segment, context-less (pooled contexts, as the JNI shim calls), TSX_MEM_HOST from a registered
source into bound-sized slots of a registered per-thread output buffer - what GpuTransformChunkEnumeration issues.  The loop is closed: a caller's next
call follows its last.

Why it can run in a process of its own: a process has ONE HIP runtime - the first one loaded.  bench.py imports torch, and torch
brings its own (HIP 7.0.2 in this image), which moves device -> host copies with blit KERNELS; the system's runtime (7.2, what a
broker's JVM loads through libtsxform.so) uses the SDMA engines.  A copy kernel needs CU slots, and on a chip full of second-long
compressor waves it waits for them: standalone, 32 callers moved 11.7 GiB/s with torch in the process and 14.3 without
(profiles/r03_copy_engine_probe.txt, r03_broker_with_and_without_torch.jsonl).  As a CHILD of bench.py it shares the device with the
parent's queues and is slower than either, so bench.py calls run() in-process by default (--broker-subprocess for the child).

  broker_leg.py --src S.npy --ivs I.npy [--expect L.npy | --sizes-out L.npy] --callers 10,20,32 [--batch 256] [--chunk 4194304] [--window 8] [--lib path]
prints one JSON line: the list of rows.  bench.py runs this FIRST, as a child that owns the device alone (before the parent initialises
HIP), so that the driver's line carries what the product's runtime does; --gen writes the source segments for it (a short-lived helper
process with torch: the chunks are generated on the device)."""
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


def run(N, nat, params, hsrc, ivs, expect, callers_list, B, CH, window, profile_note="", sizes_out=None):
    """hsrc: nseg * B * CH source bytes (registered here), ivs: (nseg * B, 12), expect: dst_len of every chunk from the device-resident run
    (None when this leg runs BEFORE that run: the sizes every caller saw are then handed back through sizes_out for the parent to compare)."""
    nseg = hsrc.size // (B * CH)
    # Output exactly as GpuTransformChunkEnumeration.java:167-201 lays it out: one bound-sized slot per chunk (align16(bound) + 16 apart)
    # in the thread's reused, registered direct buffer, TSX_MEM_HOST.  (Until round 4 this leg asked for the packed layout into a smaller
    # buffer - what the enumeration does NOT issue - and paid 256 copies per segment for it.)
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    slot = (N.transformed_bound(CH, flags) + 15) // 16 * 16 + 16
    cap = B * slot
    lock = threading.Lock()
    rows = []
    N.host_register(hsrc)
    bufs = []
    try:
        for callers in callers_list:
            while len(bufs) < callers:
                hb = np.zeros(cap, np.uint8); N.host_register(hb); bufs.append(hb)
            segs = [hsrc[(t % nseg) * B * CH:((t % nseg) + 1) * B * CH] for t in range(callers)]
            des = []
            for t in range(callers):
                dd = np.zeros(B, nat.DESC_DTYPE)
                dd["src_off"] = np.arange(B, dtype=np.uint64) * CH; dd["src_len"] = CH
                dd["dst_off"] = np.arange(B, dtype=np.uint64) * slot; dd["dst_cap"] = N.transformed_bound(CH, flags)
                dd["iv"] = ivs[(t % nseg) * B:((t % nseg) + 1) * B]
                des.append(dd)
            lat = [[] for _ in range(callers)]
            stamps = []
            stop_at = [0.0]

            def bworker(t, warm):
                while True:
                    a = time.perf_counter()
                    N.transform_batch(params, des[t], segs[t], bufs[t], cap, nat.MEM_HOST, ctx=None)
                    b_ = time.perf_counter()
                    if warm:
                        return
                    lat[t].append(b_ - a)
                    with lock:
                        stamps.append(b_)
                    if b_ >= stop_at[0]:
                        return

            th = [threading.Thread(target=bworker, args=(t, True)) for t in range(callers)]     # pooled contexts and their workspaces exist
            [x.start() for x in th]; [x.join() for x in th]
            t1 = time.perf_counter()
            stop_at[0] = t1 + window
            th = [threading.Thread(target=bworker, args=(t, False)) for t in range(callers)]
            [x.start() for x in th]; [x.join() for x in th]
            if expect is None:
                # no reference sizes yet: every caller of a segment must agree with the first one, and the parent checks those against its own run
                ok = all(bool((dd["status"] == 0).all()) and bool((dd["dst_len"] == des[t % nseg]["dst_len"]).all()) for t, dd in enumerate(des))
                if sizes_out is not None and callers >= nseg:
                    sizes_out[:] = np.concatenate([des[k]["dst_len"] for k in range(nseg)])
            else:
                ok = all(bool((dd["status"] == 0).all()) and bool((dd["dst_len"] == expect[(t % nseg) * B:((t % nseg) + 1) * B]).all())
                         for t, dd in enumerate(des))
            # rate = least-squares slope of completions over time across the middle 60 % of the run (no ramp, no drain; counting the
            # calls that end inside a fixed window would quantise: at 2.5 s per call a caller completes one or two calls in it)
            done_at = np.sort(np.asarray(stamps)) - t1
            done = len(done_at)
            k0, k1 = int(done * 0.2), max(int(done * 0.8), int(done * 0.2) + 2)
            slope = float(np.polyfit(done_at[k0:k1], np.arange(k0, min(k1, done)), 1)[0]) if done >= 4 else done / max(float(done_at[-1]), 1e-9)
            gibs = slope * B * CH / GiB
            rows.append({"callers": callers, "batch_chunks": B, "chunks_offered": callers * B, "calls": done, "seconds": round(float(done_at[-1]), 2),
                         "models": {10: "10 RLM upload threads, no read-ahead", 20: "10 RLM upload threads + one batch of read-ahead each (gpu.read.ahead, the default)",
                                    32: "16 RLM upload threads + read-ahead"}.get(callers, "%d callers" % callers),
                         "method": "slope of completions, middle 60 %", "context": "pooled (ctx = NULL), members of the device's compressor service", "dst_layout": "slots (as GpuTransformChunkEnumeration.java:167-201)",
                         "host_memory": "source and outputs registered", "gibs": round(gibs, 4),
                         "ms_per_call_median": round(float(np.median(np.concatenate([np.asarray(x) for x in lat if x]))) * 1e3, 1),
                         "whole_window_gibs": round(done * B * CH / GiB / max(float(done_at[-1]), 1e-9), 4),
                         "same_sizes_as_device_run": ok, "torch_in_process": "torch" in sys.modules})
    finally:
        for hb in bufs:
            N.host_unregister(hb)
        N.host_unregister(hsrc)
    return rows


def gen(path_src, path_ivs, nseg, B, CH, dist):
    """The leg's source segments, generated on the device exactly as bench.py generates them (segments 0 .. nseg - 1 of rank 0)."""
    import torch
    from tsxform import synth
    dev = torch.device("cuda", 0)
    out = np.empty(nseg * B * CH, np.uint8)
    for s_ in range(nseg):
        for c in range(B):
            i = s_ * B + c
            out[i * CH:(i + 1) * CH] = synth.gen_chunk(dist, 1000 + s_, s_, c, CH, device=dev).cpu().numpy()
    np.save(path_src, out)
    np.save(path_ivs, np.stack([np.frombuffer(synth.iv_for(s_, c), np.uint8) for s_ in range(nseg) for c in range(B)]))


from numa_bind import bind_to_gpu_numa_node  # noqa: E402  (tools/numa_bind.py)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--gen":
        gen(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", required=True); ap.add_argument("--ivs", required=True)
    ap.add_argument("--expect", default="", help="dst_len of every chunk from the device-resident run (omit when this leg runs first: see --sizes-out)")
    ap.add_argument("--sizes-out", default="", help="write the dst_len every caller agreed on here (.npy), for the parent to compare with its own run")
    ap.add_argument("--callers", default="10,20,32")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=4 << 20)
    ap.add_argument("--window", type=float, default=8.0)
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--lib", default="")
    ap.add_argument("--no-numa-bind", action="store_true")
    args = ap.parse_args()
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    N = nat.Native(args.lib) if args.lib else nat.Native()
    N.init(1, [0])
    numa = "not asked" if args.no_numa_bind else bind_to_gpu_numa_node(0)     # before any buffer is allocated, touched or pinned
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=args.profile)
    hsrc = np.load(args.src); ivs = np.load(args.ivs); expect = np.load(args.expect) if args.expect else None
    sizes = np.zeros(hsrc.size // args.chunk, np.uint32)
    rows = run(N, nat, params, hsrc, ivs, expect, [int(x) for x in args.callers.split(",")], args.batch, args.chunk, args.window, sizes_out=sizes)
    if args.sizes_out:
        np.save(args.sizes_out, sizes)
    for r_ in rows:
        r_["cpu_affinity"] = numa
    print(json.dumps(rows), flush=True)


if __name__ == "__main__":
    main()
