#!/usr/bin/env python3
"""A/B of compressor builds on one box (GPU): the same 2048-chunk K batch (256 distinct chunks x 8) through every library given,
three caller threads in flight as in bench.py, plus one batch at a time.  Outputs of all libraries must be identical.
METHOD (learnt the hard way in round 2): give ONE library per process and alternate processes - whatever is loaded second into a
process measures 15-20 % lower with batches in flight - and use >= 18 steps: shorter runs are dominated by the simultaneous start
of the callers' first batches.  Several libraries in one call are still useful for the digest comparison.
    python tools/sweep_libs.py [--steps 6] tools/_libs/libtsxform_a.so tools/_libs/libtsxform_b.so ..."""
import argparse, hashlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tsxform
from tsxform import synth
nat = tsxform._native
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=18)
ap.add_argument("--chunks", type=int, default=2048)
ap.add_argument("--inflight", type=int, default=3)
ap.add_argument("--dist", default="K")
ap.add_argument("--prealloc-gib", type=int, default=0, help="allocate (and keep) this much device memory before anything else: placement experiment")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
n, CH, T = a.chunks, synth.CHUNK, a.inflight
dev = torch.device("cuda", 0)
_dummy = torch.empty(a.prealloc_gib << 30, dtype=torch.uint8, device=dev) if a.prealloc_gib else None
src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
cache = "/tmp/%s256.npy" % a.dist.lower()
if os.path.exists(cache):
    src[:256 * CH] = torch.from_numpy(np.load(cache)).to(dev)
else:
    for i in range(256): src[i * CH:(i + 1) * CH] = synth.gen_chunk(a.dist, 1000, 0, i, CH, device=dev)
    np.save(cache, src[:256 * CH].cpu().numpy())
for i in range(256, n, 256): src[i * CH:(i + 256) * CH] = src[:256 * CH]
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
ref = None
for path in a.libs:
    N = nat.Native(os.path.abspath(path)); N.init(1, [0])
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=nat.ZSTD_PROFILE_1_5_7)
    d0 = np.zeros(n, nat.DESC_DTYPE); d0["src_off"] = np.arange(n, dtype=np.uint64) * CH; d0["src_len"] = CH
    d0["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d0["dst_cap"] = slot
    ds = [d0.copy() for _ in range(T)]
    dsts = [torch.empty(n * slot, dtype=torch.uint8, device=dev) for _ in range(T)]
    ctxs = [N.ctx_create(0, n, CH) for _ in range(T)]
    def step(t): N.transform_batch(params, ds[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
    for t in range(T): step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(0); step(0); torch.cuda.synchronize(); one = 2 * n * CH / 2**30 / (time.perf_counter() - t0)
    zms = N.ctx_timing(ctxs[0]).zstd_ms
    def worker(t):
        for _ in range(t, a.steps, T): step(t)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    ok = all((x["status"] == 0).all() for x in ds)
    h = hashlib.sha256()
    h.update(ds[0]["dst_len"].tobytes()); h.update(ds[0]["crc32c"].tobytes())
    for i in (0, 1, 100, 255, n - 1):
        h.update(dsts[0][i * slot:i * slot + int(ds[0]["dst_len"][i])].cpu().numpy().tobytes())
    dig = h.hexdigest()[:16]
    if ref is None: ref = dig
    print(json.dumps({"lib": os.path.basename(path), "gibs_inflight%d" % T: round(a.steps * n * CH / 2**30 / el, 3), "gibs_one_at_a_time": round(one, 3),
                      "zstd_ms_solo": round(zms, 1), "status_ok": bool(ok), "digest": dig, "same_as_first": dig == ref}), flush=True)
    for c in ctxs: N.ctx_destroy(c)
    del dsts
    torch.cuda.empty_cache()
