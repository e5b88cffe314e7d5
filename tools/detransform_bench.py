#!/usr/bin/env python3
"""Inverse path timing (fetchLogSegment side): 2048 transformed chunks resident in HBM -> tsx_detransform_batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tsxform
from tsxform import synth
nat = tsxform._native
LIBNAME = sys.argv[2] if len(sys.argv) > 2 else "libtsxform.so"
N = nat.Native(os.path.join(os.path.dirname(nat.LIB_PATH), LIBNAME) if LIBNAME == "libtsxform.so" else os.path.join(ROOT, "tools", "_libs", LIBNAME)); N.init(1, [0])
n, CH = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, synth.CHUNK
dev = torch.device("cuda", 0)
src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
if os.path.exists("/tmp/k256.npy"):
    src[:256 * CH] = torch.from_numpy(np.load("/tmp/k256.npy")).to(dev)
else:
    for i in range(256): src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
for i in range(256, n, 256): src[i * CH:(i + 256) * CH] = src[:256 * CH]
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
mid = torch.empty(n * slot, dtype=torch.uint8, device=dev)
d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
ctx = N.ctx_create(0, n, CH)
N.transform_batch(params, d, src.data_ptr(), mid.data_ptr(), mid.numel(), nat.MEM_DEVICE, ctx=ctx)
crc = d["crc32c"].copy()
back = torch.empty(n * CH, dtype=torch.uint8, device=dev)
e = np.zeros(n, nat.DESC_DTYPE); e["src_off"] = d["dst_off"]; e["src_len"] = d["dst_len"]; e["dst_off"] = np.arange(n, dtype=np.uint64) * CH; e["dst_cap"] = CH
dprof = None
if "prof2" in LIBNAME:
    import ctypes as C
    dprof = torch.zeros(n * 8, dtype=torch.int64, device=dev)
    N.lib.tsx_debug_set_dprof.restype = None; N.lib.tsx_debug_set_dprof.argtypes = [C.c_void_p]
    N.lib.tsx_debug_set_dprof(dprof.data_ptr())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N.detransform_batch(params, e, mid.data_ptr(), back.data_ptr(), back.numel(), nat.MEM_DEVICE, ctx=ctx)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    t = N.ctx_timing(ctx)
    print("detransform %d chunks: %.1f ms -> %.2f GiB/s of restored bytes (gcm %.1f ms, unzstd %.1f ms, crc %.1f ms)" % (n, el * 1e3, n * CH / 2**30 / el, t.gcm_ms, t.unzstd_ms, t.crc_ms))
assert (e["status"] == 0).all() and (e["crc32c"] == crc).all() and torch.equal(back, src)
print("round trip exact")
# the same batch from T caller threads (own tsx_ctx, own output buffer), as the reference's fetch threads would: batches in flight
import threading
for T in (2, 3):
    ctxs = [N.ctx_create(0, n, CH) for _ in range(T)]
    backs = [torch.empty(n * CH, dtype=torch.uint8, device=dev) for _ in range(T)]
    es = [e.copy() for _ in range(T)]
    tms = [[0.0, 0.0, 0.0] for _ in range(T)]
    reps = 4
    def work(t):
        for _ in range(reps):
            N.detransform_batch(params, es[t], mid.data_ptr(), backs[t].data_ptr(), backs[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
            tt = N.ctx_timing(ctxs[t]); tms[t][0] += tt.gcm_ms; tms[t][1] += tt.unzstd_ms; tms[t][2] += tt.crc_ms
    for t in range(T): work.__call__(t) if False else None
    for t in range(T):
        N.detransform_batch(params, es[t], mid.data_ptr(), backs[t].data_ptr(), backs[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
    tms = [[0.0, 0.0, 0.0] for _ in range(T)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    g = sum(x[0] for x in tms) / (T * reps); u = sum(x[1] for x in tms) / (T * reps); c = sum(x[2] for x in tms) / (T * reps)
    print("detransform, %d batches in flight: %.1f ms per batch -> %.2f GiB/s of restored bytes (per call: gcm %.1f ms, unzstd %.1f ms, crc %.1f ms)" % (T, el * 1e3 / (T * reps), T * reps * n * CH / 2**30 / el, g, u, c))
    for t in range(T):
        assert (es[t]["status"] == 0).all() and (es[t]["crc32c"] == crc).all()
        N.ctx_destroy(ctxs[t])
    del backs
if dprof is not None:
    m = dprof.cpu().numpy().reshape(n, 8).mean(axis=0)
    # three waves per chunk, one per stage, one block apart: wave 1 = literals of block k, wave 0 = sequences of block k - 1, wave 2 = execution of block k - 2
    for k, name in [(0, "wave 1: headers + literals"), (6, "wave 1: waiting"), (1, "wave 0: sequence tables"), (2, "wave 0: FSE decode"), (4, "wave 0: rest"),
                    (7, "wave 0: waiting"), (3, "wave 2: execution"), (5, "wave 2: waiting")]:
        print("  %-34s %12.0f cycles" % (name, m[k]))
