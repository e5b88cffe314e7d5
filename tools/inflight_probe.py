#!/usr/bin/env python3
"""Does a second batch in flight raise throughput?  T threads, each with its own ctx + buffers, call the synchronous
tsx_transform_batch concurrently (ctypes releases the GIL) over the same 2048 resident chunks."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tsxform
from tsxform import synth
nat = tsxform._native
N = nat.Native(os.path.join(os.path.dirname(nat.LIB_PATH), sys.argv[2] if len(sys.argv) > 2 else "libtsxform.so")); N.init(1, [0])
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, CH = 2048, synth.CHUNK
dev = torch.device("cuda", 0)
src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
if os.path.exists("/tmp/k256.npy"):
    src[:256 * CH] = torch.from_numpy(np.load("/tmp/k256.npy")).to(dev)
else:
    for i in range(256): src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
for i in range(256, n, 256): src[i * CH:(i + 256) * CH] = src[:256 * CH]
flags = nat.COMPRESS
slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
ctxs, dsts, descs = [], [], []
for t in range(T):
    ctxs.append(N.ctx_create(0, n, CH)); dsts.append(torch.empty(n * slot, dtype=torch.uint8, device=dev))
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot; descs.append(d)
def work(t, steps):
    for _ in range(steps): N.transform_batch(params, descs[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
for t in range(T): work(t, 1)
torch.cuda.synchronize()
steps = 3
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(t, steps)) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("threads %d: %.1f ms per batch-equivalent, %.2f GiB/s" % (T, el / (steps * T) * 1e3, steps * T * n * CH / 2**30 / el))
assert all((d["status"] == 0).all() for d in descs)
#assert (descs[0]["dst_len"] == descs[-1]["dst_len"]).all()
