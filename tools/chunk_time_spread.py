#!/usr/bin/env python3
"""How far apart do the chunks of a batch finish?  (libtsxform_prof.so: per-chunk cycle totals and sequence counts.)  A batch is
done when its slowest chunk is; with nothing queued behind it the slots of the chunks that finished early stay empty.  The batch
is 256 distinct K chunks x 8: the spread among the 8 copies of one content is the hardware's (placement, contention), the spread
between contents is the data's.   usage: chunk_time_spread.py [inflight=1]"""
import ctypes as C, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tsxform
from tsxform import synth
nat = tsxform._native
N = nat.Native(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", "libtsxform_prof.so")); N.init(1, [0])
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, CH = 2048, synth.CHUNK
dev = torch.device("cuda", 0)
src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
if os.path.exists("/tmp/k256.npy"):
    src[:256 * CH] = torch.from_numpy(np.load("/tmp/k256.npy")).to(dev)
else:
    for i in range(256): src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
    np.save("/tmp/k256.npy", src[:256 * CH].cpu().numpy())
for i in range(256, n, 256): src[i * CH:(i + 256) * CH] = src[:256 * CH]
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
profs = [torch.zeros(n * 24, dtype=torch.int64, device=dev) for _ in range(T)]
N.lib.tsx_debug_set_prof.restype = None; N.lib.tsx_debug_set_prof.argtypes = [C.c_void_p]
ctxs, dsts, ds = [], [], []
for t in range(T):
    ctxs.append(N.ctx_create(0, n, CH)); dsts.append(torch.empty(n * slot, dtype=torch.uint8, device=dev))
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot; ds.append(d)
def step(t): N.transform_batch(params, ds[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
N.lib.tsx_debug_set_prof(profs[0].data_ptr())          # one pointer for the library: with T > 1 the batches (same data) share it, last writer wins
for t in range(T): step(t)
torch.cuda.synchronize()
profs[0].zero_()
th = [threading.Thread(target=lambda t=t: [step(t) for _ in range(2)]) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize()
p = profs[0].cpu().numpy().reshape(n, 24)
tot, nseq = p[:, 14].astype(np.float64), p[:, 13].astype(np.float64)
q = np.percentile(tot, [0, 10, 50, 90, 99, 100])
print("batches in flight %d: per-chunk cycles min %.3g p10 %.3g median %.3g p90 %.3g p99 %.3g max %.3g | max/median %.3f, mean/max %.3f (share of the slot-time a batch uses)"
      % (T, *q, q[5] / q[2], tot.mean() / q[5]))
by = tot.reshape(8, 256)                                   # [copy, content]
print("  between contents (mean over the 8 copies): std/mean %.4f, max/median %.3f | among the copies of one content: mean std/mean %.4f, mean max/min %.3f"
      % (by.mean(0).std() / by.mean(), by.mean(0).max() / np.median(by.mean(0)), (by.std(0) / by.mean(0)).mean(), (by.max(0) / by.min(0)).mean()))
print("  sequences per chunk: min %d median %d max %d; correlation(cycles, sequences) = %.3f; cycles per sequence: p10 %.0f median %.0f p90 %.0f"
      % (nseq.min(), np.median(nseq), nseq.max(), np.corrcoef(tot, nseq)[0, 1], *np.percentile(tot / nseq, [10, 50, 90])))
# position in the launch: workgroup i of the grid
k = np.arange(n)
print("  by position in the grid (quarters): mean cycles %s" % " ".join("%.4g" % tot[k // 512 == j].mean() for j in range(4)))
print("  by workgroup id mod 8 (XCD): mean cycles %s" % " ".join("%.4g" % tot[k % 8 == j].mean() for j in range(8)))
