#!/usr/bin/env python3
"""What a continuously fed device sustains, as opposed to bench.py's timed region (K steps from a barrier: every caller starts at
the same instant, so the chip works in generations that begin and end together).  T caller threads (own tsx_ctx, own output), B
batches each, thread t starting t x stagger late; completions are time-stamped and the rate is taken over the window in which
every thread is in its steady state (after its 2nd batch, before the first thread runs out of batches).
  usage: steady_state_probe.py T B stagger_ms [chunks_per_batch]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tsxform
from tsxform import synth
nat = tsxform._native
N = nat.Native(nat.LIB_PATH); N.init(1, [0])
T, B, stagger = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]) / 1e3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
CH = synth.CHUNK
dev = torch.device("cuda", 0)
src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
base = min(n, 256)
if os.path.exists("/tmp/k256.npy"):
    src[:base * CH] = torch.from_numpy(np.load("/tmp/k256.npy"))[:base * CH].to(dev)
else:
    for i in range(base): src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
    if base == 256: np.save("/tmp/k256.npy", src[:256 * CH].cpu().numpy())
for i in range(base, n, base): src[i * CH:(i + base) * CH] = src[:base * CH]
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
ctxs, dsts, descs = [], [], []
for t in range(T):
    ctxs.append(N.ctx_create(0, n, CH)); dsts.append(torch.empty(n * slot, dtype=torch.uint8, device=dev))
    d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for i in range(n): d["iv"][i] = np.frombuffer(synth.iv_for(t, i), np.uint8)
    descs.append(d)
torch.cuda.synchronize()
N.transform_batch(params, descs[0], src.data_ptr(), dsts[0].data_ptr(), dsts[0].numel(), nat.MEM_DEVICE, ctx=ctxs[0])     # warm
done = [[] for _ in range(T)]
def work(t):
    time.sleep(t * stagger)
    for _ in range(B):
        N.transform_batch(params, descs[t], src.data_ptr(), dsts[t].data_ptr(), dsts[t].numel(), nat.MEM_DEVICE, ctx=ctxs[t])
        done[t].append(time.perf_counter())
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
el = max(d[-1] for d in done) - t0
gib = n * CH / 2**30
lo = max(d[1] for d in done); hi = min(d[-1] for d in done)
cnt = sum(1 for d in done for x in d if lo < x <= hi)
assert all((d["status"] == 0).all() for d in descs)
per = np.mean([np.diff(d).mean() for d in done])
# sustained rate without the window's batch quantisation: slope of (completions so far) over time, middle 60 % of the run
ts = np.sort(np.concatenate([np.array(d) for d in done])) - t0
k0, k1 = int(len(ts) * 0.2), int(len(ts) * 0.8)
slope = np.polyfit(ts[k0:k1], np.arange(k0, k1), 1)[0] if k1 - k0 >= 4 else 0.0
print("T=%d B=%d n=%d stagger=%4.0f ms: whole run %.2f GiB/s | steady window %.2f s, %d batches -> %.2f GiB/s | batch period %.0f ms | slope of the middle 60 %% %.2f GiB/s"
      % (T, B, n, stagger * 1e3, T * B * gib / el, hi - lo, cnt, cnt * gib / (hi - lo) if hi > lo else 0, per * 1e3, slope * gib), flush=True)
