#!/usr/bin/env python3
"""Phase laps of zb_decode_kernel (block-parallel decoder form) for a few Kafka-like chunks: libtsxform_prof2.so (`make -C csrc prof2`),
clock64() laps per (chunk, block).  Sequence wave: 0 tables, 1 window refills, 2 chain (pass 1), 3 fields (pass 2), 4 repeat offsets
(pass 3), 5 stores + sums; literal wave: 6 tree + table, 7 streams.  Prints the mean per block and the share of the wave's total."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tsxform  # noqa: E402
from tsxform import synth  # noqa: E402

nat = tsxform._native
N = nat.Native(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libs", "libtsxform_prof2.so")); N.init(1, [0])
dev = torch.device("cuda", 0)
CH = synth.CHUNK
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
for i in range(n):
    src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
mid = torch.empty(n * slot, dtype=torch.uint8, device=dev)
d = np.zeros(n, nat.DESC_DTYPE); d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH
d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
for i in range(n):
    d["iv"][i] = np.frombuffer(synth.iv_for(0, i), np.uint8)
ctx = N.ctx_create(0, n, CH)
N.transform_batch(params, d, src.data_ptr(), mid.data_ptr(), mid.numel(), nat.MEM_DEVICE, ctx=ctx)
assert (d["status"] == 0).all()
back = torch.empty(n * CH, dtype=torch.uint8, device=dev)
ZB_MAX_BLOCKS = 264
prof = torch.zeros(n * ZB_MAX_BLOCKS * 8, dtype=torch.int64, device=dev)
N.lib.tsx_debug_set_zbprof.restype = None; N.lib.tsx_debug_set_zbprof.argtypes = [C.c_void_p]
N.lib.tsx_debug_set_zbprof(prof.data_ptr())
e = np.zeros(n, nat.DESC_DTYPE); e["src_off"] = d["dst_off"]; e["src_len"] = d["dst_len"]
e["dst_off"] = np.arange(n, dtype=np.uint64) * CH; e["dst_cap"] = CH
for it in range(3):
    prof.zero_()
    N.detransform_batch(params, e, mid.data_ptr(), back.data_ptr(), back.numel(), nat.MEM_DEVICE, ctx=ctx)
torch.cuda.synchronize()
assert (e["status"] == 0).all() and torch.equal(back, src)
p = prof.cpu().numpy().reshape(n, ZB_MAX_BLOCKS, 8)
used = p[:, :, :6].sum(axis=2) > 0
m = p[used].mean(axis=0)
names = ["seq: tables", "seq: window refills", "seq: chain (pass 1)", "seq: fields (pass 2)", "seq: repeat offsets (pass 3)", "seq: stores + sums",
         "lit: tree + table", "lit: streams"]
seq_tot, lit_tot = m[:6].sum(), m[6:].sum()
print("blocks with sequences: %d of %d chunks; ctx timing unzstd %.3f ms" % (used.sum(), n, N.ctx_timing(ctx).unzstd_ms))
for k in range(8):
    print("  %-32s %12.0f ticks  (%5.1f %% of its wave)" % (names[k], m[k], 100.0 * m[k] / (seq_tot if k < 6 else lit_tot)))
print("  sequence wave total %.0f ticks, literal wave total %.0f ticks" % (seq_tot, lit_tot))
