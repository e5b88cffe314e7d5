#!/usr/bin/env python3
"""Fetch-side latency of the inverse chain (GCM verify + decrypt, Zstd frame decode, CRC32C) for small batches: 1 .. 256 chunks of
4 MiB, device resident and host -> host (registered buffers), with the block-parallel decoder form (csrc/zstd_dec_blocks.hip, the
default for batches of <= 256 chunks) and with the chunk-serial form alone (test hook dec_block_chunks = 0).  One JSON line per row."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tsxform  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tsxform import synth  # noqa: E402

nat = tsxform._native
N = nat.Native(); N.init(1, [0])
sys.path.insert(0, os.path.join(ROOT, "tools"))
from numa_bind import bind_to_gpu_numa_node
AFFINITY = bind_to_gpu_numa_node(0)                              # before any host buffer is allocated (profiles/r04_broker_numa.txt)
dev = torch.device("cuda", 0)
CH = synth.CHUNK
NMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 256
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
params = nat.Native.make_params(flags, synth.KEY, synth.AAD)
src = torch.empty(NMAX * CH, dtype=torch.uint8, device=dev)
for i in range(NMAX):
    src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i % 256, CH, device=dev)
mid = torch.empty(NMAX * slot, dtype=torch.uint8, device=dev)
d = np.zeros(NMAX, nat.DESC_DTYPE); d["src_off"] = np.arange(NMAX, dtype=np.uint64) * CH; d["src_len"] = CH
d["dst_off"] = np.arange(NMAX, dtype=np.uint64) * slot; d["dst_cap"] = slot
for i in range(NMAX):
    d["iv"][i] = np.frombuffer(synth.iv_for(0, i), np.uint8)
ctx = N.ctx_create(0, NMAX, CH)
N.transform_batch(params, d, src.data_ptr(), mid.data_ptr(), mid.numel(), nat.MEM_DEVICE, ctx=ctx)
assert (d["status"] == 0).all()
back = torch.empty(NMAX * CH, dtype=torch.uint8, device=dev)
hmid = mid.cpu().numpy(); hback = np.zeros(NMAX * CH, np.uint8)
N.host_register(hmid); N.host_register(hback)
for form in ("blocks", "chunks"):
    N.debug_config("dec_block_chunks", 256 if form == "blocks" else 0)
    for n in [x for x in (1, 2, 4, 8, 16, 64, 256) if x <= NMAX]:
        e = np.zeros(n, nat.DESC_DTYPE); e["src_off"] = d["dst_off"][:n]; e["src_len"] = d["dst_len"][:n]
        e["dst_off"] = np.arange(n, dtype=np.uint64) * CH; e["dst_cap"] = CH
        for mem in ("device", "host"):
            ts = []
            for it in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if mem == "device":
                    N.detransform_batch(params, e, mid.data_ptr(), back.data_ptr(), back.numel(), nat.MEM_DEVICE, ctx=ctx)
                else:
                    N.detransform_batch(params, e, hmid, hback, hback.size, nat.MEM_HOST, ctx=ctx)
                ts.append(time.perf_counter() - t0)
            tm = N.ctx_timing(ctx)
            ok = bool((e["status"] == 0).all() and (e["crc32c"] == d["crc32c"][:n]).all())
            if mem == "device":
                ok = ok and bool(torch.equal(back[:n * CH], src[:n * CH]))
            else:
                ok = ok and bool(np.array_equal(hback[:n * CH], src[:n * CH].cpu().numpy()))
            taken = pc.blockmode_chunks(N, ctx, n)
            print(json.dumps({"form": form, "chunks": n, "mem": mem, "ms_median": round(float(np.median(ts[1:])) * 1e3, 3), "ms_min": round(min(ts[1:]) * 1e3, 3),
                              "gibs": round(n * CH / 2**30 / float(np.median(ts[1:])), 2), "unzstd_ms": round(tm.unzstd_ms, 3), "gcm_ms": round(tm.gcm_ms, 3),
                              "decoded_by_block_form": taken, "exact": ok}), flush=True)
