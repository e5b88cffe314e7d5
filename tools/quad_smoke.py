#!/usr/bin/env python3
"""Quick device check of the four-chunks-per-wave compressor kernel (csrc/zstd_match4.h) before anything long runs on it: a handful of
chunks of awkward sizes (0, 5, one block, block boundary, 4 MiB, ragged group of 4) through the full chain with TSX_ZSTD_QUAD=1, byte for
byte against the oracle chain (libzstd + OpenSSL), and the same batch through the one-chunk kernel.  Exit code 0 = identical."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime)
import tsxform  # noqa: E402
from oracle import oracle as o  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tsxform import synth  # noqa: E402

nat = tsxform._native
o.build()
N = tsxform.get()
f = N.lib.tsx_debug_quad_launches; f.restype = ctypes.c_ulonglong
sizes = [synth.CHUNK, 300000, 0, 131072, 5, 131073, 70001, synth.CHUNK, 262144, 1 << 20, 17]
chunks = [synth.gen_chunk("K" if i % 3 else "R", 11, 0, i, s) for i, s in enumerate(sizes)]
chunks[7] = synth.gen_chunk("K", 1000, 0, 7)
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
os.environ["TSX_ZSTD_QUAD"] = "1"
before = f()
outs, d = pc.run_transform(N, flags, chunks)
assert f() > before, "the quad kernel did not run"
assert (d["status"] == 0).all(), d["status"]
os.environ["TSX_ZSTD_QUAD"] = "0"
outs1, d1 = pc.run_transform(N, flags, chunks)
assert outs == outs1, "quad and one-chunk kernels disagree"
for i, c in enumerate(chunks):
    assert outs[i] == pc.oracle_transform(o, flags, c, i), "chunk %d (%d bytes) differs from the oracle" % (i, c.size)
    assert d["crc32c"][i] == o.crc32c(c.tobytes())
print("quad smoke ok: %d chunks identical to libzstd %s + OpenSSL and to the one-chunk kernel" % (len(chunks), o.zstd_version()))
