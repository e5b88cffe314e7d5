#!/usr/bin/env python3
"""Regenerates the libzstd-derived part of vectors.json (needs libzstd 1.5.7 through the oracle; run in this container)."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tsxform  # noqa: F401
from tsxform import synth
from oracle import oracle as o
o.build()
assert o.zstd_version().startswith("1.5.7"), o.zstd_version()
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
v = json.load(open(p))
frames = []
for dist, seed, seg, chunk, n in [("K", 1000, 0, 0, 4194304), ("K", 1001, 1, 7, 4194304), ("R", 1000, 0, 0, 4194304), ("K", 5, 0, 0, 200000), ("K", 5, 0, 1, 131073)]:
    c = synth.gen_chunk(dist, seed, seg, chunk, n)
    f = o.zstd_compress_chunk(c.tobytes())
    frames.append({"dist": dist, "seed": seed, "segment": seg, "chunk": chunk, "n": n, "input_sha256": hashlib.sha256(c.tobytes()).hexdigest(),
                   "frame_len": len(f), "frame_sha256": hashlib.sha256(f).hexdigest(), "frame_head_hex": f[:16].hex()})
v["zstd_1_5_7_level3_frames"] = {"source": "libzstd 1.5.7 (Pillow wheel) driven as CompressionChunkEnumeration.java:52-61 drives zstd-jni; tests/golden/make_vectors.py",
                                 "frames": frames}
json.dump(v, open(p, "w"), indent=1)
print("wrote", p)
