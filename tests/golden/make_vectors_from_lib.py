#!/usr/bin/env python3
"""Frame digests of ANY libzstd, driven exactly as the reference drives zstd-jni (CompressionChunkEnumeration.java:50-63: fresh context per
chunk, setPledgedSrcSize, contentSizeFlag, default level, one-shot compress): the fixture that settles what this image cannot - whether the
compressor's profile `1_5_6` really is the library the reference ships (zstd-jni 1.5.6-9, core/build.gradle:29; no libzstd 1.5.6 exists here
or on the GPU box).

    python tests/golden/make_vectors_from_lib.py --lib /path/to/libzstd.so[.1.5.6] [--out tests/golden/zstd_lib_frames.json]

(zstd-jni's jar holds the library as linux/amd64/libzstd-jni-1.5.6-9.so: it exports the ZSTD_* symbols this script needs.)  Needs python,
numpy and this repository only - no oracle build, no GPU.  The fixture holds digests of seeded synthetic chunks and of their frames; then

    TSX_ZSTD_LIB_VECTORS=tests/golden/zstd_lib_frames.json python -m pytest tests/test_golden.py -k supplied_library            (CPU: the restatement)
    TSX_ZSTD_LIB_VECTORS=tests/golden/zstd_lib_frames.json python -m pytest tests/test_golden.py -k supplied_library -m gpu     (the HIP compressor)

compare the serial restatement (oracle/zstd_l3.c) and the HIP compressor with it, under the profile the library's version selects
(< 1.5.7: TSX_ZSTD_PROFILE_1_5_6, else TSX_ZSTD_PROFILE_1_5_7).  Test infrastructure."""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tsxform  # noqa: E402,F401
from tsxform import synth  # noqa: E402

# (name, how to make the chunk): full-size chunks of the three contents, the sizes around the block / window edges, a mixed chunk on which
# the 1.5.7 pre-splitter cuts, the reference's own golden input (ChunkIndexSerializationTest.java:39-61)
CASES = [
    ("golden15", lambda: np.frombuffer(bytes.fromhex("000000030000000A01000A0000001E"), np.uint8)),
    ("K_4MiB", lambda: synth.gen_chunk("K", 1000, 0, 0, 4 << 20)),
    ("K_4MiB_b", lambda: synth.gen_chunk("K", 1001, 1, 7, 4 << 20)),
    ("B_4MiB", lambda: synth.gen_chunk("B", 1000, 0, 0, 4 << 20)),
    ("B_320000", lambda: synth.gen_chunk("B", 41, 2, 1, 320000)),
    ("R_4MiB", lambda: synth.gen_chunk("R", 1000, 0, 0, 4 << 20)),
    ("K_200000", lambda: synth.gen_chunk("K", 5, 0, 0, 200000)),
    ("K_131073", lambda: synth.gen_chunk("K", 5, 0, 1, 131073)),
    ("K_131072", lambda: synth.gen_chunk("K", 5, 0, 2, 131072)),
    ("K_1000", lambda: synth.gen_chunk("K", 5, 0, 3, 1000)),
    ("mix_K_R_K", lambda: np.concatenate([synth.gen_chunk("K", 5, 0, 0, 300000), synth.gen_chunk("R", 5, 0, 0, 200000), synth.gen_chunk("K", 5, 0, 1, 300000)])),
    ("zeros_300000", lambda: np.zeros(300000, np.uint8)),
]


class Lib:
    def __init__(self, path):
        z = C.CDLL(path)
        z.ZSTD_versionString.restype = C.c_char_p
        z.ZSTD_createCCtx.restype = C.c_void_p
        z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]; z.ZSTD_CCtx_setParameter.restype = C.c_size_t
        z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]; z.ZSTD_compress2.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]; z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_isError.argtypes = [C.c_size_t]; z.ZSTD_isError.restype = C.c_uint
        self.pledge = getattr(z, "ZSTD_CCtx_setPledgedSrcSize", None)     # (the reference tolerates its absence: NoSuchMethodError, :55-58)
        if self.pledge is not None:
            self.pledge.argtypes = [C.c_void_p, C.c_ulonglong]; self.pledge.restype = C.c_size_t
        self.z = z
        self.version = z.ZSTD_versionString().decode()

    def compress_chunk(self, data: bytes) -> bytes:
        z = self.z
        ctx = z.ZSTD_createCCtx()
        try:
            if self.pledge is not None:
                self.pledge(ctx, len(data))
            z.ZSTD_CCtx_setParameter(ctx, 200, 1)                        # ZSTD_c_contentSizeFlag; the level stays the default (3)
            cap = z.ZSTD_compressBound(len(data))
            dst = C.create_string_buffer(cap)
            r = z.ZSTD_compress2(ctx, dst, cap, data, len(data))
            if z.ZSTD_isError(r):
                raise RuntimeError("ZSTD_compress2 failed")
            return dst.raw[:r]
        finally:
            z.ZSTD_freeCCtx(ctx)


def make(lib_path, names=None):
    lib = Lib(lib_path)
    frames = []
    for name, gen in CASES:
        if names and name not in names:
            continue
        c = gen().tobytes()
        f = lib.compress_chunk(c)
        frames.append({"name": name, "n": len(c), "input_sha256": hashlib.sha256(c).hexdigest(), "frame_len": len(f),
                       "frame_sha256": hashlib.sha256(f).hexdigest(), "frame_head_hex": f[:16].hex()})
    return {"lib_version": lib.version, "lib_file": os.path.basename(lib_path),
            "source": "tests/golden/make_vectors_from_lib.py: the library driven as CompressionChunkEnumeration.java:50-63 drives zstd-jni", "frames": frames}


def profile_of(version: str) -> int:
    """TSX_ZSTD_PROFILE_* a library version selects: the pre-block splitter arrived in 1.5.7."""
    v = tuple(int(x) for x in version.split(".")[:3])
    return 1 if v >= (1, 5, 7) else 0


def case_input(name):
    return dict(CASES)[name]()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_lib_frames.json"))
    a = ap.parse_args()
    v = make(a.lib)
    json.dump(v, open(a.out, "w"), indent=1)
    print("libzstd %s: %d frames -> %s (profile %s)" % (v["lib_version"], len(v["frames"]), a.out, ("1_5_6", "1_5_7")[profile_of(v["lib_version"])]))
