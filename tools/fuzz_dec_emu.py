#!/usr/bin/env python3
"""Differential fuzzing of the HIP Zstd frame decoder under the CPU emulator: structured random inputs are compressed by
the real libzstd at several levels (different block / literal / sequence-table modes, long offsets, repeat codes) and the
emulated zstd_decompress_kernel must restore every one.  Test infrastructure only (uses oracle/ and tests/emu).
    python tools/fuzz_dec_emu.py --seconds 600 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tsxform  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tests.fuzz_cases import gen_case  # noqa: E402

nat = tsxform._native
LEVELS = (1, 2, 3, 5, 9, 15, 19, 22)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    from oracle import oracle as o
    o.build()
    from tests.emu import emu_native
    emu = emu_native.get()
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); n_cases = 0; n_bytes = 0; bad = 0
    while time.time() - t0 < args.seconds:
        cases = [gen_case(rng) for _ in range(args.batch)]
        levels = [LEVELS[int(rng.integers(0, len(LEVELS)))] for _ in cases]
        blobs = [o.zstd_compress_chunk(c.tobytes(), lv) for c, lv in zip(cases, levels)]
        back, d = pc.run_detransform(emu, nat.COMPRESS, blobs, [int(c.size) for c in cases])
        for i, c in enumerate(cases):
            if d["status"][i] != 0 or back[i] != c.tobytes():
                bad += 1
                path = "/tmp/fuzz_dec_bad_%d_%d.bin" % (args.seed, n_cases + i)
                c.tofile(path)
                print("DECODE MISMATCH seed %d case %d size %d level %d status %d -> %s" % (args.seed, n_cases + i, c.size, levels[i], d["status"][i], path), flush=True)
        n_cases += len(cases); n_bytes += sum(int(c.size) for c in cases)
        if (n_cases // args.batch) % 10 == 0:
            print("[%6.0fs] seed %d: %d cases, %.1f MB, %d bad" % (time.time() - t0, args.seed, n_cases, n_bytes / 1e6, bad), flush=True)
    print("DONE seed %d: %d cases, %.1f MB, %d bad" % (args.seed, n_cases, n_bytes / 1e6, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
