#!/usr/bin/env python3
"""Differential fuzzing of the serial restatement (oracle/zstd_l3.c, profile 1.5.7) against the real libzstd 1.5.7 on structured
random inputs up to full 4 MiB chunks (window sliding, far repcodes, block splitting).  CPU only, fast (both sides are C):
    python tools/fuzz_oracle.py --seconds 600 --seed 1"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fuzz_cases import gen_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from oracle import oracle as o
    o.build()
    assert o.zstd_version().startswith("1.5.7"), o.zstd_version()
    rng = np.random.default_rng(args.seed)
    log = open(args.out, "a") if args.out else sys.stdout
    t0 = time.time(); n = 0; nb = 0; bad = 0
    while time.time() - t0 < args.seconds:
        big = rng.integers(0, 3) != 0
        c = gen_case(rng, int(rng.integers(2 << 20, (4 << 20) + 1)) if big else None)
        b = c.tobytes()
        if o.zstd_l3_compress(b, 1) != o.zstd_compress_chunk(b):
            bad += 1
            path = "/tmp/fuzz_oracle_bad_%d_%d.bin" % (args.seed, n)
            c.tofile(path)
            print("MISMATCH seed %d case %d size %d -> %s" % (args.seed, n, c.size, path), file=log, flush=True)
        n += 1; nb += c.size
        if n % 50 == 0:
            print("[%5.0fs] seed %d: %d cases, %.0f MB, %d bad" % (time.time() - t0, args.seed, n, nb / 1e6, bad), file=log, flush=True)
    print("DONE seed %d: %d cases, %.0f MB, %d bad" % (args.seed, n, nb / 1e6, bad), file=log, flush=True)


if __name__ == "__main__":
    main()
