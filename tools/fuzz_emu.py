#!/usr/bin/env python3
"""Differential fuzzing of the HIP Zstd compressor/decoder under the CPU emulator against the real libzstd (oracle).

Every case: structured random input -> emulated zstd_compress_kernel (1.5.7 profile) must give libzstd 1.5.7's frame byte for
byte, the emulated decoder must restore the input, and the full chain (compress + GCM + CRC, both fused and separate
launches) must match the oracle chain.  Test infrastructure only (uses oracle/ and tests/emu).
    python tools/fuzz_emu.py --seconds 3000 --seed 1 --out /tmp/fuzz1.log
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from oracle import oracle as o
    o.build()
    from tests.emu import emu_native
    emu = emu_native.get()
    assert o.zstd_version().startswith("1.5.7"), o.zstd_version()
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); n_cases = 0; n_bytes = 0; bad = 0
    log = open(args.out, "a") if args.out else sys.stdout
    while time.time() - t0 < args.seconds:
        cases = [gen_case(rng) for _ in range(args.batch)]
        outs, d = pc.run_transform(emu, nat.COMPRESS, cases)
        for i, c in enumerate(cases):
            exp = o.zstd_compress_chunk(c.tobytes())
            if d["status"][i] != 0 or outs[i] != exp:
                bad += 1
                path = "/tmp/fuzz_bad_%d_%d.bin" % (args.seed, n_cases + i)
                c.tofile(path)
                print("MISMATCH seed %d case %d size %d status %d -> %s" % (args.seed, n_cases + i, c.size, d["status"][i], path), file=log, flush=True)
        back, d2 = pc.run_detransform(emu, nat.COMPRESS, outs, [int(c.size) for c in cases])
        for i, c in enumerate(cases):
            if d2["status"][i] != 0 or back[i] != c.tobytes():
                bad += 1
                print("DECODE MISMATCH seed %d case %d size %d status %d" % (args.seed, n_cases + i, c.size, d2["status"][i]), file=log, flush=True)
        if (n_cases // args.batch) % 4 == 0:                       # the chain, fused and with one launch per stage
            for sep in (0, 1):
                try:
                    with emu.configured(stages_separate=sep):
                        pc.check_transform_vs_oracle(emu, o, nat.COMPRESS | nat.ENCRYPT | nat.CRC, cases[:3])
                except AssertionError as e:
                    bad += 1
                    print("CHAIN MISMATCH seed %d case %d sep=%r: %s" % (args.seed, n_cases, sep, e), file=log, flush=True)
        n_cases += len(cases); n_bytes += sum(int(c.size) for c in cases)
        if (n_cases // args.batch) % 10 == 0:
            print("[%6.0fs] seed %d: %d cases, %.1f MB, %d bad" % (time.time() - t0, args.seed, n_cases, n_bytes / 1e6, bad), file=log, flush=True)
    print("DONE seed %d: %d cases, %.1f MB, %d bad" % (args.seed, n_cases, n_bytes / 1e6, bad), file=log, flush=True)


if __name__ == "__main__":
    main()
