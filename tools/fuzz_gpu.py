#!/usr/bin/env python3
"""Differential campaign on the device: the product library against the real libzstd 1.5.7 (oracle) on structured random inputs
(tests/fuzz_cases.py) - many small/medium cases plus full 4 MiB chunks whose copies reach beyond the 2 MiB window.  Test
infrastructure (uses oracle/).   python tools/fuzz_gpu.py --cases 3000 --big 96 --seed 7 > gpurun_out/fuzz_gpu.txt"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime)
import tsxform  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tests.fuzz_cases import gen_case  # noqa: E402

nat = tsxform._native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--big", type=int, default=64)
    ap.add_argument("--seed", type=int, default=7)
    args = ap.parse_args()
    from oracle import oracle as o
    o.build()
    assert o.zstd_version().startswith("1.5.7"), o.zstd_version()
    N = tsxform.get()
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); bad = 0; done = 0; nbytes = 0

    def run(cases, label):
        nonlocal bad, done, nbytes
        outs, d = pc.run_transform(N, nat.COMPRESS, cases, mem="device")
        back, d2 = pc.run_detransform(N, nat.COMPRESS, outs, [int(c.size) for c in cases])
        for i, c in enumerate(cases):
            exp = o.zstd_compress_chunk(c.tobytes())
            if d["status"][i] != 0 or outs[i] != exp:
                bad += 1; print("MISMATCH %s case %d size %d status %d" % (label, done + i, c.size, d["status"][i]), flush=True)
                c.tofile(os.path.join(ROOT, "gpurun_out", "fuzz_gpu_bad_%s_%d.bin" % (label, done + i)))
            if d2["status"][i] != 0 or back[i] != c.tobytes():
                bad += 1; print("DECODE MISMATCH %s case %d size %d status %d" % (label, done + i, c.size, d2["status"][i]), flush=True)
        done += len(cases); nbytes += sum(int(c.size) for c in cases)

    for lo in range(0, args.cases, 256):
        run([gen_case(rng) for _ in range(min(256, args.cases - lo))], "small")
    print("[%5.0fs] %d cases, %.1f MB, %d bad" % (time.time() - t0, done, nbytes / 1e6, bad), flush=True)
    for lo in range(0, args.big, 32):
        run([gen_case(rng, 4194304 - int(rng.integers(0, 3)) * int(rng.integers(0, 70000))) for _ in range(min(32, args.big - lo))], "big")
    # the chain on a sample (CRC head + GCM tail in the compressor wave)
    sample = [gen_case(rng) for _ in range(96)] + [gen_case(rng, 4194304) for _ in range(4)]
    try:
        pc.check_transform_vs_oracle(N, o, nat.COMPRESS | nat.ENCRYPT | nat.CRC, sample)
        pc.check_roundtrip(N, nat.COMPRESS | nat.ENCRYPT | nat.CRC, sample)
    except AssertionError as e:
        bad += 1; print("CHAIN MISMATCH:", e, flush=True)
    print("DONE seed %d: %d cases (%d of 4 MiB), %.1f MB, %d bad, %.0f s; libzstd %s" % (args.seed, done, args.big, nbytes / 1e6, bad, time.time() - t0, o.zstd_version()), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
