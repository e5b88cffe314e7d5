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
    ap.add_argument("--guests", action="store_true", help="fuzz the guest waves' hand-back instead: every launch's first workgroup sits on the reserved CU and is "
                                                          "made to give its chunk back at a random block; the full chain must still equal the oracle's")
    args = ap.parse_args()
    from oracle import oracle as o
    o.build()
    from tests.emu import emu_native
    emu = emu_native.get()
    assert o.zstd_version().startswith("1.5.7"), o.zstd_version()
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); n_cases = 0; n_bytes = 0; bad = 0
    log = open(args.out, "a") if args.out else sys.stdout
    if args.guests:
        return fuzz_guests(args, emu, o, rng, log)
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


def fuzz_guests(args, emu, o, rng, log):
    """tsx_config.fetch_quiet_ms: a guest wave looks at the host's yield word before every block of its chunk and hands the chunk back when it is
    raised.  The harness raises it at the k-th look of a launch, k random: the guest (block 0 of the launch: hipemu_cu_key_shift) abandons a
    chunk anywhere from its first block to its last, with half-built hash tables, frame and entropy state in the chunk's workspace, and another
    wave starts it again.  Full chain, slot and packed layout: bytes = the oracle's (libzstd 1.5.7 + OpenSSL), every chunk counted exactly once."""
    import ctypes
    for f in ("hipemu_cu_key_shift", "hipemu_force_yield_after"):
        getattr(emu.lib, f).argtypes = [ctypes.c_int]; getattr(emu.lib, f).restype = None
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    t0 = time.time(); n_cases = 0; n_bytes = 0; bad = 0; handed = 0
    emu.debug_config("fetch_quiet_ms", 1)
    emu.lib.hipemu_cu_key_shift(3)
    try:
        while time.time() - t0 < args.seconds:
            cases = [gen_case(rng, total=int(rng.integers(1, 700000))) for _ in range(int(rng.integers(1, args.batch + 1)))]
            exp = [pc.oracle_transform(o, flags, c, i) for i, c in enumerate(cases)]
            time.sleep(0.003)                                           # quiet again: the next launch has guests
            s0 = emu.service_stats(0)
            after = int(rng.integers(1, 16))
            emu.lib.hipemu_force_yield_after(after)
            got, d = pc.run_transform(emu, flags, cases, mem="packed" if rng.integers(0, 2) else None, profile=nat.ZSTD_PROFILE_1_5_7)
            emu.lib.hipemu_force_yield_after(0)
            emu.service_quiesce(0)
            s1 = emu.service_stats(0)
            handed += s1["returned_chunks"] - s0["returned_chunks"]
            ok = (d["status"] == 0).all() and got == exp and s1["device_chunks"] - s0["device_chunks"] == len(cases) and s1["skipped_tickets"] == s0["skipped_tickets"]
            if not ok:
                bad += 1
                print("GUEST MISMATCH seed %d case %d after=%d sizes %s statuses %s chunks counted %d" % (args.seed, n_cases, after, [int(c.size) for c in cases], list(d["status"]), s1["device_chunks"] - s0["device_chunks"]), file=log, flush=True)
                for i, c in enumerate(cases):
                    c.tofile("/tmp/fuzz_guest_bad_%d_%d_%d.bin" % (args.seed, n_cases, i))
            back, d2 = pc.run_detransform(emu, flags, got, [int(c.size) for c in cases])     # (a fetch: the next launch's guests find the word raised until it is quiet again)
            if (d2["status"] != 0).any() or back != [c.tobytes() for c in cases]:
                bad += 1
                print("GUEST ROUND TRIP MISMATCH seed %d case %d" % (args.seed, n_cases), file=log, flush=True)
            n_cases += len(cases); n_bytes += sum(int(c.size) for c in cases)
            if n_cases % 20 < len(cases):
                print("[%6.0fs] seed %d: %d chunks, %.1f MB, %d handed back and started again, %d bad" % (time.time() - t0, args.seed, n_cases, n_bytes / 1e6, handed, bad), file=log, flush=True)
    finally:
        emu.lib.hipemu_cu_key_shift(0); emu.lib.hipemu_force_yield_after(0); emu.debug_config("fetch_quiet_ms", 0)
    print("DONE guests seed %d: %d chunks, %.1f MB, %d handed back and started again, %d bad" % (args.seed, n_cases, n_bytes / 1e6, handed, bad), file=log, flush=True)


if __name__ == "__main__":
    main()
