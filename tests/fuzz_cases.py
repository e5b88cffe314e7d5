"""Structured random inputs for differential tests of the Zstd kernels against the real libzstd (shared by
tools/fuzz_emu.py, the emulator tests and the GPU tests)."""
import numpy as np

from tsxform import synth


def gen_case(rng, total=None):
    """A byte string built from segments of different statistical character, with cross references at all distances."""
    if total is None:
        total = int(rng.choice([rng.integers(0, 300), rng.integers(300, 20000), rng.integers(20000, 140000), rng.integers(126000, 136000),
                                rng.integers(140000, 420000), rng.integers(255000, 270000)]))
    parts, made = [], 0
    pool = []
    while made < total:
        kind = rng.integers(0, 9)
        n = int(min(total - made, rng.choice([rng.integers(1, 40), rng.integers(40, 2000), rng.integers(2000, 60000)])))
        if kind == 0:
            seg = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:
            seg = rng.integers(0, int(rng.integers(2, 20)), n, dtype=np.uint8)
        elif kind == 2:
            seg = np.full(n, rng.integers(0, 256), np.uint8)
        elif kind == 3:
            p = rng.integers(0, 256, int(rng.integers(1, 70)), dtype=np.uint8)
            seg = np.tile(p, n // p.size + 1)[:n]
        elif kind == 4:
            seg = synth.gen_chunk("K", int(rng.integers(0, 1 << 30)), 0, 0, n)
        elif kind == 5 and pool:                                   # verbatim copy of an earlier part (match at its distance)
            src = pool[int(rng.integers(0, len(pool)))]
            o = int(rng.integers(0, max(1, src.size - 1)))
            seg = src[o:o + n].copy()
            n = seg.size
        elif kind == 6 and pool:                                   # earlier part with sparse byte edits (repcode-rich)
            src = pool[int(rng.integers(0, len(pool)))]
            o = int(rng.integers(0, max(1, src.size - 1)))
            seg = src[o:o + n].copy()
            n = seg.size
            if n:
                k = int(rng.integers(1, 2 + n // int(rng.integers(4, 200))))
                seg[rng.integers(0, n, k)] = rng.integers(0, 256, k, dtype=np.uint8)
        elif kind == 7:
            seg = np.minimum(rng.geometric(float(rng.uniform(0.05, 0.6)), n), 255).astype(np.uint8)
        else:
            seg = (np.arange(n) * int(rng.integers(1, 5)) % 256).astype(np.uint8)
        if n == 0:
            continue
        parts.append(seg); pool.append(seg); made += n
    return np.concatenate(parts)[:total] if parts else np.zeros(0, np.uint8)




# Full-size chunks on which the sliding of the 2 MiB window decides the output: repcodes / match candidates that lie between
# 2 MiB - blockSize and 2 MiB behind the block, one of them exactly AT the lowest valid index (found by tools/fuzz_gpu.py; a
# restatement that slides the window to the block's end, or excludes the bound itself, differs from libzstd 1.5.7 on every one).
WINDOW_EDGE_SEEDS = (1009, 1045, 1071, 1093, 1096, 1113)


def window_edge_case(seed):
    return gen_case(np.random.default_rng(seed), 4194304 - (seed % 3) * 40000)


def window_edge_cases():
    return [window_edge_case(s) for s in WINDOW_EDGE_SEEDS]
