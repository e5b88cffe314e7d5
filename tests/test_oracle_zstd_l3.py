"""Pins oracle/zstd_l3.c (the readable restatement of libzstd's level-3 one-shot compressor) against the REAL
library (libzstd 1.5.7, dlopen'd by oracle/zstd_ref.c), byte for byte.  The HIP compressor is then checked
against both (tests/test_emu_zstd.py, tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

from tsxform import synth


def _lib157(oracle):
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available (have %s)" % oracle.zstd_version())


def test_cparams_table_matches_ZSTD_getCParams(oracle):
    _lib157(oracle)

    class CP(C.Structure):
        _fields_ = [(n, C.c_uint) for n in "windowLog chainLog hashLog searchLog minMatch targetLength strategy".split()]
    lz = C.CDLL(oracle.lib().orc_zstd_path().decode())
    f = lz.ZSTD_getCParams; f.restype = CP; f.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
    sizes = list(range(1, 70)) + [2 ** k + d for k in range(6, 31) for d in (-1, 0, 1)] + [100000, 300000, 3000000, 4194304]
    for n in sizes:
        c = f(3, n, 0)
        assert oracle.zstd_l3_cparams(n) == (c.windowLog, c.chainLog, c.hashLog, c.searchLog, c.minMatch, c.targetLength, c.strategy), n


def _cases():
    rng = np.random.default_rng(42)
    K = synth.gen_chunk("K", 5, 0, 0, 1 << 20); R = synth.gen_chunk("R", 5, 0, 0, 1 << 19)
    yield "golden15", np.frombuffer(bytes.fromhex("000000030000000A01000A0000001E"), np.uint8)
    for n in [0, 1, 6, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1000, 4096, 10000, 16384, 16385, 65537, 131072, 131073, 262145, 300000]:
        yield "K%d" % n, K[:n]
    yield "R70000", R[:70000]
    yield "zeros", np.zeros(300000, np.uint8)
    yield "period7", np.tile(np.frombuffer(b"abcdefg", np.uint8), 40000)
    yield "mixKR", np.concatenate([K[:200000], R[:150000], K[200000:500000], np.zeros(70000, np.uint8), R[:50000], K[:300000]])
    yield "lowent", rng.integers(0, 4, 300000, dtype=np.uint8)
    yield "skewed", np.minimum(rng.geometric(0.3, 300000), 255).astype(np.uint8)
    yield "ramp", (np.arange(400000) % 256).astype(np.uint8)
    yield "far_repeat", np.concatenate([R[:100000], K[:1100000], R[:100000], K[:600000], R[:100000]])


@pytest.mark.parametrize("name,data", list(_cases()), ids=[n for n, _ in _cases()])
def test_restatement_is_byte_identical_to_libzstd_1_5_7(oracle, name, data):
    _lib157(oracle)
    d = data.tobytes()
    assert oracle.zstd_l3_compress(d, 1) == oracle.zstd_compress_chunk(d)


def test_full_chunk_and_profiles(oracle):
    _lib157(oracle)
    d = synth.gen_chunk("K", 1000, 0, 0).tobytes()
    f = oracle.zstd_l3_compress(d, 1)
    assert f == oracle.zstd_compress_chunk(d) and f[:10].hex() == "28b52ffd805800004000"
    # profile 0 (1.5.6 block loop: no pre-splitter) must still be a valid frame of the same content
    mix = np.concatenate([synth.gen_chunk("K", 5, 0, 0, 300000), synth.gen_chunk("R", 5, 0, 0, 200000), synth.gen_chunk("K", 5, 0, 1, 300000)]).tobytes()
    f0, f1 = oracle.zstd_l3_compress(mix, 0), oracle.zstd_l3_compress(mix, 1)
    assert oracle.zstd_decompress_chunk(f0) == mix and oracle.zstd_decompress_chunk(f1) == mix
    assert f1 == oracle.zstd_compress_chunk(mix)


def test_window_sliding_and_structured_random_chunks(oracle):
    """Full-size chunks where the 2 MiB window slides: since 1.5.0 libzstd slides it to the block's START, bounds match candidates
    from the block's END and accepts a candidate AT that bound - each pinned by tests/fuzz_cases.py's window-edge chunks - plus a
    sample of structured random inputs of all sizes (tools/fuzz_oracle.py runs the same comparison for as long as one likes)."""
    _lib157(oracle)
    from tests.fuzz_cases import gen_case, window_edge_cases
    rng = np.random.default_rng(31337)
    cases = window_edge_cases() + [gen_case(rng, int(rng.integers(2 << 20, (4 << 20) + 1))) for _ in range(24)] + [gen_case(rng) for _ in range(150)]
    for i, c in enumerate(cases):
        b = c.tobytes()
        assert oracle.zstd_l3_compress(b, 1) == oracle.zstd_compress_chunk(b), "case %d (n=%d)" % (i, c.size)
