"""TEST HARNESS: run under LD_PRELOAD=libasan by tests/test_emu_asan.py.  The compressor (and the fused CRC head / GCM tail) through
the AddressSanitizer build of the emulated kernels on structured random inputs and the edge sizes: any access outside the
workspace, the LDS images or the caller's slots aborts this process; every frame must also equal libzstd's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import tsxform  # noqa: E402
from oracle import oracle as o  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tests.fuzz_cases import gen_case  # noqa: E402

nat = tsxform._native
os.environ["TSX_ALLOW_ANY_ARCH"] = "1"
o.build()
N = nat.Native(sys.argv[1])
N.init()
n_cases, seed = int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
cases = [gen_case(rng) for _ in range(n_cases)] + pc.edge_chunks("K", [0, 1, 7, 8, 63, 64, 65, 4095, 4096, 4097, 16384, 16385, 70001, 131072, 131073])
real = o.zstd_version().startswith("1.5.7")
for flags, prof in ((nat.COMPRESS, nat.ZSTD_PROFILE_1_5_7), (nat.COMPRESS | nat.ENCRYPT | nat.CRC, nat.ZSTD_PROFILE_1_5_7), (nat.COMPRESS, nat.ZSTD_PROFILE_1_5_6)):
    outs, d = pc.run_transform(N, flags, cases, profile=prof)
    assert (d["status"] == 0).all()
    if flags == nat.COMPRESS:
        for c, got in zip(cases, outs):
            raw = c.tobytes()
            assert got == o.zstd_l3_compress(raw, 1 if prof == nat.ZSTD_PROFILE_1_5_7 else 0)
            if real and prof == nat.ZSTD_PROFILE_1_5_7:
                assert got == o.zstd_compress_chunk(raw)
    else:
        back, d2 = pc.run_detransform(N, flags, outs, [int(c.size) for c in cases])
        assert (d2["status"] == 0).all() and all(b == c.tobytes() for b, c in zip(back, cases)) and (d2["crc32c"] == d["crc32c"]).all()
print("asan compress check ok: %d inputs x 3 configurations, %.1f MB" % (len(cases), sum(c.size for c in cases) / 1e6))
