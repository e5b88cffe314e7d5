"""TEST HARNESS: run under LD_PRELOAD=libasan by tests/test_emu_asan.py.  Damaged and intact frames through the
AddressSanitizer build of the emulated kernels: any out-of-bounds access of the decoder aborts this process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import tsxform  # noqa: E402
from oracle import oracle as o  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tests.test_emu_zstd import CASES, _fuzzed_frames  # noqa: E402

nat = tsxform._native
os.environ["TSX_ALLOW_ANY_ARCH"] = "1"
o.build()
N = nat.Native(sys.argv[1])
N.init()
n_variants, seed = int(sys.argv[2]), int(sys.argv[3])
blobs, sizes = _fuzzed_frames(o, n_variants, seed)
outs, d = pc.run_detransform(N, nat.COMPRESS, blobs, sizes)
assert set(int(x) for x in d["status"]) <= {0, nat.E_BAD_FRAME, nat.E_BAD_SIZE, nat.E_DST_TOO_SMALL}
names = ["mixKR", "K70000", "zeros", "period7", "lowent", "R50000"]           # long runs, self-overlapping matches, raw blocks
good = [o.zstd_compress_chunk(CASES[k].tobytes(), lvl) for k in names for lvl in (1, 19)]
want = [CASES[k].tobytes() for k in names for _ in (1, 19)]
outs, d = pc.run_detransform(N, nat.COMPRESS, good, [len(w) for w in want])
assert (d["status"] == 0).all() and outs == want
print("asan decode check ok: %d damaged, %d intact frames" % (len(blobs), len(good)))
