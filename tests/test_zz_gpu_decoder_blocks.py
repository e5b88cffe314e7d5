"""Device run of the decoder's three-stage block pipeline on frames the emulated tests only see in small: full 4 MiB
chunks written by stock libzstd at levels 1, 3 and 19 (window log up to 22: offsets across the whole chunk, repeat-mode
tables, treeless literals) and frames that mix raw, RLE and compressed blocks.  (Named to run after the other GPU files.)"""
import numpy as np
import pytest

import tsxform
from tests import parity_cases as pc
from tsxform import synth

nat = tsxform._native
pytestmark = pytest.mark.gpu


def _inputs():
    K = synth.gen_chunk("K", 9, 1, 3); R = synth.gen_chunk("R", 9, 1, 4, 600000)
    mixed = np.concatenate([K[:1500000], R[:300000], np.zeros(400000, np.uint8), K[1500000:2600000], np.full(200000, 7, np.uint8),
                            R[300000:330000], K[2600000:3600000], np.tile(np.frombuffer(b"0123456789abcdef", np.uint8), 4000)])
    far = np.concatenate([R[:200000], K[:1800000], R[:200000], K[:1800000]])          # matches 2 MB back
    return {"K4M": K, "mixed": mixed[:synth.CHUNK], "far": far[:synth.CHUNK], "one block": K[:100000], "two blocks": K[:200000],
            "three blocks": K[:380000]}


@pytest.mark.parametrize("level", [1, 3, 19])
def test_full_size_and_mixed_block_frames_of_stock_libzstd(gpu, oracle, level):
    inputs = _inputs()
    blobs = [oracle.zstd_compress_chunk(v.tobytes(), level) for v in inputs.values()]
    outs, d = pc.run_detransform(gpu, nat.COMPRESS, blobs, [int(v.size) for v in inputs.values()])
    for i, (name, v) in enumerate(inputs.items()):
        assert d["status"][i] == 0 and outs[i] == v.tobytes(), (name, level)


def test_packed_host_output_on_the_device(gpu, oracle):
    """TSX_MEM_HOST_PACKED (the `.log` object assembled in the caller's buffer): same bytes as the slot-per-chunk layout and as
    the oracle chain, offsets = running sum of the sizes; full-size chunks included."""
    chunks = [synth.gen_chunk("K", 3, 0, i, n) for i, n in enumerate((70000, 1, synth.CHUNK, 0, 33333, synth.CHUNK, 65537))]
    for flags in (nat.ENCRYPT | nat.CRC, nat.COMPRESS | nat.ENCRYPT | nat.CRC):
        slots, d0 = pc.run_transform(gpu, flags, chunks)
        packed, d1 = pc.run_transform(gpu, flags, chunks, mem="packed")
        assert packed == slots and (d1["status"] == 0).all() and (d1["crc32c"] == d0["crc32c"]).all()
        ends = np.cumsum(d1["dst_len"].astype(np.int64))
        assert (d1["dst_off"].astype(np.int64) == ends - d1["dst_len"]).all()
        for i in (0, 2, 6):
            assert packed[i] == pc.oracle_transform(oracle, flags, chunks[i], i)
