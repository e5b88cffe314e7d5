"""The Kafka-shaped binary content "B" ("v2 record batches"): what it is, that it is what a reader of a log segment expects
(SegmentCompressionChecker.java:37-53 reads the first batch of the segment: magic, attributes, CRC), and the chain on it - emulated
kernels here, the device in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import tsxform
from tests import parity_cases as pc
from tsxform import synth

nat = tsxform._native


def _walk(chunk):
    """Every complete record batch of a chunk, record by record; returns (batches, records)."""
    cb = chunk.tobytes()
    nb = nr = 0
    for p, l in synth.record_batches_of(chunk):
        assert cb[p + 16] == 2 and int.from_bytes(cb[p + 21:p + 23], "big") & 7 == 0        # magic 2, no compression codec in the attributes
        nrec = int.from_bytes(cb[p + 57:p + 61], "big")
        assert int.from_bytes(cb[p + 23:p + 27], "big") == nrec - 1                          # lastOffsetDelta
        q = p + 61
        for r in range(nrec):
            assert cb[q] & 0x80
            L = ((cb[q] & 0x7F) | (cb[q + 1] << 7)) >> 1                                      # zigzag varint: record length
            body = q + 2
            assert cb[body] == 0 and cb[body + 1] == 2 * r and cb[body + 2] == 2 * r and cb[body + 3] == 16     # attributes, deltas, key length 8
            vl = ((cb[body + 12] & 0x7F) | (cb[body + 13] << 7)) >> 1
            assert L == vl + 15 and cb[body + 14 + vl] == 0                                   # value length, no headers
            q = body + L
        assert q == p + l
        nb += 1; nr += nrec
    return nb, nr


def test_b_chunks_are_valid_record_batches(oracle):
    for n in (70000, 300000):
        c = synth.gen_chunk("B", 7, 1, 2, n)
        assert c.dtype == np.uint8 and c.size == n and (c == synth.gen_chunk("B", 7, 1, 2, n)).all()
        nb, nr = _walk(c)
        assert nb >= 3 and nr >= 8 * nb
        cb = c.tobytes()
        for p, l in synth.record_batches_of(c):                          # the CRC32C of the header covers attributes .. end of the batch
            assert int.from_bytes(cb[p + 17:p + 21], "big") == oracle.crc32c(np.frombuffer(cb[p + 21:p + l], np.uint8))
    assert not (synth.gen_chunk("B", 7, 1, 2, 70000) == synth.gen_chunk("B", 7, 1, 3, 70000)).all()
    import torch
    assert (synth.gen_chunk("B", 9, 0, 0, 50000, device=torch.device("cpu")).numpy() == synth.gen_chunk("B", 9, 0, 0, 50000)).all()


def test_the_crc_kernel_validates_a_b_segments_first_batches(emu):
    """What SegmentCompressionChecker does with the first batch of a segment, for every batch of a chunk, through tsx_crc32c_batch."""
    c = synth.gen_chunk("B", 11, 0, 0, 250000)
    batches = synth.record_batches_of(c)
    buf = np.zeros(c.size + 64, np.uint8)
    sizes, offs = [], []
    at = 0
    for p, l in batches:                                                # chunk offsets must be 16-byte aligned: copy each batch's CRC range to one
        buf[at:at + l - 21] = c[p + 21:p + l]; offs.append(at); sizes.append(l - 21); at += (l - 21 + 15) // 16 * 16
        if at + 40000 > buf.size:
            buf = np.concatenate([buf, np.zeros(buf.size, np.uint8)])
    d = pc.make_descs(sizes, offs, [0] * len(sizes), [0] * len(sizes))
    emu.crc32c_batch(d, buf)
    cb = c.tobytes()
    assert [int(x) for x in d["crc32c"]] == [int.from_bytes(cb[p + 17:p + 21], "big") for p, _ in batches]


def test_chain_on_b_content_vs_oracle(emu, oracle):
    """Full chain and both Zstd profiles on B chunks: byte-identical to libzstd 1.5.7 + OpenSSL; the incompressible payloads inside make
    1.5.7's pre-splitter cut (profile 1.5.6 then differs, and both decode)."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    chunks = [synth.gen_chunk("B", 21, 0, i, s) for i, s in enumerate([60000, 150000, 320000])]
    pc.check_transform_vs_oracle(emu, oracle, nat.COMPRESS | nat.ENCRYPT | nat.CRC, chunks)
    pc.check_roundtrip(emu, nat.COMPRESS | nat.ENCRYPT | nat.CRC, chunks)
    pinned, differ = pc.check_profile_1_5_6(emu, oracle, {"B%d" % c.size: c for c in chunks})
    assert pinned + differ == 3


def test_b_chunks_are_the_committed_ones():
    """tests/golden/synth_b.json (made by tests/golden/make_synth_b.py): the generator's chunks are pinned by hash - bench.py's value_B and
    the device parity test stand on them."""
    import hashlib
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth_b.json")))
    assert len(cases) >= 4
    for c in cases:
        chunk = synth.gen_chunk("B", c["seed"], c["segment"], c["chunk"], c["size"])
        assert chunk.size == c["size"] and hashlib.sha256(chunk.tobytes()).hexdigest() == c["sha256"], c
        assert len(synth.record_batches_of(chunk)) == c["record_batches"]
