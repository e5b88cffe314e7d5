"""Kernel LOGIC under the CPU emulator (tests/emu): the exact csrc/*.hip sources, compiled with g++ against a
fiber-based HIP shim, checked bit-for-bit against the oracle at small sizes.  This is a development harness
for a GPU-less container — the parity tests proper are tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

import tsxform
from tests import parity_cases as pc

nat = tsxform._native


def test_emu_is_not_the_product_library(emu):
    assert "hipemu" in emu.version() and emu.path != nat.LIB_PATH


def test_crc32c_edge_sizes(emu, oracle):
    chunks = pc.edge_chunks("R")
    sizes = [int(c.size) for c in chunks]
    soff, _, _, st, _ = pc.layout(sizes, 0, emu)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    d = pc.make_descs(sizes, soff, [0] * len(sizes), [0] * len(sizes))
    emu.crc32c_batch(d, src)
    for i, c in enumerate(chunks):
        assert d["crc32c"][i] == oracle.crc32c(c.tobytes()), sizes[i]


def test_crc32c_kat(emu):
    src = np.zeros(64, np.uint8); src[:9] = np.frombuffer(b"123456789", np.uint8)
    src[16:48] = 0xFF
    d = pc.make_descs([9, 32], [0, 16], [0, 0], [0, 0])
    emu.crc32c_batch(d, src)
    assert d["crc32c"][0] == 0xE3069283 and d["crc32c"][1] == 0x62A8AB43


@pytest.mark.parametrize("flags", [nat.ENCRYPT, nat.ENCRYPT | nat.CRC, nat.CRC, 0])
def test_transform_no_compression_vs_oracle(emu, oracle, flags):
    pc.check_transform_vs_oracle(emu, oracle, flags, pc.edge_chunks("R"))


def test_encrypt_fixed_transformed_size(emu):
    # EncryptionChunkEnumeration.java:82-84 / EncryptionChunkEnumerationTest.java:79-86: n + 12 + 16
    chunks = pc.edge_chunks("K", [100, 4096, 70000])
    outs, d = pc.run_transform(emu, nat.ENCRYPT, chunks)
    assert [len(o_) for o_ in outs] == [128, 4124, 70028]


@pytest.mark.parametrize("flags", [nat.ENCRYPT | nat.CRC, 0])
def test_roundtrip(emu, flags):
    pc.check_roundtrip(emu, flags, pc.edge_chunks("K", [0, 1, 17, 4096, 65537, 200000]))


def test_tag_mismatch_and_short_chunk(emu):
    chunks = pc.edge_chunks("R", [1000, 2000, 3000])
    outs, _ = pc.run_transform(emu, nat.ENCRYPT, chunks)
    bad = bytearray(outs[1]); bad[500] ^= 0x40
    badtag = bytearray(outs[2]); badtag[-1] ^= 1
    back, d = pc.run_detransform(emu, nat.ENCRYPT, [outs[0], bytes(bad), bytes(badtag), b"x" * 20], [1000, 2000, 3000, 16])
    assert list(d["status"]) == [0, nat.E_TAG_MISMATCH, nat.E_TAG_MISMATCH, nat.E_SHORT_CHUNK]
    assert back[0] == chunks[0].tobytes() and d["dst_len"][1] == 0


def test_wrong_aad_or_key_fails_tag(emu):
    chunks = pc.edge_chunks("R", [5000])
    outs, _ = pc.run_transform(emu, nat.ENCRYPT, chunks)
    _, d = pc.run_detransform(emu, nat.ENCRYPT, outs, [5000], aad=bytes(32))
    assert d["status"][0] == nat.E_TAG_MISMATCH
    _, d = pc.run_detransform(emu, nat.ENCRYPT, outs, [5000], key=bytes(32))
    assert d["status"][0] == nat.E_TAG_MISMATCH


def test_dst_too_small(emu):
    chunks = pc.edge_chunks("R", [1000, 1000])
    sizes = [1000, 1000]
    soff, doff, caps, st, dt = pc.layout(sizes, nat.ENCRYPT, emu)
    caps[1] = 1027                                   # one byte short of n + 28
    src = np.zeros(st, np.uint8); dst = np.zeros(dt, np.uint8)
    d = pc.make_descs(sizes, soff, doff, caps)
    emu.transform_batch(nat.Native.make_params(nat.ENCRYPT, pc.synth.KEY, pc.synth.AAD), d, src, dst, dst.size)
    assert list(d["status"]) == [0, nat.E_DST_TOO_SMALL] and d["dst_len"][1] == 0


def test_invalid_arguments(emu):
    d = pc.make_descs([16], [8], [0], [64])          # misaligned src_off
    src = np.zeros(64, np.uint8); dst = np.zeros(64, np.uint8)
    with pytest.raises(nat.TsxError) as e:
        emu.transform_batch(nat.Native.make_params(0), d, src, dst, 64)
    assert e.value.code == nat.E_INVAL
    d = pc.make_descs([16], [0], [0], [128])         # slot beyond dst_size
    with pytest.raises(nat.TsxError):
        emu.transform_batch(nat.Native.make_params(0), d, src, dst, 64)


def test_device_memory_mode_matches_host_mode(emu, oracle):
    pc.check_transform_vs_oracle(emu, oracle, nat.ENCRYPT | nat.CRC, pc.edge_chunks("K", [10, 70000]), mem="device")


def test_explicit_context_and_timing(emu):
    ctx = emu.ctx_create(0, 4, 1 << 16)
    chunks = pc.edge_chunks("R", [4096])
    sizes = [4096]
    soff, doff, caps, st, dt = pc.layout(sizes, nat.ENCRYPT, emu)
    src = np.zeros(st, np.uint8); dst = np.zeros(dt, np.uint8)
    d = pc.make_descs(sizes, soff, doff, caps)
    emu.transform_batch(nat.Native.make_params(nat.ENCRYPT, pc.synth.KEY, pc.synth.AAD), d, src, dst, dst.size, ctx=ctx)
    t = emu.ctx_timing(ctx)
    assert t.gcm_launches == 2 and t.total_ms >= 0
    emu.ctx_destroy(ctx)
