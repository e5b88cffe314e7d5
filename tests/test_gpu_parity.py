"""Parity tests proper: the HIP library on a real MI355X, through the C ABI, against the CPU oracle.
Bit-exact everywhere (integer/byte work)."""
import numpy as np
import pytest

import tsxform
from tests import parity_cases as pc
from tsxform import synth

nat = tsxform._native
pytestmark = pytest.mark.gpu

CHUNK = synth.CHUNK


def test_runs_on_the_hip_library_and_gfx950(gpu):
    assert gpu.path == nat.LIB_PATH and "hipemu" not in gpu.version()
    import torch
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_crc32c_kat_and_edges(gpu, oracle):
    src = np.zeros(64, np.uint8); src[:9] = np.frombuffer(b"123456789", np.uint8); src[16:48] = 0xFF
    d = pc.make_descs([9, 32, 32], [0, 16, 48 - 48], [0, 0, 0], [0, 0, 0])
    d["src_off"][2] = 48; d["src_len"][2] = 0
    gpu.crc32c_batch(d, src)
    assert d["crc32c"][0] == 0xE3069283 and d["crc32c"][1] == 0x62A8AB43 and d["crc32c"][2] == 0
    chunks = pc.edge_chunks("R")
    sizes = [int(c.size) for c in chunks]
    soff, _, _, st, _ = pc.layout(sizes, 0, gpu)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    d = pc.make_descs(sizes, soff, [0] * len(sizes), [0] * len(sizes))
    gpu.crc32c_batch(d, src)
    for i, c in enumerate(chunks):
        assert d["crc32c"][i] == oracle.crc32c(c.tobytes()), sizes[i]


@pytest.mark.parametrize("flags", [nat.ENCRYPT | nat.CRC, nat.ENCRYPT, nat.CRC, 0])
def test_transform_no_compression_edges_vs_oracle(gpu, oracle, flags):
    pc.check_transform_vs_oracle(gpu, oracle, flags, pc.edge_chunks("R"))


def test_full_size_chunks_vs_oracle(gpu, oracle):
    chunks = [synth.gen_chunk("K", 1000, 0, 0), synth.gen_chunk("R", 1000, 0, 1), synth.gen_chunk("K", 1000, 0, 2, CHUNK - 5)]
    pc.check_transform_vs_oracle(gpu, oracle, nat.ENCRYPT | nat.CRC, chunks)
    pc.check_roundtrip(gpu, nat.ENCRYPT | nat.CRC, chunks)


def test_errors(gpu):
    chunks = pc.edge_chunks("R", [1000, 2000, 3000])
    outs, _ = pc.run_transform(gpu, nat.ENCRYPT, chunks)
    bad = bytearray(outs[1]); bad[500] ^= 0x40
    back, d = pc.run_detransform(gpu, nat.ENCRYPT, [outs[0], bytes(bad), b"x" * 20], [1000, 2000, 16])
    assert list(d["status"]) == [0, nat.E_TAG_MISMATCH, nat.E_SHORT_CHUNK] and back[0] == chunks[0].tobytes()
    _, d = pc.run_detransform(gpu, nat.ENCRYPT, outs[:1], [1000], aad=bytes(32))
    assert d["status"][0] == nat.E_TAG_MISMATCH


def _segment_on_gpu(dist, segment, nchunks):
    import torch
    return torch.cat([synth.gen_chunk(dist, 1000 + segment, segment, c, CHUNK, device="cuda") for c in range(nchunks)])


def test_one_gib_segment_device_resident(gpu, oracle):
    """BASELINE configs[1]/[2] at full size: 1 GiB segment = 256 x 4 MiB, device resident (torch tensors share
    the HIP runtime with libtsxform).  All 256 CRCs and a sample of GCM chunks against the oracle, the rest
    through size-independent properties (fixed transformed size, full round trip, CRC of restored bytes)."""
    import torch
    n = 256
    seg = _segment_on_gpu("K", 0, n)
    assert seg.numel() == n * CHUNK
    flags = nat.ENCRYPT | nat.CRC
    slot = (CHUNK + 28 + 15) // 16 * 16
    out = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
    d = np.zeros(n, nat.DESC_DTYPE)
    d["src_off"] = np.arange(n, dtype=np.uint64) * CHUNK; d["src_len"] = CHUNK
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for c in range(n):
        d["iv"][c] = np.frombuffer(synth.iv_for(0, c), np.uint8)
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    gpu.transform_batch(p, d, seg.data_ptr(), out.data_ptr(), out.numel(), nat.MEM_DEVICE)
    torch.cuda.synchronize()
    assert (d["status"] == 0).all() and (d["dst_len"] == CHUNK + 28).all()      # fixed-size index
    host = seg.cpu().numpy()
    for c in range(n):
        assert d["crc32c"][c] == oracle.crc32c(host[c * CHUNK:(c + 1) * CHUNK]), c
    enc = out.cpu().numpy()
    for c in [0, 1, 127, 255]:
        exp = oracle.gcm_encrypt_chunk(synth.KEY, synth.iv_for(0, c), synth.AAD, host[c * CHUNK:(c + 1) * CHUNK], openssl=True)
        assert enc[c * slot:c * slot + CHUNK + 28].tobytes() == exp, c
    # inverse on the device: every chunk restores, CRC(restored) == CRC(original)
    back = torch.empty(n * CHUNK, dtype=torch.uint8, device="cuda")
    d2 = np.zeros(n, nat.DESC_DTYPE)
    d2["src_off"] = d["dst_off"]; d2["src_len"] = d["dst_len"]; d2["dst_off"] = d["src_off"]; d2["dst_cap"] = CHUNK
    gpu.detransform_batch(p, d2, out.data_ptr(), back.data_ptr(), back.numel(), nat.MEM_DEVICE)
    torch.cuda.synchronize()
    assert (d2["status"] == 0).all() and (d2["crc32c"] == d["crc32c"]).all()
    assert torch.equal(back, seg)


def test_context_timing_reports_kernel_time(gpu):
    import torch
    ctx = gpu.ctx_create(0, 64, CHUNK)
    seg = _segment_on_gpu("R", 1, 16)
    d = np.zeros(16, nat.DESC_DTYPE)
    d["src_off"] = np.arange(16, dtype=np.uint64) * CHUNK; d["src_len"] = CHUNK
    gpu.crc32c_batch(d, seg.data_ptr(), nat.MEM_DEVICE, ctx=ctx)
    t = gpu.ctx_timing(ctx)
    assert t.crc_launches == 2 and 0 < t.crc_ms < 1000
    gpu.ctx_destroy(ctx)
