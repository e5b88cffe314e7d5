"""Parity tests proper: the HIP library on a real MI355X, through the C ABI, against the CPU oracle.
Bit-exact everywhere (integer/byte work)."""
import os

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


# ---- Zstandard stages -------------------------------------------------------------------------------------
def _zcases():
    rng = np.random.default_rng(7)
    K = synth.gen_chunk("K", 5, 0, 0, 1 << 20); R = synth.gen_chunk("R", 5, 0, 0, 1 << 19)
    return {
        "golden15": np.frombuffer(bytes.fromhex("000000030000000A01000A0000001E"), np.uint8),
        "empty": K[:0], "one": K[:1], "K7": K[:7], "K8": K[:8], "K64": K[:64], "K256": K[:256], "K1000": K[:1000], "K16385": K[:16385],
        "K70000": K[:70000], "K131073": K[:131073], "K262145": K[:262145], "K1M": K, "R300000": R[:300000],
        "zeros": np.zeros(500000, np.uint8), "period7": np.tile(np.frombuffer(b"abcdefg", np.uint8), 60000),
        "mixKR": np.concatenate([K[:200000], R[:150000], K[200000:500000], np.zeros(70000, np.uint8), R[:50000], K[:300000]]),
        "lowent": rng.integers(0, 4, 700000, dtype=np.uint8),
        "skewed": np.minimum(rng.geometric(0.3, 800000), 255).astype(np.uint8),
        "ramp": (np.arange(1000000) % 256).astype(np.uint8),
        "far_repeat": np.concatenate([R[:100000], K[:1100000], R[:100000], K[:600000], R[:100000]]),
    }


def test_zstd_compressor_byte_identical_to_libzstd(gpu, oracle):
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    cases = _zcases(); names = list(cases)
    outs, d = pc.run_transform(gpu, nat.COMPRESS, [cases[n] for n in names])
    for i, n in enumerate(names):
        assert d["status"][i] == 0, n
        assert outs[i] == oracle.zstd_compress_chunk(cases[n].tobytes()), "%s: frame differs from libzstd %s" % (n, oracle.zstd_version())
    # 1.5.6 profile (no pre-block splitter) against the serial restatement
    outs, d = pc.run_transform(gpu, nat.COMPRESS, [cases["mixKR"], cases["K1M"]], profile=nat.ZSTD_PROFILE_1_5_6)
    assert outs[0] == oracle.zstd_l3_compress(cases["mixKR"].tobytes(), 0) and outs[1] == oracle.zstd_l3_compress(cases["K1M"].tobytes(), 0)


def test_zstd_profile_1_5_6_whole_case_set_and_fuzz(gpu, oracle):
    """the profile the Java class can ship (1.5.6) had two GPU assertions.  Here: every case of _zcases(), 40 fuzzed
    inputs, the window-edge chunks and two full 4 MiB chunks through BOTH profiles - profile 0 equals the restatement everywhere and
    the real libzstd 1.5.7 on every input the 1.5.7 pre-splitter leaves alone; where it cuts, the real library decodes both."""
    from tests import fuzz_cases
    cases = dict(_zcases())
    rng = np.random.default_rng(20260923)
    for k in range(40):
        cases["fuzz%d" % k] = fuzz_cases.gen_case(rng)
    for seed in fuzz_cases.WINDOW_EDGE_SEEDS[:3]:
        cases["window_edge_%d" % seed] = fuzz_cases.window_edge_case(seed)
    cases["K4M"] = synth.gen_chunk("K", 1000, 0, 7); cases["mix4M"] = np.concatenate([cases["far_repeat"], cases["mixKR"], cases["skewed"]])[:CHUNK]
    pinned, differ = pc.check_profile_1_5_6(gpu, oracle, cases)
    assert differ >= 1
    if oracle.zstd_version().startswith("1.5.7"):
        assert pinned >= len(cases) // 2
    print("profile 1.5.6: %d of %d inputs byte-identical to the real libzstd 1.5.7, %d differ only through the pre-splitter" % (pinned, len(cases), differ))


@pytest.mark.parametrize("gcm", ["in_compressor_wave", "separate_kernels"])
def test_zstd_full_size_chunks_and_full_chain(gpu, oracle, gcm):
    """With compression each compressor wave also checksums its chunk and encrypts its frame, unless the test hook stages_separate
    asks for one launch per stage."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    with gpu.configured(stages_separate=1 if gcm == "separate_kernels" else 0):
        _zstd_full_size_chunks_and_full_chain(gpu, oracle)


def _zstd_full_size_chunks_and_full_chain(gpu, oracle):
    small = [synth.gen_chunk("K", 7, 1, i, s) for i, s in enumerate([0, 1, 15, 16, 17, 1000, 65536, 70001, 300007])]
    pc.check_transform_vs_oracle(gpu, oracle, nat.COMPRESS | nat.ENCRYPT | nat.CRC, small)
    outs, d = pc.run_transform(gpu, nat.COMPRESS | nat.ENCRYPT, [synth.gen_chunk("R", 7, 1, 0, 50000), small[5]], dst_caps=[50000, None])
    assert d["status"][0] == nat.E_DST_TOO_SMALL and d["dst_len"][0] == 0 and d["status"][1] == 0
    assert outs[1] == pc.oracle_transform(oracle, nat.COMPRESS | nat.ENCRYPT, small[5], 1)
    chunks = [synth.gen_chunk("K", 1000, 0, 0), synth.gen_chunk("R", 1000, 0, 1), synth.gen_chunk("K", 1000, 0, 2, CHUNK - 5)]
    outs, d = pc.check_transform_vs_oracle(gpu, oracle, nat.COMPRESS | nat.ENCRYPT | nat.CRC, chunks)
    assert outs[0][:12] == synth.iv_for(0, 0) and d["dst_len"][1] == CHUNK + 106 + 28          # raw blocks: n + 10 + 3*32, + IV + tag
    pc.check_roundtrip(gpu, nat.COMPRESS | nat.ENCRYPT | nat.CRC, chunks)
    pc.check_roundtrip(gpu, nat.COMPRESS, chunks)


@pytest.mark.parametrize("level", [0, 1, 19])
def test_zstd_decoder_accepts_libzstd_frames(gpu, oracle, level):
    cases = _zcases(); names = list(cases)
    blobs = [oracle.zstd_compress_chunk(cases[n].tobytes(), level) for n in names]
    outs, d = pc.run_detransform(gpu, nat.COMPRESS, blobs, [int(cases[n].size) for n in names])
    for i, n in enumerate(names):
        assert d["status"][i] == 0 and outs[i] == cases[n].tobytes(), (n, level)


def test_zstd_decoder_errors(gpu, oracle):
    K = synth.gen_chunk("K", 5, 0, 0, 70000)
    good = oracle.zstd_compress_chunk(K.tobytes())
    no_size = b"\x28\xb5\x2f\xfd\x00\x58" + b"\x01\x00\x00"
    outs, d = pc.run_detransform(gpu, nat.COMPRESS, [good, b"\x00" + good[1:], good[:len(good) // 2], no_size, good], [70000, 70000, 70000, 16, 100])
    assert list(d["status"]) == [0, nat.E_BAD_FRAME, nat.E_BAD_FRAME, nat.E_BAD_SIZE, nat.E_DST_TOO_SMALL] and outs[0] == K.tobytes()


def test_quarter_gib_full_chain_device_resident(gpu, oracle):
    """64 x 4 MiB chunks through Zstd -> GCM -> CRC on the device; sampled chunks byte-exact vs libzstd + OpenSSL,
    all chunks through the round trip and the CRC-of-restored property."""
    import torch
    n = 64
    seg = _segment_on_gpu("K", 2, n)
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    slot = (gpu.transformed_bound(CHUNK, flags) + 63) // 64 * 64
    out = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
    d = np.zeros(n, nat.DESC_DTYPE)
    d["src_off"] = np.arange(n, dtype=np.uint64) * CHUNK; d["src_len"] = CHUNK
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for c in range(n):
        d["iv"][c] = np.frombuffer(synth.iv_for(2, c), np.uint8)
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    gpu.transform_batch(p, d, seg.data_ptr(), out.data_ptr(), out.numel(), nat.MEM_DEVICE)
    assert (d["status"] == 0).all() and (d["dst_len"] < CHUNK // 2).all()
    host = seg.cpu().numpy(); enc = out.cpu().numpy()
    if oracle.zstd_version().startswith("1.5.7"):
        for c in [0, 31, 63]:
            exp, crc = oracle.transform_chunk(oracle.COMPRESS | oracle.ENCRYPT | oracle.CRC | oracle.OPENSSL, synth.KEY, synth.AAD, synth.iv_for(2, c),
                                              host[c * CHUNK:(c + 1) * CHUNK].tobytes())
            assert enc[c * slot:c * slot + d["dst_len"][c]].tobytes() == exp and d["crc32c"][c] == crc, c
    back = torch.empty(n * CHUNK, dtype=torch.uint8, device="cuda")
    d2 = np.zeros(n, nat.DESC_DTYPE)
    d2["src_off"] = d["dst_off"]; d2["src_len"] = d["dst_len"]; d2["dst_off"] = d["src_off"]; d2["dst_cap"] = CHUNK
    gpu.detransform_batch(p, d2, out.data_ptr(), back.data_ptr(), back.numel(), nat.MEM_DEVICE)
    assert (d2["status"] == 0).all() and (d2["dst_len"] == CHUNK).all() and (d2["crc32c"] == d["crc32c"]).all()
    assert torch.equal(back, seg)


@pytest.mark.timeout(600)
def test_zstd_decoder_survives_corrupt_frames(gpu, oracle):
    """Same fuzz as the emulated run, 600 variants on the device: per-chunk errors, no hang, no fault."""
    from tests.test_emu_zstd import _fuzzed_frames
    blobs, sizes = _fuzzed_frames(oracle, 600, 23)
    outs, d = pc.run_detransform(gpu, nat.COMPRESS, blobs, sizes)
    assert set(int(x) for x in d["status"]) <= {0, nat.E_BAD_FRAME, nat.E_BAD_SIZE, nat.E_DST_TOO_SMALL}
    assert (d["status"] != 0).sum() >= 200
    good = oracle.zstd_compress_chunk(np.arange(5000, dtype=np.uint8).tobytes())       # the device still works afterwards
    outs, d = pc.run_detransform(gpu, nat.COMPRESS, [good], [5000])
    assert d["status"][0] == 0 and outs[0] == np.arange(5000, dtype=np.uint8).tobytes()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_zstd_differential_fuzz_vs_libzstd(gpu, oracle):
    """Several hundred structured random inputs (tests/fuzz_cases.py: segments of differing statistics with verbatim and edited
    copies at all distances, sizes around the block boundaries) + the regression inputs: every frame must be libzstd 1.5.7's byte
    for byte, the device decoder must restore it, and the full chain must match the oracle chain."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    import glob
    import os
    from tests.fuzz_cases import gen_case
    here = os.path.dirname(os.path.abspath(__file__))
    cases = [np.fromfile(f, np.uint8) for f in sorted(glob.glob(os.path.join(here, "golden", "fuzz_regress", "*.bin")))]
    assert len(cases) >= 5
    rng = np.random.default_rng(4242)
    cases += [gen_case(rng) for _ in range(600)]
    # full-size chunks: the window-edge pins and a sample whose copies reach beyond the 2 MiB window
    from tests.fuzz_cases import window_edge_cases
    cases += window_edge_cases() + [gen_case(rng, 4194304 - int(rng.integers(0, 3)) * int(rng.integers(0, 70000))) for _ in range(58)]
    for lo in range(0, len(cases), 128):
        part = cases[lo:lo + 128]
        outs, d = pc.run_transform(gpu, nat.COMPRESS, part)
        back, d2 = pc.run_detransform(gpu, nat.COMPRESS, outs, [int(c.size) for c in part])
        for i, c in enumerate(part):
            assert d["status"][i] == 0 and outs[i] == oracle.zstd_compress_chunk(c.tobytes()), "case %d (n=%d): frame differs from libzstd" % (lo + i, c.size)
            assert d2["status"][i] == 0 and back[i] == c.tobytes(), "case %d (n=%d): round trip" % (lo + i, c.size)
    pc.check_transform_vs_oracle(gpu, oracle, nat.COMPRESS | nat.ENCRYPT | nat.CRC, cases[:64])


# ---- the front end on the device (twins of tests/test_emu_boundary.py) ----------------------------------------
def test_detransform_on_a_fresh_context_of_highly_compressible_chunks(gpu, oracle):
    """160 all-zero 4 MiB chunks compress to ~150 bytes each; the CRC of the restored bytes runs over 4 MiB slots
    (16 partial sums per chunk) on a context that has only ever seen those tiny inputs."""
    import torch
    n = 160
    zero_frame = oracle.zstd_compress_chunk(bytes(CHUNK))
    blob = oracle.gcm_encrypt_chunk(synth.KEY, synth.iv_for(0, 0), synth.AAD, zero_frame, openssl=True)
    stride = (len(blob) + 15) // 16 * 16 + 16
    src = np.zeros(n * stride, np.uint8)
    for i in range(n):
        src[i * stride:i * stride + len(blob)] = np.frombuffer(blob, np.uint8)
    dsrc = torch.from_numpy(src).cuda()
    back = torch.full((n * CHUNK,), 0x5A, dtype=torch.uint8, device="cuda")
    guard = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")          # a neighbour allocation that must stay untouched
    d = np.zeros(n, nat.DESC_DTYPE)
    d["src_off"] = np.arange(n, dtype=np.uint64) * stride; d["src_len"] = len(blob)
    d["dst_off"] = np.arange(n, dtype=np.uint64) * CHUNK; d["dst_cap"] = CHUNK
    ctx = gpu.ctx_create(0, 0, 0)
    try:
        gpu.detransform_batch(nat.Native.make_params(nat.COMPRESS | nat.ENCRYPT | nat.CRC, synth.KEY, synth.AAD), d, dsrc.data_ptr(), back.data_ptr(),
                              back.numel(), nat.MEM_DEVICE, ctx=ctx)
        assert gpu.lib.tsx_debug_key_residue(ctx) == 0
    finally:
        gpu.ctx_destroy(ctx)
    assert (d["status"] == 0).all() and (d["dst_len"] == CHUNK).all() and (d["crc32c"] == oracle.crc32c(bytes(CHUNK))).all()
    assert int(back.max()) == 0 and int(guard.max()) == 0


def test_forged_chunk_is_scrubbed_from_a_device_slot(gpu):
    import torch
    chunks = pc.edge_chunks("R", [3000, 2 << 20, 70001])
    outs, _ = pc.run_transform(gpu, nat.ENCRYPT, chunks)
    forged = bytearray(outs[1]); forged[1 << 20] ^= 1
    blobs = [outs[0], bytes(forged), outs[2]]
    soff, st = [], 0
    for b in blobs:
        soff.append(st); st += (len(b) + 15) // 16 * 16 + 16
    src = np.zeros(st, np.uint8)
    for b, o_ in zip(blobs, soff):
        src[o_:o_ + len(b)] = np.frombuffer(b, np.uint8)
    doff = [0, 4096, 4096 + (2 << 20) + 4096]
    total = doff[2] + 70016
    d = pc.make_descs([len(b) for b in blobs], soff, doff, [3008, (2 << 20) + 16, 70016])
    dsrc = torch.from_numpy(src).cuda(); ddst = torch.full((total,), 0xAB, dtype=torch.uint8, device="cuda")
    gpu.detransform_batch(nat.Native.make_params(nat.ENCRYPT, synth.KEY, synth.AAD), d, dsrc.data_ptr(), ddst.data_ptr(), total, nat.MEM_DEVICE)
    back = ddst.cpu().numpy()
    assert list(d["status"]) == [0, nat.E_TAG_MISMATCH, 0] and d["dst_len"][1] == 0
    assert back[0:3000].tobytes() == chunks[0].tobytes() and back[doff[2]:doff[2] + 70001].tobytes() == chunks[2].tobytes()
    assert not back[4096:4096 + (2 << 20)].any(), "unauthenticated plaintext left in the caller's device slot"


@pytest.mark.parametrize("flags", [nat.ENCRYPT | nat.CRC, nat.COMPRESS | nat.ENCRYPT | nat.CRC])
def test_staged_host_pipeline_on_the_device(gpu, oracle, flags):
    """Host-memory batches in pieces (three streams; members of the compressor service) = one shot = the oracle; pageable and registered
    buffers; packed output."""
    chunks = [synth.gen_chunk("K", 11, 0, i, s) for i, s in enumerate([CHUNK, 1 << 20, 70001, 0, 17, CHUNK - 5, 300000, 1 << 16, 4096, 2 << 20, 12345, 1 << 20])]
    with gpu.configured(no_pipeline=1):
        ref, dref = pc.run_transform(gpu, flags, chunks)
    with gpu.configured(sub_bytes=3 << 20):
        _staged_host_pipeline(gpu, oracle, flags, chunks, ref, dref)


def _staged_host_pipeline(gpu, oracle, flags, chunks, ref, dref):
    got, dgot = pc.check_transform_vs_oracle(gpu, oracle, flags, chunks)
    gotp, dpk = pc.run_transform(gpu, flags, chunks, mem="packed")
    assert got == ref == gotp and (dgot["crc32c"] == dref["crc32c"]).all() and (dpk["dst_len"] == dref["dst_len"]).all()
    back, d2 = pc.run_detransform(gpu, flags, got, [int(c.size) for c in chunks])
    assert (d2["status"] == 0).all() and back == [c.tobytes() for c in chunks]
    # the same through registered (pinned) buffers
    sizes = [int(c.size) for c in chunks]
    soff, doff, caps, st, dt = pc.layout(sizes, flags, gpu)
    src = np.zeros(st, np.uint8); dst = np.zeros(dt, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    gpu.host_register(src); gpu.host_register(dst)
    try:
        d = pc.make_descs(sizes, soff, doff, caps)
        gpu.transform_batch(nat.Native.make_params(flags, synth.KEY, synth.AAD), d, src, dst, dst.size)
        assert [dst[doff[i]:doff[i] + d["dst_len"][i]].tobytes() for i in range(len(sizes))] == ref
    finally:
        gpu.host_unregister(src); gpu.host_unregister(dst)


def test_encrypt_only_batches_write_into_registered_buffers_on_the_device(gpu, oracle):
    """The producers-compress chain (RemoteStorageManager.java:381-398: encryption only): the GCM kernel's waves store IV || C || TAG through
    the device alias of the caller's registered buffer - full-size chunks, an empty one, one whose slot is too small; same bytes as the copy
    path and the oracle (the body is the emulator test's: tests/parity_cases.py)."""
    pc.check_encrypt_only_zero_copy(gpu, oracle, [CHUNK, 0, 70001, 17, CHUNK - 5, 4096, 1 << 20, 3])


@pytest.mark.parametrize("kind", ["K6", "K10", "mixed6", "mixed10", "sparse64", "K64"])
def test_chunks_beyond_4_MiB(gpu, oracle, kind):
    """chunk.size above 4 MiB - up to a whole segment as ONE chunk (RemoteStorageManagerConfig.java:122-130, chunk.size = 0 in
    BaseTransformChunkEnumeration.java:85-89; the reference's integration matrix has a 10 MiB segment as one chunk): the full chain
    equals libzstd 1.5.7 + OpenSSL for both Zstd profiles (wherever 1.5.7's pre-splitter is idle), decodes back, and the host-memory
    pipeline cut into pieces gives the same bytes."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    x = pc.big_chunk(kind)
    small = synth.gen_chunk("K", 13, 0, 1, 300000)                     # a small neighbour in the same batch
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    outs, d = pc.check_transform_vs_oracle(gpu, oracle, flags, [x, small])
    back, d2 = pc.run_detransform(gpu, flags, outs, [int(x.size), int(small.size)])
    assert (d2["status"] == 0).all() and back[0] == x.tobytes() and back[1] == small.tobytes() and d2["crc32c"][0] == d["crc32c"][0]
    pinned, differ = pc.check_profile_1_5_6(gpu, oracle, {kind: x})
    with gpu.configured(sub_bytes=3 << 20):                           # pieces smaller than the chunk: one chunk never straddles two
        outs2, _ = pc.run_transform(gpu, flags, [x, small], mem="packed")
    assert outs2 == outs
    dev, _ = pc.run_transform(gpu, flags, [x, small], mem="device")
    assert dev == outs


def test_descriptors_beyond_the_source_buffer_are_rejected_on_the_device(gpu):
    """ABI 3 on the product library: src_size bounds every descriptor; CRC-only batches publish their statuses."""
    src = np.zeros(4096, np.uint8); dst = np.zeros(8192, np.uint8)
    p = nat.Native.make_params(nat.ENCRYPT | nat.CRC, synth.KEY, synth.AAD)
    d = pc.make_descs([64], [4096 - 32], [0], [256]); d["status"] = -7
    for call in (lambda: gpu.transform_batch(p, d, src, dst, dst.size), lambda: gpu.detransform_batch(p, d, src, dst, dst.size),
                 lambda: gpu.crc32c_batch(d, src)):
        with pytest.raises(nat.TsxError) as e:
            call()
        assert e.value.code == nat.E_INVAL and d["status"][0] == -7
    src[:9] = np.frombuffer(b"123456789", np.uint8)
    d = pc.make_descs([9, 0], [0, 16], [0, 0], [0, 0]); d["status"] = -7
    gpu.crc32c_batch(d, src)
    assert list(d["status"]) == [0, 0] and d["crc32c"][0] == 0xE3069283


def test_every_chunk_of_a_segment_equals_libzstd_and_openssl(gpu, oracle):
    """at full size the oracle saw a handful of chunks.  Here every one of the 256 chunks of a 1 GiB Kafka-like
    segment (+ 16 incompressible ones) through Zstd -> GCM -> CRC on the device is compared byte for byte with libzstd 1.5.7 + OpenSSL,
    for BOTH Zstd profiles (the content never makes 1.5.7's pre-splitter cut, so profile 1.5.6 must give the same bytes)."""
    import torch
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    n = 272
    seg = torch.empty(n * CHUNK, dtype=torch.uint8, device="cuda")
    for c in range(n):
        seg[c * CHUNK:(c + 1) * CHUNK] = synth.gen_chunk("K" if c < 256 else "R", 1234, 3, c, CHUNK, device="cuda")
    host = seg.cpu().numpy()
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    slot = (gpu.transformed_bound(CHUNK, flags) + 63) // 64 * 64
    out = torch.empty(n * slot, dtype=torch.uint8, device="cuda")
    of = oracle.COMPRESS | oracle.ENCRYPT | oracle.CRC | oracle.OPENSSL
    expected = None
    for profile in (nat.ZSTD_PROFILE_1_5_7, nat.ZSTD_PROFILE_1_5_6):
        d = np.zeros(n, nat.DESC_DTYPE)
        d["src_off"] = np.arange(n, dtype=np.uint64) * CHUNK; d["src_len"] = CHUNK
        d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
        for c in range(n):
            d["iv"][c] = np.frombuffer(synth.iv_for(3, c), np.uint8)
        gpu.transform_batch(nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=profile), d, seg.data_ptr(), out.data_ptr(), out.numel(), nat.MEM_DEVICE)
        assert (d["status"] == 0).all()
        got = out.cpu().numpy()
        if expected is None:
            expected = [oracle.transform_chunk(of, synth.KEY, synth.AAD, synth.iv_for(3, c), host[c * CHUNK:(c + 1) * CHUNK].tobytes()) for c in range(n)]
        for c in range(n):
            exp, crc = expected[c]
            assert d["crc32c"][c] == crc, c
            assert got[c * slot:c * slot + int(d["dst_len"][c])].tobytes() == exp, "chunk %d, profile %d" % (c, profile)


def test_concurrent_ctxless_host_batches(gpu, oracle):
    """The JVM case in small: 8 threads, no context of their own (ctx == NULL -> pooled contexts), host buffers, full chain and
    encrypt-only batches interleaved, the staged pipeline cut into small pieces - every result equals the single-threaded one."""
    import threading
    old_sub = gpu.debug_config("sub_bytes", 2 << 20)
    try:
        sets = {}
        for flags in (nat.ENCRYPT | nat.CRC, nat.COMPRESS | nat.ENCRYPT | nat.CRC):
            chunks = [synth.gen_chunk("K", 21, 0, i, s) for i, s in enumerate([1 << 20, 70001, 1 << 19, 17, 0, 300000, 1 << 20, 4096] * 4)]
            sets[flags] = (chunks, pc.run_transform(gpu, flags, chunks)[0])
        errors = []

        def worker(k):
            try:
                for rep in range(4):
                    flags = (nat.ENCRYPT | nat.CRC) if (k + rep) % 2 else (nat.COMPRESS | nat.ENCRYPT | nat.CRC)
                    chunks, ref = sets[flags]
                    outs, d = pc.run_transform(gpu, flags, chunks, mem="packed" if rep % 2 else None)
                    if outs != ref or (d["status"] != 0).any():
                        errors.append((k, rep, "transform"))
                    back, d2 = pc.run_detransform(gpu, flags, outs, [int(c.size) for c in chunks])
                    if back != [c.tobytes() for c in chunks] or (d2["status"] != 0).any():
                        errors.append((k, rep, "detransform"))
            except Exception as e:                                    # noqa: BLE001 - reported through the list
                errors.append((k, repr(e)))

        th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errors, errors[:5]
        s = gpu.pool_stats(0)
        assert s["in_use"] == 0 and 1 <= s["idle"] <= 32
    finally:
        gpu.debug_config("sub_bytes", old_sub)


def test_kafka_shaped_binary_content_on_the_device(gpu, oracle):
    """Content "B" (v2 record batches: binary headers, varint-framed records, one record in 32 with an incompressible payload - tsxform/synth.py):
    the full chain equals libzstd 1.5.7 + OpenSSL on full-size and smaller chunks, both Zstd profiles hold what they promise (1.5.7's
    pre-splitter cuts on this content: the profiles differ and both decode), everything round-trips, and the CRC kernel confirms the
    batch CRCs of a chunk as SegmentCompressionChecker.java:37-53 would check the first batch of a segment."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    sizes = [CHUNK, CHUNK, CHUNK - 4099] + [131072 + 29989 * i for i in range(21)]
    chunks = [synth.gen_chunk("B", 41, 2, i, s) for i, s in enumerate(sizes)]
    outs, d = pc.check_transform_vs_oracle(gpu, oracle, flags, chunks)
    back, d2 = pc.run_detransform(gpu, flags, outs, sizes)
    assert (d2["status"] == 0).all() and back == [c.tobytes() for c in chunks] and (d2["crc32c"] == d["crc32c"]).all()
    pinned, differ = pc.check_profile_1_5_6(gpu, oracle, {"B%d_%d" % (i, c.size): c for i, c in enumerate(chunks[1:10])})
    assert differ >= 1, "the pre-splitter never cut on B content: the test does not exercise the difference between the profiles"
    ratio = sum(len(o_) for o_ in outs) / float(sum(sizes))
    print("B content: %d chunks, transformed / original = %.3f, profile 1.5.6: %d pinned to the real library, %d differ through the pre-splitter" % (len(chunks), ratio, pinned, differ))
    c = chunks[3]
    batches = synth.record_batches_of(c)
    buf = np.zeros(2 * c.size + 64, np.uint8); lens, offs, at = [], [], 0
    for p, l in batches:
        buf[at:at + l - 21] = c[p + 21:p + l]; offs.append(at); lens.append(l - 21); at += (l - 21 + 15) // 16 * 16
    dd = pc.make_descs(lens, offs, [0] * len(lens), [0] * len(lens))
    gpu.crc32c_batch(dd, buf)
    cb = c.tobytes()
    assert [int(x) for x in dd["crc32c"]] == [int.from_bytes(cb[p + 17:p + 21], "big") for p, _ in batches]
