"""The oracle against every golden vector / known-answer test available for this path."""
import base64

import numpy as np
import pytest

from tsxform import synth


def test_crc32c_kats(oracle):
    # java.util.zip.CRC32C == iSCSI CRC (RFC 3720 B.4)
    assert oracle.crc32c(b"123456789") == 0xE3069283
    assert oracle.crc32c(bytes(32)) == 0x8A9136AA
    assert oracle.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert oracle.crc32c(bytes(range(32))) == 0x46DD794E
    assert oracle.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert oracle.crc32c(b"") == 0


def test_crc32c_table_vs_bitwise(oracle):
    rng = np.random.default_rng(3)
    for n in [1, 7, 8, 9, 63, 64, 65, 1000, 4097]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.crc32c(d) == oracle.crc32c_bitwise(d)


def test_aes256_fips197_c3(oracle):
    key = bytes(range(32))
    assert oracle.aes256_encrypt_block(key, bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == \
        "8ea2b7ca516745bfeafc49904b496089"


GCM_VECTORS = [  # McGrew & Viega, "The Galois/Counter Mode of Operation", test cases 13-16 (AES-256)
    ("00" * 32, "00" * 12, "", "", "", "530f8afbc74536b9a963b4f1c4cb738b"),
    ("00" * 32, "00" * 12, "00" * 16, "", "cea7403d4d606b6e074ec5d3baf39d18", "d0d1c8a799996bf0265b98b5d48ab919"),
    ("feffe9928665731c6d6a8f9467308308feffe9928665731c6d6a8f9467308308", "cafebabefacedbaddecaf888",
     "d9313225f88406e5a55909c5aff5269a86a7a9531534f7da2e4c303d8a318a721c3c0c95956809532fcf0e2449a6b525b16aedf5aa0de657ba637b391aafd255",
     "",
     "522dc1f099567d07f47f37a32a84427d643a8cdcbfe5c0c97598a2bd2555d1aa8cb08e48590dbb3da7b08b1056828838c5f61e6393ba7a0abcc9f662898015ad",
     "b094dac5d93471bdec1a502270e3cc6c"),
    ("feffe9928665731c6d6a8f9467308308feffe9928665731c6d6a8f9467308308", "cafebabefacedbaddecaf888",
     "d9313225f88406e5a55909c5aff5269a86a7a9531534f7da2e4c303d8a318a721c3c0c95956809532fcf0e2449a6b525b16aedf5aa0de657ba637b39",
     "feedfacedeadbeeffeedfacedeadbeefabaddad2",
     "522dc1f099567d07f47f37a32a84427d643a8cdcbfe5c0c97598a2bd2555d1aa8cb08e48590dbb3da7b08b1056828838c5f61e6393ba7a0abcc9f662",
     "76fc6ece0f4e1768cddf8853bb2d551b"),
]


@pytest.mark.parametrize("key,iv,pt,aad,ct,tag", GCM_VECTORS)
@pytest.mark.parametrize("openssl", [False, True])
def test_gcm_spec_vectors(oracle, key, iv, pt, aad, ct, tag, openssl):
    out = oracle.gcm_encrypt_chunk(bytes.fromhex(key), bytes.fromhex(iv), bytes.fromhex(aad), bytes.fromhex(pt), openssl=openssl)
    # layout of EncryptionChunkEnumeration.java:66-84: IV || C || TAG
    assert out[:12].hex() == iv and out[12:-16].hex() == ct and out[-16:].hex() == tag
    assert oracle.gcm_decrypt_chunk(bytes.fromhex(key), bytes.fromhex(aad), out, openssl=openssl).hex() == pt


def test_gcm_restatement_vs_openssl(oracle):
    rng = np.random.default_rng(5)
    for n in [0, 1, 15, 16, 17, 100, 4096, 65537, 300001]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        iv = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
        a = oracle.gcm_encrypt_chunk(synth.KEY, iv, synth.AAD, d)
        assert a == oracle.gcm_encrypt_chunk(synth.KEY, iv, synth.AAD, d, openssl=True)
        assert len(a) == n + 28                       # EncryptionChunkEnumerationTest.java:79-86
        assert oracle.gcm_decrypt_chunk(synth.KEY, synth.AAD, a) == d
        # CTR half of GCM (BASELINE config 3): ciphertext == AES-CTR keystream from counter IV||2
        assert a[12:-16] == oracle.aes256_ctr(synth.KEY, iv, 2, d)


def test_gcm_bad_tag_and_short(oracle):
    c = bytearray(oracle.gcm_encrypt_chunk(synth.KEY, bytes(12), synth.AAD, b"hello world"))
    c[14] ^= 1
    with pytest.raises(oracle.BadTag):
        oracle.gcm_decrypt_chunk(synth.KEY, synth.AAD, bytes(c))
    with pytest.raises(oracle.BadTag):
        oracle.gcm_decrypt_chunk(synth.KEY, synth.AAD, bytes(c), openssl=True)
    with pytest.raises(RuntimeError):
        oracle.gcm_decrypt_chunk(synth.KEY, synth.AAD, b"x" * 27)


def test_gf128_table_mul_vs_bitwise(oracle):
    rng = np.random.default_rng(9)
    L = oracle.lib()
    for _ in range(50):
        x = rng.integers(0, 256, 16, dtype=np.uint8); h = rng.integers(0, 256, 16, dtype=np.uint8)
        a = x.copy(); b = x.copy()
        L.orc_gf128_mul(a.ctypes.data, h.ctypes.data)
        L.orc_gf128_mul_tab(b.ctypes.data, h.ctypes.data)
        assert a.tobytes() == b.tobytes()


def test_zstd_reference_golden_frame(oracle):
    # core/src/test/java/io/aiven/kafka/tieredstorage/manifest/index/ChunkIndexSerializationTest.java:39-61
    raw = bytes.fromhex("000000030000000A01000A0000001E")
    frame = oracle.zstd_compress_chunk(raw)
    assert base64.b64encode(frame) == b"KLUv/SAPeQAAAAAAAwAAAAoBAAoAAAAe"
    assert oracle.zstd_decompress_chunk(frame) == raw


def test_zstd_4mib_header_and_raw_blocks(oracle):
    # 4 MiB chunk -> header 28B52FFD 80 58 00004000; incompressible -> raw blocks, n + 10 + 3*32
    r = synth.gen_chunk("R", 1000, 0, 0)
    f = oracle.zstd_compress_chunk(r.tobytes())
    assert f[:10].hex() == "28b52ffd805800004000" and len(f) == 4194304 + 10 + 96
    assert oracle.zstd_decompress_chunk(f) == r.tobytes()


def test_zstd_invalid_size(oracle):
    with pytest.raises(RuntimeError, match="Invalid decompressed size"):
        oracle.zstd_decompress_chunk(b"\x00" * 20)


@pytest.mark.parametrize("flags", [0, 2, 1, 3])
def test_chain_roundtrip_like_TransformsEndToEndTest(oracle, flags):
    # CT/transform/TransformsEndToEndTest.java:32-117 — detransform(transform(x)) == x, order compress -> encrypt
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, 181200, dtype=np.uint8).tobytes()
    for chunk in [1024, 5123, len(data) - 1, len(data) * 2]:
        parts = [data[i:i + chunk] for i in range(0, len(data), chunk)]
        back = b""
        for i, p in enumerate(parts):
            t, _ = oracle.transform_chunk(flags, synth.KEY, synth.AAD, synth.iv_for(0, i), p)
            if flags & oracle.ENCRYPT and not flags & oracle.COMPRESS:
                assert len(t) == len(p) + 28
            b, _ = oracle.detransform_chunk(flags, synth.KEY, synth.AAD, t)
            back += b
        assert back == data


def test_threaded_chain_matches_single(oracle):
    src = np.concatenate([synth.gen_chunk("K", 1, 0, c, 65536) for c in range(6)])
    ivs = np.frombuffer(b"".join(synth.iv_for(0, c) for c in range(6)), np.uint8).copy()
    flags = oracle.COMPRESS | oracle.ENCRYPT | oracle.CRC | oracle.OPENSSL
    secs, sizes, crcs, dst, stride = oracle.chain_run_threads(flags, synth.KEY, synth.AAD, src, 65536, ivs, 3)
    for c in range(6):
        exp, crc = oracle.transform_chunk(flags & ~oracle.OPENSSL, synth.KEY, synth.AAD, synth.iv_for(0, c), src[c * 65536:(c + 1) * 65536].tobytes())
        assert dst[c * stride:c * stride + sizes[c]].tobytes() == exp and crcs[c] == crc


def test_two_independent_builds_of_libzstd_1_5_7_agree(oracle):
    """The parity library is the libzstd 1.5.7 bundled with Pillow (an unoptimised build: ~30 MiB/s at level 3).  pyarrow carries its
    own, optimised, statically linked 1.5.7: its level-3 frames must be the same bytes - the checker is not an artefact of one build."""
    pa = pytest.importorskip("pyarrow")
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    from tsxform import synth
    rng = np.random.default_rng(5)
    cases = [synth.gen_chunk("K", 5, 0, 0, 1 << 20), synth.gen_chunk("R", 5, 0, 1, 300000), synth.gen_chunk("K", 5, 0, 2, 70001),
             np.concatenate([synth.gen_chunk("K", 6, 0, 0, 200000), synth.gen_chunk("R", 6, 0, 1, 150000), np.zeros(70000, np.uint8), synth.gen_chunk("K", 6, 0, 2, 300000)]),
             rng.integers(0, 4, 500000, dtype=np.uint8), np.frombuffer(bytes.fromhex("000000030000000A01000A0000001E"), np.uint8)]
    codec = pa.Codec("zstd", compression_level=3)
    for c in cases:
        assert codec.compress(c.tobytes(), asbytes=True) == oracle.zstd_compress_chunk(c.tobytes()), c.size
