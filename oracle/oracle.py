"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package never does (tests/test_boundary.py greps for that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

COMPRESS, ENCRYPT, CRC, OPENSSL = 1, 2, 4, 0x100


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
        L.orc_crc32c.restype = u32; L.orc_crc32c.argtypes = [u8p, sz]
        L.orc_crc32c_bitwise.restype = u32; L.orc_crc32c_bitwise.argtypes = [u8p, sz]
        L.orc_crc32c_update.restype = u32; L.orc_crc32c_update.argtypes = [u32, u8p, sz]
        L.orc_aes256_expand_key.argtypes = [u8p, u8p]
        L.orc_aes256_encrypt_block.argtypes = [u8p, u8p, u8p]
        L.orc_gf128_mul.argtypes = [u8p, u8p]
        L.orc_gf128_mul_tab.argtypes = [u8p, u8p]
        L.orc_aes256_ctr.argtypes = [u8p, u8p, u32, u8p, sz, u8p]
        for name in ("orc_gcm_encrypt_chunk", "orc_gcm_encrypt_chunk_openssl"):
            f = getattr(L, name); f.restype = sz; f.argtypes = [u8p, u8p, u8p, sz, u8p, sz, u8p]
        for name in ("orc_gcm_decrypt_chunk", "orc_gcm_decrypt_chunk_openssl"):
            f = getattr(L, name); f.restype = C.c_long; f.argtypes = [u8p, u8p, sz, u8p, sz, u8p]
        L.orc_zstd_open.restype = C.c_int; L.orc_zstd_open.argtypes = [C.c_char_p]
        L.orc_zstd_version.restype = C.c_char_p
        L.orc_zstd_path.restype = C.c_char_p
        L.orc_zstd_compress_bound.restype = sz; L.orc_zstd_compress_bound.argtypes = [sz]
        L.orc_zstd_compress_chunk.restype = sz; L.orc_zstd_compress_chunk.argtypes = [u8p, sz, u8p, sz, C.c_int]
        L.orc_zstd_decompress_chunk.restype = C.c_longlong; L.orc_zstd_decompress_chunk.argtypes = [u8p, sz, u8p, sz]
        L.orc_zstd_frame_content_size.restype = C.c_longlong; L.orc_zstd_frame_content_size.argtypes = [u8p, sz]
        L.orc_chain_bound.restype = sz; L.orc_chain_bound.argtypes = [sz, C.c_uint]
        L.orc_transform_chunk.restype = sz
        L.orc_transform_chunk.argtypes = [C.c_uint, u8p, u8p, sz, u8p, u8p, sz, u8p, sz, u8p, C.POINTER(u32)]
        L.orc_detransform_chunk.restype = C.c_longlong
        L.orc_detransform_chunk.argtypes = [C.c_uint, u8p, u8p, sz, u8p, sz, u8p, sz, u8p, C.POINTER(u32)]
        L.orc_chain_run_threads.restype = C.c_double
        L.orc_chain_run_threads.argtypes = [C.c_uint, u8p, u8p, sz, u8p, sz, sz, u8p, u8p, sz, u8p, u8p, C.c_int]
        L.orc_chain_set_sink.restype = None; L.orc_chain_set_sink.argtypes = [C.c_char_p]
        L.orc_l3_compress.restype = sz; L.orc_l3_compress.argtypes = [u8p, sz, u8p, sz, C.c_int]
        L.orc_l3_compress_bound.restype = sz; L.orc_l3_compress_bound.argtypes = [sz]
        L.orc_l3_cparams.restype = None; L.orc_l3_cparams.argtypes = [C.c_uint64, C.POINTER(C.c_uint32 * 7)]
        L.orc_l3_set_tap.restype = None; L.orc_l3_set_tap.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _buf(x):
    """numpy uint8 view + pointer of bytes-like input (keeps the array alive for the call)."""
    a = np.frombuffer(x, dtype=np.uint8) if not isinstance(x, np.ndarray) else np.ascontiguousarray(x).view(np.uint8).reshape(-1)
    return a, a.ctypes.data


def crc32c(data) -> int:
    a, p = _buf(data)
    return lib().orc_crc32c(p, a.size)


def crc32c_bitwise(data) -> int:
    a, p = _buf(data)
    return lib().orc_crc32c_bitwise(p, a.size)


def aes256_expand_key(key: bytes) -> bytes:
    rk = np.zeros(240, np.uint8)
    k, kp = _buf(key)
    lib().orc_aes256_expand_key(kp, rk.ctypes.data)
    return rk.tobytes()


def aes256_encrypt_block(key: bytes, block: bytes) -> bytes:
    rk, rkp = _buf(aes256_expand_key(key))
    b, bp = _buf(block)
    out = np.zeros(16, np.uint8)
    lib().orc_aes256_encrypt_block(rkp, bp, out.ctypes.data)
    return out.tobytes()


def aes256_ctr(key: bytes, iv: bytes, ctr0: int, data) -> bytes:
    k, kp = _buf(key); i, ip = _buf(iv); d, dp = _buf(data)
    out = np.zeros(d.size, np.uint8)
    lib().orc_aes256_ctr(kp, ip, ctr0, dp, d.size, out.ctypes.data)
    return out.tobytes()


def gcm_encrypt_chunk(key: bytes, iv: bytes, aad: bytes, pt, openssl=False) -> bytes:
    """EncryptionChunkEnumeration.nextElement: IV || C || TAG."""
    k, kp = _buf(key); i, ip = _buf(iv); a, ap = _buf(aad); d, dp = _buf(pt)
    out = np.zeros(d.size + 28, np.uint8)
    f = lib().orc_gcm_encrypt_chunk_openssl if openssl else lib().orc_gcm_encrypt_chunk
    r = f(kp, ip, ap, a.size, dp, d.size, out.ctypes.data)
    assert r == d.size + 28
    return out.tobytes()


class BadTag(RuntimeError):
    """javax.crypto.AEADBadTagException wrapped in RuntimeException (DecryptionChunkEnumeration.java:59-61)."""


def gcm_decrypt_chunk(key: bytes, aad: bytes, chunk, openssl=False) -> bytes:
    k, kp = _buf(key); a, ap = _buf(aad); d, dp = _buf(chunk)
    out = np.zeros(max(d.size - 28, 0), np.uint8)
    f = lib().orc_gcm_decrypt_chunk_openssl if openssl else lib().orc_gcm_decrypt_chunk
    r = f(kp, ap, a.size, dp, d.size, out.ctypes.data)
    if r == -1:
        raise BadTag("Tag mismatch")
    if r < 0:
        raise RuntimeError("chunk shorter than IV+TAG")
    return out.tobytes()


def zstd_open(path=None) -> bool:
    return lib().orc_zstd_open(path.encode() if path else None) == 0


def zstd_version() -> str:
    return lib().orc_zstd_version().decode()


def zstd_compress_bound(n: int) -> int:
    return lib().orc_zstd_compress_bound(n)


def zstd_compress_chunk(data, level=0) -> bytes:
    """CompressionChunkEnumeration.nextElement."""
    d, dp = _buf(data)
    cap = zstd_compress_bound(d.size)
    out = np.zeros(cap, np.uint8)
    r = lib().orc_zstd_compress_chunk(dp, d.size, out.ctypes.data, cap, level)
    if r == C.c_size_t(-1).value:
        raise RuntimeError("zstd compress failed")
    return out[:r].tobytes()


def zstd_decompress_chunk(frame, cap=None) -> bytes:
    """DecompressionChunkEnumeration.nextElement."""
    d, dp = _buf(frame)
    size = lib().orc_zstd_frame_content_size(dp, d.size)
    if size < 0:
        raise RuntimeError("Invalid decompressed size: %d" % size)
    out = np.zeros(max(size, 1), np.uint8)
    r = lib().orc_zstd_decompress_chunk(dp, d.size, out.ctypes.data, size)
    if r < 0:
        raise RuntimeError("zstd decompress failed: %d" % r)
    return out[:r].tobytes()


def transform_chunk(flags, key, aad, iv, data):
    """One chunk through the reference chain.  Returns (transformed bytes, crc32c of original or None)."""
    out = bytes(data) if not isinstance(data, bytes) else data
    crc = crc32c(out) if flags & CRC else None
    if flags & COMPRESS:
        out = zstd_compress_chunk(out)
    if flags & ENCRYPT:
        out = gcm_encrypt_chunk(key, iv, aad, out, openssl=bool(flags & OPENSSL))
    return out, crc


def detransform_chunk(flags, key, aad, data):
    out = bytes(data) if not isinstance(data, bytes) else data
    if flags & ENCRYPT:
        out = gcm_decrypt_chunk(key, aad, out, openssl=bool(flags & OPENSSL))
    if flags & COMPRESS:
        out = zstd_decompress_chunk(out)
    crc = crc32c(out) if flags & CRC else None
    return out, crc


def chain_run_threads(flags, key, aad, src: np.ndarray, chunk: int, ivs: np.ndarray, nthreads: int, sink=None):
    """cpu_baseline leg: returns (seconds, sizes, crcs, dst, stride).  sink: path of a file that also receives every transformed chunk
    (BASELINE configs[0] "to filesystem backend": tmpfs on the bench box); None = in memory only."""
    lib().orc_chain_set_sink(sink.encode() if sink else None)
    n = src.size // chunk
    stride = lib().orc_chain_bound(chunk, flags)
    dst = np.empty(n * stride, np.uint8)
    dst[::4096] = 0                                     # pre-fault the output pages: the baseline times the chain, not the kernel's page faults
    sizes = np.zeros(n, np.uint32); crcs = np.zeros(n, np.uint32)
    k, kp = _buf(key); a, ap = _buf(aad)
    secs = lib().orc_chain_run_threads(flags, kp, ap, a.size, src.ctypes.data, chunk, n, ivs.ctypes.data,
                                       dst.ctypes.data, stride, sizes.ctypes.data, crcs.ctypes.data, nthreads)
    return secs, sizes, crcs, dst, stride


def zstd_l3_compress(data, profile=1) -> bytes:
    """oracle/zstd_l3.c: serial restatement of libzstd's level-3 one-shot compressor (profile 0 = 1.5.6, 1 = 1.5.7)."""
    d, dp = _buf(data)
    cap = lib().orc_l3_compress_bound(d.size) + 64
    out = np.zeros(cap, np.uint8)
    r = lib().orc_l3_compress(dp, d.size, out.ctypes.data, cap, profile)
    if r == 0:
        raise RuntimeError("zstd_l3 failed")
    return out[:r].tobytes()


def zstd_l3_cparams(n: int):
    a = (C.c_uint32 * 7)()
    lib().orc_l3_cparams(n, C.byref(a))
    return tuple(a)


_SEQ_TAP = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t, C.c_size_t)


def zstd_l3_sequences(data, profile=1):
    """Per block: (list of (litLength, matchLength, offBase), litSize, blockSize) as found by the restatement."""
    blocks = []

    def tap(ctx, bi, seqs, nb, lit, bs):
        arr = np.ctypeslib.as_array(seqs, shape=(nb * 3,)).reshape(-1, 3).copy() if nb else np.zeros((0, 3), np.uint32)
        blocks.append(([(int(r[1]), int(r[2]) + 3, int(r[0])) for r in arr], int(lit), int(bs)))
    cb = _SEQ_TAP(tap)
    lib().orc_l3_set_tap(C.cast(cb, C.c_void_p), None)
    try:
        zstd_l3_compress(data, profile)
    finally:
        lib().orc_l3_set_tap(None, None)
    return blocks
