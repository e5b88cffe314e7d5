"""Deterministic synthetic log-segment content, identical on numpy (CPU) and torch (GPU).

Two distributions, both functions of (seed, segment, chunk, byte position) through a counter-based hash so
that a chunk can be generated anywhere — on the host for the oracle, directly in HBM for bench.py:
  K  Kafka-like compressible: newline-delimited JSON-ish records
         {"id":0000001234,"user":"u00042","event":"click","ts":1700000000123,"payload":"<8-63 x a-z>"}
     every chunk starts on a record boundary; the last record of a chunk is cut at the chunk end.
  R  incompressible: uniform random bytes (what the reference's own tests use,
     core/src/test/java/io/aiven/kafka/tieredstorage/transform/TransformsEndToEndTest.java:38-42).
Crypto material (deterministic for parity; production takes IVs from SecureRandom):
  key = bytes 0..31, aad = bytes 32..63, IV(segment s, chunk c) = BE32(s) || BE64(c).
"""
import numpy as np

CHUNK = 4 << 20
KEY = bytes(range(32))
AAD = bytes(range(32, 64))

_PREFIX = b'{"id":0000000000,"user":"u00000","event":"click","ts":0000000000000,"payload":"'
_SUFFIX = b'"}\n'
_EVENTS = [b"click", b"views", b"login", b"query"]
_F = len(_PREFIX)
_ID_POS = _PREFIX.index(b"0000000000")
_USER_POS = _PREFIX.index(b"u00000") + 1
_EVENT_POS = _PREFIX.index(b"click")
_TS_POS = _PREFIX.index(b"0000000000000")
_MASK63 = (1 << 63) - 1


def iv_for(segment: int, chunk: int) -> bytes:
    return int(segment).to_bytes(4, "big") + int(chunk).to_bytes(8, "big")


class _NP:
    """numpy flavour of the few array ops the generator needs (int64 everywhere)."""
    int64 = np.int64

    @staticmethod
    def arange(n): return np.arange(n, dtype=np.int64)
    @staticmethod
    def full(n, v): return np.full(n, v, dtype=np.int64)
    @staticmethod
    def cumsum(a): return np.cumsum(a, dtype=np.int64)
    @staticmethod
    def searchsorted_right(a, v): return np.searchsorted(a, v, side="right").astype(np.int64)
    @staticmethod
    def where(c, a, b): return np.where(c, a, b)
    @staticmethod
    def const(b): return np.frombuffer(b, dtype=np.uint8).astype(np.int64)
    @staticmethod
    def to_u8(a): return a.astype(np.uint8)
    @staticmethod
    def concat(xs): return np.concatenate(xs)
    @staticmethod
    def zeros(n): return np.zeros(n, dtype=np.int64)


def _torch_ops(device):
    import torch

    class _T:
        int64 = torch.int64
        @staticmethod
        def arange(n): return torch.arange(n, dtype=torch.int64, device=device)
        @staticmethod
        def full(n, v): return torch.full((n,), v, dtype=torch.int64, device=device)
        @staticmethod
        def cumsum(a): return torch.cumsum(a, 0)
        @staticmethod
        def searchsorted_right(a, v): return torch.searchsorted(a, v, right=True)
        @staticmethod
        def where(c, a, b): return torch.where(c, a, b)
        @staticmethod
        def const(b): return torch.tensor(list(b), dtype=torch.int64, device=device)
        @staticmethod
        def to_u8(a): return a.to(torch.uint8)
        @staticmethod
        def concat(xs): return torch.cat(xs)
        @staticmethod
        def zeros(n): return torch.zeros(n, dtype=torch.int64, device=device)
    return _T


def _mix(x, k):
    """splitmix64-style finaliser on int64 with wrap-around (logical shifts emulated with masks)."""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    C1, C2 = -4658895280553007687, -7723592293110705685      # 0xBF58476D1CE4E5B9, 0x94D049BB133111EB as int64
    x = x + k
    x = (x ^ lsr(x, 30)) * C1
    x = (x ^ lsr(x, 27)) * C2
    x = x ^ lsr(x, 31)
    return x & _MASK63


def _chunk_key(seed, segment, chunk):
    k = (seed * 0x9E3779B97F4A7C15 + segment * 0x632BE59BD9B4E019 + chunk * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & _MASK63
    return int(k)


def _gen_random(xp, n, key):
    pos = xp.arange((n + 7) // 8)
    w = _mix(pos, key)                       # 63 random bits per word; take 8 bytes from two mixes for full range
    w2 = _mix(pos, key ^ 0x5555555555555555)
    parts = []
    for b in range(4):
        parts.append((w >> (8 * b + 3)) & 0xFF)
    for b in range(4):
        parts.append((w2 >> (8 * b + 3)) & 0xFF)
    if xp is _NP:
        out = np.stack(parts, axis=1).reshape(-1)[:n]
    else:
        import torch
        out = torch.stack(parts, dim=1).reshape(-1)[:n]
    return xp.to_u8(out)


def _gen_kafka(xp, n, key, segment, chunk):
    # upper bound of records in the chunk: shortest record is F + 8 + 3 bytes
    max_rec = n // (_F + 8 + len(_SUFFIX)) + 2
    rec = xp.arange(max_rec)
    plen = 8 + (_mix(rec, key ^ 0x1111) % 56)                        # 8..63
    rlen = plen + (_F + len(_SUFFIX))
    ends = xp.cumsum(rlen)                                           # end offset (exclusive) of each record
    pos = xp.arange(n)
    r = xp.searchsorted_right(ends, pos)                             # record index of every byte
    starts = ends - rlen
    col = pos - starts[r]
    pl = plen[r]
    # record fields
    rid = (segment * 1000003 + chunk * 50000 + r) % 10000000000
    user = _mix(r, key ^ 0x2222) % 1000
    ev = _mix(r, key ^ 0x3333) % 4
    ts = 1700000000000 + (segment * 256 + chunk) * 100000 + r * 37 + (_mix(r, key ^ 0x4444) % 29)
    prefix = xp.const(_PREFIX)
    suffix = xp.const(_SUFFIX)
    events = xp.const(b"".join(_EVENTS))
    cc = xp.where(col < _F, col, xp.zeros(n))
    out = prefix[cc]

    def digits(out, value, start, width):
        k = col - start
        inside = (k >= 0) & (k < width)
        kk = xp.where(inside, k, xp.zeros(n))
        p10 = xp.full(n, 1)
        # 10^(width-1-k) without pow: iterative (width <= 13)
        e = (width - 1) - kk
        for bit in range(4):
            p10 = xp.where(((e >> bit) & 1) == 1, p10 * (10 ** (1 << bit)), p10)
        d = (value // p10) % 10 + 48
        return xp.where(inside, d, out)

    out = digits(out, rid, _ID_POS, 10)
    out = digits(out, user, _USER_POS, 5)
    out = digits(out, ts, _TS_POS, 13)
    k = col - _EVENT_POS
    inside = (k >= 0) & (k < 5)
    kk = xp.where(inside, k, xp.zeros(n))
    out = xp.where(inside, events[ev * 5 + kk], out)
    # payload letters and suffix
    in_payload = (col >= _F) & (col < _F + pl)
    letter = 97 + (_mix(pos, key ^ 0x7777) % 26)
    out = xp.where(in_payload, letter, out)
    sk = col - (_F + pl)
    in_suffix = sk >= 0
    skk = xp.where(in_suffix, sk, xp.zeros(n))
    out = xp.where(in_suffix, suffix[skk], out)
    return xp.to_u8(out)


def gen_chunk(dist: str, seed: int, segment: int, chunk: int, n: int = CHUNK, device=None):
    """One chunk of synthetic content.  device None -> numpy uint8 array; else a torch uint8 tensor there."""
    xp = _NP if device is None else _torch_ops(device)
    key = _chunk_key(seed, segment, chunk)
    if n == 0:
        return xp.to_u8(xp.zeros(0))
    if dist == "R":
        return _gen_random(xp, n, key)
    if dist == "K":
        return _gen_kafka(xp, n, key, segment, chunk)
    raise ValueError("dist must be 'K' or 'R'")


def gen_segment(dist: str, seed: int, segment: int, nchunks: int, chunk_size: int = CHUNK, device=None):
    xp = _NP if device is None else _torch_ops(device)
    return xp.concat([gen_chunk(dist, seed, segment, c, chunk_size, device) for c in range(nchunks)])
