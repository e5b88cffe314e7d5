"""Deterministic synthetic log-segment content, identical on numpy (CPU) and torch (GPU).

Three distributions, all functions of (seed, segment, chunk, byte position) through a counter-based hash so
that a chunk can be generated anywhere — on the host for the oracle, directly in HBM for bench.py (B: on the host, then uploaded):
  K  Kafka-like compressible: newline-delimited JSON-ish records
         {"id":0000001234,"user":"u00042","event":"click","ts":1700000000123,"payload":"<8-63 x a-z>"}
     every chunk starts on a record boundary; the last record of a chunk is cut at the chunk end.
  B  Kafka-shaped binary: back-to-back v2 RECORD BATCHES as a broker's log segment holds them (SURVEY.md 8d "v2 record batches") -
     61-byte header (baseOffset, batchLength, magic 2, CRC32C over attributes .. end, timestamps, producer fields, record count), 8 - 63
     records each: varint-framed (length, attributes, timestamp / offset deltas, 8-byte binary key, value, no headers); a value is a 24-byte
     binary struct + JSON-ish text of 40 - 500 bytes, and one record in 32 carries a 1 - 6 KiB incompressible payload (a producer that
     compressed or encrypted its own messages).  Every chunk starts on a batch boundary; the last batch is cut at the chunk end.
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


_B_TEMPLATE = (b'{"orderId":"A0000000","sku":"ZX-44110","qty":3,"price":"19.99","currency":"EUR","status":"SHIPPED","warehouse":"FRA-2",'
               b'"customer":{"id":"c-000000","tier":"gold"},"notes":"')
_CRC32C_TABLE = None


def _crc32c_table():
    global _CRC32C_TABLE
    if _CRC32C_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
        _CRC32C_TABLE = t
    return _CRC32C_TABLE


_CRC_W = 256
_CRC_ZW = None


def _crc32c_many(buf, start, length):
    """CRC32C of buf[start[i] : start[i] + length[i]] for every i (uint32 array; lengths >= 4).  The recurrence is byte-serial, so the
    regions are cut into blocks of _CRC_W bytes aligned to their ENDS (zeros in front of a message do not change a CRC that starts from
    0; the initial value 0xFFFFFFFF is the same as complementing the first four bytes): all blocks of all regions advance in lockstep,
    then each region folds its blocks front to back with the "append _CRC_W zero bytes" operator (linear: four 256-entry tables)."""
    global _CRC_ZW
    T = _crc32c_table()
    W = _CRC_W
    if _CRC_ZW is None:
        z = np.concatenate([np.arange(256, dtype=np.uint32) << np.uint32(8 * k) for k in range(4)])
        for _ in range(W):
            z = T[z & 0xFF] ^ (z >> 8)
        _CRC_ZW = z.reshape(4, 256)
    start = np.asarray(start, np.int64); length = np.asarray(length, np.int64)
    nblk = (length + W - 1) // W
    first = np.cumsum(nblk) - nblk                                    # index of a region's first (front, possibly short) block
    tot = int(nblk.sum())
    reg = np.repeat(np.arange(start.size), nblk)
    j = np.arange(tot) - first[reg]                                   # block index inside its region, front to back
    bend_ = (start + length)[reg] - (nblk[reg] - 1 - j) * W           # one past the block's last byte
    lo = start[reg]
    s = np.zeros(tot, np.uint32)
    head = np.zeros(tot, bool); head[first] = True
    for c in range(W):
        p = bend_ - W + c
        ok = p >= lo
        byte = np.where(ok, buf[np.where(ok, p, 0)], 0).astype(np.uint32)
        byte = np.where(ok & (p < lo + 4), byte ^ 0xFF, byte)         # the initial value
        s = T[(s ^ byte) & 0xFF] ^ (s >> 8)
    Z = _CRC_ZW
    crc = s[first].copy()
    for k in range(1, int(nblk.max())):
        act = k < nblk
        nxt = Z[0][crc & 0xFF] ^ Z[1][(crc >> 8) & 0xFF] ^ Z[2][(crc >> 16) & 0xFF] ^ Z[3][crc >> 24] ^ s[np.where(act, first + k, 0)]
        crc = np.where(act, nxt, crc)
    return ~crc


def _be(value, width):
    """(len(value), width) uint8: big-endian bytes of an int64 array (two's complement)."""
    v = value.astype(np.int64)
    return np.stack([((v >> (8 * (width - 1 - k))) & 0xFF) for k in range(width)], axis=1).astype(np.uint8)


def record_batches_of(buf):
    """(start, length) of every COMPLETE v2 record batch in a B chunk (numpy uint8): what a reader of the segment walks."""
    out, p, n = [], 0, len(buf)
    while p + 61 <= n:
        blen = int.from_bytes(bytes(buf[p + 8:p + 12]), "big") + 12
        if p + blen > n:
            break
        out.append((p, blen)); p += blen
    return out


def _gen_record_batches(n, key, segment, chunk):
    """numpy only (the per-batch CRC32C is a byte-serial recurrence: vectorised across the chunk's batches, not along them)."""
    xp = _NP
    B = n // (61 + 8 * 81) + 2                                       # upper bound of batches: the shortest has 8 records of 64 + 17 bytes
    b = xp.arange(B)
    nrec = 8 + (_mix(b, key ^ 0xB001) % 56)                          # 8..63 records per batch
    rec_base = np.cumsum(nrec) - nrec                                # index of a batch's first record among the chunk's records
    R = int(nrec.sum())
    r = xp.arange(R)
    rb = np.searchsorted(np.cumsum(nrec), r, side="right").astype(np.int64)      # batch of every record
    ri = r - rec_base[rb]                                            # its index inside the batch
    h = _mix(r, key ^ 0xB002)
    big = (h % 32) == 0
    vlen = np.where(big, 1024 + (_mix(r, key ^ 0xB003) % 5120), 64 + (_mix(r, key ^ 0xB004) % 460)).astype(np.int64)
    rtot = vlen + 17                                                 # length varint (2) + body (vlen + 15)
    blen = 61 + np.bincount(rb, weights=rtot, minlength=B).astype(np.int64)
    bend = np.cumsum(blen); bstart = bend - blen
    rec_off = np.cumsum(rtot) - rtot - (np.cumsum(np.bincount(rb, weights=rtot, minlength=B).astype(np.int64)) - np.bincount(rb, weights=rtot, minlength=B).astype(np.int64))[rb]
    rstart = bstart[rb] + 61 + rec_off                               # position of every record in the chunk
    rend = rstart + rtot
    pos = xp.arange(n)
    pb = np.minimum(np.searchsorted(bend, pos, side="right"), B - 1).astype(np.int64)
    ob = pos - bstart[pb]
    # ---- headers ----
    base_off = (segment * 1000003 + chunk * 70001) * 64 + rec_base
    base_ts = 1700000000000 + (segment * 256 + chunk) * 100000 + rec_base * 3
    hdr = np.zeros((B, 61), np.uint8)
    hdr[:, 0:8] = _be(base_off, 8)
    hdr[:, 8:12] = _be(blen - 12, 4)                                 # batchLength: everything behind this field
    hdr[:, 16] = 2                                                   # magic (partitionLeaderEpoch 12..15 = 0; crc 17..20 below; attributes 21..22 = 0: no compression)
    hdr[:, 23:27] = _be(nrec - 1, 4)                                 # lastOffsetDelta
    hdr[:, 27:35] = _be(base_ts, 8)
    hdr[:, 35:43] = _be(base_ts + nrec - 1, 8)
    hdr[:, 43:57] = 0xFF                                             # producerId, producerEpoch, baseSequence = -1
    hdr[:, 57:61] = _be(nrec, 4)
    out = np.zeros(n, np.int64)
    in_hdr = ob < 61
    out[in_hdr] = hdr[pb[in_hdr], ob[in_hdr]]
    # ---- records ----
    pr = np.minimum(np.searchsorted(rend, pos, side="right"), R - 1).astype(np.int64)
    col = pos - rstart[pr]
    inrec = (~in_hdr) & (col >= 0)
    L = vlen[pr] + 15
    zzL = 2 * L; zzV = 2 * vlen[pr]
    v = col - 16
    vl = vlen[pr]
    ts = base_ts[rb][pr] + ri[pr]
    body = np.zeros(n, np.int64)
    body = np.where(col == 0, (zzL & 0x7F) | 0x80, body)
    body = np.where(col == 1, zzL >> 7, body)
    body = np.where(col == 3, 2 * ri[pr], body)                      # timestampDelta (zigzag, one byte: < 64 records)
    body = np.where(col == 4, 2 * ri[pr], body)                      # offsetDelta
    body = np.where(col == 5, 16, body)                              # keyLength 8
    kk = col - 6
    kid = (_mix(pr, key ^ 0xB005) % 5000) + 100000 * (segment % 7)
    body = np.where((kk >= 0) & (kk < 8), (kid >> (8 * np.clip(7 - kk, 0, 7))) & 0xFF, body)
    body = np.where(col == 14, (zzV & 0x7F) | 0x80, body)
    body = np.where(col == 15, zzV >> 7, body)
    # value: 24-byte struct, then text - or, for the big records, incompressible bytes
    sv = np.clip(v, 0, None)
    struct = np.where(sv < 8, (ts >> (8 * np.clip(sv, 0, 7))) & 0xFF,
             np.where(sv < 12, ((pr % 7) >> (8 * np.clip(sv - 8, 0, 3))) & 0xFF,
             np.where(sv < 16, ((ri[pr] * 13 + 5) >> (8 * np.clip(sv - 12, 0, 3))) & 0xFF, (0x0101000000000001 >> (8 * np.clip(sv - 16, 0, 7))) & 0xFF)))
    tmpl = np.frombuffer(_B_TEMPLATE, np.uint8).astype(np.int64)
    tpos = (np.clip(sv - 24, 0, None) + (h[pr] % 7)) % len(tmpl)
    letter = np.where((_mix(pos, key ^ 0xB006) % 5) == 0, 97 + (_mix(pos, key ^ 0xB007) % 26), tmpl[tpos])
    text = np.where(sv < 24, struct, letter)
    rnd = _mix(pos, key ^ 0xB008) & 0xFF
    in_val = (v >= 0) & (v < vl)
    body = np.where(in_val, np.where(big[pr], rnd, text), body)     # (the last byte of a record, headers count, stays 0; attributes at col 2 too)
    out = np.where(inrec, body, out)
    out = out.astype(np.uint8)
    # ---- CRC32C of every complete batch over [attributes .. end) -> header bytes 17..20 ----
    full = bend <= n
    idx = np.nonzero(full)[0]
    if idx.size:
        crc = _crc32c_many(out, bstart[idx] + 21, blen[idx] - 21)
        cb = _be(crc.astype(np.int64), 4)
        for k in range(4):
            out[bstart[idx] + 17 + k] = cb[:, k]
    return out


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
    if dist == "B":
        host = _gen_record_batches(n, key, segment, chunk)
        if device is None:
            return host
        import torch
        return torch.from_numpy(host).to(device)
    raise ValueError("dist must be 'K', 'B' or 'R'")


def gen_segment(dist: str, seed: int, segment: int, nchunks: int, chunk_size: int = CHUNK, device=None):
    xp = _NP if device is None else _torch_ops(device)
    return xp.concat([gen_chunk(dist, seed, segment, c, chunk_size, device) for c in range(nchunks)])
