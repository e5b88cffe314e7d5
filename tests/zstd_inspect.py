"""Independent Zstandard frame parser/decoder (RFC 8878) in plain Python — a debugging and test aid.

It decodes a frame all the way down (block layout, literal-section modes, Huffman weights, FSE table
descriptions, every sequence) so that two frames — ours and libzstd's — can be compared structurally when
their bytes differ, and so that tests can assert properties of the frames the HIP compressor emits.
Not part of the product and not the oracle: the oracle for Zstd is the real libzstd (oracle/zstd_ref.c).
"""
from dataclasses import dataclass, field

LL_BASE = list(range(16)) + [16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
LL_BITS = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
ML_BASE = list(range(3, 35)) + [35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539]
ML_BITS = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
LL_DEFAULT = [4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1]
OF_DEFAULT = [1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1]
ML_DEFAULT = [1, 4, 3, 2, 2, 2, 2, 2, 2] + [1] * 37 + [-1] * 7


class BackBits:
    """Backward bitstream reader: the last byte holds a final 1-bit marker."""

    def __init__(self, data: bytes):
        if not data or data[-1] == 0:
            raise ValueError("bad backward bitstream")
        self.v = int.from_bytes(data, "little")
        self.pos = len(data) * 8 - (10 - data[-1].bit_length())  # index of the bit just below the marker

    def read(self, n):
        """n bits, most significant first from the current position; zero-filled past the start."""
        if n == 0:
            return 0
        self.pos -= n
        lo = self.pos + 1
        if lo >= 0:
            return (self.v >> lo) & ((1 << n) - 1)
        have = lo + n
        return ((self.v & ((1 << have) - 1)) << (n - have)) if have > 0 else 0

    def bits_left(self):
        return self.pos + 1


class FwdBits:
    def __init__(self, data, off):
        self.d = data; self.bitpos = off * 8

    def peek(self, n):
        v = 0
        for i in range(4):
            b = self.bitpos // 8 + i
            v |= (self.d[b] if b < len(self.d) else 0) << (8 * i)
        return (v >> (self.bitpos % 8)) & ((1 << n) - 1)

    def skip(self, n):
        self.bitpos += n


def read_ncount(data, off, max_symbol):
    """FSE table description -> (norm counts list, accuracy log, bytes consumed)."""
    fb = FwdBits(data, off)
    al = fb.peek(4) + 5; fb.skip(4)
    remaining = (1 << al) + 1
    threshold = 1 << al
    nbits = al + 1
    norm = []
    prev0 = False
    while remaining > 1 and len(norm) <= max_symbol:
        if prev0:
            while True:
                r = fb.peek(2); fb.skip(2)
                norm += [0] * r
                if r != 3:
                    break
            prev0 = False
            continue
        mx = (2 * threshold - 1) - remaining
        lo = fb.peek(nbits - 1)
        if lo < mx:
            count = lo; fb.skip(nbits - 1)
        else:
            count = fb.peek(nbits)
            if count >= threshold:
                count -= mx
            fb.skip(nbits)
        count -= 1
        remaining -= abs(count)
        norm.append(count)
        prev0 = count == 0
        while remaining < threshold:
            nbits -= 1; threshold >>= 1
    if remaining != 1:
        raise ValueError("bad NCount (remaining=%d)" % remaining)
    return norm, al, (fb.bitpos + 7) // 8 - off


def build_fse_dtable(norm, al):
    size = 1 << al
    sym = [0] * size
    high = size - 1
    nxt = []
    for s, c in enumerate(norm):
        if c == -1:
            sym[high] = s; high -= 1; nxt.append(1)
        else:
            nxt.append(c)
    step = (size >> 1) + (size >> 3) + 3
    pos = 0
    for s, c in enumerate(norm):
        for _ in range(max(c, 0)):
            sym[pos] = s
            pos = (pos + step) & (size - 1)
            while pos > high:
                pos = (pos + step) & (size - 1)
    table = []
    for u in range(size):
        s = sym[u]
        ns = nxt[s]; nxt[s] += 1
        nb = al - (ns.bit_length() - 1)
        table.append((s, nb, (ns << nb) - size))
    return table


@dataclass
class Block:
    last: bool
    btype: str                 # raw / rle / compressed
    size: int                  # header size field
    regen: int = 0             # decompressed size of the block
    lit_type: str = ""
    lit_regen: int = 0
    lit_csize: int = 0
    lit_streams: int = 0
    huf_weights: list = field(default_factory=list)
    huf_header: str = ""       # 'fse' / 'direct' / ''
    nbseq: int = 0
    modes: tuple = ()
    tables: dict = field(default_factory=dict)    # 'll'/'of'/'ml' -> (norm, al) for compressed mode
    seqs: list = field(default_factory=list)      # (ll, ml, offBase)
    literals: bytes = b""


class FrameState:
    def __init__(self):
        self.huf = None
        self.fse = {"ll": None, "of": None, "ml": None}
        self.rep = [1, 4, 8]


def _huf_dtable(weights):
    total = sum((1 << (w - 1)) for w in weights if w)
    max_bits = total.bit_length()
    left = (1 << max_bits) - total
    if left & (left - 1) or left == 0:
        raise ValueError("bad huffman weights")
    last_w = left.bit_length()
    weights = weights + [last_w]
    size = 1 << max_bits
    table = [None] * size
    rank_start = [0] * (max_bits + 2)
    cnt = [0] * (max_bits + 2)
    for w in weights:
        cnt[w] += 1
    nxt = 0
    for w in range(1, max_bits + 1):
        rank_start[w] = nxt
        nxt += cnt[w] << (w - 1)
    for s, w in enumerate(weights):
        if not w:
            continue
        length = (1 << w) >> 1
        nb = max_bits + 1 - w
        for i in range(rank_start[w], rank_start[w] + length):
            table[i] = (s, nb)
        rank_start[w] += length
    return table, max_bits, weights


def _huf_decode_stream(data, table, max_bits, count):
    bb = BackBits(data)
    out = bytearray()
    for _ in range(count):
        # peek max_bits (zero padded at the start of the stream)
        save = bb.pos
        v = bb.read(max_bits)
        s, nb = table[v]
        bb.pos = save - nb
        out.append(s)
    if bb.pos != -1:
        raise ValueError("huffman stream not fully consumed (pos=%d)" % bb.pos)
    return bytes(out)


def _read_huf_weights(data, off):
    hb = data[off]
    if hb >= 128:
        n = hb - 127
        raw = data[off + 1: off + 1 + (n + 1) // 2]
        w = []
        for i in range(n):
            b = raw[i // 2]
            w.append(b >> 4 if i % 2 == 0 else b & 15)
        return w, 1 + (n + 1) // 2, "direct"
    norm, al, used = read_ncount(data, off + 1, 12)
    tab = build_fse_dtable(norm, al)
    bb = BackBits(data[off + 1 + used: off + 1 + hb])
    s1 = bb.read(al); s2 = bb.read(al)
    w = []
    while True:
        sym, nb, base = tab[s1]
        w.append(sym)
        if bb.bits_left() < nb:
            w.append(tab[s2][0]); break
        s1 = base + bb.read(nb)
        sym, nb, base = tab[s2]
        w.append(sym)
        if bb.bits_left() < nb:
            w.append(tab[s1][0]); break
        s2 = base + bb.read(nb)
    return w, 1 + hb, "fse"


def _parse_compressed_block(data, blk, st: FrameState, decode=True):
    p = 0
    b0 = data[0]
    lt = b0 & 3
    sf = (b0 >> 2) & 3
    blk.lit_type = ["raw", "rle", "compressed", "treeless"][lt]
    if lt < 2:
        if sf in (0, 2):
            regen = b0 >> 3; hl = 1
        elif sf == 1:
            regen = (b0 >> 4) + (data[1] << 4); hl = 2
        else:
            regen = (b0 >> 4) + (data[1] << 4) + (data[2] << 12); hl = 3
        blk.lit_regen = regen
        if lt == 0:
            blk.literals = bytes(data[hl:hl + regen]); p = hl + regen
        else:
            blk.literals = bytes([data[hl]]) * regen; p = hl + 1
        blk.lit_streams = 0
    else:
        if sf == 0:
            v = int.from_bytes(data[0:3], "little"); hl = 3; bits = 10; streams = 1
        elif sf == 1:
            v = int.from_bytes(data[0:3], "little"); hl = 3; bits = 10; streams = 4
        elif sf == 2:
            v = int.from_bytes(data[0:4], "little"); hl = 4; bits = 14; streams = 4
        else:
            v = int.from_bytes(data[0:5], "little"); hl = 5; bits = 18; streams = 4
        regen = (v >> 4) & ((1 << bits) - 1)
        csize = v >> (4 + bits)
        blk.lit_regen, blk.lit_csize, blk.lit_streams = regen, csize, streams
        q = hl
        if lt == 2:
            w, used, kind = _read_huf_weights(data, q)
            blk.huf_header = kind
            table, mb, wfull = _huf_dtable(w)
            blk.huf_weights = wfull
            st.huf = (table, mb)
            q += used
        elif st.huf is None:
            raise ValueError("treeless literals without a previous table")
        table, mb = st.huf
        payload = data[q: hl + csize]
        if decode:
            if streams == 1:
                blk.literals = _huf_decode_stream(payload, table, mb, regen)
            else:
                s1, s2, s3 = (int.from_bytes(payload[i:i + 2], "little") for i in (0, 2, 4))
                seg = (regen + 3) // 4
                parts = [payload[6:6 + s1], payload[6 + s1:6 + s1 + s2], payload[6 + s1 + s2:6 + s1 + s2 + s3], payload[6 + s1 + s2 + s3:]]
                counts = [seg, seg, seg, regen - 3 * seg]
                blk.literals = b"".join(_huf_decode_stream(pp, table, mb, c) for pp, c in zip(parts, counts))
        p = hl + csize
    # sequences section
    b = data[p]
    if b == 0:
        nbseq = 0; p += 1
    elif b < 128:
        nbseq = b; p += 1
    elif b < 255:
        nbseq = ((b - 128) << 8) + data[p + 1]; p += 2
    else:
        nbseq = data[p + 1] + (data[p + 2] << 8) + 0x7F00; p += 3
    blk.nbseq = nbseq
    if nbseq == 0:
        if p != len(data):
            raise ValueError("trailing bytes after empty sequence section")
        return
    modes = data[p]; p += 1
    m = ((modes >> 6) & 3, (modes >> 4) & 3, (modes >> 2) & 3)
    blk.modes = tuple(["predefined", "rle", "compressed", "repeat"][x] for x in m)
    tabs = {}
    for name, mode, default, dal, maxsym in (("ll", m[0], LL_DEFAULT, 6, 35), ("of", m[1], OF_DEFAULT, 5, 31), ("ml", m[2], ML_DEFAULT, 6, 52)):
        if mode == 0:
            tabs[name] = (build_fse_dtable(default, dal), dal)
        elif mode == 1:
            tabs[name] = ([(data[p], 0, 0)], 0); blk.tables[name] = ("rle", data[p]); p += 1
        elif mode == 2:
            norm, al, used = read_ncount(data, p, maxsym)
            blk.tables[name] = (norm, al)
            tabs[name] = (build_fse_dtable(norm, al), al); p += used
        else:
            if st.fse[name] is None:
                raise ValueError("repeat mode without previous table")
            tabs[name] = st.fse[name]
        st.fse[name] = tabs[name]
    if not decode:
        return
    bb = BackBits(data[p:])
    (llt, lla), (oft, ofa), (mlt, mla) = tabs["ll"], tabs["of"], tabs["ml"]
    sl = bb.read(lla); so = bb.read(ofa); sm = bb.read(mla)
    seqs = []
    for i in range(nbseq):
        oc = oft[so][0]; mc = mlt[sm][0]; lc = llt[sl][0]
        offbase = (1 << oc) + bb.read(oc)
        ml = ML_BASE[mc] + bb.read(ML_BITS[mc])
        ll = LL_BASE[lc] + bb.read(LL_BITS[lc])
        seqs.append((ll, ml, offbase))
        if i != nbseq - 1:
            _, nb, base = llt[sl]; sl = base + bb.read(nb)
            _, nb, base = mlt[sm]; sm = base + bb.read(nb)
            _, nb, base = oft[so]; so = base + bb.read(nb)
    if bb.pos != -1:
        raise ValueError("sequence bitstream not fully consumed (pos=%d)" % bb.pos)
    blk.seqs = seqs


def parse_frame(frame: bytes, decode=True):
    """-> (header dict, [Block], decoded bytes or None)."""
    if frame[:4] != b"\x28\xb5\x2f\xfd":
        raise ValueError("bad magic")
    fhd = frame[4]
    p = 5
    single = (fhd >> 5) & 1
    hdr = {"fhd": fhd, "single_segment": bool(single), "checksum": bool(fhd & 4), "dict_flag": fhd & 3}
    if not single:
        wd = frame[p]; p += 1
        hdr["window_log"] = 10 + (wd >> 3); hdr["window_mantissa"] = wd & 7
    p += [0, 1, 2, 4][fhd & 3]
    fcs_flag = fhd >> 6
    fcs_len = [1 if single else 0, 2, 4, 8][fcs_flag]
    if fcs_len:
        v = int.from_bytes(frame[p:p + fcs_len], "little")
        hdr["content_size"] = v + 256 if fcs_len == 2 else v
        p += fcs_len
    hdr["header_size"] = p
    st = FrameState()
    blocks = []
    out = bytearray()
    while True:
        h = int.from_bytes(frame[p:p + 3], "little"); p += 3
        last, bt, size = h & 1, (h >> 1) & 3, h >> 3
        blk = Block(bool(last), ["raw", "rle", "compressed", "reserved"][bt], size)
        if bt == 0:
            blk.regen = size
            if decode:
                out += frame[p:p + size]
            p += size
        elif bt == 1:
            blk.regen = size
            if decode:
                out += frame[p:p + 1] * size
            p += 1
        elif bt == 2:
            _parse_compressed_block(frame[p:p + size], blk, st, decode)
            p += size
            if decode:
                start = len(out)
                lp = 0
                for ll, ml, ob in blk.seqs:
                    out += blk.literals[lp:lp + ll]; lp += ll
                    if ob > 3:
                        off = ob - 3; st.rep = [off, st.rep[0], st.rep[1]]
                    else:
                        idx = ob - 1 + (1 if ll == 0 else 0)
                        if idx == 0:
                            off = st.rep[0]
                        elif idx == 1:
                            off = st.rep[1]; st.rep = [off, st.rep[0], st.rep[2]]
                        elif idx == 2:
                            off = st.rep[2]; st.rep = [off, st.rep[0], st.rep[1]]
                        else:
                            off = st.rep[0] - 1; st.rep = [off, st.rep[0], st.rep[1]]
                    if off > len(out) or off <= 0:
                        raise ValueError("offset %d beyond output (%d)" % (off, len(out)))
                    for _ in range(ml):
                        out.append(out[-off])
                out += blk.literals[lp:]
                blk.regen = len(out) - start
        else:
            raise ValueError("reserved block type")
        blocks.append(blk)
        if last:
            break
    hdr["frame_size"] = p
    return hdr, blocks, (bytes(out) if decode else None)


def summarize(frame: bytes):
    hdr, blocks, _ = parse_frame(frame, decode=True)
    lines = ["header %s" % hdr]
    for i, b in enumerate(blocks):
        lines.append("block %d: %s size=%d regen=%d last=%s lit=%s(%d->%d,x%d,%s) nbseq=%d modes=%s" % (
            i, b.btype, b.size, b.regen, b.last, b.lit_type, b.lit_regen, b.lit_csize, b.lit_streams, b.huf_header, b.nbseq, b.modes))
    return "\n".join(lines)
