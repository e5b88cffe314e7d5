"""Shared parity checks: the same assertions run against the emulated kernels (CPU, small) and the real
library (-m gpu).  `N` is a tsxform._native.Native; `o` is the oracle module."""
import numpy as np

import tsxform
from tsxform import synth

nat = tsxform._native


def layout(sizes, flags, N, slack=0):
    """16-byte aligned src/dst slots for a list of chunk sizes."""
    soff, doff, st, dt = [], [], 0, 0
    caps = []
    for s in sizes:
        soff.append(st); st += (s + 15) // 16 * 16 + 16
        cap = N.transformed_bound(s, flags) + slack
        caps.append(cap)
        doff.append(dt); dt += (cap + 15) // 16 * 16 + 16
    return soff, doff, caps, st, dt


def make_descs(sizes, soff, doff, caps, segment=0):
    d = np.zeros(len(sizes), nat.DESC_DTYPE)
    d["src_off"] = soff; d["src_len"] = sizes; d["dst_off"] = doff; d["dst_cap"] = caps
    for i in range(len(sizes)):
        d["iv"][i] = np.frombuffer(synth.iv_for(segment, i), np.uint8)
    return d


def run_transform(N, flags, chunks, key=synth.KEY, aad=synth.AAD, mem=None, profile=nat.ZSTD_PROFILE_1_5_7, dst_caps=None):
    """chunks: list of numpy uint8 arrays.  Returns (list of transformed bytes, descs).  dst_caps: per-chunk override of the
    slot capacity handed to the library (None = the library's own bound)."""
    sizes = [int(c.size) for c in chunks]
    soff, doff, caps, st, dt = layout(sizes, flags, N)
    if dst_caps:
        caps = [c if o_ is None else o_ for c, o_ in zip(caps, dst_caps)]
    src = np.zeros(max(st, 16), np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    dst = np.zeros(max(dt, 16), np.uint8)
    d = make_descs(sizes, soff, doff, caps)
    p = nat.Native.make_params(flags, key, aad, zstd_profile=profile)
    if mem == "device":
        ds, dd = N.device_malloc(src.size), N.device_malloc(dst.size)
        N.h2d(ds, src)
        N.transform_batch(p, d, ds, dd, dst.size, nat.MEM_DEVICE)
        N.d2h(dst, dd)
        N.device_free(ds); N.device_free(dd)
    elif mem == "packed":                                               # TSX_MEM_HOST_PACKED: outputs back to back, offsets are results
        N.transform_batch(p, d, src, dst, dst.size, nat.MEM_HOST_PACKED)
        return [dst[int(d["dst_off"][i]):int(d["dst_off"][i]) + int(d["dst_len"][i])].tobytes() for i in range(len(sizes))], d
    else:
        N.transform_batch(p, d, src, dst, dst.size)
    outs = [dst[doff[i]:doff[i] + d["dst_len"][i]].tobytes() for i in range(len(sizes))]
    return outs, d


def run_detransform(N, flags, blobs, out_sizes, key=synth.KEY, aad=synth.AAD):
    sizes = [len(b) for b in blobs]
    soff, st = [], 0
    for s in sizes:
        soff.append(st); st += (s + 15) // 16 * 16 + 16
    doff, dt = [], 0
    for s in out_sizes:
        doff.append(dt); dt += (s + 15) // 16 * 16 + 16
    src = np.zeros(max(st, 16), np.uint8)
    for b, o_ in zip(blobs, soff):
        src[o_:o_ + len(b)] = np.frombuffer(b, np.uint8)
    dst = np.zeros(max(dt, 16), np.uint8)
    d = make_descs(sizes, soff, doff, out_sizes)
    p = nat.Native.make_params(flags, key, aad)
    N.detransform_batch(p, d, src, dst, dst.size)
    outs = [dst[doff[i]:doff[i] + d["dst_len"][i]].tobytes() for i in range(len(sizes))]
    return outs, d


def oracle_transform(o, flags, chunk, i, segment=0):
    oflags = (o.COMPRESS if flags & nat.COMPRESS else 0) | (o.ENCRYPT if flags & nat.ENCRYPT else 0) | (o.CRC if flags & nat.CRC else 0)
    return o.transform_chunk(oflags, synth.KEY, synth.AAD, synth.iv_for(segment, i), chunk.tobytes())[0]


def check_transform_vs_oracle(N, o, flags, chunks, **kw):
    outs, d = run_transform(N, flags, chunks, **kw)
    oflags = (o.COMPRESS if flags & nat.COMPRESS else 0) | (o.ENCRYPT if flags & nat.ENCRYPT else 0) | (o.CRC if flags & nat.CRC else 0)
    for i, c in enumerate(chunks):
        exp, crc = o.transform_chunk(oflags, synth.KEY, synth.AAD, synth.iv_for(0, i), c.tobytes())
        assert d["status"][i] == 0, (i, c.size, d["status"][i])
        assert outs[i] == exp, "chunk %d (n=%d): transformed bytes differ from the oracle" % (i, c.size)
        if flags & nat.CRC:
            assert d["crc32c"][i] == crc, "chunk %d crc" % i
    return outs, d


def check_roundtrip(N, flags, chunks):
    outs, d = run_transform(N, flags, chunks)
    back, d2 = run_detransform(N, flags, outs, [int(c.size) for c in chunks])
    for i, c in enumerate(chunks):
        assert d2["status"][i] == 0, (i, d2["status"][i])
        assert back[i] == c.tobytes(), "chunk %d round trip" % i
        if flags & nat.CRC:
            assert d2["crc32c"][i] == d["crc32c"][i]
    return outs


EDGE_SIZES = [0, 1, 2, 3, 5, 13, 15, 16, 17, 31, 32, 33, 255, 1024, 2048, 4095, 4096, 4097, 5123, 65535, 65536, 65537, 70001,
              131072, 262144, 262145, 300007]


def edge_chunks(dist="R", sizes=EDGE_SIZES):
    return [synth.gen_chunk(dist, 7, 1, i, s) for i, s in enumerate(sizes)]


def check_profile_1_5_6(N, o, named_chunks):
    """The profile the reference would ship with (libzstd 1.5.6 inside zstd-jni 1.5.6-9, core/build.gradle:29) has no real library
    to be compared with here.  What CAN be pinned: profile 0 is the profile-1 code minus 1.5.7's block pre-splitter, so
      (a) its frames equal the serial restatement's (oracle/zstd_l3.c, profile 0) - always;
      (b) wherever the pre-splitter changes nothing for an input (restatement: profile 0 == profile 1), its frame is byte-identical
          to what the REAL libzstd 1.5.7 emits;
      (c) where the splitter does cut, the two profiles' frames differ and the real library decodes both to the source.
    Returns (#inputs pinned to the real library through (b), #inputs where the profiles differ)."""
    names = list(named_chunks)
    chunks = [named_chunks[n] for n in names]
    outs0, d0 = run_transform(N, nat.COMPRESS, chunks, profile=nat.ZSTD_PROFILE_1_5_6)
    outs1, d1 = run_transform(N, nat.COMPRESS, chunks, profile=nat.ZSTD_PROFILE_1_5_7)
    real157 = o.zstd_version().startswith("1.5.7")
    pinned = differ = 0
    for i, n in enumerate(names):
        raw = chunks[i].tobytes()
        assert d0["status"][i] == 0 and d1["status"][i] == 0, n
        r0, r1 = o.zstd_l3_compress(raw, 0), o.zstd_l3_compress(raw, 1)
        assert outs0[i] == r0, "%s: profile 1.5.6 frame differs from the restatement" % n
        assert outs1[i] == r1, "%s: profile 1.5.7 frame differs from the restatement" % n
        assert o.zstd_decompress_chunk(outs0[i], len(raw)) == raw, n
        if r0 == r1:
            if real157:
                assert outs0[i] == o.zstd_compress_chunk(raw), "%s: split-free input, yet profile 1.5.6 != real libzstd 1.5.7" % n
                pinned += 1
        else:
            differ += 1
            assert outs0[i] != outs1[i] and o.zstd_decompress_chunk(outs1[i], len(raw)) == raw, n
    return pinned, differ
