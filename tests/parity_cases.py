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


def run_transform(N, flags, chunks, key=synth.KEY, aad=synth.AAD, mem=None, profile=nat.ZSTD_PROFILE_1_5_7, dst_caps=None, ctx=None):
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
        N.transform_batch(p, d, src, dst, dst.size, ctx=ctx)
    outs = [dst[doff[i]:doff[i] + d["dst_len"][i]].tobytes() for i in range(len(sizes))]
    return outs, d


def run_detransform(N, flags, blobs, out_sizes, key=synth.KEY, aad=synth.AAD, ctx=None):
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
    N.detransform_batch(p, d, src, dst, dst.size, ctx=ctx)
    outs = [dst[doff[i]:doff[i] + d["dst_len"][i]].tobytes() for i in range(len(sizes))]
    return outs, d


def blockmode_chunks(N, ctx, n):
    """Test hook of the library: how many of the first n chunks of ctx's last detransform batch the block-parallel decoder form
    (csrc/zstd_dec_blocks.hip) decoded; -1 when the batch did not use that form."""
    import ctypes as C
    f = N.lib.tsx_debug_blockmode_chunks
    f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_uint32]
    return f(ctx, n)


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


MiB = 1 << 20


def big_chunk(kind):
    """Chunks beyond 4 MiB: the reference allows chunk.size up to 2^30 - 1 (RemoteStorageManagerConfig.java:122-130), disables chunking
    with size 0 (BaseTransformChunkEnumeration.java:85-89: the whole segment is ONE chunk) and its integration matrix transforms a
    10 MiB segment as one chunk (RemoteStorageManagerTest.java:190-242).  Beyond 4 MiB the 2 MiB match window slides several times, the
    table entries keep fewer tag bits (9 at 4 MiB, 8 at 10 MiB, 5 at 64 MiB) and the frame is no longer single-segment."""
    R = synth.gen_chunk("R", 11, 0, 0, 4 * MiB); K = synth.gen_chunk("K", 11, 0, 1, 4 * MiB)
    Z = np.zeros(MiB // 2, np.uint8)
    if kind == "K6":                 # Kafka-like throughout (the bench content), 6 MiB
        return np.concatenate([K, synth.gen_chunk("K", 11, 0, 2, 2 * MiB)])
    if kind == "K10":
        return np.concatenate([K, synth.gen_chunk("K", 11, 0, 2, 4 * MiB), synth.gen_chunk("K", 11, 0, 3, 2 * MiB)])
    if kind == "mixed6":             # K / R pieces; the 1 MiB of K at the start comes back 1.5 MiB later (inside the window) and 4 MiB later (outside)
        return np.concatenate([K[:MiB], R[:MiB // 2], K[:MiB], R[MiB:2 * MiB], Z, K[MiB:2 * MiB], K[:MiB]])[:6 * MiB]
    if kind == "mixed10":            # far repeats at 1.5 / 3 / 7 MiB distance over five window slides, raw (R) and RLE (zeros) blocks between compressed ones
        parts = [K[:MiB], R[:MiB // 2], K[:MiB // 2], R[MiB:2 * MiB], K[MiB:2 * MiB], R[:MiB // 2], Z, K[:MiB], R[2 * MiB:3 * MiB],
                 K[2 * MiB:3 * MiB + MiB // 2], R[:MiB], K[:MiB]]
        return np.concatenate(parts)[:10 * MiB]
    if kind == "sparse64":           # 64 MiB: mostly incompressible with repeats 1 MiB (in the window) and 3+ MiB (outside) apart, zeros, some K
        parts = []
        for i in range(16):
            parts += [R[:2 * MiB], R[MiB // 2:MiB // 2 + MiB], Z, K[i * 65536:(i + 4) * 65536], R[3 * MiB:3 * MiB + MiB // 4]]
        return np.concatenate([np.concatenate(parts), R])[:64 * MiB]
    if kind == "K64":                # the bench content as ONE 64 MiB chunk (a segment with chunking disabled)
        return np.concatenate([synth.gen_chunk("K", 12, 0, c, 4 * MiB) for c in range(16)])
    raise ValueError(kind)


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


def check_encrypt_only_zero_copy(N, o, sizes):
    """Encrypt-only host batches in slot layout (chunk 3 of `sizes` gets a slot that is too small): whole buffer registered -> the GCM waves
    write into it (tsx_debug_last_zero_copy says so); unregistered / switched off -> the copies; same bytes, same failure behaviour."""
    import ctypes
    flags = nat.ENCRYPT | nat.CRC
    chunks = [synth.gen_chunk("K" if i % 2 else "R", 23, 0, i, s) for i, s in enumerate(sizes)]
    soff, doff, caps, st, dt = layout(sizes, flags, N)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    ctx = N.ctx_create(0, 0, 0)
    zc = N.lib.tsx_debug_last_zero_copy; zc.restype = ctypes.c_int; zc.argtypes = [ctypes.c_void_p]
    try:
        res = {}
        for mode in ("registered", "unregistered", "copies"):
            dst = np.full(dt + 64, 0xEE, np.uint8)
            if mode != "unregistered":
                N.host_register(dst)
            try:
                d = make_descs(sizes, soff, doff, caps); d["dst_cap"][3] = 20          # chunk 3 needs 17 + 28 bytes
                with N.configured(no_zero_copy_out=1 if mode == "copies" else 0):
                    N.transform_batch(p, d, src, dst, dst.size, nat.MEM_HOST, ctx=ctx)
                assert zc(ctx) == (1 if mode == "registered" else 0), mode
            finally:
                if mode != "unregistered":
                    N.host_unregister(dst)
            assert d["status"][3] == nat.E_DST_TOO_SMALL and d["dst_len"][3] == 0 and (np.delete(d["status"], 3) == 0).all(), (mode, d["status"])
            assert (dst[doff[3]:doff[3] + 64] == 0xEE).all(), mode
            res[mode] = ([dst[doff[i]:doff[i] + int(d["dst_len"][i])].tobytes() for i in range(len(sizes))], d["crc32c"].copy())
        assert res["registered"][0] == res["copies"][0] == res["unregistered"][0] and (res["registered"][1] == res["copies"][1]).all()
        for i in (0, 2, 4):
            assert res["registered"][0][i] == oracle_transform(o, flags, chunks[i], i)
    finally:
        N.ctx_destroy(ctx)


def b_batch_256():
    """256 chunks of content "B" (Kafka v2 record batches), 40-400 KB each: the batch VERDICT r4 #7 asks the device parity set to hold."""
    sizes = [40000 + (i * 7919) % 360000 for i in range(256)]
    return [synth.gen_chunk("B", 43, 3, i, s) for i, s in enumerate(sizes)]


def check_b_batch_256(N, o):
    """... through the full chain (= libzstd 1.5.7 + OpenSSL, byte for byte), back again, and through both Zstd profiles.  Returns
    (transformed / original, #chunks where 1.5.7's pre-splitter cut, #chunks pinned to the real library under profile 1.5.6)."""
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    chunks = b_batch_256()
    outs, d = check_transform_vs_oracle(N, o, flags, chunks)
    back, d2 = run_detransform(N, flags, outs, [int(c.size) for c in chunks])
    assert (d2["status"] == 0).all() and back == [c.tobytes() for c in chunks] and (d2["crc32c"] == d["crc32c"]).all()
    pinned, differ = check_profile_1_5_6(N, o, {"B256_%d" % i: c for i, c in enumerate(chunks)})
    return sum(len(x) for x in outs) / float(sum(int(c.size) for c in chunks)), differ, pinned
