"""Device run of the decoder's three-stage block pipeline on frames the emulated tests only see in small: full 4 MiB
chunks written by stock libzstd at levels 1, 3 and 19 (window log up to 22: offsets across the whole chunk, repeat-mode
tables, treeless literals) and frames that mix raw, RLE and compressed blocks.  (Named to run after the other GPU files.)"""
import numpy as np
import pytest

import tsxform
from tests import parity_cases as pc
from tsxform import synth

nat = tsxform._native
pytestmark = pytest.mark.gpu


def _inputs():
    K = synth.gen_chunk("K", 9, 1, 3); R = synth.gen_chunk("R", 9, 1, 4, 600000)
    mixed = np.concatenate([K[:1500000], R[:300000], np.zeros(400000, np.uint8), K[1500000:2600000], np.full(200000, 7, np.uint8),
                            R[300000:330000], K[2600000:3600000], np.tile(np.frombuffer(b"0123456789abcdef", np.uint8), 4000)])
    far = np.concatenate([R[:200000], K[:1800000], R[:200000], K[:1800000]])          # matches 2 MB back
    return {"K4M": K, "mixed": mixed[:synth.CHUNK], "far": far[:synth.CHUNK], "one block": K[:100000], "two blocks": K[:200000],
            "three blocks": K[:380000]}


@pytest.mark.parametrize("level", [1, 3, 19])
def test_full_size_and_mixed_block_frames_of_stock_libzstd(gpu, oracle, level):
    inputs = _inputs()
    blobs = [oracle.zstd_compress_chunk(v.tobytes(), level) for v in inputs.values()]
    outs, d = pc.run_detransform(gpu, nat.COMPRESS, blobs, [int(v.size) for v in inputs.values()])
    for i, (name, v) in enumerate(inputs.items()):
        assert d["status"][i] == 0 and outs[i] == v.tobytes(), (name, level)


def test_packed_host_output_on_the_device(gpu, oracle):
    """TSX_MEM_HOST_PACKED (the `.log` object assembled in the caller's buffer): same bytes as the slot-per-chunk layout and as
    the oracle chain, offsets = running sum of the sizes; full-size chunks included."""
    chunks = [synth.gen_chunk("K", 3, 0, i, n) for i, n in enumerate((70000, 1, synth.CHUNK, 0, 33333, synth.CHUNK, 65537))]
    for flags in (nat.ENCRYPT | nat.CRC, nat.COMPRESS | nat.ENCRYPT | nat.CRC):
        slots, d0 = pc.run_transform(gpu, flags, chunks)
        packed, d1 = pc.run_transform(gpu, flags, chunks, mem="packed")
        assert packed == slots and (d1["status"] == 0).all() and (d1["crc32c"] == d0["crc32c"]).all()
        ends = np.cumsum(d1["dst_len"].astype(np.int64))
        assert (d1["dst_off"].astype(np.int64) == ends - d1["dst_len"]).all()
        for i in (0, 2, 6):
            assert packed[i] == pc.oracle_transform(oracle, flags, chunks[i], i)


def test_both_decoder_forms_agree_on_the_device(gpu, oracle, monkeypatch):
    """Batches of up to 256 chunks decode one workgroup per BLOCK with execution by pointer jumping (csrc/zstd_dec_blocks.hip), larger
    ones one workgroup per chunk (csrc/zstd_dec.hip); the block form hands what it does not like back to the chunk form.  Full-size
    frames of stock libzstd (levels 1 / 3 / 19), frames of many mixed blocks, damaged frames and a 10 MiB chunk through both forms on
    the device: identical statuses and bytes, every undamaged frame decoded by the block form, a 300-chunk batch by the chunk form."""
    from tests.test_emu_zstd import _fuzzed_frames
    inputs = _inputs()
    plain = list(inputs.values()) + [pc.big_chunk("mixed10")]
    blobs, sizes = [], []
    for lvl in (1, 3, 19):
        for v in plain:
            blobs.append(oracle.zstd_compress_chunk(v.tobytes(), lvl)); sizes.append(int(v.size))
    good = len(blobs)
    fb, fs = _fuzzed_frames(oracle, 60, 31)
    big = bytearray(blobs[0]); big[len(big) // 3] ^= 0x5A; big[len(big) // 2] ^= 0xC3                  # a damaged full-size frame
    blobs += fb + [bytes(big)]; sizes += fs + [sizes[0]]
    ctx = gpu.ctx_create(0, 0, 0)
    try:
        outs, d = pc.run_detransform(gpu, nat.COMPRESS, blobs, sizes, ctx=ctx)
        # the block form takes frames of up to 264 blocks; libzstd 1.5.7's pre-splitter cuts the 10 MiB mixed chunk into 553 at level 3:
        # that one is the chunk form's by design
        from tests import zstd_inspect as zi
        fits = sum(1 for b in blobs[:good] if len(zi.parse_frame(b, decode=False)[1]) <= 264)
        assert fits >= good - 2 and pc.blockmode_chunks(gpu, ctx, good) == fits
        with gpu.configured(dec_block_chunks=0):
            outs0, d0 = pc.run_detransform(gpu, nat.COMPRESS, blobs, sizes, ctx=ctx)
            assert pc.blockmode_chunks(gpu, ctx, len(blobs)) == -1
        # above the threshold the batch takes the chunk form by itself
        many = [blobs[3]] * 300
        outs2, d2 = pc.run_detransform(gpu, nat.COMPRESS, many, [sizes[3]] * 300, ctx=ctx)
        assert pc.blockmode_chunks(gpu, ctx, 300) == -1 and (d2["status"] == 0).all() and all(o == plain[3].tobytes() for o in outs2)
    finally:
        gpu.ctx_destroy(ctx)
    assert list(d["status"]) == list(d0["status"]) and (d["status"][:good] == 0).all() and (d["status"][good:] != 0).sum() >= 15
    for i in range(len(blobs)):
        if d["status"][i] == 0:
            assert outs[i] == outs0[i], i
    for i in range(good):
        assert outs[i] == plain[i % len(plain)].tobytes(), i


def test_block_form_takes_deep_copy_chains_on_the_device(gpu, oracle):
    """4 MiB chunks whose copy chains are 4 Mi / 2 Mi / 600 k hops deep (runs at offset 1, 2 and 7), frames of stock libzstd and of the
    product compressor: exact bytes, and the block form keeps every one of them - with in-place pointer jumping a pass guarantees three
    hops, so ceil(log3(size)) + 1 passes are queued (round 3 queued log4 + 1 = 13: a 2 Mi-hop chain could end up in the 31 ms path)."""
    from tests.test_emu_zstd import _deep_chain_inputs
    inputs = _deep_chain_inputs(synth.CHUNK)
    vals = list(inputs.values())
    blobs = [oracle.zstd_compress_chunk(v.tobytes(), 3) for v in vals] + [oracle.zstd_compress_chunk(v.tobytes(), 19) for v in vals]
    ours, d0 = pc.run_transform(gpu, nat.COMPRESS, vals)
    assert (d0["status"] == 0).all()
    blobs += ours
    ctx = gpu.ctx_create(0, 0, 0)
    try:
        for rep in range(3):                                            # (the order of the in-place stores differs from launch to launch)
            outs, d = pc.run_detransform(gpu, nat.COMPRESS, blobs, [synth.CHUNK] * len(blobs), ctx=ctx)
            assert pc.blockmode_chunks(gpu, ctx, len(blobs)) == len(blobs), "a deep chain fell back to the chunk-serial decoder"
            assert (d["status"] == 0).all()
            for i in range(len(blobs)):
                assert outs[i] == vals[i % 3].tobytes(), (rep, i)
    finally:
        gpu.ctx_destroy(ctx)


def test_ctxless_compressing_callers_are_members_of_the_service_on_the_device(gpu, oracle):
    """16 threads, context-less 64-chunk compressing batches from / to pinned host buffers: every batch equals the single-threaded
    result - each member of the compressor service returns on its own completion flag, raised by its last wave while the persistent kernel
    goes on for the others (and re-reads descriptors, keys and source bytes that the host and the copy engine have rewritten since the
    wave's previous chunk): stale ciphertext, a stale descriptor or a stale source line (a missing release / acquire) would show here."""
    import threading
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    T, reps, B = 16, 3, 64
    CH = 1 << 20
    chunks = [synth.gen_chunk("K", 77, 0, i, CH) for i in range(B)]
    ref, dref = pc.run_transform(gpu, flags, chunks, mem="packed")
    sizes = [CH] * B
    soff, _, _, st, _ = pc.layout(sizes, flags, gpu)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    gpu.host_register(src)
    dsts = [np.zeros(B * (CH + 4096), np.uint8) for _ in range(T)]
    for h in dsts:
        gpu.host_register(h)
    s0 = gpu.service_stats(0)
    errors = []
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)

    def worker(t):
        try:
            for _ in range(reps):
                d = pc.make_descs(sizes, soff, [0] * B, [0] * B)
                gpu.transform_batch(p, d, src, dsts[t], dsts[t].size, nat.MEM_HOST_PACKED)
                got = [dsts[t][int(d["dst_off"][i]):int(d["dst_off"][i]) + int(d["dst_len"][i])].tobytes() for i in range(B)]
                if got != ref or (d["status"] != 0).any() or (d["crc32c"] != dref["crc32c"]).any():
                    errors.append((t, "differs"))
        except Exception as e:                                          # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    gpu.service_quiesce(0)
    s1 = gpu.service_stats(0)
    for h in dsts:
        gpu.host_unregister(h)
    gpu.host_unregister(src)
    assert not errors, errors[:4]
    # every batch went through the queue (a 64-chunk host batch is cut into 4 members: piece k + 1's input copy overlaps piece k's waves),
    # every chunk was counted on the device, and far fewer kernels were launched than batches: the waves stay while there is work
    assert s1["members"] - s0["members"] == 4 * T * reps and s1["chunks"] - s0["chunks"] == T * reps * B == s1["device_chunks"] - s0["device_chunks"], (s0, s1)
    assert 1 <= s1["launches"] - s0["launches"] <= T * reps and s1["skipped_tickets"] == 0, (s0, s1)


def test_zero_copy_output_into_registered_buffers_on_the_device(gpu, oracle):
    """Round 4: when the device can address the caller's output buffer (tsx_host_register), the compressor waves write IV || C || TAG
    straight into it - context-less calls (per-member completion flag, no end-of-kernel release to lean on) and an explicit context, the
    slot layout GpuTransformChunkEnumeration issues and the packed layout (packed down in place when the buffer has room for the slots).
    Bytes, sizes, CRCs and packed offsets equal the copy path's and the oracle chain's; a slot that is too small fails its chunk only."""
    import ctypes as C
    import threading
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    sizes = [synth.CHUNK, 300000, 0, 131072, 5, synth.CHUNK, 70001, 1 << 20] * 4
    chunks = [synth.gen_chunk("K" if i % 5 else "R", 91, 0, i, s) for i, s in enumerate(sizes)]
    n = len(chunks)
    ref, dref = pc.run_transform(gpu, flags, chunks)                      # pageable numpy buffers: the copy path
    soff, doff, caps, st, dt = pc.layout(sizes, flags, gpu)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    gpu.host_register(src)
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    slot = (gpu.transformed_bound(max(sizes), flags) + 63) // 64 * 64
    errors = []

    def one(kind, ctx, tag):
        dst = np.full(max(dt, n * slot) + 64, 0xEE, np.uint8)
        gpu.host_register(dst)
        try:
            d = pc.make_descs(sizes, soff, doff, caps)
            gpu.transform_batch(p, d, src, dst, dst.size, kind, ctx=ctx)
            got = [dst[int(d["dst_off"][i]):int(d["dst_off"][i]) + int(d["dst_len"][i])].tobytes() for i in range(n)]
            if got != ref or (d["status"] != 0).any() or (d["crc32c"] != dref["crc32c"]).any():
                errors.append((tag, "bytes"))
            if kind == nat.MEM_HOST_PACKED:
                ends = np.cumsum(d["dst_len"].astype(np.int64))
                if not (d["dst_off"].astype(np.int64) == ends - d["dst_len"]).all():
                    errors.append((tag, "packed offsets"))
            elif not (dst[int(d["dst_off"][2]) + int(d["dst_len"][2]):int(d["dst_off"][3])] == 0xEE).all():
                errors.append((tag, "bytes written beyond a chunk's dst_len"))
        except Exception as e:                                          # noqa: BLE001
            errors.append((tag, repr(e)))
        finally:
            gpu.host_unregister(dst)

    ctx = gpu.ctx_create(0, 0, 0)
    try:
        for kind, name in ((nat.MEM_HOST, "slots"), (nat.MEM_HOST_PACKED, "packed")):
            one(kind, None, "ctxless " + name)
            one(kind, ctx, "ctx " + name)
        with gpu.configured(zero_copy_packed=1):                      # an explicit context packs in place only on request (a whole batch is ~90 ms of memmove)
            one(nat.MEM_HOST_PACKED, ctx, "ctx packed in place")
        th = [threading.Thread(target=one, args=(nat.MEM_HOST if t % 2 else nat.MEM_HOST_PACKED, None, "thread %d" % t)) for t in range(8)]
        [x.start() for x in th]; [x.join() for x in th]
        # one slot too small: that chunk fails, nothing of it is written, its neighbours are whole
        dst = np.full(dt + 64, 0xEE, np.uint8); gpu.host_register(dst)
        d = pc.make_descs(sizes, soff, doff, caps); d["dst_cap"][1] = 1000
        gpu.transform_batch(p, d, src, dst, dst.size, nat.MEM_HOST, ctx=None)
        assert d["status"][1] == nat.E_DST_TOO_SMALL and d["dst_len"][1] == 0 and (np.delete(d["status"], 1) == 0).all()
        assert (dst[doff[1]:doff[1] + 1000] == 0xEE).all() and dst[doff[0]:doff[0] + int(d["dst_len"][0])].tobytes() == ref[0]
        gpu.host_unregister(dst)
    finally:
        gpu.ctx_destroy(ctx)
        gpu.host_unregister(src)
    assert not errors, errors[:6]
    for i in (0, 1, 4, 7):
        assert ref[i] == pc.oracle_transform(oracle, flags, chunks[i], i)
