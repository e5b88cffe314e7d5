"""The C-ABI front end (csrc/tsx_api.hip) under the CPU emulator: device selection of the ctx-less calls, pooled contexts,
the staged host-memory pipeline, key hygiene, failed-chunk scrubbing and argument validation.  Same source as the product
library; the GPU twins of the data-path cases are in tests/test_gpu_parity.py."""
import os
import subprocess
import sys
import textwrap

import time

import numpy as np
import pytest

import tsxform
from tests import parity_cases as pc
from tsxform import synth

nat = tsxform._native
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_py(code, **env):
    e = dict(os.environ, TSX_ALLOW_ANY_ARCH="1", **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_ctxless_calls_use_every_device(emu):
    """tsx_init(2): ctx-less batches alternate between the devices (least loaded, ties round-robin), a thread's device hint pins
    them, a burst of callers leaves at most 32 idle contexts per device (and at most 128 GiB of idle workspaces) (a broker JVM must reach all 8 GPUs)."""
    out = _run_py("""
        import threading, numpy as np
        import tsxform
        from tests import parity_cases as pc
        from tests.emu import emu_native
        nat = tsxform._native
        N = nat.Native(emu_native.build())
        assert N.init(2) == 2 and "2 device(s)" in N.version()
        src = np.zeros(64, np.uint8); src[:9] = np.frombuffer(b"123456789", np.uint8)
        def crc():
            d = pc.make_descs([9], [0], [0], [0]); N.crc32c_batch(d, src); assert d["crc32c"][0] == 0xE3069283
        for _ in range(6): crc()
        assert [N.pool_stats(i)["batches"] for i in (0, 1)] == [3, 3]
        N.set_thread_device(1)
        for _ in range(4): crc()
        assert [N.pool_stats(i)["batches"] for i in (0, 1)] == [3, 7]
        N.set_thread_device(-1)
        try:
            N.set_thread_device(2); raise SystemExit("device 2 accepted")
        except nat.TsxError as e:
            assert e.code == nat.E_INVAL
        c = N.ctx_create(1); assert N.ctx_device(c) == 1; N.ctx_destroy(c)
        # the calling thread's current device is the caller's business: every entry point puts it back
        import ctypes
        cur = ctypes.c_int(-1); get = N.lib._Z12hipGetDevicePi; get.argtypes = [ctypes.POINTER(ctypes.c_int)]
        get(ctypes.byref(cur)); assert cur.value == 0, cur.value
        th = [threading.Thread(target=lambda: [crc() for _ in range(3)]) for _ in range(80)]
        [t.start() for t in th]; [t.join() for t in th]
        s = [N.pool_stats(i) for i in (0, 1)]
        assert all(x["in_use"] == 0 and 1 <= x["idle"] <= 32 for x in s), s
        assert s[0]["batches"] + s[1]["batches"] == 10 + 240 and min(x["batches"] for x in s) >= 60, s
        N.shutdown()
        assert "uninitialised" in N.version()
        print("ok")
    """, HIPEMU_DEVICES=2)
    assert out.strip().endswith("ok")


def test_detransform_crc_on_a_fresh_context_with_generous_slots(emu):
    """the CRC of the restored bytes runs over dst_cap-sized slots; a fresh context sized its partial sums from the
    (small) transformed chunks only.  64 tiny chunks restored into 4 MiB slots on a new ctx."""
    n, cap = 64, 4 << 20
    chunks = [synth.gen_chunk("K", 3, 0, i, 700 + i) for i in range(n)]
    outs, d0 = pc.run_transform(emu, nat.ENCRYPT | nat.CRC, chunks)
    soff, st = [], 0
    for b in outs:
        soff.append(st); st += (len(b) + 15) // 16 * 16 + 16
    src = np.zeros(st, np.uint8)
    for b, o_ in zip(outs, soff):
        src[o_:o_ + len(b)] = np.frombuffer(b, np.uint8)
    dst = np.zeros(n * cap, np.uint8)
    d = pc.make_descs([len(b) for b in outs], soff, [i * cap for i in range(n)], [cap] * n)
    ctx = emu.ctx_create(0, 0, 0)
    try:
        emu.detransform_batch(nat.Native.make_params(nat.ENCRYPT | nat.CRC, synth.KEY, synth.AAD), d, src, dst, dst.size, ctx=ctx)
    finally:
        emu.ctx_destroy(ctx)
    assert (d["status"] == 0).all() and (d["crc32c"] == d0["crc32c"]).all()
    for i, c in enumerate(chunks):
        assert dst[i * cap:i * cap + c.size].tobytes() == c.tobytes()


def test_failed_tag_check_leaves_nothing_in_a_device_slot_and_no_key_behind(emu):
    chunks = pc.edge_chunks("R", [3000, 5000, 70001])
    outs, _ = pc.run_transform(emu, nat.ENCRYPT, chunks)
    forged = bytearray(outs[1]); forged[100] ^= 1
    blobs = [outs[0], bytes(forged), outs[2]]
    soff, st = [], 0
    for b in blobs:
        soff.append(st); st += (len(b) + 15) // 16 * 16 + 16
    src = np.zeros(st, np.uint8)
    for b, o_ in zip(blobs, soff):
        src[o_:o_ + len(b)] = np.frombuffer(b, np.uint8)
    doff = [0, 4096, 16384]; total = 16384 + 70016
    d = pc.make_descs([len(b) for b in blobs], soff, doff, [3008, 5008, 70016])
    ds, dd = emu.device_malloc(src.size), emu.device_malloc(total)
    emu.h2d(ds, src); emu.h2d(dd, np.full(total, 0xAB, np.uint8))
    ctx = emu.ctx_create(0, 0, 0)
    try:
        emu.detransform_batch(nat.Native.make_params(nat.ENCRYPT, synth.KEY, synth.AAD), d, ds, dd, total, nat.MEM_DEVICE, ctx=ctx)
        assert emu.lib.tsx_debug_key_residue(ctx) == 0          # round keys, H powers, raw key and AAD are gone
    finally:
        emu.ctx_destroy(ctx)
    back = np.zeros(total, np.uint8); emu.d2h(back, dd)
    emu.device_free(ds); emu.device_free(dd)
    assert list(d["status"]) == [0, nat.E_TAG_MISMATCH, 0] and d["dst_len"][1] == 0
    assert back[0:3000].tobytes() == chunks[0].tobytes() and back[16384:16384 + 70001].tobytes() == chunks[2].tobytes()
    assert not back[4096:4096 + 5008].any(), "unauthenticated plaintext left in the caller's slot"


def test_key_is_wiped_after_a_failed_batch_too(emu):
    ctx = emu.ctx_create(0, 0, 0)
    try:
        src = np.zeros(64, np.uint8); dst = np.zeros(16, np.uint8)
        d = pc.make_descs([32], [0], [0], [16])                      # slot too small: per-chunk failure
        emu.transform_batch(nat.Native.make_params(nat.ENCRYPT, synth.KEY, synth.AAD), d, src, dst, dst.size, ctx=ctx)
        assert d["status"][0] == nat.E_DST_TOO_SMALL and emu.lib.tsx_debug_key_residue(ctx) == 0
    finally:
        emu.ctx_destroy(ctx)


def test_validation_is_overflow_safe_and_leaves_descriptors_alone(emu):
    src = np.zeros(4096, np.uint8); dst = np.zeros(8192, np.uint8)
    p = nat.Native.make_params(nat.ENCRYPT, synth.KEY, synth.AAD)
    d = pc.make_descs([100], [0], [(1 << 64) - 16], [128])         # dst_off + dst_cap wraps around
    with pytest.raises(nat.TsxError) as e:
        emu.transform_batch(p, d, src, dst, dst.size)
    assert e.value.code == nat.E_INVAL
    d = pc.make_descs([100], [0], [8192 - 64], [128])               # ends 64 bytes beyond dst
    with pytest.raises(nat.TsxError):
        emu.transform_batch(p, d, src, dst, dst.size)
    d = pc.make_descs([100, 100], [0, 8], [777, 999], [5, 6])        # packed: unaligned src_off -> INVAL, dst fields untouched
    with pytest.raises(nat.TsxError):
        emu.transform_batch(p, d, src, dst, dst.size, nat.MEM_HOST_PACKED)
    assert list(d["dst_off"]) == [777, 999] and list(d["dst_cap"]) == [5, 6]


def test_descriptors_beyond_the_source_buffer_are_rejected(emu):
    """ABI 3: every batch entry point takes src_size; a chunk that reaches beyond it (or whose src_off + src_len wraps) fails the call
    with TSX_E_INVAL and nothing is touched - the array bounds of the reference's byte[] chunks."""
    src = np.zeros(4096, np.uint8); dst = np.zeros(8192, np.uint8)
    p = nat.Native.make_params(nat.ENCRYPT | nat.CRC, synth.KEY, synth.AAD)
    for off, ln in ((4096 - 32, 64), (4096 + 16, 0), ((1 << 64) - 16, 64)):
        d = pc.make_descs([ln], [off], [0], [256]); d["status"] = -7; d["dst_len"] = 5
        for call in (lambda: emu.transform_batch(p, d, src, dst, dst.size), lambda: emu.detransform_batch(p, d, src, dst, dst.size),
                     lambda: emu.crc32c_batch(d, src)):
            with pytest.raises(nat.TsxError) as e:
                call()
            assert e.value.code == nat.E_INVAL and d["status"][0] == -7 and d["dst_len"][0] == 5
    d = pc.make_descs([64], [4096 - 64], [0], [256])                # ends exactly at the end of src: fine
    emu.transform_batch(p, d, src, dst, dst.size)
    assert d["status"][0] == 0
    d = pc.make_descs([64], [0], [0], [256])                        # the caller states a smaller source than the array: that bound counts
    with pytest.raises(nat.TsxError):
        emu.transform_batch(p, d, src, dst, dst.size, src_size=32)


def test_compressing_batches_are_members_of_one_service(emu, oracle):
    """Every compressing batch of a device - pooled or explicit context, host or device memory, packed or slots - is a member of that
    device's compressor service: tickets in a ring, persistent waves that pull them (csrc/tsx_internal.h).  12 threads, each with its own
    key, content and memory kind, 5 batches each: every result equals the single-threaded one, every batch was counted as a member and
    every chunk as done, waves that found themselves on the reserved compute unit left without work, and a member's key is gone afterwards."""
    import threading
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    s0 = emu.service_stats(0)
    T, reps = 12, 5
    sets = []
    old_quiet = emu.debug_config("fetch_quiet_ms", 0)                   # the reservation in force whether or not something has fetched lately
    for t in range(T):
        chunks = [synth.gen_chunk("K" if (t + i) % 3 else "R", 40 + t, t, i, sz) for i, sz in enumerate([30000, 1, 0, 70001, 4096, 12345][: 3 + t % 4])]
        key = bytes((b + t) & 0xFF for b in synth.KEY)
        mem = (None, "packed", "device")[t % 3]
        sets.append((chunks, key, mem, pc.run_transform(emu, flags, chunks, key=key, mem=mem)[0]))
    emu.debug_config("fetch_quiet_ms", old_quiet)
    s1 = emu.service_stats(0)
    nchunks = sum(len(x[0]) for x in sets)
    assert s1["members"] - s0["members"] == T and s1["chunks"] - s0["chunks"] == nchunks == s1["device_chunks"] - s0["device_chunks"]
    assert s1["launches"] - s0["launches"] == T                        # one caller at a time on the harness: the kernel ends with each batch
    assert s1["reserved_cus"] == 1 and s1["cu_keys_seen"] == s1["compute_units"] == 4
    assert s1["reserved_exits"] - s0["reserved_exits"] == T * s1["waves"] // 4       # a quarter of every launch met the reserved CU
    assert s1["skipped_tickets"] == 0 and s1["watchdog_launches"] == s0["watchdog_launches"]
    errors = []

    def worker(t):
        chunks, key, mem, ref = sets[t]
        try:
            for _ in range(reps):
                outs, d = pc.run_transform(emu, flags, chunks, key=key, mem=mem)
                if outs != ref or (d["status"] != 0).any():
                    errors.append((t, "differs"))
        except Exception as e:                                          # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errors, errors[:4]
    s2 = emu.service_stats(0)
    assert s2["members"] - s1["members"] == T * reps and s2["chunks"] - s1["chunks"] == nchunks * reps and s2["running"] == 0
    back, d2 = pc.run_detransform(emu, flags, sets[0][3], [int(c.size) for c in sets[0][0]], key=sets[0][1])
    assert (d2["status"] == 0).all() and back == [c.tobytes() for c in sets[0][0]]


def test_a_launch_that_met_only_reserved_cus_is_started_again(emu):
    """HIP promises nothing about where workgroups land: a launch of the service whose every wave found itself on a reserved compute unit
    (a chip busy elsewhere) ends without having taken a ticket.  The waiting caller's watchdog starts the kernel again; same bytes."""
    import ctypes
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    chunks = [synth.gen_chunk("K", 3, 0, i, 5000 + 700 * i) for i in range(5)]
    ref, _ = pc.run_transform(emu, flags, chunks)
    s0 = emu.service_stats(0)
    emu.lib.hipemu_force_reserved_launches.argtypes = [ctypes.c_int]; emu.lib.hipemu_force_reserved_launches.restype = None
    emu.lib.hipemu_force_reserved_launches(2)
    with emu.configured(fetch_quiet_ms=0):                              # (the reserved CUs are left alone whether or not a fetch has been seen lately)
        got, d = pc.run_transform(emu, flags, chunks)
    s1 = emu.service_stats(0)
    assert got == ref and (d["status"] == 0).all()
    assert s1["watchdog_launches"] - s0["watchdog_launches"] == 2 and s1["launches"] - s0["launches"] == 3
    assert s1["reserved_exits"] - s0["reserved_exits"] == 2 * s1["waves"] + s1["waves"] // 4


def test_ticket_counters_wrap(emu):
    """Tickets are 32-bit counters compared wrap-safely: a device that has compressed 2^32 chunks (ten days at full rate) goes on."""
    import ctypes
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    chunks = [synth.gen_chunk("K", 8, 0, i, 3000 + 100 * i) for i in range(9)]
    ref, _ = pc.run_transform(emu, flags, chunks)
    seed = emu.lib.tsx_debug_service_seed; seed.restype = ctypes.c_int; seed.argtypes = [ctypes.c_int, ctypes.c_uint32]
    assert seed(0, 0xFFFFFFFF - 3) == 0
    for _ in range(2):
        got, d = pc.run_transform(emu, flags, chunks)
        assert got == ref and (d["status"] == 0).all()
    assert seed(0, 0) == 0


def test_crc_only_batches_publish_their_status(emu):
    """A descriptor that comes back from tsx_crc32c_batch says TSX_OK whatever it said before (callers reuse descriptors)."""
    buf = np.zeros(64, np.uint8); buf[:9] = np.frombuffer(b"123456789", np.uint8)
    d = pc.make_descs([9, 0], [0, 16], [0, 0], [0, 0]); d["status"] = -7
    emu.crc32c_batch(d, buf)
    assert list(d["status"]) == [0, 0] and d["crc32c"][0] == 0xE3069283 and d["crc32c"][1] == 0


@pytest.mark.parametrize("flags", [nat.ENCRYPT | nat.CRC, nat.CRC, 0, nat.COMPRESS | nat.ENCRYPT | nat.CRC, nat.COMPRESS])
def test_staged_pipeline_equals_single_shot(emu, flags):
    """TSX_MEM_HOST batches are cut into pieces (copy-in / kernels / copy-out overlapped); forced down to 4 KiB pieces here so that 23
    chunks make ~20 of them (with compression: 4 members of the compressor service, each published when its input has landed).  Outputs,
    descriptors and the inverse must equal the un-pipelined run."""
    chunks = pc.edge_chunks("K", [0, 1, 17, 300, 4096, 5000, 65537, 12, 70001, 33, 2048, 9000, 100, 4097, 1, 31000, 16, 15, 8191, 8192, 8193, 700, 64])
    with emu.configured(no_pipeline=1):
        ref, dref = pc.run_transform(emu, flags, chunks)
        refp, dpk = pc.run_transform(emu, flags, chunks, mem="packed")
    ctx = emu.ctx_create(0, 0, 0)
    try:
        with emu.configured(sub_bytes=4096, comp_pieces=4):
            got, dgot = pc.run_transform(emu, flags, chunks)
            gotp, dgp = pc.run_transform(emu, flags, chunks, mem="packed")
            gotc, dgc = pc.run_transform(emu, flags, chunks, ctx=ctx)
            if flags & nat.COMPRESS:
                import ctypes
                lm = emu.lib.tsx_debug_last_members; lm.restype = ctypes.c_int; lm.argtypes = [ctypes.c_void_p]
                assert lm(ctx) == 4
    finally:
        emu.ctx_destroy(ctx)
    assert gotc == ref and (dgc["dst_len"] == dref["dst_len"]).all() and (dgc["status"] == dref["status"]).all()
    assert got == ref and gotp == refp == ref
    for f in ("dst_len", "crc32c", "status"):
        assert (dgot[f] == dref[f]).all() and (dgp[f] == dpk[f]).all(), f
    assert (dgp["dst_off"] == dpk["dst_off"]).all()
    back, d2 = pc.run_detransform(emu, flags, got, [int(c.size) for c in chunks])
    assert (d2["status"] == 0).all() and [b for b in back] == [c.tobytes() for c in chunks]
    sizes = [int(c.size) for c in chunks]
    soff, _, _, st, _ = pc.layout(sizes, 0, emu)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    d = pc.make_descs(sizes, soff, [0] * len(sizes), [0] * len(sizes))
    emu.crc32c_batch(d, src)
    if flags & nat.CRC:
        assert (d["crc32c"] == dref["crc32c"]).all()


def test_host_register_roundtrip(emu):
    buf = np.zeros(1 << 16, np.uint8)
    emu.host_register(buf)
    d = pc.make_descs([9], [0], [0], [0]); buf[:9] = np.frombuffer(b"123456789", np.uint8)
    emu.crc32c_batch(d, buf)
    assert d["crc32c"][0] == 0xE3069283
    emu.host_unregister(buf)


def test_cached_workspaces_are_not_a_reason_for_nomem(emu):
    """idle pooled contexts only CACHE memory.  An allocation that fails while the idle pool holds contexts drains the pool
    and is tried again (the batch succeeds, the pool is empty afterwards); with nothing cached the failure is reported as before."""
    out = _run_py("""
        import ctypes, numpy as np
        import tsxform
        from tests import parity_cases as pc
        from tests.emu import emu_native
        from tsxform import synth
        nat = tsxform._native
        N = nat.Native(emu_native.build()); N.init()
        N.lib.hipemu_fail_alloc_at.argtypes = [ctypes.c_long]; N.lib.hipemu_fail_alloc_at.restype = None
        flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
        chunks = [synth.gen_chunk("K", 7, 0, i, 2500 + 10 * i) for i in range(12)]
        want, _ = pc.run_transform(N, flags, chunks)                     # a ctx-less call: leaves one idle pooled context
        assert N.pool_stats(0)["idle"] >= 1
        ctx = N.ctx_create(0, 0, 0)
        # the descriptor arrays come first (5 allocations, plain failures); the 6th is the first workspace that grows on demand
        N.lib.hipemu_fail_alloc_at(6)
        got, _ = pc.run_transform(N, flags, chunks, ctx=ctx)
        N.lib.hipemu_fail_alloc_at(0)
        assert got == want
        assert N.pool_stats(0)["idle"] == 0, N.pool_stats(0)              # the cache paid for the retry
        ctx2 = N.ctx_create(0, 0, 0)
        N.lib.hipemu_fail_alloc_at(6)
        try:
            pc.run_transform(N, flags, chunks, ctx=ctx2); raise SystemExit("nothing was cached, yet the batch succeeded")
        except nat.TsxError as e:
            assert e.code == nat.E_NOMEM, e.code
        N.lib.hipemu_fail_alloc_at(0)
        got, _ = pc.run_transform(N, flags, chunks, ctx=ctx2)
        assert got == want
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_idle_contexts_do_not_all_keep_a_block_form_workspace(emu):
    """the block-parallel decoder's workspace is 37 MiB per 4 MiB chunk; of the idle pooled contexts at most 4 keep theirs."""
    out = _run_py("""
        import threading, numpy as np
        import tsxform
        from tests import parity_cases as pc
        from tests.emu import emu_native
        from tsxform import synth
        nat = tsxform._native
        N = nat.Native(emu_native.build()); N.init()
        flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
        chunks = [synth.gen_chunk("K", 9, 0, i, 3000) for i in range(3)]
        blobs, d = pc.run_transform(N, flags, chunks)
        gate = threading.Barrier(10)
        def fetch():
            gate.wait()                                                 # ten contexts out at once
            back, _ = pc.run_detransform(N, flags, blobs, [c.size for c in chunks])
            assert all(a == b.tobytes() for a, b in zip(back, chunks))
        th = [threading.Thread(target=fetch) for _ in range(10)]
        [t.start() for t in th]; [t.join() for t in th]
        N.lib.tsx_debug_pool_bwork.restype = int
        s = N.pool_stats(0)
        assert s["in_use"] == 0 and s["idle"] >= 5, s
        assert 1 <= N.lib.tsx_debug_pool_bwork(0) <= 4, N.lib.tsx_debug_pool_bwork(0)
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_mid_size_fetch_batches_decode_in_co_resident_pieces(emu, oracle):
    """a host-memory inverse batch of 16 .. 256 chunks is cut into up to 8 pieces on the context's compute streams (copy-in
    of piece k + 1, block-form decode of piece k, copy-out of piece k - 1 overlap).  40 chunks of mixed sizes incl. an empty one and a
    forged one: same bytes and statuses as the uncut batch, every good chunk decoded by the block form, CRCs of the restored bytes right."""
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    sizes = [3000 + 977 * i for i in range(40)]; sizes[7] = 0; sizes[23] = 1
    chunks = [synth.gen_chunk("K" if i % 4 else "R", 31, 0, i, s) for i, s in enumerate(sizes)]
    blobs, d0 = pc.run_transform(emu, flags, chunks)
    forged = bytearray(blobs[11]); forged[40] ^= 2; blobs[11] = bytes(forged)
    ctx = emu.ctx_create(0, 0, 0)
    try:
        outs, d = pc.run_detransform(emu, flags, blobs, sizes, ctx=ctx)
        taken = pc.blockmode_chunks(emu, ctx, len(blobs))
        import ctypes
        pieces = emu.lib.tsx_debug_blockmode_pieces; pieces.restype = ctypes.c_int; pieces.argtypes = [ctypes.c_void_p]
        assert pieces(ctx) == 5, pieces(ctx)                            # 40 chunks: pieces of 8
        with emu.configured(no_dec_pieces=1):
            outs1, d1 = pc.run_detransform(emu, flags, blobs, sizes, ctx=ctx)
            assert pc.blockmode_chunks(emu, ctx, len(blobs)) == taken and pieces(ctx) == 1
    finally:
        emu.ctx_destroy(ctx)
    assert list(d["status"]) == list(d1["status"]) and d["status"][11] == nat.E_TAG_MISMATCH and (np.delete(d["status"], 11) == 0).all()
    assert taken >= 38                                                  # (the forged chunk never reaches the decoder; the empty frame may go either way)
    for i, c in enumerate(chunks):
        if i != 11:
            assert outs[i] == c.tobytes() == outs1[i], i
            assert d["crc32c"][i] == oracle.crc32c(c.tobytes())
    assert outs[11] == b"" or not any(outs[11])


@pytest.mark.parametrize("kind", ["slots", "packed"])
@pytest.mark.parametrize("ctxless", [True, False])
@pytest.mark.parametrize("flags", [nat.COMPRESS | nat.ENCRYPT | nat.CRC, nat.COMPRESS | nat.CRC])
def test_zero_copy_output_equals_the_copy_path(emu, oracle, kind, ctxless, flags):
    """The compressor waves write into the caller's buffer when the device can address ALL of it (a buffer pinned with tsx_host_register);
    the test hook no_zero_copy_out keeps the device output buffer + copies.  Same bytes, sizes, CRCs and packed offsets either way, with and
    without a context, with and without encryption; what lies between a chunk's last byte and the next slot is untouched; a packed buffer
    too small for the slots, an unregistered buffer and one of which only the first half is registered take the copy path by themselves."""
    import ctypes
    sizes = [9000, 0, 131072, 5, 40001, 70000, 1, 20000, 3000]
    chunks = [synth.gen_chunk("K" if i % 3 else "R", 17, 0, i, s) for i, s in enumerate(sizes)]
    n = len(chunks)
    soff, doff, caps, st, dt = pc.layout(sizes, flags, emu)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    slot = (emu.transformed_bound(max(sizes), flags) + 63) // 64 * 64
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    mem = nat.MEM_HOST if kind == "slots" else nat.MEM_HOST_PACKED
    ctx = emu.ctx_create(0, 0, 0)                                       # (the pooled path is reached with ctx=None; this one is asked about afterwards)
    zc = emu.lib.tsx_debug_last_zero_copy; zc.restype = ctypes.c_int; zc.argtypes = [ctypes.c_void_p]
    try:
        res = {}
        for mode in ("zero_copy", "copies", "unregistered", "half_registered", "small_packed_buffer"):
            if mode == "small_packed_buffer" and kind != "packed":
                continue
            size = dt + 64 if kind == "slots" else (n * slot + 64 if mode != "small_packed_buffer" else sum(int(x) for x in caps) // 2)
            dst = np.full(size, 0xEE, np.uint8)
            reg = None
            if mode in ("zero_copy", "copies", "small_packed_buffer"):
                reg = dst
            elif mode == "half_registered":
                reg = dst[:size // 2]
            if reg is not None:
                emu.host_register(reg)
            try:
                d = pc.make_descs(sizes, soff, doff, caps)
                # (an explicit context packs in place only on request: a whole batch is ~90 ms of memmove)
                with emu.configured(zero_copy_packed=1, no_zero_copy_out=1 if mode == "copies" else 0):
                    emu.transform_batch(p, d, src, dst, dst.size, mem, ctx=None if ctxless else ctx)
                    if not ctxless:
                        assert zc(ctx) == (1 if mode == "zero_copy" else 0), mode
            finally:
                if reg is not None:
                    emu.host_unregister(reg)
            assert (d["status"] == 0).all(), (mode, d["status"])
            res[mode] = ([dst[int(d["dst_off"][i]):int(d["dst_off"][i]) + int(d["dst_len"][i])].tobytes() for i in range(n)], d.copy())
            if kind == "slots":
                assert (dst[doff[0] + int(d["dst_len"][0]):doff[1]] == 0xEE).all(), mode
        ref, dref = res["copies"]
        for mode, (got, d) in res.items():
            assert got == ref and (d["dst_len"] == dref["dst_len"]).all() and (d["crc32c"] == dref["crc32c"]).all() and (d["dst_off"] == dref["dst_off"]).all(), mode
        for i in (0, 2, 4):
            assert ref[i] == pc.oracle_transform(oracle, flags, chunks[i], i)
    finally:
        emu.ctx_destroy(ctx)


def test_reserved_cus_change_nothing_but_where_the_waves_run(emu):
    """tsx_config.fetch_reserved_cus (default: one compute unit per shader engine that a fetch under full upload load finds free)
    decides which waves of the service leave at once (fetch_quiet_ms = 0: always; otherwise - the default since round 6, 2000 ms - while fetches are about);
    bytes and statuses are the same with a reservation, without one, and with the environment's override, context-less and with a context, slot and packed
    layout; the fetch side is untouched.  With fetch_quiet_ms set, a process that has not fetched for that long compresses on the reserved CUs too (guest waves)."""
    code = """
        import os, hashlib, numpy as np
        import tsxform
        from tests import parity_cases as pc
        from tests.emu import emu_native
        from tsxform import synth
        nat = tsxform._native
        N = nat.Native(emu_native.build()); N.init(%s)
        flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
        chunks = [synth.gen_chunk("K", 5, 0, i, 4000 + 1500 * i) for i in range(10)]
        ctx = N.ctx_create(0, 0, 0)
        a, da = pc.run_transform(N, flags, chunks)
        b, db = pc.run_transform(N, flags, chunks, ctx=ctx)
        c, dc = pc.run_transform(N, flags, chunks, mem="packed")
        s = N.service_stats(0)
        back, d2 = pc.run_detransform(N, flags, a, [int(x.size) for x in chunks], ctx=ctx)
        assert a == b == c and (da["status"] == 0).all() and back == [x.tobytes() for x in chunks]
        e, de = pc.run_transform(N, flags, chunks)                      # after a fetch: the reserved CUs are left alone
        s2 = N.service_stats(0)
        assert e == a
        print(hashlib.sha256(b"".join(a)).hexdigest(), s["reserved_cus"], s["reserved_exits"] > 0, s["launches"], s["wave_starts"], s["reserved_exits"], s["waves"],
              s["guest_launches"], s2["launches"] - s["launches"], s2["guest_launches"] - s["guest_launches"], s2["reserved_exits"] - s["reserved_exits"])
        """
    engaged = _run_py(code % "fetch_quiet_ms=0").strip().splitlines()[-1].split()
    without = _run_py(code % "fetch_reserved_cus=0").strip().splitlines()[-1].split()
    by_env = _run_py(code % "fetch_reserved_cus=0", TSX_FETCH_RESERVED_CUS=1, TSX_FETCH_QUIET_MS=0).strip().splitlines()[-1].split()
    assert engaged[0] == without[0] == by_env[0]
    assert engaged[1:3] == ["1", "True"] and without[1:3] == ["0", "False"] and by_env[1:3] == ["1", "True"]      # (the harness has 4 CUs: at most one is reserved)
    assert engaged[7] == "0" and without[7] == "0"
    # fetch_quiet_ms != 0 (the default is 2000): nobody has fetched yet - every launch's waves use the reserved CU as well (creating a context needs room
    # for a moment but is no fetch); after the first fetch they leave it alone
    for cfg in ("fetch_quiet_ms=10000", ""):
        guests = _run_py(code % cfg).strip().splitlines()[-1].split()
        assert guests[0] == engaged[0] and guests[1:3] == ["1", "False"] and guests[7] == guests[3] and int(guests[3]) >= 3, (cfg, guests)
        assert int(guests[8]) >= 1 and guests[9] == "0" and int(guests[10]) > 0, (cfg, guests)
    # ... and some of the compressor's waves may stay on a reserved CU all the same (tsx_config.fetch_shared_cu_waves: a CU shared between
    # fetches and uploads): exactly that many per launch, counted afresh by every launch
    kept = _run_py(code % "fetch_shared_cu_waves=2, fetch_quiet_ms=0").strip().splitlines()[-1].split()
    launches, starts, exits, waves = (int(x) for x in kept[3:7])
    dl, ds_, de, dw = (int(x) for x in engaged[3:7])
    assert kept[0] == engaged[0] and waves == dw and starts + exits == launches * waves and ds_ + de == dl * dw
    assert exits == launches * (de // dl - 2) and launches >= 2


def test_guest_waves_hand_their_chunks_back_when_a_fetch_arrives(emu, oracle):
    """The reservation follows the traffic (tsx_config.fetch_quiet_ms, opt-in): with no fetch about, the waves on a reserved CU compress too -
    as guests that look at the host's yield word before every block of their chunk.  Here the first workgroup of every launch sits on the
    reserved CU (hipemu_cu_key_shift) and the word is raised while it is in the middle of a chunk (hipemu_force_yield_after): the chunk goes back to the
    queue with half-built tables, frame and entropy state in its workspace, another wave starts it again from its first byte - the bytes are
    the oracle's all the same, every chunk is counted once, and the next launch (nobody has fetched: the word was raised by the harness, and
    the front end clears it) has guests again."""
    import ctypes
    for f, t in (("hipemu_cu_key_shift", [ctypes.c_int]), ("hipemu_force_yield_after", [ctypes.c_int])):
        getattr(emu.lib, f).argtypes = t; getattr(emu.lib, f).restype = None
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    sizes = [400000, 131072 * 2 + 5, 70001, 17, 0, 200000]
    chunks = [synth.gen_chunk("K" if i % 3 else "B", 31, 1, i, s) for i, s in enumerate(sizes)]
    with emu.configured(fetch_quiet_ms=1):
        time.sleep(0.01)                                                # (whatever fetched before this test: quiet again)
        ref, dref = pc.check_transform_vs_oracle(emu, oracle, flags, chunks)
        emu.service_quiesce(0)
        s0 = emu.service_stats(0)
        emu.lib.hipemu_cu_key_shift(3)
        try:
            for after in (5, 3, 8):                                     # looks: at the wave's start, in svc_take, then one per block
                time.sleep(0.01)
                emu.lib.hipemu_force_yield_after(after)
                got, d = pc.run_transform(emu, flags, chunks)
                assert got == ref and (d["status"] == 0).all() and (d["crc32c"] == dref["crc32c"]).all(), after
                got, d = pc.run_transform(emu, flags, chunks, mem="packed")      # the launch after: guests again (yield cleared), nothing forced
                assert got == ref and (d["status"] == 0).all()
        finally:
            emu.lib.hipemu_cu_key_shift(0); emu.lib.hipemu_force_yield_after(0)
        emu.service_quiesce(0)
        s1 = emu.service_stats(0)
    assert s1["yielded_waves"] - s0["yielded_waves"] >= 2 and s1["returned_chunks"] - s0["returned_chunks"] >= 2, (s0, s1)
    assert s1["device_chunks"] - s0["device_chunks"] == 6 * len(chunks) and s1["skipped_tickets"] == s0["skipped_tickets"]
    assert s1["guest_launches"] - s0["guest_launches"] >= 6


def test_a_wave_that_finds_itself_on_a_reserved_cu_hands_its_chunk_back(emu, oracle):
    """A compressor wave does not move by itself, but the hardware's scheduler may save a queue's waves and restore them on other compute units;
    round 5's "kernel of a fetch that does not start, once in a few hundred fetches" was compressor waves sitting on the reserved CUs after
    such a restore (profiles/r06_stuck_fetch_trace.txt).  So every wave asks where it is - between two chunks and before every block of a
    chunk - and leaves a reserved CU it did not start on: the chunk in progress goes back to the queue like a guest's.  Here the harness
    moves the first workgroup onto the reserved CU at its n-th look (hipemu_relocate_after): bytes and checksums stay the oracle's, every chunk
    is counted once, the wave is counted as relocated."""
    import ctypes
    emu.lib.hipemu_relocate_after.argtypes = [ctypes.c_int]; emu.lib.hipemu_relocate_after.restype = None
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    sizes = [400000, 131072 * 2 + 5, 70001, 17, 0, 200000]
    chunks = [synth.gen_chunk("K" if i % 3 else "B", 32, 1, i, s) for i, s in enumerate(sizes)]
    with emu.configured(fetch_quiet_ms=0):
        ref, dref = pc.check_transform_vs_oracle(emu, oracle, flags, chunks)
        emu.service_quiesce(0)
        s0 = emu.service_stats(0)
        try:
            for after in (65, 67, 68, 73):                              # looks: 64 at the wave's start (every lane), then lane 0's: before every ticket, before every block of a chunk
                emu.lib.hipemu_relocate_after(after)
                got, d = pc.run_transform(emu, flags, chunks)
                assert got == ref and (d["status"] == 0).all() and (d["crc32c"] == dref["crc32c"]).all(), after
        finally:
            emu.lib.hipemu_relocate_after(0)
        emu.service_quiesce(0)
        s1 = emu.service_stats(0)
    assert s1["relocated_waves"] - s0["relocated_waves"] >= 3, (s0, s1)
    assert s1["returned_chunks"] - s0["returned_chunks"] >= 1 and s1["yielded_waves"] == s0["yielded_waves"], (s0, s1)      # handed back mid-chunk at least once; nobody was a guest
    assert s1["device_chunks"] - s0["device_chunks"] == 4 * len(chunks) and s1["skipped_tickets"] == s0["skipped_tickets"]


def test_on_a_quiet_device_a_relocated_wave_finishes_its_chunk(emu, oracle):
    """The same restore onto a reserved CU while no fetch has been seen for fetch_quiet_ms: nobody wants the CU, so the wave does not throw a
    second of work away (measured: a lone 2048-chunk batch with 304 relocated waves took 2.8 s instead of 0.95 s, its chunks started three times,
    profiles/r06_ticket_storm.txt) - it reads the yield word before every block, finishes the chunk, and leaves between two chunks.  A fetch that
    arrives meanwhile (hipemu_force_yield_after) gets the CU at the next block boundary: the chunk goes back to the queue then."""
    import ctypes
    import time
    emu.lib.hipemu_relocate_after.argtypes = [ctypes.c_int]; emu.lib.hipemu_relocate_after.restype = None
    emu.lib.hipemu_force_yield_after.argtypes = [ctypes.c_int]; emu.lib.hipemu_force_yield_after.restype = None
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    sizes = [400000, 131072 * 2 + 5, 70001, 17, 0, 200000]
    chunks = [synth.gen_chunk("K" if i % 3 else "B", 32, 1, i, s) for i, s in enumerate(sizes)]
    with emu.configured(fetch_quiet_ms=0):
        ref, dref = pc.check_transform_vs_oracle(emu, oracle, flags, chunks)
    with emu.configured(fetch_quiet_ms=1):
        time.sleep(0.05)
        emu.service_quiesce(0)
        s0 = emu.service_stats(0)
        try:
            for after in (65, 67, 68, 73):
                time.sleep(0.01)
                emu.lib.hipemu_relocate_after(after)
                got, d = pc.run_transform(emu, flags, chunks)
                assert got == ref and (d["status"] == 0).all() and (d["crc32c"] == dref["crc32c"]).all(), after
        finally:
            emu.lib.hipemu_relocate_after(0)
        emu.service_quiesce(0)
        s1 = emu.service_stats(0)
        assert s1["relocated_waves"] - s0["relocated_waves"] >= 3, (s0, s1)
        assert s1["returned_chunks"] == s0["returned_chunks"], (s0, s1)       # nothing was handed back: the chunks were finished where the waves found themselves
        assert s1["device_chunks"] - s0["device_chunks"] == 4 * len(chunks) and s1["skipped_tickets"] == s0["skipped_tickets"]


def test_encrypt_only_batches_write_into_registered_buffers_too(emu, oracle):
    """Producers compress -> the broker's chain is encryption only (RemoteStorageManager.java:381-398): the GCM kernel's waves write
    IV || C || TAG straight into the caller's slots when the whole buffer is registered (slot layout); same bytes as the copy path and the
    oracle, a too-small slot fails its chunk only and leaves the slot untouched, an unregistered buffer takes the copies."""
    pc.check_encrypt_only_zero_copy(emu, oracle, [5000, 0, 70001, 17, 131072, 4096])


def test_tsx_config_sizes_and_the_environment():
    """tsx_init_ex reads tsx_config by its struct_size: the struct as it was before fetch_quiet_ms (24 bytes) is accepted and leaves the new
    field at its default, a shorter one is refused; a field at TSX_CFG_DEFAULT keeps the library's default; the environment has the last word."""
    out = _run_py("""
        import ctypes as C
        import tsxform
        from tests.emu import emu_native
        nat = tsxform._native
        N = nat.Native(emu_native.build())
        class Old(C.Structure):
            _fields_ = [("struct_size", C.c_uint32), ("fetch_reserved_cus", C.c_uint32), ("service_max_launch_ms", C.c_uint32), ("fetch_shared_cu_waves", C.c_uint32), ("pool_idle_bytes", C.c_uint64)]
        assert C.sizeof(Old) == 24 and C.sizeof(nat.Config) == 32
        f = N.lib.tsx_init_ex; f.argtypes = [C.c_int, C.c_void_p, C.c_void_p]; f.restype = C.c_int
        short = Old(16, 0, 0, 0, 0)
        assert f(1, None, C.byref(short)) == nat.E_INVAL and N.lib.tsx_device_count() == 0
        old = Old(24, 0, 1234, 3, nat.CFG_DEFAULT64)
        assert f(1, None, C.byref(old)) == 1
        vals = {k: N.debug_config(k, 7) for k in ("reserved_cus", "svc_max_launch_ms", "svc_keep_waves", "fetch_quiet_ms")}
        print(vals["reserved_cus"], vals["svc_max_launch_ms"], vals["svc_keep_waves"], vals["fetch_quiet_ms"])
        N.lib.tsx_shutdown()
        new = nat.Config(32, nat.CFG_DEFAULT, nat.CFG_DEFAULT, 99, nat.CFG_DEFAULT64, 2500, 0)
        assert f(1, None, C.byref(new)) == 1
        vals = {k: N.debug_config(k, 7) for k in ("reserved_cus", "svc_max_launch_ms", "svc_keep_waves", "fetch_quiet_ms")}
        print(vals["reserved_cus"], vals["svc_max_launch_ms"], vals["svc_keep_waves"], vals["fetch_quiet_ms"])
        N.lib.tsx_shutdown()
        """, TSX_SERVICE_MAX_LAUNCH_MS=777).strip().splitlines()
    assert out[-2].split() == ["0", "777", "3", "2000"]                 # the environment overrides service_max_launch_ms; fetch_quiet_ms untouched by the old struct (its default: 2000)
    assert out[-1].split() == ["7", "777", "8", "2500"]                 # (7: what the first process-wide debug_config left - reserved_cus at TSX_CFG_DEFAULT keeps the current value; 99 waves are capped at 8)
