"""TEST HARNESS: run under LD_PRELOAD=libasan by tests/test_emu_asan.py.  Every allocation a batch makes on a context is made to
fail once (hipemu_fail_alloc_at): the call must come back with TSX_E_NOMEM / TSX_E_DEVICE, free nothing twice, leave no dangling
workspace pointer behind, and the SAME context must then serve the same batch correctly."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import tsxform  # noqa: E402
from tests import parity_cases as pc  # noqa: E402
from tsxform import synth  # noqa: E402

nat = tsxform._native
os.environ["TSX_ALLOW_ANY_ARCH"] = "1"
N = nat.Native(sys.argv[1])
N.init()
N.lib.hipemu_fail_alloc_at.argtypes = [ctypes.c_long]; N.lib.hipemu_fail_alloc_at.restype = None
N.lib.hipemu_alloc_calls.restype = ctypes.c_long
flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
small = [synth.gen_chunk("K", 3, 0, 0, 900)]
large = [synth.gen_chunk("K", 4, 0, i, 3000 + 20 * i) for i in range(18)]         # more chunks, longer chunks: every workspace regrows
# (references through an explicit context: an idle POOLED context would be drained to retry a failed allocation - tsx_api.hip
#  reserve_or_drain, tests/test_emu_boundary.py::test_cached_workspaces_are_not_a_reason_for_nomem - and this script wants the failures)
_c = N.ctx_create(0, 0, 0)
want_small, _ = pc.run_transform(N, flags, small, ctx=_c)
want_large, _ = pc.run_transform(N, flags, large, ctx=_c)
N.ctx_destroy(_c)
assert N.pool_stats(0)["idle"] == 0


def attempt(ctx, chunks, mem):
    sizes = [int(c.size) for c in chunks]
    soff, doff, caps, st, dt = pc.layout(sizes, flags, N)
    src = np.zeros(st, np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    dst = np.zeros(dt, np.uint8)
    d = pc.make_descs(sizes, soff, doff, caps)
    p = nat.Native.make_params(flags, synth.KEY, synth.AAD)
    try:
        if mem == "packed":
            N.transform_batch(p, d, src, dst, dst.size, nat.MEM_HOST_PACKED, ctx=ctx)
        else:       # "device": the emulator has one address space, numpy memory is device memory - no staging buffers then
            N.transform_batch(p, d, src, dst, dst.size, nat.MEM_DEVICE if mem == "device" else nat.MEM_HOST, ctx=ctx)
    except nat.TsxError as e:
        assert e.code in (nat.E_NOMEM, nat.E_DEVICE), e.code
        return None
    assert (d["status"] == 0).all()
    return [dst[int(d["dst_off"][i]):int(d["dst_off"][i]) + int(d["dst_len"][i])].tobytes() for i in range(len(sizes))]


faults = 0
for mem in ("host", "packed"):                   # device-memory batches allocate a subset of these (no staging buffers)
    # how many allocations does the growth from the small batch to the large one take?
    ctx = N.ctx_create(0, 0, 0)
    assert attempt(ctx, small, mem) == want_small
    N.lib.hipemu_fail_alloc_at(0)
    assert attempt(ctx, large, mem) == want_large
    total = N.lib.hipemu_alloc_calls()
    N.ctx_destroy(ctx)
    assert total >= 5, total
    for k in range(1, (total if mem == "host" else 3) + 1):
        ctx = N.ctx_create(0, 0, 0)
        assert attempt(ctx, small, mem) == want_small
        N.lib.hipemu_fail_alloc_at(k)
        got = attempt(ctx, large, mem)
        N.lib.hipemu_fail_alloc_at(0)
        assert got is None, "allocation %d of %d failed and the batch still succeeded" % (k, total)
        faults += 1
        assert attempt(ctx, large, mem) == want_large, "context unusable after allocation %d failed" % k
        N.ctx_destroy(ctx)
    # context creation itself
    for k in range(1, 8):
        N.lib.hipemu_fail_alloc_at(k)
        try:
            c = N.ctx_create(0, 64, 70000)
            N.ctx_destroy(c)
        except nat.TsxError as e:
            assert e.code in (nat.E_NOMEM, nat.E_DEVICE)
            faults += 1
        N.lib.hipemu_fail_alloc_at(0)
# ctx-less calls: a pooled context that failed goes back to the pool and serves the next caller
xl = [synth.gen_chunk("K", 5, 0, i, 5000 + 50 * i) for i in range(40)]           # outgrows the pooled context the lines above used
N.lib.hipemu_fail_alloc_at(1)
assert attempt(None, xl, "host") is None
N.lib.hipemu_fail_alloc_at(0)
got = attempt(None, xl, "host")
assert got is not None and attempt(None, xl, "packed") == got and attempt(None, large, "host") == want_large
s = N.pool_stats(0)
assert s["in_use"] == 0, s
print("asan alloc faults ok: %d injected failures" % faults)
