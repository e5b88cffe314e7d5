"""N > 1 path on CPU: two gloo ranks split one segment by chunk range, transform their chunks through the emulated
kernels (same C ABI as the GPU library), all-gather the transformed sizes and assemble the .log object - it must equal
the single-process oracle chain and its chunk index (bench.py --gpus N uses the same partitioning)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_main(rank, world, port, n_chunks, chunk_size, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["TSX_ALLOW_ANY_ARCH"] = "1"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import tsxform
    from tests import parity_cases as pc
    from tests.emu import emu_native
    from tsxform import shard, synth
    nat = tsxform._native
    N = emu_native.get()
    lo, hi = shard.chunk_range_of_rank(n_chunks, rank, world)
    chunks = [synth.gen_chunk("K", 9, 0, c, chunk_size) for c in range(lo, hi)]
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    # IVs are a function of the GLOBAL chunk id, whoever transforms the chunk
    sizes = [int(c.size) for c in chunks]
    soff, doff, caps, st, dt = pc.layout(sizes, flags, N)
    src = np.zeros(max(st, 16), np.uint8)
    for c, o_ in zip(chunks, soff):
        src[o_:o_ + c.size] = c
    dst = np.zeros(max(dt, 16), np.uint8)
    d = pc.make_descs(sizes, soff, doff, caps)
    for i, c in enumerate(range(lo, hi)):
        d["iv"][i] = np.frombuffer(synth.iv_for(0, c), np.uint8)
    N.transform_batch(nat.Native.make_params(flags, synth.KEY, synth.AAD), d, src, dst, dst.size)
    assert (d["status"] == 0).all()
    all_sizes, positions, base = shard.exchange_transformed_sizes(d["dst_len"], n_chunks, rank, world, dist)
    mine = b"".join(dst[doff[i]:doff[i] + d["dst_len"][i]].tobytes() for i in range(hi - lo))
    assert base == positions[lo] if hi > lo else True
    # the optional second exchange: the owner of the upload stream (rank 0) receives the other slices in place
    packed = shard.pack_slice(dst, [doff[i] for i in range(hi - lo)], d["dst_len"])
    assert packed.tobytes() == mine
    obj = shard.gather_object_to_owner(packed, all_sizes, n_chunks, rank, world, 0, dist)
    assert (obj is None) == (rank != 0)
    if rank == 0:
        with open(os.path.join(tmpdir, "object.bin"), "wb") as f:
            f.write(obj.numpy().tobytes())
    np.save(os.path.join(tmpdir, "sizes_%d.npy" % rank), all_sizes)
    with open(os.path.join(tmpdir, "slice_%d.bin" % rank), "wb") as f:
        f.write(base.to_bytes(8, "little") + mine)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks", [5, 8])
def test_two_ranks_split_a_segment_and_agree_on_the_chunk_index(oracle, emu, tmp_path, n_chunks):
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    import torch.multiprocessing as mp
    from tsxform import synth
    chunk_size, world = 30000, 2
    mp.spawn(_rank_main, args=(world, _free_port(), n_chunks, chunk_size, str(tmp_path)), nprocs=world, join=True)
    of = oracle.COMPRESS | oracle.ENCRYPT
    expected = [oracle.transform_chunk(of, synth.KEY, synth.AAD, synth.iv_for(0, c), synth.gen_chunk("K", 9, 0, c, chunk_size).tobytes())[0]
                for c in range(n_chunks)]
    s0, s1 = np.load(tmp_path / "sizes_0.npy"), np.load(tmp_path / "sizes_1.npy")
    assert (s0 == s1).all() and list(s0) == [len(e) for e in expected]             # every rank holds the whole size list
    obj = bytearray(sum(len(e) for e in expected))
    for r in range(world):
        raw = (tmp_path / ("slice_%d.bin" % r)).read_bytes()
        base = int.from_bytes(raw[:8], "little")
        obj[base:base + len(raw) - 8] = raw[8:]                                      # slices land at their exchanged bases
    assert bytes(obj) == b"".join(expected)
    assert (tmp_path / "object.bin").read_bytes() == b"".join(expected)               # ... and the owner rank holds exactly that object


def test_partition_helpers():
    from tsxform import shard
    assert [shard.segment_owner(s, 8) for s in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert shard.segments_of_rank(64, 3, 8) == list(range(3, 64, 8))
    for n, w in [(256, 8), (5, 2), (3, 8), (1, 4)]:
        rs = [shard.chunk_range_of_rank(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
    sizes, pos, base = shard.exchange_transformed_sizes([10, 20, 30], 3, 0, 1)
    assert list(pos) == [0, 10, 30] and base == 0                                     # ChunkIndexSerializationTest.java:104-123 positions
