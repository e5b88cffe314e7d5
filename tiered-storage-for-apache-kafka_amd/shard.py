"""Multi-GPU partitioning of the chunk-transform path - one process per GPU under torch.distributed.

Chunks are independent by construction (fresh Zstd context and fresh cipher per chunk:
core/.../transform/CompressionChunkEnumeration.java:52, EncryptionChunkEnumeration.java:67), so:

* segment-major (segments >= ranks): segment s belongs to rank s % world.  No data-path collective at all.
* chunk-range (one segment split over ranks): rank g takes the contiguous chunks [g*n/G, (g+1)*n/G).  The only
  cross-rank state is the running sum of transformed sizes (the chunk index, AbstractChunkIndex.java:52-72), so the ranks
  all-gather their int32 transformed sizes (1 KiB per 256-chunk segment) and each exclusive-scans its
  transformedPosition base.  RCCL ("nccl" on ROCm) on GPUs, gloo in the CPU tests.
"""
import numpy as np


def segment_owner(segment: int, world: int) -> int:
    return segment % world


def segments_of_rank(n_segments: int, rank: int, world: int):
    return [s for s in range(n_segments) if segment_owner(s, world) == rank]


def chunk_range_of_rank(n_chunks: int, rank: int, world: int):
    """Contiguous chunk ids [lo, hi) of `rank`; every rank's output is then a contiguous slice of the .log object."""
    return (n_chunks * rank) // world, (n_chunks * (rank + 1)) // world


def exchange_transformed_sizes(local_sizes, n_chunks: int, rank: int, world: int, dist=None, device="cpu"):
    """All-gather the transformed chunk sizes of a segment split by chunk_range_of_rank.

    Returns (sizes[n_chunks] int64, positions[n_chunks] int64, my_base): the full size list in chunk order, every chunk's
    transformedPosition, and the offset at which this rank's slice starts in the transformed object."""
    lo, hi = chunk_range_of_rank(n_chunks, rank, world)
    local = np.asarray(local_sizes, dtype=np.int32)
    assert local.shape == (hi - lo,)
    if world == 1 and dist is None:
        sizes = local.astype(np.int64)
    else:                                                # (world 1 with a process group: bench.py --force-dist runs the collective on one GPU)
        import torch
        per = max(chunk_range_of_rank(n_chunks, r, world)[1] - chunk_range_of_rank(n_chunks, r, world)[0] for r in range(world))
        buf = torch.zeros(per, dtype=torch.int32, device=device)
        buf[:hi - lo] = torch.from_numpy(local).to(device)
        out = [torch.zeros(per, dtype=torch.int32, device=device) for _ in range(world)]
        dist.all_gather(out, buf)
        parts = []
        for r in range(world):
            a, b = chunk_range_of_rank(n_chunks, r, world)
            parts.append(out[r][:b - a].cpu().numpy())
        sizes = np.concatenate(parts).astype(np.int64)
    positions = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    return sizes, positions, int(positions[lo]) if hi > lo else int(sizes.sum())


def pack_slice(dst, dst_offs, dst_lens):
    """This rank's transformed chunks back to back - its contiguous slice of the .log object - from the slotted output buffer
    (torch tensor on the device, or numpy array): one narrow copy per chunk, no host round trip for device tensors."""
    lens = [int(x) for x in dst_lens]
    total = sum(lens)
    if isinstance(dst, np.ndarray):
        out = np.empty(total, np.uint8)
        at = 0
        for o, n in zip(dst_offs, lens):
            out[at:at + n] = dst[int(o):int(o) + n]; at += n
        return out
    import torch
    out = torch.empty(total, dtype=torch.uint8, device=dst.device)
    at = 0
    for o, n in zip(dst_offs, lens):
        out[at:at + n] = dst[int(o):int(o) + n]; at += n
    return out


def gather_object_to_owner(my_slice, sizes, n_chunks: int, rank: int, world: int, owner: int = 0, dist=None, device="cpu"):
    """The optional second exchange of a split segment: the rank that owns the upload stream receives every other rank's
    slice straight into its place in the transformed object (point-to-point send / recv - RCCL over xGMI on GPUs, gloo in the CPU
    tests; the object is the concatenation of the chunks in id order, AbstractChunkIndex.java:52-72).  `sizes` is the all-gathered
    size list of exchange_transformed_sizes.  Returns the whole object (uint8 tensor) on `owner`, None elsewhere."""
    import torch
    sizes = np.asarray(sizes, dtype=np.int64)
    bounds = [chunk_range_of_rank(n_chunks, r, world) for r in range(world)]
    slice_bytes = [int(sizes[a:b].sum()) for a, b in bounds]
    bases = np.concatenate([[0], np.cumsum(slice_bytes)[:-1]]).astype(np.int64)
    mine = my_slice if isinstance(my_slice, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(my_slice)).to(device)
    assert mine.numel() == slice_bytes[rank], (mine.numel(), slice_bytes[rank])
    if world == 1:
        if dist is None:
            return mine
        # bench.py --force-dist: one rank, one GPU - the slice travels through the process group's point-to-point path to this same
        # rank (a grouped send + recv to self: ncclSend / ncclRecv on the device), so that the first multi-GPU run is not the first time
        # that code executes
        if dist.get_backend() != "nccl":                 # gloo has no pair to itself: the CPU rehearsal keeps the slice as it is
            return mine
        obj = torch.empty_like(mine)
        if mine.numel():
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.contiguous(), 0), dist.P2POp(dist.irecv, obj, 0)]):
                w.wait()
        return obj
    if rank != owner:
        if slice_bytes[rank]:
            dist.send(mine.contiguous(), dst=owner)
        return None
    obj = torch.empty(int(sizes.sum()), dtype=torch.uint8, device=mine.device)
    obj[int(bases[rank]):int(bases[rank]) + slice_bytes[rank]] = mine
    for r in range(world):
        if r == owner or slice_bytes[r] == 0:
            continue
        dist.recv(obj[int(bases[r]):int(bases[r]) + slice_bytes[r]], src=r)        # a contiguous view: received in place
    return obj
