"""RCCL on real hardware, before the first multi-GPU run does it.  The GPU box has ONE MI355X, so the process group has
one rank: `bench.py --gpus 1 --force-dist --split-segments --gather-object` under the driver's own launcher makes that rank run
init_process_group("nccl"), the barrier, the max-over-ranks all-reduce on a device tensor, the all-gather of transformed sizes and a
grouped send + recv of its slice of the object (to itself) - every torch.distributed call the N > 1 path of bench.py / shard.py makes."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(600)
def test_rccl_collectives_run_on_one_gpu(gpu, oracle):
    cmd = ["timeout", "420", sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--split-segments", "--gather-object",
           "--segments", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-sustained", "--no-end-to-end", "--verify-chunks", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    c = j["config"]
    pg = c["process_group"]
    assert pg["backend"].startswith("nccl") and pg["forced_on_one_rank"] is True
    assert pg["ran"] == ["barrier", "all_reduce(MAX)", "all_gather(sizes)", "p2p slice -> owner"]
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and c["chunks_of_rank0"] == 256
    assert j["detransform"]["round_trip_exact"] is True and c["verified_chunks_vs_oracle"] >= 4
    # the chunk index that came out of the all-gather and the object that came back through the send / recv pair: what the product
    # library itself produces for the same segment in this process
    import torch
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    n, CH = 256, synth.CHUNK
    dev = torch.device("cuda", 0)
    src = torch.empty(n * CH, dtype=torch.uint8, device=dev)
    for i in range(n):
        src[i * CH:(i + 1) * CH] = synth.gen_chunk("K", 1000, 0, i, CH, device=dev)
    slot = (gpu.transformed_bound(CH, flags) + 63) // 64 * 64
    dst = torch.empty(n * slot, dtype=torch.uint8, device=dev)
    d = np.zeros(n, nat.DESC_DTYPE)
    d["src_off"] = np.arange(n, dtype=np.uint64) * CH; d["src_len"] = CH; d["dst_off"] = np.arange(n, dtype=np.uint64) * slot; d["dst_cap"] = slot
    for i in range(n):
        d["iv"][i] = np.frombuffer(synth.iv_for(0, i), np.uint8)
    gpu.transform_batch(nat.Native.make_params(flags, synth.KEY, synth.AAD), d, src.data_ptr(), dst.data_ptr(), dst.numel(), nat.MEM_DEVICE)
    assert (d["status"] == 0).all()
    sizes = d["dst_len"].astype(np.int64)
    pos = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    assert c["chunk_index_positions_sha"] == hashlib.sha256(pos.tobytes()).hexdigest()[:16]
    host = dst.cpu().numpy()
    whole = b"".join(host[i * slot:i * slot + int(sizes[i])].tobytes() for i in range(n))
    assert c["object_gathered_on_rank0_sha"] == hashlib.sha256(whole).hexdigest()[:16]
