#!/usr/bin/env python3
"""bench.py — GiB/s of the chunk-transform hot path on MI355X (BASELINE.json metric).

A step = one pass of the chain over one batch of synthetic 4 MiB chunks that is ALREADY RESIDENT IN HBM when
the timed region starts (device pointers through the C ABI; the PCIe-inclusive rate is reported separately in
DESIGN.md, never as `value`).  Workloads (config.workload):
  full     BASELINE configs[3]: 8 x 1 GiB segments per GPU, Zstd(level 3) -> AES-256-GCM -> CRC32C
  gcm_crc  BASELINE configs[2]: 1 GiB segment, AES-256-GCM + CRC32C
  crc      BASELINE configs[1]: 1 GiB segment, CRC32C only
N > 1: one process per GPU (torch.distributed.run), segments sharded segment-major, no data-path collective
 — weak scaling: every GPU gets the same number of segments.  Rank 0 prints ONE JSON line.
"""
import argparse
import subprocess
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes every stream of the process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share
# one run their kernels one after the other.  The library's shared lanes + copy streams (ctx-less calls) and the caller contexts of
# the legs below are more than 4: the process asks for 16, as INTEGRATION.md tells a broker's launcher to (must be set before the
# runtime initialises, i.e. before torch is imported; an explicit setting wins).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# N > 1: RCCL shares device memory between the ranks through dmabuf handles; the pool's host driver supports no other kind
# (`hipIpcGetMemHandle: invalid argument` without this).  Exported on the GPU boxes already - a default for a launcher whose environment lacks it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# torch and the HIP tools want a temporary directory; a box whose /tmp is missing or full (seen once on the GPU pool) must not cost the line
try:
    import tempfile
    tempfile.gettempdir()
except (OSError, FileNotFoundError):
    for _d in ("/dev/shm", ROOT):
        if os.path.isdir(_d) and os.access(_d, os.W_OK):
            os.environ["TMPDIR"] = _d
            tempfile.tempdir = None
            break
GiB = float(1 << 30)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="auto", choices=["auto", "full", "gcm_crc", "crc"])
    ap.add_argument("--segments", type=int, default=0, help="1 GiB segments per GPU (default: 8 for full, 1 otherwise)")
    ap.add_argument("--dist", default="K", choices=["K", "B", "R"], help="synthetic content: K Kafka-like JSON lines, B Kafka v2 record batches (binary), R random")
    ap.add_argument("--no-value-b", action="store_true", help="skip the extra leg on Kafka-shaped binary content (value_B)")
    ap.add_argument("--profile", default="1.5.7", choices=["1.5.6", "1.5.7"], help="libzstd release reproduced")
    ap.add_argument("--inflight", type=int, default=5,
                    help="caller threads, each with its own tsx_ctx + output buffer, that submit the steps concurrently (the reference "
                         "calls the path from >= 10 RLM upload threads; the Zstd kernel is latency bound, so batches in flight are its "
                         "latency cover).  1 = strictly one batch at a time")
    ap.add_argument("--chunk-bytes", type=int, default=0, help="chunk size (default 4 MiB: the metric's configuration; smaller only with --rehearse)")
    ap.add_argument("--chunks-per-segment", type=int, default=0, help="default 256 (1 GiB segments); smaller only with --rehearse")
    ap.add_argument("--split-segments", action="store_true",
                    help="segments < GPUs (BASELINE configs[4] tail): every segment is cut by chunk range over ALL ranks, the ranks all-gather the "
                         "transformed sizes (the one exchange of the path, tsxform.shard) inside the timed step; total work fixed -> scaling strong")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU rehearsal of the multi-rank logic: the kernel sources compiled for the CPU emulator (tests/emu), host buffers, gloo. "
                         "Exercises rank->segment mapping, IVs, barrier + max-over-ranks timing and the size exchange - NOT a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--verify-chunks", type=int, default=0, help="chunks of the timed batch compared byte for byte with the oracle after the timed region (0 = ALL of them, "
                    "on every usable host core: ~2-3 s for 2048 chunks)")
    ap.add_argument("--no-line-rate-probe", action="store_true", help="skip tools/ubench/line_rate (the box's random-line rate for the parser's access mix, measured right before the timed region)")
    ap.add_argument("--gather-object", action="store_true",
                    help="with --split-segments: rank 0 (the owner of the upload stream) also receives every rank's slice of the transformed object inside the step (send / recv)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the continuously-fed measurement (5 callers, 10 batches each) after the timed region")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host->host (PCIe-inclusive) measurement after the timed region")
    ap.add_argument("--broker-inprocess", action="store_true", help="run the broker-shaped leg in THIS process, after the timed region (torch's bundled HIP runtime: D2H copies "
                    "are blit kernels), instead of first, in a torch-free child that owns the device alone (the default since round 4)")
    ap.add_argument("--broker-subprocess", action="store_true", help="run the broker-shaped leg as tools/broker_leg.py in a child process without torch (the system's HIP runtime "
                    "instead of the one torch bundles); slower while this process holds the device too - see the comment at the leg")
    ap.add_argument("--no-broker", action="store_true", help="skip the broker-shaped leg of end_to_end (10 / 20 callers x 256-chunk segments, pooled contexts, registered buffers)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the barrier / max-over-ranks reduction (and, with --split-segments / --gather-object, the "
                         "size all-gather and a point-to-point loop-back) even with ONE rank: RCCL on a single GPU (tests/test_gpu_parity.py)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process group of the N > 1 barrier / max-over-ranks (gloo: CPU rehearsal)")
    ap.add_argument("--no-inverse", action="store_true", help="skip the detransform (fetch side) measurement after the timed region")
    ap.add_argument("--no-mixed-load", action="store_true", help="skip the mixed-load leg (fetch latency while the timed region's callers keep the chip full)")
    ap.add_argument("--mixed-load-seconds", type=float, default=8.0)
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[1] / configs[2] legs (CRC32C only; AES-256-GCM + CRC32C on one 1 GiB segment)")
    ap.add_argument("--settle-seconds", type=float, default=10.0,
                    help="idle time between the broker-shaped children (which saturate the chip for ~100 s) and this process's own measurement")
    return ap.parse_args()


def kernel_source_sha(names):
    """sha256 (16 hex) over the kernel sources a PMC record was measured on (tools/pmc_traffic.py writes the same)."""
    import hashlib
    h = hashlib.sha256()
    for nm in names:
        with open(os.path.join(ROOT, "tiered-storage-for-apache-kafka_amd", "csrc", nm), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_record(key):
    """(record, fresh): the committed PMC pass for this workload, and whether the kernel source is still the one it was measured on.
    A stale record is not quoted: `traffic` is null until tools/pmc_zstd.sh + tools/pmc_traffic.py have run on the new kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(key)
        if not rec:
            return None, False
        fresh = "kernel_source_sha" in rec and rec["kernel_source_sha"] == kernel_source_sha(rec.get("kernel_sources", []))
        return rec, fresh
    except (OSError, ValueError, KeyError):
        return None, False


def _gen_b_chunk(a):
    """(worker of a spawned process pool) one chunk of content B."""
    from tsxform import synth
    return synth.gen_chunk("B", a[0], a[1], a[2], a[3])


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity, capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cfs quota) - oversubscribing a throttled container makes a CPU baseline look slower than it is."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: become the launch line the contract names (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) instead of silently measuring one rank."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def broker_leg_first(args):
    """The broker-shaped leg, BEFORE this process touches the device: a helper child generates two source segments on the GPU (torch, exits),
    then tools/broker_leg.py runs without torch - the system's HIP runtime, as a broker's JVM loads it through libtsxform.so - and owns the
    device alone.  (In this process the first HIP runtime loaded is the one torch bundles, whose device -> host copies are blit KERNELS that
    wait for wave slots on a chip full of compressor waves; and a child that shares the device with a parent that holds queues on it is
    time-sliced against them: profiles/r03_broker_with_and_without_torch.jsonl.)  Returns (rows, sizes) - sizes = the dst_len of the two
    segments' chunks as the child's callers saw them, compared with this process's own run later - or (error rows, None)."""
    import shutil
    import subprocess
    import tempfile
    B, CH = 256, 4 << 20
    tmpd = tempfile.mkdtemp(prefix="tsx_broker_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    env = dict(os.environ)
    try:
        src, ivs, sizes = (os.path.join(tmpd, f) for f in ("src.npy", "ivs.npy", "sizes.npy"))
        leg = os.path.join(ROOT, "tools", "broker_leg.py")
        g = subprocess.run([sys.executable, leg, "--gen", src, ivs, "2", str(B), str(CH), args.dist], capture_output=True, text=True, timeout=300, env=env)
        if g.returncode != 0:
            return [{"error": "source generation failed (rc %d): %s" % (g.returncode, g.stderr.strip()[-300:])}], None
        cp = subprocess.run([sys.executable, leg, "--src", src, "--ivs", ivs, "--sizes-out", sizes, "--callers", "10,20,32", "--batch", str(B), "--chunk", str(CH),
                             "--profile", "1" if args.profile == "1.5.7" else "0"], capture_output=True, text=True, timeout=600, env=env)
        lines = [ln for ln in cp.stdout.strip().splitlines() if ln.startswith("[")]
        if cp.returncode != 0 or not lines:
            return [{"error": "tools/broker_leg.py failed (rc %d): %s" % (cp.returncode, cp.stderr.strip()[-300:])}], None
        return json.loads(lines[-1]), np.load(sizes)
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return [{"error": "tools/broker_leg.py: %r" % (e,)}], None
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)


def _device_state_under_load(out, after_s):
    """rocm-smi's view of the GPU a few seconds into a saturated leg (clocks, power, cap, temperatures), into `out`.  Reported, never used:
    a slow box should be told from a slow process by something other than the result."""
    time.sleep(after_s)
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showtemp", "--showperflevel", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        j = json.loads(r.stdout.decode() or "{}")
        dev_i = int(os.environ.get("LOCAL_RANK", "0"))
        card = j.get("card%d" % dev_i) or (next(iter(j.values())) if j else {})
        keep = ("sclk", "mclk", "fclk", "socclk", "power", "temperature", "performance level")
        out.update({k: v for k, v in card.items() if any(w in k.lower() for w in keep)})
        out["sampled_at_s"] = after_s
    except Exception as e:                                 # noqa: BLE001 - a diagnostic
        out["error"] = repr(e)[:160]


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args)                                     # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): the line would misreport n_gpus" % (args.gpus, world))
    # the broker-shaped leg of end_to_end comes FIRST, in children, while this process has not initialised HIP yet (see broker_leg_first)
    broker_first = None
    if (rank == 0 and world == 1 and not args.rehearse and args.workload in ("auto", "full") and not args.no_broker and not args.no_end_to_end
            and not args.broker_inprocess and args.inflight > 1 and args.steps > 1 and not args.split_segments):
        broker_first = broker_leg_first(args)
        time.sleep(max(0.0, args.settle_seconds))                        # the children kept the chip saturated: let it idle before the timed region
    import torch  # before libtsxform: one shared HIP runtime
    import torch.distributed as dist
    rehearse = args.rehearse
    if rehearse:
        assert args.backend == "gloo" or world == 1, "--rehearse runs on CPU: use --backend gloo"
        os.environ["TSX_ALLOW_ANY_ARCH"] = "1"
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback; --rehearse is a logic rehearsal, not a measurement)"
        torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        if "MASTER_ADDR" not in os.environ:                                # --force-dist without a launcher: a one-rank rendezvous on loopback
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # NCCL = RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    if rehearse:
        from tests.emu import emu_native                                   # test harness: same sources, compiled for the CPU emulator
        N = emu_native.get()
    else:
        N = nat.Native()
        N.init(1, [local_rank])
    # host-path legs (end_to_end): this rank's threads and the buffers they first-touch and pin stay on the NUMA node of ITS GPU - the far
    # socket of a two-socket host costs a third of the rate (profiles/r04_broker_numa.txt).  The timed region is device resident: unaffected.
    cpu_affinity = "not asked"
    if not rehearse:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from numa_bind import bind_to_numa_node_of_pci
            pr = torch.cuda.get_device_properties(local_rank)
            cpu_affinity = bind_to_numa_node_of_pci(getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception as e:                                           # noqa: BLE001 - a convenience, never a reason to lose the line
            cpu_affinity = "not bound: %r" % (e,)
    dev = None if rehearse else torch.device("cuda", local_rank)
    MEM = nat.MEM_HOST if rehearse else nat.MEM_DEVICE

    class Mem:                                                            # resident buffers: HBM tensors, or host arrays in a rehearsal
        @staticmethod
        def empty(nbytes):
            return np.zeros(nbytes, np.uint8) if rehearse else torch.empty(nbytes, dtype=torch.uint8, device=dev)
        @staticmethod
        def ptr(b):
            return b if rehearse else b.data_ptr()
        @staticmethod
        def host(b, lo, hi):
            return b[lo:hi] if rehearse else b[lo:hi].cpu().numpy()
        @staticmethod
        def equal(a, b):
            return bool(np.array_equal(a, b)) if rehearse else bool(torch.equal(a, b))
        @staticmethod
        def sync():
            if not rehearse:
                torch.cuda.synchronize()
        @staticmethod
        def max_over_ranks(x):
            tt = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if args.backend == "gloo" else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
    have_zstd = getattr(tsxform, "HAVE_ZSTD", False)
    workload = args.workload
    if workload == "auto":
        workload = "full" if have_zstd else "gcm_crc"
    flags = {"full": nat.COMPRESS | nat.ENCRYPT | nat.CRC, "gcm_crc": nat.ENCRYPT | nat.CRC, "crc": nat.CRC}[workload]
    nseg = args.segments or (8 if workload == "full" else 1)
    CH = args.chunk_bytes or synth.CHUNK
    cps = args.chunks_per_segment or 256                 # chunks per segment (256 x 4 MiB = 1 GiB)
    if (CH != synth.CHUNK or cps != 256) and not rehearse:
        raise SystemExit("the metric is quoted on 1 GiB segments of 4 MiB chunks: --chunk-bytes / --chunks-per-segment need --rehearse")
    from tsxform import shard
    split = args.split_segments
    if split:
        # segments < GPUs: the job has nseg segments in total, every one cut by chunk range over all ranks (strong scaling)
        my_segments = list(range(nseg))
        ranges = [shard.chunk_range_of_rank(cps, rank, world) for _ in range(nseg)]
    else:
        # weak scaling, segment-major: with world*nseg segments in the job, rank r owns segments r, r+world, ...
        my_segments = shard.segments_of_rank(world * nseg, rank, world)
        ranges = [(0, cps)] * nseg
    work = [(s_, c) for s_, (lo, hi) in zip(range(nseg), ranges) for c in range(lo, hi)]     # (local segment slot, chunk id) of every chunk here
    n = len(work)

    # ---- synthetic segments, generated directly in HBM -------------------------------------------------
    src = Mem.empty(max(n, 1) * CH)
    for i, (s_, c) in enumerate(work):
        gs = my_segments[s_]
        src[i * CH:(i + 1) * CH] = synth.gen_chunk(args.dist, 1000 + gs, gs, c, CH, device=dev)
    slot = (N.transformed_bound(CH, flags) + 63) // 64 * 64
    T = max(1, min(args.inflight, args.steps)) if workload == "full" else 1
    if split:
        T = 1                                            # the size exchange is a collective: one caller thread per rank
    dsts = [Mem.empty(max(n, 1) * slot if workload != "crc" else 64) for _ in range(T)]
    dst = dsts[0]
    d = np.zeros(n, nat.DESC_DTYPE)
    d["src_off"] = np.arange(n, dtype=np.uint64) * CH
    d["src_len"] = CH
    d["dst_off"] = np.arange(n, dtype=np.uint64) * slot
    d["dst_cap"] = slot
    for i, (s_, c) in enumerate(work):
        d["iv"][i] = np.frombuffer(synth.iv_for(my_segments[s_], c), np.uint8)      # IV = f(global segment, chunk id), whoever transforms it
    profile = nat.ZSTD_PROFILE_1_5_7 if args.profile == "1.5.7" else nat.ZSTD_PROFILE_1_5_6
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=profile)
    ctxs = [N.ctx_create(0, max(n, 1), CH) for _ in range(T)]
    ds = [d] + [d.copy() for _ in range(T - 1)]
    ctx = ctxs[0]
    index = {}                                           # split mode: per segment (sizes, positions, base) after the exchange
    objects = {}                                         # split mode with --gather-object: the whole transformed object of each segment, on rank 0

    def step(t=0):
        if n:
            if workload == "crc":
                N.crc32c_batch(ds[t], Mem.ptr(src), MEM, ctx=ctxs[t])
            else:
                N.transform_batch(params, ds[t], Mem.ptr(src), Mem.ptr(dsts[t]), dsts[t].size if rehearse else dsts[t].numel(), MEM, ctx=ctxs[t])
        if split and workload != "crc":
            # the path's one exchange: every rank learns every chunk's transformed size, so that it can place its slice of the
            # .log object and build the whole chunk index (AbstractChunkIndex.java:52-72)
            at = 0
            for s_, (lo, hi) in zip(range(nseg), ranges):
                index[s_] = shard.exchange_transformed_sizes(ds[t]["dst_len"][at:at + hi - lo], cps, rank, world, dist if dist_on else None,
                                                             device="cpu" if args.backend == "gloo" else dev)
                if args.gather_object:
                    # the optional second exchange: this rank's slice, packed, straight into its place on the owner rank
                    mine = shard.pack_slice(dsts[t], ds[t]["dst_off"][at:at + hi - lo], ds[t]["dst_len"][at:at + hi - lo])
                    objects[s_] = shard.gather_object_to_owner(mine, index[s_][0], cps, rank, world, 0, dist if dist_on else None,
                                                               device="cpu" if args.backend == "gloo" else dev)
                at += hi - lo

    def fence():
        Mem.sync()
        if dist_on:
            dist.barrier()
        Mem.sync()

    # ---- what THIS box sustains for the parser's memory access mix, measured now (VERDICT r5 #5a): tools/ubench/line_rate in a child, waves that
    # do nothing but touch random 64-B lines of 768 KiB tables in the workspace layout, reads : rewrites : blind stores as the compressor's PMC
    # passes count them.  roofline.binding_resource.peak below is this number - same box, same minute - not a band from an older box.
    line_rate = None
    if rank == 0 and world == 1 and workload == "full" and not rehearse and not args.no_line_rate_probe:
        exe = os.path.join(ROOT, "tools", "ubench", "line_rate")
        if not os.path.exists(exe):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools", "ubench"), "line_rate"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            runs = []
            for mixargs in (["6144", "20000", "25", "18", "6"], ["6144", "20000", "25", "0", "0"], ["6144", "20000", "25", "18", "6", "4"]):   # the parser's mix; reads alone; the mix with four reads in flight per lane
                # (four sets of tables for the parser's mix, the best one counts: pairs of multi-GB allocations share a "side" of device memory or not -
                #  37.6 / 45.7 / 50.3 G requests/s by set, profiles/r06_workspace_placement.txt - and a ceiling is the best case)
                pr_ = subprocess.run([exe] + mixargs, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120,
                                     env=dict(os.environ, LINE_RATE_SETS="4") if mixargs[3] != "0" else None)
                runs.append(json.loads(pr_.stdout.strip().splitlines()[-1]))
            line_rate = {"parser_mix": runs[0], "reads_only": runs[1], "parser_mix_pipelined": runs[2], "tool": "tools/ubench/line_rate.hip (child process on the same GPU, before the warm-up)"}
        except Exception as e:                                           # noqa: BLE001 - a diagnosis, never a reason to lose the line
            line_rate = {"error": repr(e)[:200]}
    import threading
    for w in range(max(args.warmup, 1) if T > 1 else args.warmup):
        for t in range(T if w == 0 else 1):              # every context's workspace is allocated before the timed region
            step(t)
    fence()
    stage = {"crc": 0.0, "zstd": 0.0, "gcm": 0.0}
    launches = {"crc": 0, "zstd": 0, "gcm": 0}
    lock = threading.Lock()

    def worker(t):
        # caller thread t submits steps t, t + T, ...: EXACTLY args.steps steps in total; each call is synchronous for its
        # caller (returns when that batch is done on the device), concurrency is across callers as in the reference
        for _ in range(t, args.steps, T):
            step(t)
            tm = N.ctx_timing(ctxs[t])
            with lock:
                stage["crc"] += tm.crc_ms; stage["zstd"] += tm.zstd_ms; stage["gcm"] += tm.gcm_ms
                launches["crc"] += 1; launches["zstd"] += 1; launches["gcm"] += 1

    # the compressor service's kernel of the warm-up is gone before the clock starts: the launches counted below are the timed region's
    svc0 = None
    if flags & nat.COMPRESS:
        N.service_quiesce(0)
        svc0 = N.service_stats(0)
    t0 = time.perf_counter()
    if T == 1:
        worker(0)
    else:
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
    fence()
    elapsed = time.perf_counter() - t0
    svc = None
    if svc0 is not None:
        N.service_quiesce(0)
        svc1 = N.service_stats(0)
        svc = {k: svc1[k] - svc0[k] for k in ("launches", "watchdog_launches", "members", "chunks", "kernel_ms", "device_chunks", "wave_starts", "reserved_exits", "guest_launches", "yielded_waves", "relocated_waves")}
        svc.update({k: svc1[k] for k in ("waves", "compute_units", "cu_keys_seen", "reserved_cus")})
    for t in range(1, T):
        assert (ds[t]["status"] == 0).all() and (ds[t]["dst_len"] == d["dst_len"]).all() and (ds[t]["crc32c"] == d["crc32c"]).all()
    # for the record (outside the timed region): the same batch strictly one at a time
    single = None
    if T > 1:
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step(0)
        fence()
        single = float(n) * CH * 2 / GiB / (time.perf_counter() - t1)
    # ---- a continuously fed device (outside the timed region, never `value`) ---------------------------------------------
    # The timed region is K steps from a barrier: the callers start at the same instant and each waits for its whole batch - the
    # stragglers of a batch hold its slots while nothing is queued behind them.  A broker's >= 10 upload threads (README.md:221 of
    # the reference) keep work queued: 5 callers, started a quarter of a second apart, 10 batches each; the rate is the slope of
    # (batches completed) over time across the middle 60 % of the run, i.e. without the ramp at either end.
    sustained = None
    box_state = {}
    if rank == 0 and world == 1 and workload == "full" and T > 1 and not split and not args.no_sustained:
        TS, BS, stag = (5, 10, 0.25) if not rehearse else (5, 3, 0.01)
        nbytes = (lambda b: b.size) if rehearse else (lambda b: b.numel())
        xctx = [N.ctx_create(0, n, CH) for _ in range(TS - T)]
        xdst = [Mem.empty(n * slot) for _ in range(TS - T)]
        sctx, sdst, sds = ctxs + xctx, dsts + xdst, ds + [d.copy() for _ in range(TS - T)]
        for t in range(T, TS):                             # their workspaces exist before the clock starts
            N.transform_batch(params, sds[t], Mem.ptr(src), Mem.ptr(sdst[t]), nbytes(sdst[t]), MEM, ctx=sctx[t])
        fence()
        stamps = [[] for _ in range(TS)]

        def feeder(t):
            time.sleep(t * stag)
            for _ in range(BS):
                N.transform_batch(params, sds[t], Mem.ptr(src), Mem.ptr(sdst[t]), nbytes(sdst[t]), MEM, ctx=sctx[t])
                stamps[t].append(time.perf_counter())

        ts0 = time.perf_counter()
        th = [threading.Thread(target=feeder, args=(t,)) for t in range(TS)]
        # one look at the device UNDER this load (clocks, power, power cap, temperatures): the same build gives 18.5 - 21.8 GiB/s by box
        smi = threading.Thread(target=_device_state_under_load, args=(box_state, 3.0)) if not rehearse else None
        [x.start() for x in th]
        if smi is not None:
            smi.start()
        [x.join() for x in th]
        if smi is not None:
            smi.join()
        fence()
        whole = time.perf_counter() - ts0
        for t in range(TS):
            assert (sds[t]["status"] == 0).all() and (sds[t]["dst_len"] == d["dst_len"]).all() and (sds[t]["crc32c"] == d["crc32c"]).all()
        done_at = np.sort(np.concatenate([np.asarray(x) for x in stamps])) - ts0
        k0, k1 = int(len(done_at) * 0.2), int(len(done_at) * 0.8)
        slope = float(np.polyfit(done_at[k0:k1], np.arange(k0, k1), 1)[0])
        batch_gib = float(n) * CH / GiB
        sustained = {"metric": "GiB/s of original bytes with work always queued (%d callers %.2f s apart, %d batches of %d chunks each)" % (TS, stag, BS, n),
                     "value": round(slope * batch_gib, 4), "unit": "GiB/s", "callers": TS, "batches": TS * BS,
                     "method": "least-squares slope of batches completed over time, completions %d..%d of %d (no ramp-up, no drain)" % (k0, k1 - 1, len(done_at)),
                     "whole_run_gibs_incl_ramp_and_drain": round(TS * BS * batch_gib / whole, 4),
                     "ms_between_completions": round(1e3 / slope, 2), "device_state_under_this_load": box_state or None}
        for c in xctx:
            N.ctx_destroy(c)
        del xdst, sdst
    # ---- mixed load (outside the timed region): what a consumer's fetch costs while uploads keep the chip full -----------------------
    # A broker tiers and serves from the same GPU: fetchLogSegment -> ChunkCache.java:85-108 waits get.timeout.ms (10 s) for a chunk.  The
    # timed region's callers go on submitting their batches (more chunks queued than the chip holds) while this thread restores 1 and 4
    # chunks host -> host through a context of its own.  The compressor service leaves `reserved_cus` compute units alone for exactly this.
    mixed = None
    if rank == 0 and world == 1 and workload == "full" and T > 1 and not split and not args.no_mixed_load and not rehearse and n >= 4:
        try:
            hfr = dst[:4 * slot].cpu().numpy(); hbk = np.zeros(4 * CH, np.uint8)
            N.host_register(hfr); N.host_register(hbk)
            fctx = N.ctx_create(0, 4, CH)
            want4 = src[:4 * CH].cpu().numpy()

            def fetch(k):
                e_ = np.zeros(k, nat.DESC_DTYPE); e_["src_off"] = d["dst_off"][:k]; e_["src_len"] = d["dst_len"][:k]; e_["iv"] = d["iv"][:k]
                e_["dst_off"] = np.arange(k, dtype=np.uint64) * CH; e_["dst_cap"] = CH
                t1 = time.perf_counter()
                N.detransform_batch(params, e_, hfr, hbk, hbk.size, nat.MEM_HOST, ctx=fctx)
                dt_ = time.perf_counter() - t1
                assert (e_["status"] == 0).all()
                return dt_ * 1e3

            for k_ in (1, 4):
                fetch(k_)
            idle_ms = {k_: round(float(np.median([fetch(k_) for _ in range(7)])), 3) for k_ in (1, 4)}
            msv0 = N.service_stats(0)
            stop = [False]
            done = [0] * T

            mstamps = []

            def loader(t):
                while not stop[0]:
                    step(t)
                    done[t] += 1
                    with lock:
                        mstamps.append(time.perf_counter())

            th = [threading.Thread(target=loader, args=(t,)) for t in range(T)]
            tm0 = time.perf_counter()
            [x.start() for x in th]
            time.sleep(2.0)                                              # the chip is full
            lat = {1: [], 4: []}
            while time.perf_counter() - tm0 < 2.0 + args.mixed_load_seconds:
                for k_ in (1, 4):
                    lat[k_].append(fetch(k_))
                time.sleep(0.03)
            stop[0] = True
            [x.join() for x in th]
            el_ = time.perf_counter() - tm0
            exact = bool(np.array_equal(hbk, want4))
            N.host_unregister(hfr); N.host_unregister(hbk); N.ctx_destroy(fctx)
            msv1 = N.service_stats(0)
            rot_ = msv1["rotations"] - msv0["rotations"]
            mixed = {"metric": "latency of a fetch (tsx_detransform_batch of 1 / 4 chunks, host -> host, own context) while %d callers keep %d compressor chunks queued" % (T, T * n),
                     "reserved_cus": None if svc is None else svc["reserved_cus"], "compress_callers": T, "chunks_offered": T * n,
                     "compress_gibs_while_fetching": round(sum(done) * float(n) * CH / GiB / el_, 3),
                     "compress_gibs_while_fetching_note": "whole window incl. the callers' ramp and drain; `_slope` = least-squares slope of batch completions over the middle 60 %, as `sustained`",
                     "fetch_idle_ms": {str(k_): v for k_, v in idle_ms.items()}, "restored_bytes_exact": exact, "unit": "ms",
                     "compressor_launches_asked_to_end_early_by_a_waiting_fetch": int(rot_)}
            da_ = np.sort(np.asarray(mstamps)) - tm0
            if da_.size >= 8:
                q0, q1 = int(da_.size * 0.2), int(da_.size * 0.8)
                mixed["compress_gibs_while_fetching_slope"] = round(float(np.polyfit(da_[q0:q1], np.arange(q0, q1), 1)[0]) * float(n) * CH / GiB, 3)
                if sustained:
                    mixed["frac_of_sustained"] = round(mixed["compress_gibs_while_fetching_slope"] / sustained["value"], 3)
            for k_ in (1, 4):
                a_ = np.asarray(lat[k_])
                mixed["fetch_%d_under_load_ms" % k_] = {"n": int(a_.size), "p50": round(float(np.median(a_)), 2), "p95": round(float(np.percentile(a_, 95)), 2), "max": round(float(a_.max()), 2)}
        except Exception as ex:                                          # noqa: BLE001 - reported, never fatal for the line
            mixed = {"error": repr(ex)[:300]}
    # ---- the same chain on Kafka-shaped BINARY content (never `value`): v2 record batches, tsxform/synth.py "B" ------------------------
    # How far does K's number carry?  Sequence density sets the GiB/s; B has binary headers, varint framing and incompressible payloads.
    value_b = None
    if rank == 0 and world == 1 and workload == "full" and T > 1 and not split and not args.no_value_b and args.dist == "K" and (n >= 256 or rehearse):
        try:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
            from oracle import oracle as o
            DIST = min(64, n) if not rehearse else min(4, n)               # distinct chunks, replicated over the batch (generated on the host, ~2.5 s each:
            tg0 = time.perf_counter()                                      # a pool of fresh interpreters - this process holds a HIP runtime and must not fork)
            if rehearse:
                hb = [synth.gen_chunk("B", 1000, 0, c_, CH) for c_ in range(DIST)]
            else:
                with ProcessPoolExecutor(max(1, min(DIST, usable_cores(), 32)), mp_context=mp.get_context("spawn")) as ex:
                    hb = list(ex.map(_gen_b_chunk, [(1000, 0, c_, CH) for c_ in range(DIST)]))
            gen_s = time.perf_counter() - tg0
            srcb = Mem.empty(n * CH)
            for i in range(n):
                srcb[i * CH:(i + 1) * CH] = hb[i % DIST] if rehearse else torch.from_numpy(hb[i % DIST]).to(dev)
            bdst = [Mem.empty(n * slot) for _ in range(T)]               # (the timed region's outputs stay as they are: verified and restored below)
            # as many batches as the timed region has steps (the driver's command: 5 callers x 4 = 20), so that value_B reads next to `value` -
            # both include the ramp and the drain of their callers; never fewer than two per caller
            reps_b = max(2, -(-args.steps // T))
            legs = {}
            for pname, pval in (("1_5_7", nat.ZSTD_PROFILE_1_5_7), ("1_5_6", nat.ZSTD_PROFILE_1_5_6)):
                pb = nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=pval)
                dbs = [d.copy() for _ in range(T)]
                for x_ in dbs:
                    x_["status"] = 0; x_["dst_len"] = 0

                def bstep(t, pb=pb, dbs=dbs):
                    N.transform_batch(pb, dbs[t], Mem.ptr(srcb), Mem.ptr(bdst[t]), bdst[t].size if rehearse else bdst[t].numel(), MEM, ctx=ctxs[t])

                bstep(0); fence()
                tb0 = time.perf_counter()
                th = [threading.Thread(target=lambda t=t: [bstep(t) for _ in range(reps_b)]) for t in range(T)]
                [x.start() for x in th]
                [x.join() for x in th]
                fence()
                el_b = time.perf_counter() - tb0
                okb = all(bool((x_["status"] == 0).all()) for x_ in dbs)

                def check_b(i, pval=pval, dbs=dbs):
                    # byte equality on every distinct chunk: profile 1.5.7 with the REAL libzstd 1.5.7 + OpenSSL; profile 1_5_6 with the serial restatement
                    # (oracle/zstd_l3.c, profile 0: no libzstd 1.5.6 exists here) + OpenSSL - and with the real 1.5.7 wherever its splitter is idle
                    got = Mem.host(bdst[0], i * slot, i * slot + int(dbs[0]["dst_len"][i])).tobytes()
                    raw = hb[i].tobytes()
                    if pval == nat.ZSTD_PROFILE_1_5_7:
                        exp, _ = o.transform_chunk(o.COMPRESS | o.ENCRYPT | o.OPENSSL, synth.KEY, synth.AAD, dbs[0]["iv"][i].tobytes(), raw)
                    else:
                        exp = o.gcm_encrypt_chunk(synth.KEY, dbs[0]["iv"][i].tobytes(), synth.AAD, o.zstd_l3_compress(raw, 0), openssl=True)
                    return got == exp
                with ThreadPoolExecutor(max(1, min(32, usable_cores()))) as ex:
                    okb = okb and all(ex.map(check_b, range(DIST)))
                legs[pname] = {"value": round(T * reps_b * float(n) * CH / GiB / el_b, 4), "unit": "GiB/s", "batches": T * reps_b,
                               "mean_transformed_chunk_bytes": round(float(dbs[0]["dst_len"].astype(np.int64).mean()), 1), "exact_vs_oracle": bool(okb),
                               "checked_against": "libzstd %s + OpenSSL" % o.zstd_version() if pname == "1_5_7" else "oracle/zstd_l3.c profile 0 (unverified stand-in for libzstd 1.5.6) + OpenSSL"}
            value_b = {"metric": "GiB/s of original bytes, same chain and batch shape, content B (Kafka v2 record batches, %d distinct chunks replicated)" % DIST,
                       "distinct_chunks": DIST, "generated_in_s": round(gen_s, 1), **legs["1_5_7"], "value_B_1_5_6": legs["1_5_6"],
                       "note": "timed like `value`: %d callers x %d batches one after the other, ramp and drain of the callers included (rounds 5 and 6 before this line timed 2 batches per caller); "
                               "profile 1_5_6 = 1.5.7 without the pre-block splitter: what a broker with the reference's zstd-jni 1.5.6-9 would select" % (T, reps_b)}
            del srcb, bdst
        except Exception as ex:                                          # noqa: BLE001 - reported, never fatal for the line
            value_b = {"error": repr(ex)[:300]}
    # the timed region's extra callers are done: their workspaces (12.7 GiB each) and output buffers (8.5 GiB each) go back before the
    # legs below allocate their own (pooled contexts of the broker leg, host staging buffers)
    for c_ in ctxs[1:]:
        N.ctx_destroy(c_)
    del ctxs[1:], dsts[1:], ds[1:]
    if not rehearse:
        torch.cuda.empty_cache()
    if dist_on:
        elapsed = Mem.max_over_ranks(elapsed)
    assert (d["status"] == 0).all(), "chunk failures: %s" % d["status"][d["status"] != 0][:8]
    # whole-job bytes: weak scaling - every rank has nseg segments; split mode - the job is nseg segments in total
    job_chunks = nseg * cps if split else n * world
    total_bytes = float(job_chunks) * CH * args.steps
    value = total_bytes / GiB / elapsed
    out_sizes = d["dst_len"].astype(np.int64) if workload != "crc" else np.zeros(n, np.int64)

    # ---- parity spot-check against the oracle (outside the timed region) ------------------------------
    verified = None
    if rank == 0 and not args.no_verify:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as o
        # EVERY chunk of the timed batch (--verify-chunks k: k of them, spread over the batch): CRC32C and transformed bytes against
        # libzstd + OpenSSL, on all usable host cores (the oracle's C code runs without the interpreter lock)
        if n and 0 < args.verify_chunks < n:
            idx = sorted(set([0, 1, n // 2, n - 1] + [int(k) for k in np.linspace(0, n - 1, args.verify_chunks)]))
        else:
            idx = list(range(n))
        of = (o.COMPRESS if flags & nat.COMPRESS else 0) | o.ENCRYPT | o.OPENSSL

        def check(i):
            chunk = Mem.host(src, i * CH, (i + 1) * CH)
            if d["crc32c"][i] != o.crc32c(chunk):
                return "crc mismatch chunk %d" % i
            if workload != "crc":
                got = Mem.host(dst, i * slot, i * slot + int(d["dst_len"][i])).tobytes()
                exp, _ = o.transform_chunk(of, synth.KEY, synth.AAD, d["iv"][i].tobytes(), chunk.tobytes())
                if got != exp:
                    return "transformed bytes differ from the oracle for chunk %d (libzstd %s)" % (i, o.zstd_version())
            return None
        tv0 = time.perf_counter()
        with ThreadPoolExecutor(max(1, min(32, usable_cores()))) as ex:
            bad = [r for r in ex.map(check, idx) if r]
        assert not bad, bad[:4]
        verified = len(idx)
        verify_seconds = round(time.perf_counter() - tv0, 2)

    # ---- the inverse chain (fetchLogSegment side), outside the timed region: BASELINE configs[4] asks for the round trip --
    # tsx_detransform_batch over the batch just produced (GCM tag check + decrypt, Zstd frame decode, CRC32C of the restored
    # bytes), device resident like the forward step; restored bytes must equal the source segment byte for byte.
    inverse = None
    if workload != "crc" and not args.no_inverse:
        back = Mem.empty(max(n, 1) * CH)
        e = np.zeros(n, nat.DESC_DTYPE)
        e["src_off"] = d["dst_off"]; e["src_len"] = d["dst_len"]; e["iv"] = d["iv"]
        e["dst_off"] = np.arange(n, dtype=np.uint64) * CH; e["dst_cap"] = CH
        bsz = back.size if rehearse else back.numel()
        if n:
            N.detransform_batch(params, e, Mem.ptr(dst), Mem.ptr(back), bsz, MEM, ctx=ctx)     # warm-up
        fence()
        reps = 3
        t1 = time.perf_counter()
        for _ in range(reps):
            if n:
                N.detransform_batch(params, e, Mem.ptr(dst), Mem.ptr(back), bsz, MEM, ctx=ctx)
        fence()
        inv_s = (time.perf_counter() - t1) / reps
        if dist_on:
            inv_s = Mem.max_over_ranks(inv_s)
        tm = N.ctx_timing(ctx)
        exact = bool((e["status"] == 0).all() and (e["dst_len"] == CH).all() and (e["crc32c"] == d["crc32c"]).all() and Mem.equal(back, src))
        if dist_on:
            exact = Mem.max_over_ranks(0.0 if exact else 1.0) == 0.0                      # every rank's round trip
        # reported, not asserted: a fetch-side failure must not take the forward measurement's line with it
        inverse = {"metric": "GiB/s of restored bytes, tsx_detransform_batch (GCM verify+decrypt, Zstd decode, CRC32C), one batch at a time",
                   "value": round(float(job_chunks) * CH / GiB / inv_s, 4), "unit": "GiB/s", "ms_per_batch": round(inv_s * 1e3, 3),
                   "stage_ms": {"gcm": round(tm.gcm_ms, 3), "unzstd": round(tm.unzstd_ms, 3), "crc": round(tm.crc_ms, 3)},
                   "round_trip_exact": exact}
        # roofline of the inverse chain's dominant kernel (the frame decoder): algorithmic bytes = frame read + chunk written
        alg_inv = float(n) * (CH + float(d["dst_len"].astype(np.int64).mean()) - (28 if flags & nat.ENCRYPT else 0))
        dom_ms = max(tm.unzstd_ms, tm.gcm_ms, tm.crc_ms)
        dom_k = "zstd_decompress_kernel" if dom_ms == tm.unzstd_ms else ("gcm_ctr_ghash_kernel" if dom_ms == tm.gcm_ms else "crc32c_partial_kernel")
        rec, fresh = pmc_record("detransform/%s/%d" % (args.dist, n))
        inv_traffic = rec["hbm_bytes_per_launch"] if rec and fresh else None
        ach = alg_inv / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # fetch latency: a single chunk and a 4-chunk prefetch window (16 MiB, ChunkCache's default window is a few chunks), host buffers
        # in -> host buffers out, registered; small batches take the block-parallel decoder form (csrc/zstd_dec_blocks.hip)
        if rank == 0 and world == 1 and not rehearse and n >= 4:
            try:
                kmax = 256 if n >= 256 else 4
                hfr = dst[:kmax * slot].cpu().numpy(); hbk = np.zeros(kmax * CH, np.uint8)
                N.host_register(hfr); N.host_register(hbk)
                lat = {}
                for k_ in ((1, 4, 64, 256) if kmax == 256 else (1, 4)):
                    ee = e[:k_].copy(); ee["dst_off"] = np.arange(k_, dtype=np.uint64) * CH
                    tt = []
                    for _ in range(7):
                        t1 = time.perf_counter()
                        N.detransform_batch(params, ee, hfr, hbk, hbk.size, nat.MEM_HOST, ctx=ctx)
                        tt.append(time.perf_counter() - t1)
                    okk = bool((ee["status"] == 0).all() and np.array_equal(hbk[:k_ * CH], src[:k_ * CH].cpu().numpy()))
                    lat[k_] = (round(float(np.median(tt[2:])) * 1e3, 3), okk)
                N.host_unregister(hfr); N.host_unregister(hbk)
                inverse["single_chunk_ms"] = lat[1][0]; inverse["window4_ms"] = lat[4][0]
                if 256 in lat:
                    # a consumer catching up (ChunkCache.java:159-184 with a large prefetch.max.size): 64 chunks and a whole segment, cut into
                    # co-resident pieces (copy-in, block-form decode and copy-out overlap: csrc/tsx_api.hip, run_batch_inner)
                    inverse["window64_ms"] = lat[64][0]; inverse["segment_ms"] = lat[256][0]
                    inverse["segment_gibs"] = round(256.0 * CH / GiB / (lat[256][0] * 1e-3), 3)
                inverse["small_batches_exact"] = all(v[1] for v in lat.values())
                inverse["small_batch_note"] = "host -> host, registered buffers, median of 5; batches of <= 256 chunks decode one workgroup per block"
            except nat.TsxError as ex:
                inverse["small_batch_error"] = str(ex)
        inverse["roofline"] = {"bound": "hbm", "kernel": dom_k, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": inv_traffic,
                               "traffic_source": None if not rec else (rec.get("source") if fresh else "STALE: %s was measured on another build of the kernel" % rec.get("source")),
                               "ms_per_launch": round(dom_ms, 4),
                               "algorithmic_bytes_per_launch": int(alg_inv)}
        del back

    # ---- roofline of the dominant kernel (HIP events on the library's own stream, per launch) ----------
    if rehearse:
        stage["zstd" if workload == "full" else workload.split("_")[0]] += 1e-9          # (HIP events of the emulator say nothing)
    dom = max(stage, key=lambda k: stage[k])
    ms = stage[dom] / max(launches[dom], 1)
    mean_out = float(out_sizes.mean()) if workload != "crc" else 0.0
    if dom == "crc":
        alg = n * (CH + 4.0)                                                   # N read + 4 B written per chunk
    elif dom == "gcm":
        m = mean_out - 28 if workload != "crc" else CH                        # GCM input = frame (or chunk)
        alg = n * (m + m + 28.0)                                               # m read + m + 28 written
    else:
        # the whole chain of a chunk runs in its compressor wave (CRC head, GCM tail): N bytes read, the transformed chunk
        # (frame, + IV and tag when encrypting) written
        alg = n * (CH + mean_out)
    launches_meta = None
    if dom == "zstd" and svc is not None and svc["launches"] > 0 and svc["kernel_ms"] > 0:
        # the compressor is ONE persistent kernel per device (csrc/tsx_internal.h): a launch does as many chunks as were queued while it
        # lived.  Per launch: the timed region's chunks / its launches, against the launches' average duration - HIP events on the service's
        # own stream, read through tsx_service_stats (the same launches rocprofv3 --kernel-trace sees as zstd_service_kernel).
        cpl = float(svc["chunks"]) / svc["launches"]
        ms = svc["kernel_ms"] / svc["launches"]
        alg = cpl * (CH + mean_out)
        launches_meta = {"launches_in_timed_region": int(svc["launches"]), "chunks_per_launch": round(cpl, 1), "started_by_watchdog": int(svc["watchdog_launches"]),
                         "waves_per_launch": int(svc["waves"]), "reserved_cus": int(svc["reserved_cus"]),
                         "launches_with_guest_waves_on_the_reserved_cus": int(svc["guest_launches"]), "compute_units": int(svc["compute_units"]), "cu_keys_seen": int(svc["cu_keys_seen"]),
                         # waves that the hardware's scheduler saved and restored onto a reserved CU, counted when they leave it: a timed region that
                         # met such an event is up to 10 % slower
                         "waves_that_left_reserved_cus_after_a_save_restore": int(svc["relocated_waves"])}
    achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM bytes per launch of that kernel from the committed PMC passes (tools/pmc_zstd.sh -> profiles/pmc_traffic.json:
    # FETCH_SIZE + WRITE_SIZE of the same kernel build and workload); null when no such measurement is recorded
    traffic = None
    binding = None
    rec, fresh = pmc_record("%s/%s/%d" % (workload, args.dist, n))
    if rec and fresh:
        traffic = rec["hbm_bytes_per_launch"]
        scale = 1.0
        if dom == "zstd" and launches_meta is not None and rec.get("chunks"):
            scale = launches_meta["chunks_per_launch"] / float(rec["chunks"])      # the PMC pass was one launch of rec["chunks"] chunks
            traffic = int(traffic * scale)
        if dom == "zstd" and "tcc_ea_rdreq" in rec:
            # The resource that actually binds the hash-table parser: 64-B lines moved between L2 and memory at random addresses (one
            # per 4-byte table probe / insertion).  Requests per launch from the committed PMC passes (TCC_EA0_RDREQ + WRREQ), rate over
            # the whole timed region (all launches, as they overlapped).  The ceiling is what the chip sustains for the same mix with
            # waves that do nothing else (tools/ubench/mix.hip): 36-46 G lines/s by box and run - a band, so this is a diagnosis,
            # not a roofline; the roofline above is the contract's (HBM streaming peak).
            req = float(rec["tcc_ea_rdreq"] + rec["tcc_ea_wrreq"]) * scale
            rate = req * (svc["launches"] if launches_meta is not None else args.steps) / elapsed / 1e9
            binding = {"resource": "random 64-B line requests L2<->HBM (table probes + insertions)", "requests_per_launch": int(req),
                       "requests_per_sequence": rec.get("requests_per_sequence"), "achieved": round(rate, 2), "unit": "G requests/s"}
            if line_rate and "parser_mix" in line_rate:
                pk = max(float(line_rate["parser_mix"].get("g_requests_per_s_best_set", line_rate["parser_mix"]["g_requests_per_s"])),
                         float((line_rate.get("parser_mix_pipelined") or {}).get("g_requests_per_s_best_set", 0.0)))      # (the better of: every iteration waits for its data / four reads in flight)
                binding.update({"peak": pk, "frac": round(rate / pk, 3), "peak_reads_only": line_rate["reads_only"]["g_requests_per_s"],
                                "peak_source": "tools/ubench/line_rate on this box right before the timed region: %d waves x %d iterations of %d reads + %d rewrites + %d blind stores, the best of four sets of tables"
                                               % (line_rate["parser_mix"]["waves"], line_rate["parser_mix"]["iters"], line_rate["parser_mix"]["reads"],
                                                  line_rate["parser_mix"]["rewrites"], line_rate["parser_mix"]["blind_stores"])})
                if rate > pk:
                    binding["note"] = ("the compressor moves MORE line requests per second than the uniform-random microbenchmark: that benchmark is not an upper bound for this "
                                       "access pattern (a probe's line is rewritten while it is still in L2; several lanes of a step share lines), so 'bound by the random-line "
                                       "rate' is a diagnosis of where the time goes (PMC: x75 the algorithmic bytes), not a proven ceiling")
            else:
                binding.update({"peak": None, "frac": None, "peak_source": "tools/ubench/line_rate did not run: %s" % (line_rate or {}).get("error", "skipped")})
    roofline = {"bound": "hbm", "kernel": {"crc": "crc32c_partial_kernel", "gcm": "gcm_ctr_ghash_kernel", "zstd": "zstd_service_kernel"}[dom],
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": None if not rec else (rec.get("source") if fresh else "STALE: %s was measured on another build of the kernel - rerun tools/pmc_zstd.sh + tools/pmc_traffic.py" % rec.get("source")),
                "ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(alg),
                "launches_in_flight": 1 if launches_meta is not None else T, "achieved_aggregate": round(achieved * (1 if launches_meta is not None else T), 2),
                "service": launches_meta, "callers": T,
                "stage_ms_per_step": {k: round(v / args.steps, 4) for k, v in stage.items()}, "binding_resource": binding, "line_rate_probe": line_rate}

    # ---- CPU baseline: the oracle port (libzstd + OpenSSL GCM + CRC32C) on 1, 10 and all usable host cores, bounded samples ---------
    # (T = 1 is the per-thread rate of the reference's chain, T = 10 the reference's default RLM copier pool, "all" what the
    # box could do if every core ran uploads).  The headline object is the all-cores leg.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not rehearse:
        from oracle import oracle as o
        cores = usable_cores()
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except (OSError, IndexError):
            model = "unknown"
        of = ((o.COMPRESS if flags & nat.COMPRESS else 0) | (o.ENCRYPT if flags & nat.ENCRYPT else 0) | o.CRC | o.OPENSSL)
        per_thread = {"full": 12, "gcm_crc": 64, "crc": 256}[workload]          # chunks per thread: ~0.5-1 s of work each
        # Which libzstd is TIMED: parity is checked against libzstd 1.5.7, but the only loadable 1.5.7 here is the copy bundled with
        # Pillow, which runs level 3 five to six times slower than the distribution's library on the same input (28 vs 168 MiB/s per
        # thread on the build container; an unoptimised build).  A baseline that slow would flatter the device, so the chain is timed
        # with the FASTEST real libzstd that can be loaded (same level-3 double-fast work), and the Zstd stage of an optimised 1.5.7
        # (the copy inside pyarrow - its level-3 frames are byte-identical to the parity library's) is reported next to it.
        timed_lib, timed_ver, probe = None, o.zstd_version() if flags & nat.COMPRESS else None, {}
        if flags & nat.COMPRESS:
            one = src[:CH].cpu().numpy().tobytes()
            best = None
            for cand in (None, "/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"):
                if cand is not None and not os.path.exists(cand):
                    continue
                if not o.zstd_open(cand):
                    continue
                t1 = time.perf_counter(); o.zstd_compress_chunk(one); dt = time.perf_counter() - t1
                probe[o.zstd_version() + (" (parity library)" if cand is None else "")] = round(CH / (1 << 20) / dt, 1)
                if best is None or dt < best[0]:
                    best = (dt, cand, o.zstd_version())
            timed_lib, timed_ver = best[1], best[2]
            o.zstd_open(timed_lib)
        legs = []
        # BASELINE configs[0] times the reference "to filesystem backend": the transformed chunks also go to a file on tmpfs (what
        # FileSystemStorage.upload does with the transformed stream), in-memory rate next to it
        sink_dir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        sink = os.path.join(sink_dir, "tsx_cpu_baseline_%d.log" % os.getpid()) if sink_dir else None
        try:
            for nthr in sorted(set([1, min(10, cores), cores])):
                sample = min(n, max(per_thread * nthr, 24 if workload == "full" else 256))
                host = src[:sample * CH].cpu().numpy()
                ivs = np.ascontiguousarray(d["iv"][:sample]).reshape(-1)
                secs_mem, _, _, _, _ = o.chain_run_threads(of, synth.KEY, synth.AAD, host, CH, ivs, nthr)
                secs = o.chain_run_threads(of, synth.KEY, synth.AAD, host, CH, ivs, nthr, sink=sink)[0] if sink else secs_mem
                legs.append({"threads": nthr, "value": round(sample * CH / GiB / secs, 4), "unit": "GiB/s", "sample_chunks": sample,
                             "MiB_per_s_per_thread": round(sample * CH / (1 << 20) / secs / nthr, 1),
                             "in_memory_only_gibs": round(sample * CH / GiB / secs_mem, 4)})
        finally:
            if flags & nat.COMPRESS:
                o.zstd_open(None)                                         # parity checks go back to 1.5.7
        zstd157 = None
        if flags & nat.COMPRESS:
            try:                                                          # Zstd stage alone, optimised 1.5.7 build (pyarrow releases the GIL)
                import pyarrow as pa
                from concurrent.futures import ThreadPoolExecutor
                codec = pa.Codec("zstd", compression_level=3)
                same = codec.compress(one, asbytes=True) == o.zstd_compress_chunk(one)
                rows = []
                for nthr in sorted(set([1, min(10, cores), cores])):
                    sample = min(n, max(4 * nthr, 8))
                    bufs = [src[i * CH:(i + 1) * CH].cpu().numpy().tobytes() for i in range(sample)]
                    t1 = time.perf_counter()
                    with ThreadPoolExecutor(nthr) as ex:
                        list(ex.map(lambda b: len(codec.compress(b, asbytes=True)), bufs))
                    dt = time.perf_counter() - t1
                    rows.append({"threads": nthr, "GiB_per_s": round(sample * CH / GiB / dt, 4), "MiB_per_s_per_thread": round(sample * CH / (1 << 20) / dt / nthr, 1)})
                zstd157 = {"library": "libzstd inside pyarrow %s, level 3" % pa.__version__, "frames_identical_to_parity_library": bool(same), "by_threads": rows}
            except Exception as e:                                        # reported, never fatal
                zstd157 = {"error": str(e)[:200]}
        top = legs[-1]
        cpu = {"value": top["value"], "unit": "GiB/s", "cores": top["threads"], "kind": "port",
               "sample": "%d x 4 MiB chunks (%s) through oracle/chain.c: libzstd %s level 3 + OpenSSL AES-256-GCM + CRC32C, %d threads, transformed chunks written to %s"
                         % (top["sample_chunks"], args.dist, timed_ver if flags & nat.COMPRESS else "n/a", top["threads"], "a tmpfs file (/dev/shm)" if sink else "memory only (no tmpfs)"),
               "sink": "tmpfs file" if sink else "memory", "jvm_on_box": __import__("shutil").which("java"),
               "not_the_reference_java_chain": "no JDK in this image or on the GPU box (java not on PATH): a C port of the chain with the real libzstd / OpenSSL stands in",
               "by_threads": legs, "nproc": os.cpu_count(), "usable_cores": cores, "cpu_model": model,
               "libzstd_timed": timed_ver, "libzstd_timed_path": timed_lib or "parity library", "libzstd_parity": o.zstd_version() if flags & nat.COMPRESS else None,
               "libzstd_one_chunk_probe_MiB_per_s": probe, "zstd_stage_alone_optimised_1_5_7": zstd157}

    # ---- end to end: the same batch host -> host through TSX_MEM_HOST / TSX_MEM_HOST_PACKED (what the JNI shim uses), PCIe included.
    # Never `value`.  Pageable buffers first (the runtime stages them), then the same buffers pinned with tsx_host_register.
    e2e = None
    if rank == 0 and world == 1 and workload != "crc" and not args.no_end_to_end and not rehearse:
        PCIE = 64.0                                                       # GB/s per direction, PCIe 5.0 x16
        hsrc = src.cpu().numpy()
        hdst = np.zeros(n * slot, np.uint8)
        rows = []
        for kind, label in ((nat.MEM_HOST, "slots"), (nat.MEM_HOST_PACKED, "packed")):
            for pinned in (False, True):
                if pinned:
                    try:
                        N.host_register(hsrc); N.host_register(hdst)
                    except nat.TsxError:
                        break
                de = d.copy(); de["status"] = 0; de["dst_len"] = 0
                best = None
                for _ in range(2):
                    t1 = time.perf_counter()
                    N.transform_batch(params, de, hsrc, hdst, hdst.size, kind, ctx=ctx)
                    el = time.perf_counter() - t1
                    best = el if best is None else min(best, el)
                tm = N.ctx_timing(ctx)
                ok = bool((de["status"] == 0).all() and (de["dst_len"] == d["dst_len"]).all())
                moved = (float(n) * CH + float(de["dst_len"].sum())) / 1e9
                rows.append({"dst_layout": label, "host_memory": "registered" if pinned else "pageable", "ms": round(best * 1e3, 2),
                             "gibs": round(float(n) * CH / GiB / best, 4), "pcie_frac": round(moved / best / (2 * PCIE), 4),
                             "kernels_ms": round(tm.crc_ms + tm.zstd_ms + tm.gcm_ms, 2), "same_sizes_as_device_run": ok})
                if pinned:
                    N.host_unregister(hsrc); N.host_unregister(hdst)
        # ... and the way a broker drives it: T caller threads, each with its own context and its own host output buffer, batches in
        # flight - one thread's copies overlap the other threads' kernels
        conc = None
        if T > 1:
            conc = []
            # T callers as in the timed region, then T + 1 with the source buffer pinned (tsx_host_register, what the JVM side's reusable
            # direct buffers are): a caller's own copy-in precedes its kernel, so one more caller keeps T batches on the device
            Th = min(T, 3)                                                # (each caller owns an 8.5 GiB host output buffer)
            # ... and a third row with the output buffers registered as well: the compressor waves then write their chunks straight into
            # them (zero-copy output) - what the JVM side's reused direct buffers get
            for callers, pin_src, pin_dst in ((Th, False, False), (Th + 1, True, False), (Th + 1, True, True)):
                cx = list(ctxs) + [N.ctx_create(0, n, CH) for _ in range(callers - len(ctxs))]
                hdsts = [hdst] + [np.zeros(n * slot, np.uint8) for _ in range(callers - 1)]
                des = [d.copy() for _ in range(callers)]
                reps = 2
                pinned = False
                pinned_dst = []
                if pin_src:
                    try:
                        N.host_register(hsrc); pinned = True
                    except nat.TsxError:
                        pass
                if pin_dst:
                    try:
                        for hb_ in hdsts:
                            N.host_register(hb_); pinned_dst.append(hb_)
                    except nat.TsxError:
                        pass

                def hworker(t):
                    for _ in range(reps):
                        N.transform_batch(params, des[t], hsrc, hdsts[t], hdsts[t].size, nat.MEM_HOST_PACKED, ctx=cx[t])

                for t in range(callers):                                  # first touch of every output page outside the timed part
                    N.transform_batch(params, des[t], hsrc, hdsts[t], hdsts[t].size, nat.MEM_HOST_PACKED, ctx=cx[t])
                t1 = time.perf_counter()
                th = [threading.Thread(target=hworker, args=(t,)) for t in range(callers)]
                [x.start() for x in th]
                [x.join() for x in th]
                el = time.perf_counter() - t1
                ok = all(bool((de["status"] == 0).all() and (de["dst_len"] == d["dst_len"]).all()) for de in des)
                conc.append({"callers": callers, "batches": callers * reps, "dst_layout": "packed",
                             "host_memory": ("source and outputs registered (zero-copy output)" if pinned_dst else "source registered, outputs pageable") if pinned else "pageable",
                             "ms_per_batch": round(el / (callers * reps) * 1e3, 2), "gibs": round(float(n) * CH * callers * reps / GiB / el, 4),
                             "pcie_frac": round((float(n) * CH + float(d["dst_len"].sum())) * callers * reps / 1e9 / el / (2 * PCIE), 4),
                             "same_sizes_as_device_run": ok})
                if pinned:
                    N.host_unregister(hsrc)
                for hb_ in pinned_dst:
                    N.host_unregister(hb_)
                for c_ in cx[len(ctxs):]:
                    N.ctx_destroy(c_)
                del hdsts
        # ... and in the broker's real shape (reference README.md:218-222, RemoteStorageManager.java:400-432: >= 10 RLM upload threads, one
        # segment each): `callers` threads, every call ONE 256-chunk segment, context-less (pooled contexts, as the JNI shim calls),
        # TSX_MEM_HOST_PACKED from a registered source into a registered per-thread output buffer - what GpuTransformChunkEnumeration
        # issues (20 callers = 10 threads with one batch of read-ahead each; 32 = 16 such threads).  The loop is closed - a caller's next
        # call follows its last - so the row is bounded by the chunks the callers OFFER: 2560 / 5120 / 8192 against the 6144 the chip holds.
        # The leg lives in tools/broker_leg.py.  One thing about THIS process distorts it: a process has one HIP runtime, the first one loaded,
        # and here that is the one torch bundles (7.0.2), which moves device -> host copies with blit KERNELS; the system's runtime (7.2, what a
        # broker's JVM loads through libtsxform.so) uses the SDMA engines.  A copy kernel needs CU slots and waits for them on a chip full of
        # second-long compressor waves: standalone, the same leg at 32 callers reads 11.7 GiB/s with torch in the process and 14.3 without, and
        # no longer falls below the 20-caller row (profiles/r03_broker_with_and_without_torch.jsonl, r03_copy_engine_probe.txt).
        # --broker-subprocess runs the leg as a torch-free child - but while this process holds the device as well the child is slower than
        # either (two processes' queues are time-sliced: 8.9 / 13.5 / 12.8), so the default stays in-process: the rows are a LOWER bound of
        # what a torch-free process sees from 32 callers up.
        broker = None
        if broker_first is not None:
            broker, bsizes = broker_first
            for b_ in broker:
                if "gibs" in b_:
                    # the child ran before this process knew the sizes: compare what its callers saw with this process's device-resident run
                    b_["same_sizes_as_device_run"] = bool(b_.get("same_sizes_as_device_run")) and bsizes is not None and n >= bsizes.size and \
                        bool((bsizes.astype(np.int64) == d["dst_len"][:bsizes.size].astype(np.int64)).all())
                    b_["frac_of_device_resident_value"] = round(b_["gibs"] / value, 3)
                    b_["process"] = "tools/broker_leg.py as a child that ran FIRST and owned the device alone, no torch: the system's HIP runtime, as a JVM loads it"
        elif T > 1 and n >= 256 and not args.no_broker:
            B = 256
            bseg = min(2, n // B) if args.broker_subprocess else n // B    # distinct segments, the callers take them in turn
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import broker_leg
            if not args.broker_subprocess:
                broker = broker_leg.run(N, nat, params, hsrc[:bseg * B * CH], d["iv"][:bseg * B], d["dst_len"][:bseg * B], (10, 20, 32), B, CH, 8.0)
            else:
                pass  # (subprocess: module-level import)
                import tempfile
                shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
                tmpd = tempfile.mkdtemp(prefix="tsx_broker_", dir=shm)
                try:
                    np.save(os.path.join(tmpd, "src.npy"), hsrc[:bseg * B * CH]); np.save(os.path.join(tmpd, "ivs.npy"), d["iv"][:bseg * B])
                    np.save(os.path.join(tmpd, "expect.npy"), d["dst_len"][:bseg * B])
                    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "broker_leg.py"), "--src", os.path.join(tmpd, "src.npy"),
                                         "--ivs", os.path.join(tmpd, "ivs.npy"), "--expect", os.path.join(tmpd, "expect.npy"), "--callers", "10,20,32",
                                         "--batch", str(B), "--chunk", str(CH), "--profile", str(profile)], capture_output=True, text=True, timeout=600)
                    lines = [ln for ln in cp.stdout.strip().splitlines() if ln.startswith("[")]
                    if cp.returncode == 0 and lines:
                        broker = json.loads(lines[-1])
                    else:
                        broker = [{"error": "tools/broker_leg.py failed (rc %d): %s" % (cp.returncode, cp.stderr.strip()[-300:])}]
                except (OSError, subprocess.SubprocessError, ValueError) as e:
                    broker = [{"error": "tools/broker_leg.py: %r" % (e,)}]
                finally:
                    import shutil
                    shutil.rmtree(tmpd, ignore_errors=True)
            for b_ in broker:
                if "gibs" in b_:
                    b_["frac_of_device_resident_value"] = round(b_["gibs"] / value, 3)
                    b_["process"] = "tools/broker_leg.py, no torch: the system's HIP runtime, as a JVM loads it" if args.broker_subprocess else "this one (torch's bundled HIP runtime: D2H copies are blit kernels)"
        e2e = {"metric": "GiB/s of original bytes, host buffers in -> host buffers out (PCIe inclusive)",
               "chunks": n, "pcie_peak_GBs_per_direction": PCIE, "one_batch_at_a_time": rows, "batches_in_flight": conc, "broker": broker,
               "value": max([r["gibs"] for r in rows] + [c_["gibs"] for c_ in (conc or [])] + [b_["gibs"] for b_ in (broker or []) if "gibs" in b_]) if rows else None, "unit": "GiB/s"}
        del hsrc, hdst

    # ---- BASELINE configs[1] and configs[2] on one 1 GiB segment (never `value`): CRC32C only; AES-256-GCM + CRC32C - what a broker runs
    # whenever the producers compress (RemoteStorageManager.java:381-398 leaves Zstd out then, EncryptionChunkEnumeration.java:66-84) ----
    configs = None
    if rank == 0 and world == 1 and workload == "full" and not rehearse and not args.no_configs and n >= 256:
        configs = {}
        m = 256
        from oracle import oracle as o
        for name, fl in (("crc", nat.CRC), ("gcm_crc", nat.ENCRYPT | nat.CRC)):
            try:
                sl = (N.transformed_bound(CH, fl) + 63) // 64 * 64
                dd = d[:m].copy(); dd["dst_off"] = np.arange(m, dtype=np.uint64) * sl; dd["dst_cap"] = sl; dd["status"] = 0; dd["dst_len"] = 0
                outb = Mem.empty(m * sl if name != "crc" else 64)
                pr = nat.Native.make_params(fl, synth.KEY, synth.AAD)

                def run_dev():
                    if name == "crc":
                        N.crc32c_batch(dd, Mem.ptr(src), MEM, ctx=ctx)
                    else:
                        N.transform_batch(pr, dd, Mem.ptr(src), Mem.ptr(outb), outb.numel(), MEM, ctx=ctx)

                run_dev(); fence()
                reps = 10
                kms = 0.0
                t1 = time.perf_counter()
                for _ in range(reps):
                    run_dev()
                    tm_ = N.ctx_timing(ctx)
                    kms += tm_.crc_ms if name == "crc" else tm_.gcm_ms
                fence()
                dt_ = (time.perf_counter() - t1) / reps
                kms /= reps
                okc = bool((dd["status"] == 0).all() and (dd["crc32c"] == d["crc32c"][:m]).all())
                for i in (0, m - 1):                                      # two chunks against the oracle chain (OpenSSL GCM, CRC32C)
                    chunk = Mem.host(src, i * CH, (i + 1) * CH)
                    okc = okc and int(dd["crc32c"][i]) == o.crc32c(chunk)
                    if name != "crc":
                        exp, _ = o.transform_chunk(o.ENCRYPT | o.OPENSSL, synth.KEY, synth.AAD, dd["iv"][i].tobytes(), chunk.tobytes())
                        okc = okc and Mem.host(outb, i * sl, i * sl + int(dd["dst_len"][i])).tobytes() == exp
                alg_ = m * (CH + 4.0) if name == "crc" else m * (2.0 * CH + 28.0)
                ach_ = alg_ / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
                rec_, fresh_ = pmc_record("%s/%s/%d" % (name, args.dist, m))
                # host -> host, registered buffers (what the JNI shim hands over)
                hs_ = src[:m * CH].cpu().numpy(); hd_ = np.zeros(m * sl if name != "crc" else 64, np.uint8)
                N.host_register(hs_); N.host_register(hd_)
                best = None
                for _ in range(3):
                    de_ = dd.copy()
                    t1 = time.perf_counter()
                    if name == "crc":
                        N.crc32c_batch(de_, hs_, nat.MEM_HOST, ctx=ctx)
                    else:
                        N.transform_batch(pr, de_, hs_, hd_, hd_.size, nat.MEM_HOST, ctx=ctx)
                    el_ = time.perf_counter() - t1
                    best = el_ if best is None else min(best, el_)
                okh = bool((de_["status"] == 0).all() and (de_["crc32c"] == d["crc32c"][:m]).all())
                N.host_unregister(hs_); N.host_unregister(hd_)
                configs[name] = {"workload": "1 GiB segment, 256 x 4 MiB chunks, %s (BASELINE configs[%d])" % ("CRC32C only" if name == "crc" else "AES-256-GCM + CRC32C", 1 if name == "crc" else 2),
                                 "device_resident_gibs": round(m * float(CH) / GiB / dt_, 3), "ms_per_batch": round(dt_ * 1e3, 3),
                                 "host_to_host_gibs": round(m * float(CH) / GiB / best, 3), "exact_vs_oracle": bool(okc and okh),
                                 "roofline": {"bound": "hbm", "kernel": "crc32c_partial_kernel" if name == "crc" else "gcm_ctr_ghash_kernel", "achieved": round(ach_, 2), "peak": HBM_PEAK_GBS,
                                              "unit": "GB/s", "frac": round(ach_ / HBM_PEAK_GBS, 5), "ms_per_launch": round(kms, 4), "algorithmic_bytes_per_launch": int(alg_),
                                              "traffic": rec_["hbm_bytes_per_launch"] if rec_ and fresh_ else None,
                                              "traffic_source": None if not rec_ else (rec_.get("source") if fresh_ else "STALE: measured on another build of the kernel")}}
                del outb, hs_, hd_
            except Exception as ex:                                      # noqa: BLE001 - reported, never fatal for the line
                configs[name] = {"error": repr(ex)[:300]}

    if rank == 0:
        line = {
            "metric": "GiB/s segment chunk transform (Zstd+AES+CRC), 4MiB chunks",
            "value": round(value, 4), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if split else "weak",
            "vs_baseline": None, "dtype": "u8",
            "data": "synthetic" if not rehearse else "synthetic; REHEARSAL on the CPU emulator (tests/emu) - exercises the rank logic, measures nothing",
            "config": {"workload": {"full": "%dx1GiB segments/GPU, 4 MiB chunks, Zstd(L3)+AES-256-GCM+CRC32C (BASELINE configs[3])" % nseg,
                                    "gcm_crc": "1 GiB segment, 4 MiB chunks, AES-256-GCM+CRC32C (BASELINE configs[2])",
                                    "crc": "1 GiB segment, 4 MiB chunks, CRC32C only (BASELINE configs[1])"}[workload],
                       "stages": workload, "segments_per_gpu": None if split else nseg, "segments_total": nseg if split else nseg * world, "chunks_per_gpu": n, "chunk_bytes": CH, "content": args.dist,
                       "zstd_profile": args.profile if flags & nat.COMPRESS else None,
                       "mean_transformed_chunk_bytes": round(mean_out, 1), "residency": "device (HBM) in/out",
                       "parallelism": ("chunk-range split of %d segment(s) over %d rank(s), all-gather of transformed sizes per step" % (nseg, world)) if split
                                      else "segment-major shard, %d rank(s), no data-path collective" % world,
                       "segments_of_rank0": [int(x) for x in my_segments], "chunks_of_rank0": n,
                       "chunk_index_positions_sha": None if not index else __import__("hashlib").sha256(
                           b"".join(np.asarray(index[k][1], np.int64).tobytes() for k in sorted(index))).hexdigest()[:16],
                       "object_gathered_on_rank0_sha": None if not objects else __import__("hashlib").sha256(
                           b"".join((objects[k].cpu().numpy() if hasattr(objects[k], "cpu") else np.asarray(objects[k])).tobytes() for k in sorted(objects))).hexdigest()[:16],
                       "batches_in_flight": T, "gibs_one_batch_at_a_time": None if single is None else round(single, 4),
                       "hip_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "cpu_affinity": cpu_affinity,
                       "process_group": None if not dist_on else {
                           "backend": args.backend + (" (= RCCL)" if args.backend == "nccl" else ""), "world": world, "forced_on_one_rank": bool(args.force_dist and world == 1),
                           "ran": ["barrier", "all_reduce(MAX)"] + (["all_gather(sizes)"] if split else []) + (["p2p slice -> owner"] if split and args.gather_object and (world > 1 or args.backend == "nccl") else [])},
                       "verified_chunks_vs_oracle": verified, "verify_seconds": None if verified is None else verify_seconds},
            "roofline": roofline, "cpu_baseline": cpu, "sustained": sustained, "value_B": value_b, "mixed_load": mixed, "configs": configs, "end_to_end": e2e, "detransform": inverse,
        }
        print(json.dumps(line))
    for c in ctxs:
        N.ctx_destroy(c)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
