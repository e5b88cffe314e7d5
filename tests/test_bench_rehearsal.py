"""bench.py's N > 1 logic, rehearsed on CPU: the driver's own launch line (torch.distributed.run, one process per
rank) with --rehearse --backend gloo - the kernel sources compiled for the CPU emulator, host buffers, tiny chunks.  Asserts the
rank -> segment mapping, per-rank IVs (through the oracle check and the round trip), the barrier + max-over-ranks reduction (one JSON
line from rank 0 with whole-job bytes) and, for segments < GPUs, the chunk-range split with the all-gather of transformed sizes."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _launch(world, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--rehearse", "--backend", "gloo", "--workload", "full"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line, got %d" % len(lines)
    return json.loads(lines[0])


def test_two_ranks_segment_major(emu, oracle):
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    CH, cps, nseg = 20000, 3, 2
    j = _launch(2, ["--chunk-bytes", str(CH), "--chunks-per-segment", str(cps), "--segments", str(nseg), "--inflight", "2"])
    c = j["config"]
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and "REHEARSAL" in j["data"]
    assert c["segments_of_rank0"] == [0, 2] and c["chunks_of_rank0"] == nseg * cps and c["segments_total"] == 4     # segment s -> rank s mod 2
    assert c["verified_chunks_vs_oracle"] == nseg * cps              # rank 0's chunks equal libzstd + OpenSSL with IV(global segment, chunk)
    assert j["detransform"]["round_trip_exact"] is True             # max-over-ranks of every rank's round trip
    # value = whole-job bytes (both ranks) / max-over-ranks time
    assert abs(j["value"] - 2 * nseg * cps * CH / 2**30 / (j["ms_per_step"] * 1e-3)) <= 5.1e-5 + 1e-3 * j["value"]      # (value is rounded to 4 places)
    assert j["cpu_baseline"] is None and j["end_to_end"] is None


def test_gpus_2_without_a_launcher_starts_two_ranks(emu, oracle):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run: two ranks, the
    process group, the barrier and the max-over-ranks reduction really run, and the line says n_gpus 2 (it used to run ONE
    rank and print n_gpus 1).  A launcher whose WORLD_SIZE disagrees with --gpus is refused."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse", "--backend", "gloo",
           "--workload", "full", "--chunk-bytes", "20000", "--chunks-per-segment", "3", "--segments", "2", "--inflight", "2"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["segments_of_rank0"] == [0, 2] and j["config"]["segments_total"] == 4
    assert j["detransform"]["round_trip_exact"] is True
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse", "--backend", "gloo"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 4" in (r.stdout + r.stderr)


def test_one_segment_split_over_two_ranks(emu, oracle):
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    from tsxform import synth
    CH, cps = 20000, 5
    j = _launch(2, ["--chunk-bytes", str(CH), "--chunks-per-segment", str(cps), "--segments", "1", "--split-segments", "--gather-object"])
    c = j["config"]
    assert j["scaling"] == "strong" and c["segments_total"] == 1 and c["chunks_of_rank0"] == 2 and "chunk-range split" in c["parallelism"]
    assert abs(j["value"] - cps * CH / 2**30 / (j["ms_per_step"] * 1e-3)) <= 5.1e-5 + 1e-3 * j["value"]          # the job is ONE segment
    of = oracle.COMPRESS | oracle.ENCRYPT
    sizes = [len(oracle.transform_chunk(of, synth.KEY, synth.AAD, synth.iv_for(0, k), synth.gen_chunk("K", 1000, 0, k, CH).tobytes())[0])
             for k in range(cps)]
    pos = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    assert c["chunk_index_positions_sha"] == hashlib.sha256(pos.tobytes()).hexdigest()[:16]     # rank 0 holds the whole chunk index
    whole = b"".join(oracle.transform_chunk(of, synth.KEY, synth.AAD, synth.iv_for(0, k), synth.gen_chunk("K", 1000, 0, k, CH).tobytes())[0] for k in range(cps))
    assert c["object_gathered_on_rank0_sha"] == hashlib.sha256(whole).hexdigest()[:16]          # ... and, with --gather-object, the whole .log object
    assert j["detransform"]["round_trip_exact"] is True


def test_single_rank_line_has_every_leg(emu, oracle):
    """N = 1 as the driver runs it at round end (plain `python bench.py ...`, no launcher): one JSON line whose legs outside the
    timed region are all present - here on the emulator, where they measure nothing but must run."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--rehearse", "--backend", "gloo",
           "--chunk-bytes", "8192", "--chunks-per-segment", "3", "--segments", "2"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["config"]["batches_in_flight"] == 4 and j["config"]["verified_chunks_vs_oracle"] == 6
    s = j["sustained"]
    assert s["callers"] == 5 and s["batches"] == 15 and s["value"] >= 0 and "slope" in s["method"]
    assert j["detransform"]["round_trip_exact"] is True and j["roofline"]["bound"] == "hbm"
    assert j["value_B"]["exact_vs_oracle"] is True and j["value_B"]["batches"] == 8 and j["value_B"]["value"] > 0, j["value_B"]
    assert j["value_B"]["value_B_1_5_6"]["exact_vs_oracle"] is True and j["value_B"]["value_B_1_5_6"]["value"] > 0, j["value_B"]      # the profile a zstd-jni 1.5.6 broker selects


def test_broker_leg_runs_without_torch_and_checks_sizes(emu, tmp_path):
    """tools/broker_leg.py - the broker-shaped leg of bench.py's end_to_end object - as the child process bench.py --broker-subprocess
    starts: no torch in the process (a process has one HIP runtime; torch's bundled one moves D2H copies with kernels), pooled contexts,
    packed output between registered buffers, sizes checked against the run that produced `expect`.  Here on the emulated library with
    tiny chunks; a wrong expectation must show in same_sizes_as_device_run."""
    import tsxform
    from tsxform import synth
    nat = tsxform._native
    CH, B, nseg = 32768, 3, 2
    flags = nat.COMPRESS | nat.ENCRYPT | nat.CRC
    params = nat.Native.make_params(flags, synth.KEY, synth.AAD, zstd_profile=0)
    src = np.concatenate([synth.gen_chunk("K", 1000, 0, i, CH) for i in range(nseg * B)])
    ivs = np.stack([np.frombuffer(synth.iv_for(0, i), np.uint8) for i in range(nseg * B)])
    d = np.zeros(nseg * B, nat.DESC_DTYPE); d["src_off"] = np.arange(nseg * B, dtype=np.uint64) * CH; d["src_len"] = CH; d["iv"] = ivs
    slot = (emu.transformed_bound(CH, flags) + 63) // 64 * 64
    d["dst_off"] = np.arange(nseg * B, dtype=np.uint64) * slot; d["dst_cap"] = slot
    dst = np.zeros(nseg * B * slot, np.uint8)
    emu.transform_batch(params, d, src, dst, dst.size, nat.MEM_HOST, ctx=None)
    assert (d["status"] == 0).all()
    lib = [m for m in open("/proc/self/maps").read().split() if m.endswith("libtsxform_emu.so")][0]
    for name, arr in (("src", src), ("ivs", ivs), ("expect", d["dst_len"]), ("wrong", d["dst_len"] + 1)):
        np.save(str(tmp_path / (name + ".npy")), arr)
    for expect, ok in (("expect", True), ("wrong", False)):
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "broker_leg.py"), "--src", str(tmp_path / "src.npy"), "--ivs", str(tmp_path / "ivs.npy"),
                             "--expect", str(tmp_path / (expect + ".npy")), "--callers", "2,3", "--batch", str(B), "--chunk", str(CH), "--window", "0.5", "--lib", lib],
                            cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert cp.returncode == 0, cp.stderr[-2000:]
        rows = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("[")][-1])
        assert [r["callers"] for r in rows] == [2, 3]
        for r in rows:
            assert r["same_sizes_as_device_run"] is ok and r["torch_in_process"] is False and r["calls"] >= r["callers"] and r["gibs"] > 0


def _force_dist_cmd(extra):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--split-segments", "--gather-object",
            "--segments", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-sustained", "--no-end-to-end"] + extra


def test_force_dist_runs_the_collectives_with_one_rank(emu, oracle):
    """--force-dist makes ONE rank initialise the process group and run the barrier, the max-over-ranks all-reduce, the
    all-gather of transformed sizes and the point-to-point path of --gather-object (a grouped send + recv to itself) - here with gloo on
    the emulator; tests/test_gpu_parity.py::test_rccl_collectives_run_on_one_gpu is the same command with the nccl backend on the device."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    from tsxform import synth
    CH, cps = 20000, 4
    r = subprocess.run(_force_dist_cmd(["--rehearse", "--backend", "gloo", "--chunk-bytes", str(CH), "--chunks-per-segment", str(cps)]),
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    pg = j["config"]["process_group"]
    assert pg["forced_on_one_rank"] is True and pg["ran"] == ["barrier", "all_reduce(MAX)", "all_gather(sizes)"]     # (gloo has no pair to itself: the p2p loop-back is nccl only)
    of = oracle.COMPRESS | oracle.ENCRYPT
    whole = b"".join(oracle.transform_chunk(of, synth.KEY, synth.AAD, synth.iv_for(0, k), synth.gen_chunk("K", 1000, 0, k, CH).tobytes())[0] for k in range(cps))
    assert j["config"]["object_gathered_on_rank0_sha"] == hashlib.sha256(whole).hexdigest()[:16]     # what came back through the loop-back IS the object
    assert j["detransform"]["round_trip_exact"] is True


@pytest.mark.timeout(1500)
def test_eight_ranks_as_the_driver_launches_them(emu, oracle):
    """BASELINE configs[4] in small: `bench.py --gpus 8` under the driver's own launcher, eight gloo ranks on the emulator - segment-major
    (64 segments over 8 ranks in the real run: rank r owns segments r, r + 8, ...) and a segment count below the rank count
    (--split-segments: every segment cut by chunk range over all eight ranks, sizes all-gathered, object gathered on rank 0)."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    CH, cps, nseg = 6000, 2, 2
    j = _launch(8, ["--chunk-bytes", str(CH), "--chunks-per-segment", str(cps), "--segments", str(nseg), "--inflight", "2"])
    c = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and c["segments_of_rank0"] == [0, 8] and c["segments_total"] == 16
    assert c["verified_chunks_vs_oracle"] == nseg * cps and j["detransform"]["round_trip_exact"] is True
    assert abs(j["value"] - 8 * nseg * cps * CH / 2**30 / (j["ms_per_step"] * 1e-3)) <= 5.1e-5 + 1e-3 * j["value"]
    cps = 16
    j = _launch(8, ["--chunk-bytes", str(CH), "--chunks-per-segment", str(cps), "--segments", "1", "--split-segments", "--gather-object"])
    c = j["config"]
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and c["segments_total"] == 1 and c["chunks_of_rank0"] == 2
    assert c["process_group"]["world"] == 8 and c["process_group"]["ran"][:3] == ["barrier", "all_reduce(MAX)", "all_gather(sizes)"]
    assert c["chunk_index_positions_sha"] and c["object_gathered_on_rank0_sha"] and j["detransform"]["round_trip_exact"] is True


def test_device_state_sample_reads_rocm_smi_json(tmp_path, monkeypatch):
    """bench.py's look at the device under load (`sustained.device_state_under_this_load`): clocks, power, cap and temperatures of THIS rank's
    card out of `rocm-smi --json`; a missing or failing tool costs a field, never the line."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\ncat <<'J'\n" + json.dumps({
        "card0": {"sclk clock speed:": "(2397Mhz)", "mclk clock speed:": "(2000Mhz)", "Max Graphics Package Power (W)": "1400.0",
                  "Current Socket Graphics Package Power (W)": "1231.0", "Temperature (Sensor junction) (C)": "51.0", "Unique ID": "0xabc"},
        "card1": {"sclk clock speed:": "(132Mhz)", "Current Socket Graphics Package Power (W)": "140.0"}}) + "\nJ\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    out = {}
    b._device_state_under_load(out, 0.0)
    assert out["sclk clock speed:"] == "(2397Mhz)" and out["Max Graphics Package Power (W)"] == "1400.0" and "Unique ID" not in out
    monkeypatch.setenv("LOCAL_RANK", "1")
    out = {}
    b._device_state_under_load(out, 0.0)
    assert out["sclk clock speed:"] == "(132Mhz)"
    fake.write_text("#!/bin/sh\nexit 3\n")
    out = {}
    b._device_state_under_load(out, 0.0)
    assert "sclk clock speed:" not in out                          # nothing to report; no exception
