#!/bin/bash
# One parameterised script for the GPU box (run through gpurun from the repo root):  bash tools/gpu_call.sh <tag> <section>...
# Sections write under gpurun_out/<tag>/ ; copy what is to be judged into profiles/ afterwards.
#   tests[:expr]   pytest -m gpu (optionally -k expr)
#   smoke          __graft_entry__.smoke()
#   bench          the driver's default bench.py line (all legs)
#   regime         timed-region regime sweep: callers x steps, no side legs
#   broker         tools/broker_probe.py (10/20/24 callers x 256-chunk batches), device + host memory, GPU_MAX_HW_QUEUES variants
#   pmc            PMC passes over the compressor + decoder (tools/pmc_zstd.sh, tools/pmc_dec.sh)
#   trace          rocprofv3 --kernel-trace --stats of the bench's timed region
#   prio           per-block wave priority modes in the sustained regime (tools/steady_state_probe.py)
#   detr[:n]       big-batch inverse chain (tools/detransform_bench.py);  pmcdec: PMC passes over the decoder only;  zblaps[:n] / dectrace[:n]: block-form laps / kernel stats
#   dec[:nmax]     fetch-side latency of 1 .. nmax chunks, block-parallel vs chunk-serial decoder form (tools/dec_latency.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
for sec in "$@"; do
  name=${sec%%:*}; arg=""; [[ "$sec" == *:* ]] && arg=${sec#*:}
  echo "=== $sec $(date +%T)"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" > $O/pytest_gpu.log 2>&1; else timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; fi
      tail -4 $O/pytest_gpu.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench)
      timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err ;;
    regime)
      for cfg in ${arg:-"3x18 4x48 5x60 6x72"}; do
        t=${cfg%x*}; k=${cfg#*x}
        timeout 600 python bench.py --inflight $t --steps $k --no-cpu-baseline --no-end-to-end --no-inverse --no-sustained --no-verify > $O/regime_$cfg.json 2> $O/regime_$cfg.err
        python - <<PY
import json
try:
    j = json.loads(open("$O/regime_$cfg.json").read().strip().splitlines()[-1]); print("regime $cfg", j["value"], j["ms_per_step"], j["config"]["gibs_one_batch_at_a_time"])
except Exception as e: print("regime $cfg failed", e)
PY
      done ;;
    broker)
      for q in "" 16; do
        for mem in host; do
          extra="--ctxless"
          ( [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q; timeout 400 python tools/broker_probe.py --mem $mem $extra --configs ${arg:-10x256,20x256,40x128} --pool-chunks 512 --seconds 5 --tag "q=$q" >> $O/broker.jsonl 2>> $O/broker.err )
        done
      done
      cat $O/broker.jsonl ;;
    pmc)
      bash tools/pmc_zstd.sh > $O/pmc_zstd.log 2>&1; bash tools/pmc_dec.sh > $O/pmc_dec.log 2>&1
      python tools/show_pmc.py gpurun_out/pmc | tee $O/pmc_zstd_summary.txt; python tools/show_pmc.py gpurun_out/pmc_dec | sed 's/^/dec /' | tee $O/pmc_dec_summary.txt ;;
    trace)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-end-to-end --no-inverse --no-sustained > $O/bench_under_rocprofv3.json 2> $O/trace.err )
      f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_rocprofv3_kernel_stats.csv && head -8 $f
      find $O/trace -name "*kernel_trace.csv" -delete ;;
    dectrace)
      # per-kernel time of a single-chunk (or arg-chunk) inverse call, both decoder forms (kernel names tell them apart)
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/dectrace -o dec --output-format csv -- python $R/tools/dec_latency.py ${arg:-1} > $O/dec_under_rocprofv3.jsonl 2> $O/dectrace.err )
      f=$(find $O/dectrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/dec_rocprofv3_kernel_stats.csv && head -20 $f
      find $O/dectrace -name "*kernel_trace.csv" -delete ;;
    detr)
      timeout 300 python tools/detransform_bench.py ${arg:-2048} 2>&1 | grep -v amdgpu.ids | tee $O/detransform_bench.txt ;;
    pmcdec)
      bash tools/pmc_dec.sh > $O/pmc_dec.log 2>&1; python tools/show_pmc.py gpurun_out/pmc_dec | sed 's/^/dec /' | tee $O/pmc_dec_summary.txt ;;
    zblaps)
      timeout 300 python tools/zb_phase_laps.py ${arg:-4} 2>&1 | grep -v amdgpu.ids | tee $O/zb_phase_laps.txt ;;
    dec)
      timeout 600 python tools/dec_latency.py ${arg:-256} > $O/dec_latency.jsonl 2> $O/dec_latency.err; cat $O/dec_latency.jsonl; tail -3 $O/dec_latency.err ;;
    prio)
      # per-block wave priorities (zs_block_priority) in the continuously fed regime, alternating processes
      for m in "4,32,0" "4,32,2" "4,32,3" "4,32,0" "4,32,2"; do
        echo -n "TSX_ZSTD_SCHED=$m: "; TSX_ZSTD_SCHED=$m timeout 150 python tools/steady_state_probe.py 5 10 250 2>&1 | tail -1
      done | tee $O/prio_sustained.txt ;;
    sched)
      # speculation predictor / schedule A/B: alternating processes, timed region only (5 callers x 40 steps)
      for round in 1 2 3; do
        for m in ${arg:-0,0,0 0,0,1 4,32,1}; do
          echo -n "round $round TSX_ZSTD_SCHED=$m: "
          TSX_ZSTD_SCHED=$m timeout 400 python bench.py --steps 40 --no-cpu-baseline --no-end-to-end --no-inverse --no-sustained --no-verify 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['gibs_one_batch_at_a_time'])"
        done
      done | tee $O/sched_ab.txt ;;
    *) echo "unknown section $name" ;;
  esac
done
echo "=== done $(date +%T)"
