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
      bash tools/pmc_zstd.sh > $O/pmc_zstd.log 2>&1; bash tools/pmc_dec.sh > $O/pmc_dec.log 2>&1; bash tools/pmc_small.sh > $O/pmc_small.log 2>&1
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
    brokernt)
      # the broker shape in a process WITHOUT torch (the system's HIP runtime, as a JVM loads it) and, for comparison, with it
      export GPU_MAX_HW_QUEUES=16
      timeout 200 python tools/broker_probe.py --mem host --ctxless --pool-chunks 512 --src-file /dev/shm/tsx_k512.npy --gen-only > /dev/null 2> $O/brokernt.err
      timeout 300 python tools/broker_probe.py --mem host --ctxless --configs ${arg:-10x256,20x256,32x256} --pool-chunks 512 --seconds 6 --src-file /dev/shm/tsx_k512.npy --tag notorch 2>> $O/brokernt.err | tee $O/brokernt.jsonl
      timeout 300 python tools/broker_probe.py --mem host --ctxless --configs ${arg:-10x256,20x256,32x256} --pool-chunks 512 --seconds 6 --tag torch 2>> $O/brokernt.err | tee -a $O/brokernt.jsonl
      rm -f /dev/shm/tsx_k512.npy; unset GPU_MAX_HW_QUEUES ;;
    copytorch)
      for mode in registered torchpinned; do
        D=$O/ct_$mode
        ( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $D -o c --output-format csv -- python $R/tools/ubench/copy_engine_torch.py $mode 2> /dev/null | grep "GB/s" )
        k=$(grep -h copyBuffer $D/*kernel_stats.csv 2>/dev/null | awk -F, '{s+=$2} END {print s+0}' | tr -d '"'); m=$(grep -hc MEMORY_COPY $D/*memory_copy_trace.csv 2>/dev/null | awk '{s+=$1} END {print s+0}')
        echo "   -> blit kernels: $k   SDMA copies: $m"; rm -rf $D
      done 2>&1 | tee $O/copy_engine_torch.txt ;;
    copyenv)
      # ... with hipHostRegisterPortable (what tsx_host_register asks for) and with GPU_MAX_HW_QUEUES=16 (what bench.py / a broker's launcher set)
      for cfg in "registered 4" "registered 16" "portable 4" "portable 16"; do set -- $cfg
        D=$O/cv_$1_$2
        ( cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=$2 && timeout 60 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $D -o c --output-format csv -- $R/tools/ubench/copy_engine d2h $1 1300000 256 2> /dev/null | grep "GB/s" | sed "s/^/GPU_MAX_HW_QUEUES=$2 /" )
        k=$(grep -h copyBuffer $D/*kernel_stats.csv 2>/dev/null | awk -F, '{s+=$2} END {print s+0}' | tr -d '"'); m=$(grep -hc MEMORY_COPY $D/*memory_copy_trace.csv 2>/dev/null | awk '{s+=$1} END {print s+0}')
        echo "   -> blit kernels: $k   SDMA copies: $m"; rm -rf $D
      done 2>&1 | tee $O/copy_engine_env.txt ;;
    copybg)
      # ... and with the other direction busy on another stream (bg 1) / a kernel holding every CU slot (bg 2)
      for cfg in ${arg:-"d2h,1300000,0,1 d2h,1300001,3,1 h2d,1300000,0,1 d2h,1300000,0,2 d2h,1300001,3,2 h2d,1300000,0,2"}; do set -- ${cfg//,/ }
        D=$O/cb_$1_$2_$4
        ( cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $D -o c --output-format csv -- $R/tools/ubench/copy_engine $1 registered $2 256 0 $3 $4 2> /dev/null | grep "GB/s" )
        k=$(grep -h copyBuffer $D/*kernel_stats.csv 2>/dev/null | awk -F, '{s+=$2} END {print s+0}' | tr -d '"'); m=$(grep -hc MEMORY_COPY $D/*memory_copy_trace.csv 2>/dev/null | awk '{s+=$1} END {print s+0}')
        echo "   -> blit kernels: $k   SDMA copies: $m"; rm -rf $D
      done 2>&1 | tee $O/copy_engine_bg.txt ;;
    copyalign)
      # ... and when sizes / host addresses are not multiples of 4 (packed output)
      for cfg in "1300000 0" "1300001 0" "1300002 0" "1300000 1" "1300000 2" "1300001 3" "1300032 0" "1300000 32"; do set -- $cfg; for dir in d2h h2d; do
        D=$O/ca_${dir}_$1_$2
        ( cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $D -o c --output-format csv -- $R/tools/ubench/copy_engine $dir registered $1 256 0 $2 2> /dev/null | grep "GB/s" )
        k=$(grep -h copyBuffer $D/*kernel_stats.csv 2>/dev/null | awk -F, '{s+=$2} END {print s+0}' | tr -d '"'); m=$(grep -hc MEMORY_COPY $D/*memory_copy_trace.csv 2>/dev/null | awk '{s+=$1} END {print s+0}')
        echo "   -> blit kernels: $k   SDMA copies: $m"; rm -rf $D
      done; done 2>&1 | tee $O/copy_engine_align.txt ;;
    copyhog)
      # ... and when other streams of the process have copied before (do they keep the SDMA engines?)
      for hog in 0 1 2 4 8; do for dir in d2h h2d; do
        D=$O/ch_${dir}_$hog
        ( cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $D -o c --output-format csv -- $R/tools/ubench/copy_engine $dir registered 1300000 256 $hog 2> /dev/null | grep "GB/s" )
        k=$(grep -h copyBuffer $D/*kernel_stats.csv 2>/dev/null | awk -F, '{s+=$2} END {print s+0}' | tr -d '"'); m=$(grep -hc MEMORY_COPY $D/*memory_copy_trace.csv 2>/dev/null | awk '{s+=$1} END {print s+0}')
        echo "   -> blit kernels: $k   SDMA copies: $m"; rm -rf $D
      done; done 2>&1 | tee $O/copy_engine_hog.txt ;;
    copyengine)
      # SDMA or blit kernel?  tools/ubench/copy_engine per direction / host memory kind / size under rocprofv3 (kernel + memory-copy trace)
      for dir in d2h h2d; do for kind in malloc registered; do for sz in 65536 1300000 16777216 268435456; do
        cnt=256; [ $sz -ge 16777216 ] && cnt=16; [ $sz -ge 268435456 ] && cnt=4
        D=$O/ce_${dir}_${kind}_$sz
        ( cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $D -o c --output-format csv -- $R/tools/ubench/copy_engine $dir $kind $sz $cnt 2> /dev/null | grep "GB/s" )
        k=$(grep -h copyBuffer $D/*kernel_stats.csv 2>/dev/null | awk -F, '{s+=$2} END {print s+0}' | tr -d '"'); m=$(grep -hc MEMORY_COPY $D/*memory_copy_trace.csv 2>/dev/null | awk '{s+=$1} END {print s+0}')
        echo "   -> blit kernels: $k   SDMA copies: $m"; rm -rf $D
      done; done; done 2>&1 | tee $O/copy_engine_probe.txt ;;
    brokertrace)
      # kernel + memory-copy trace of the broker shape (default 32 callers x 256 chunks, ctx-less, registered host buffers): who runs when
      ( cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16 && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/btrace -o b --output-format csv -- python $R/tools/broker_probe.py --mem host --ctxless --configs ${arg:-32x256} --pool-chunks 512 --seconds 6 > $O/btrace.jsonl 2> $O/btrace.err )
      cat $O/btrace.jsonl; for f in $(find $O/btrace -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do gzip -c $f > $O/$(basename $f).gz; done
      rm -rf $O/btrace; ls -la $O ;;
    e2e)
      # the host-path legs of the bench line only (one batch at a time, batches in flight, broker rows)
      timeout 600 python bench.py --steps ${arg:-10} --no-cpu-baseline --no-inverse --no-sustained --no-verify > $O/bench_e2e.json 2> $O/bench_e2e.err
      python - <<PY
import json
j = json.loads(open("$O/bench_e2e.json").read().strip().splitlines()[-1]); e = j["end_to_end"]
print("value", j["value"]); print("one at a time", [(r.get("mem"), r.get("ms"), r.get("gibs")) for r in e["one_batch_at_a_time"]]); print("in flight", [(c.get("callers"), c.get("gibs")) for c in e["batches_in_flight"] or []])
for b in e["broker"] or []: print("broker", b["callers"], b["gibs"], b["frac_of_device_resident_value"], b["ms_per_call_median"], b["calls"])
PY
      ;;
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
    probe)
      # what tsx_init found: CU keys of the probe launch, reserved CUs, waves per launch of the compressor service
      timeout 300 python -c "
import torch, tsxform, json
N = tsxform.get(); print(N.version()); print(json.dumps(N.service_stats(0)))" 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt ;;
    guestbench)
      # the timed region and the sustained leg with the opt-in guest waves (TSX_FETCH_QUIET_MS: no fetch ever runs in this process before them)
      TSX_FETCH_QUIET_MS=10000 timeout 300 python bench.py --steps ${arg:-30} --no-cpu-baseline --no-end-to-end --no-inverse --no-configs --no-mixed-load --no-value-b --no-verify > $O/bench_guests.json 2> $O/bench_guests.err
      python - <<PY
import json
try:
    j = json.loads(open("$O/bench_guests.json").read().strip().splitlines()[-1])
    print("guests: value", j["value"], "ms/step", j["ms_per_step"], "sustained", (j.get("sustained") or {}).get("value"), (j.get("sustained") or {}).get("whole_run_gibs_incl_ramp_and_drain"), "launch meta", json.dumps(j["roofline"].get("service"))[:500])
except Exception as e: print("guestbench failed", e)
PY
      tail -2 $O/bench_guests.err ;;
    benchq)
      # the timed region + sustained + mixed-load legs only (no CPU baseline, no host-path legs, no broker children)
      timeout 600 python bench.py ${arg:+--steps $arg} --no-cpu-baseline --no-end-to-end --no-inverse --no-configs > $O/bench_quick.json 2> $O/bench_quick.err
      python - <<PY
import json
try:
    j = json.loads(open("$O/bench_quick.json").read().strip().splitlines()[-1])
    print("value", j["value"], "ms/step", j["ms_per_step"], "one at a time", j["config"]["gibs_one_batch_at_a_time"]); print("roofline", json.dumps(j["roofline"])[:900])
    print("sustained", json.dumps(j["sustained"])[:400]); print("mixed_load", json.dumps(j["mixed_load"]))
except Exception as e: print("benchq failed", e); print(open("$O/bench_quick.err").read()[-1500:])
PY
      ;;
    mixed)
      # fetch latency under upload load, tools/mixed_load_probe.py: arg = "shape,callers[,reserved]" ... (default: both shapes, library default)
      for cfg in ${arg:-"batches,5 broker,32"}; do set -- ${cfg//,/ }
        nf=""; [ "$4" = "nofetch" ] && nf="--no-fetch"
        timeout 300 python tools/mixed_load_probe.py --shape $1 --callers $2 ${3:+--reserved-cus $3} $nf --seconds ${MIXED_SECONDS:-12} --tag "$cfg" 2>> $O/mixed.err | tee -a $O/mixed.jsonl
      done ;;
    mixednt)
      # fetch latency under upload load in a process WITHOUT torch (the system's HIP runtime): arg = "shape,callers[,reserved[,nofetch]]" ...
      export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/mixednt.err
      for cfg in ${arg:-"broker,32 batches,5"}; do set -- ${cfg//,/ }
        nf=""; [ "$4" = "nofetch" ] && nf="--no-fetch"
        timeout 300 python tools/mixed_load_notorch.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --shape $1 --callers $2 ${3:+--reserved-cus $3} $nf --seconds ${MIXED_SECONDS:-12} ${MIXED_MAX_LAUNCH_MS:+--max-launch-ms $MIXED_MAX_LAUNCH_MS} --tag "$cfg" 2>> $O/mixednt.err | tee -a $O/mixednt.jsonl
      done ;;
    pmcenc)
      # the passes the forward-side records of profiles/pmc_traffic.json need (service kernel, CRC32C, GCM); the decoder's are tools/pmc_dec.sh
      bash tools/pmc_zstd.sh > $O/pmc_zstd.log 2>&1; bash tools/pmc_small.sh > $O/pmc_small.log 2>&1
      python tools/show_pmc.py gpurun_out/pmc | tee $O/pmc_zstd_summary.txt | tail -5 ;;
    pmctraffic)
      # profiles/pmc_traffic.json from the PMC passes just made (the bench that follows quotes roofline.traffic for the sources as they are);
      # the same command at home over the merged gpurun_out/pmc* gives the same file
      python tools/pmc_traffic.py --tag ${arg:-r05} > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -4 $O/pmc_traffic.log | cut -c1-200 ;;
    guests)
      # the reservation follows the traffic: uploads alone (guest waves on the reserved CUs), then the same with fetches arriving 2 s in; and the
      # reservation always in force (fetch_quiet_ms=0) for comparison.  Device-resident 2048-chunk batches, 5 callers, no torch
      export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/guests.err
      for v in ${arg:-500,nofetch 500,fetch 0,nofetch}; do set -- ${v//,/ }
        nf=""; [ "$2" = "nofetch" ] && nf="--no-fetch"
        timeout 120 python tools/mixed_load_notorch.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --shape batches --callers 5 --seconds ${MIXED_SECONDS:-12} --config fetch_quiet_ms=$1 $nf --tag "fetch_quiet_ms=$1 $2" 2>> $O/guests.err | tee -a $O/guests.jsonl
      done ;;
    guestsrep)
      # the no-fetch guest-wave run repeated arg times (default 6) with completion stamps: does the window lose seconds, and where?
      export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/guests.err
      for i in $(seq 1 ${arg:-6}); do
        timeout 120 python tools/mixed_load_notorch.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --shape batches --callers 5 --seconds ${MIXED_SECONDS:-12} --config fetch_quiet_ms=500 --no-fetch --sample-service --tag "rep $i fetch_quiet_ms=500 nofetch" 2>> $O/guests.err | tee -a $O/guestsrep.jsonl | cut -c1-700
      done ;;
    decab)
      # fetch-side latency, round 4's tree (tools/_ab/r4tree, built at home from a72b84f) against HEAD, alternating processes on this box
      for i in 1 2; do
        for t in r4 head; do
          if [ $t = r4 ]; then ( cd $R/tools/_ab/r4tree && timeout 300 python tools/dec_latency.py ${arg:-256} 2>> $O/decab.err | sed "s/^{/{\"tree\": \"r4\", \"round\": $i, /" ) >> $O/decab.jsonl
          else timeout 300 python tools/dec_latency.py ${arg:-256} 2>> $O/decab.err | sed "s/^{/{\"tree\": \"head\", \"round\": $i, /" >> $O/decab.jsonl; fi
        done
      done
      python - <<PY
import json
rows = [json.loads(l) for l in open("$O/decab.jsonl") if l.startswith("{")]
for form in ("blocks", "chunks"):
    for mem in ("host", "device"):
        for n in (1, 4, 64, 256):
            r = {t: sorted(x["ms_median"] for x in rows if x["tree"] == t and x["form"] == form and x["mem"] == mem and x["chunks"] == n) for t in ("r4", "head")}
            print(form, mem, n, "r4", r["r4"], "head", r["head"])
PY
      ;;
    prof)
      # lap timers of the compressor wave (tools/_libs/libtsxform_prof.so, built at home with `make -C .../csrc prof`): arg = "dist,profile[,chain]" ...
      for v in ${arg//+/ }; do set -- ${v//,/ }
        ch=""; [ "$3" = "chain" ] && ch="--chain"
        u=256; [ "$1" = "B" ] && u=8
        timeout 400 python tools/prof_zstd.py --chunks ${PROF_CHUNKS:-2048} --dist $1 --profile $2 --uniq $u $ch --out $O/prof_$1_$2${3:+_$3}.json > /dev/null 2>> $O/prof.err
        python tools/show_prof.py $O/prof_$1_$2${3:+_$3}.json 2>/dev/null || tail -c 1500 $O/prof_$1_$2${3:+_$3}.json
      done ;;
    enchost)
      # encrypt-only host -> host: zero-copy output vs copy engines, piece sizes, system runtime vs torch's (tools/enc_host_probe.py)
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/enchost.err
      timeout 300 python tools/enc_host_probe.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --tag head 2>> $O/enchost.err | tee -a $O/enchost.jsonl | cut -c1-420
      timeout 300 python tools/enc_host_probe.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --with-torch --tag head 2>> $O/enchost.err | tee -a $O/enchost.jsonl | cut -c1-420
      if [ -d tools/_ab/r4tree ]; then ( cd tools/_ab/r4tree && cp $R/tools/enc_host_probe.py tools/enc_host_probe.py && timeout 300 python tools/enc_host_probe.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --tag r4tree 2>> $O/enchost.err | tee -a $O/enchost.jsonl | cut -c1-420 ); fi ;;
    stucktrace)
      # fetches next to saturating uploads under rocprofv3 --kernel-trace (torch-free process): which kernel of a fetch was not placed, and next to what?
      export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/stuck.err
      for i in $(seq 1 ${arg:-2}); do
        ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace -d $O/stuck_$i -o m --output-format csv -- python $R/tools/mixed_load_notorch.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --shape batches --callers 5 --seconds ${MIXED_SECONDS:-40} --config fetch_quiet_ms=0 --tag "traced $i" 2>> $O/stuck.err | tee -a $O/stucktrace.jsonl | cut -c1-900 )
        f=$(find $O/stuck_$i -name "*kernel_trace.csv" | head -1)
        [ -n "$f" ] && python tools/trace_gaps.py $f --ms 20 | tee $O/stuck_gaps_$i.txt && gzip -c $f > $O/stuck_kernel_trace_$i.csv.gz
        rm -rf $O/stuck_$i
      done ;;
    lone)
      # one batch at a time with / without guest waves (tools/lone_batch_probe.py), and with no reservation at all
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/lone.err
      timeout 300 python tools/lone_batch_probe.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --configs ${arg:-fetch_quiet_ms=0 fetch_quiet_ms=2000} 2>> $O/lone.err | tee -a $O/lone.jsonl | cut -c1-1500
      TSX_FETCH_RESERVED_CUS=0 timeout 200 python tools/lone_batch_probe.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --configs fetch_quiet_ms=0 --batches 3 2>> $O/lone.err | sed 's/^{/{"env": "TSX_FETCH_RESERVED_CUS=0", /' | tee -a $O/lone.jsonl | cut -c1-1500 ;;
    where)
      # where do the chunks of a lone batch run, and which are slow?  prof flavour, guests on (default) and off
      for cfg in ${arg:-fetch_quiet_ms=2000 fetch_quiet_ms=0}; do
        for rep in 1 2 3; do
          timeout 300 python tools/prof_zstd.py --chunks 2048 --dist K --chain --where --config $cfg --data /dev/shm/prof_k256.npy --out $O/where_${cfg}_$rep.json > /dev/null 2>> $O/where.err
          python - <<PY
import json
j = json.load(open("$O/where_${cfg}_$rep.json")); print("$cfg rep $rep wall", round(j["wall_ms_1"]), json.dumps(j["where"])[:1400])
PY
        done
      done ;;
    lonev)
      # lone batches under variants "ENVVALUE:config" (ENVVALUE = TSX_FETCH_RESERVED_CUS or - for the default), e.g. lonev:0:fetch_quiet_ms=0+-:fetch_quiet_ms=2000
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/lonev.err
      for v in ${arg//+/ }; do
        e=${v%%:*}; cfg=${v#*:}
        pre=""; case "$cfg" in *@*) pre="--pre-config ${cfg#*@}"; cfg=${cfg%%@*};; esac
        ( [ "$e" != "-" ] && export TSX_FETCH_RESERVED_CUS=$e; timeout 200 python tools/lone_batch_probe.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --no-sampler --batches ${LONE_BATCHES:-8} $pre --configs $cfg 2>> $O/lonev.err | sed "s/^{/{\"reserved_cus_env\": \"$e\", /" >> $O/lonev.jsonl )
      done
      python - <<PY
import json
for l in open("$O/lonev.jsonl"):
    j = json.loads(l); print("reserved env", j["reserved_cus_env"], j["config"], "pre", j.get("pre_config"), "waves", j["waves"], [b["ms"] for b in j["batches"]], "guest launches", sum(b["guest_launches"] for b in j["batches"]), "reserved exits", j["batches"][0]["reserved_exits"])
PY
      ;;
    keepwaves)
      # compressor waves that stay on the reserved CU of every shader engine (svc_keep_waves): fetch latency and upload rate, device-resident 2048-chunk batches, no torch
      export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/keepwaves.err
      for k in ${arg//,/ }; do
        timeout 120 python tools/mixed_load_notorch.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --shape batches --callers 5 --seconds ${MIXED_SECONDS:-14} --config svc_keep_waves=$k --tag "keep_waves=$k" 2>> $O/keepwaves.err | tee -a $O/keepwaves.jsonl
      done ;;
    benchdrv)
      # the driver's command, verbatim
      timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
      python - <<PY
import json
try:
    j = json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
    print("value", j["value"], "ms/step", j["ms_per_step"], "one at a time", j["config"]["gibs_one_batch_at_a_time"], "sustained", j["sustained"]["value"], "value_B", j["value_B"]["value"], j["value_B"]["value_B_1_5_6"]["value"])
    print("roofline", json.dumps({k: v for k, v in j["roofline"].items() if k != "line_rate_probe"})[:1200])
    print("mixed_load", json.dumps(j["mixed_load"])[:900])
except Exception as e: print("benchdrv failed", e); print(open("$O/bench_driver_cmd.err").read()[-1500:])
PY
      ;;
    satcfg)
      # the saturated regime (5 callers x 2048-chunk device-resident batches, no fetches) under configurations set BEFORE tsx_init, e.g. satcfg:svc_waves_per_cu=23+svc_waves_per_cu=24
      export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
      [ -f /dev/shm/tsx_mix_src.npy ] || timeout 200 python tools/broker_leg.py --gen /dev/shm/tsx_mix_src.npy /dev/shm/tsx_mix_ivs.npy 1 256 4194304 K > /dev/null 2>> $O/satcfg.err
      for cfg in ${arg//+/ }; do
        timeout 120 python tools/mixed_load_notorch.py --src /dev/shm/tsx_mix_src.npy --ivs /dev/shm/tsx_mix_ivs.npy --shape batches --callers ${SAT_CALLERS:-5} --seconds ${MIXED_SECONDS:-14} --no-fetch ${SAT_SAMPLE:+--sample-service} --config $cfg --tag "$cfg" 2>> $O/satcfg.err | tee -a $O/satcfg.jsonl | cut -c1-700
      done ;;
    *) echo "unknown section $name" ;;
  esac
done
echo "=== done $(date +%T)"
