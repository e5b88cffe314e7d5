#!/bin/bash
# First GPU call of the next round (about 4 minutes): does per-block wave priority (zs_block_priority, zstd_enc.hip) pay in the regime
# that has no tails?  tools/steady_state_probe.py = 5 callers, work always queued, rate = slope of completions; alternating processes.
#   gpurun --timeout 400 -- 'bash tools/next_round/01_priority_sustained.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/prio_sustained; mkdir -p $O
for round in 1 2; do
  for m in "4,32,0" "4,32,2" "4,32,3"; do
    echo -n "round $round TSX_ZSTD_SCHED=$m: "
    TSX_ZSTD_SCHED=$m timeout 120 python tools/steady_state_probe.py 5 10 250 2>&1 | tail -1
  done
done | tee $O/result.txt
# and the timed region's arrangement (three callers from a barrier), for the record
for m in "4,32,0" "4,32,2"; do echo -n "3 in flight, TSX_ZSTD_SCHED=$m: "; TSX_ZSTD_SCHED=$m timeout 60 python tools/sweep_libs.py tiered-storage-for-apache-kafka_amd/libtsxform.so 2>/dev/null | cut -c30-130; done | tee -a $O/result.txt
