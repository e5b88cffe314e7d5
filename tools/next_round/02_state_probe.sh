#!/bin/bash
# The same build measures 17.3-17.7 or 18.8-19.7 GiB/s with three batches in flight, process by process (a lone batch: 727 ms always).
# Is it the box (clocks / power cap under the heavier load) or the callers' phase?  Samples rocm-smi twice a second while the
# in-flight harness runs, four processes in a row, and prints each process's rate next to its mean / min clocks and power.
#   gpurun --timeout 300 -- 'bash tools/next_round/02_state_probe.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/state_probe; mkdir -p $O
for i in 1 2 3 4; do
  ( while :; do /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2; sleep 0.5; done ) > $O/smi_$i.csv &
  S=$!
  timeout 90 python tools/sweep_libs.py tiered-storage-for-apache-kafka_amd/libtsxform.so 2>/dev/null | cut -c30-110 > $O/run_$i.txt
  kill $S; wait $S 2>/dev/null
  echo "process $i: $(cat $O/run_$i.txt)"
  python - <<PY
import csv, statistics as st
rows = [r for r in csv.reader(open("$O/smi_$i.csv")) if r and r[0].startswith("card")]
print("   %d samples; columns of the last one: %s" % (len(rows), rows[-1] if rows else None))
PY
done
/opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | head -1 > $O/header.csv; cat $O/header.csv
