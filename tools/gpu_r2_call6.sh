#!/bin/bash
# round-2 GPU call 6: rocprofv3 kernel statistics of the default bench command (18 steps, 3 batches in flight; kernel trace only)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=gpurun_out/r2c6; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -type f ! -name "*stats.csv" -size +1M -delete
head -3 $O/kernel_stats.csv | cut -c1-260; tail -3 $O/rocprof.err | cut -c1-200
python - <<PY
import json
j = json.loads(open("$O/bench_under_rocprof.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "ms_per_launch", j["roofline"]["ms_per_launch"])
PY
