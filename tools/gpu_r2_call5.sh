#!/bin/bash
# round-2 GPU call 5 (HEAD after the front-end fixes): the whole GPU suite, smoke, the default bench line, and the rocprofv3 kernel
# statistics of the same bench command (kernel trace only - no counters in this run)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; 2>/dev/null
find $O/prof -type f ! -name "*stats.csv" -size +2M -delete 2>/dev/null
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; head -5 $O/kernel_stats.csv; python - <<PY
import json
for f in ("bench_default.json", "bench_under_rocprof.json"):
    try:
        j = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, "value", j["value"], "ms/step", j["ms_per_step"], "roofline", j["roofline"])
    except Exception as e:
        print(f, "unreadable", e)
PY
