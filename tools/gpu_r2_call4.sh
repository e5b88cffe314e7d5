#!/bin/bash
# round-2 GPU call 4: the whole GPU suite (new front-end twins, every-chunk parity), smoke, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err
tail -5 $O/pytest_gpu.log; tail -2 $O/smoke.log; python - <<PY
import json
j = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "inflight", j["config"]["batches_in_flight"], "verified", j["config"]["verified_chunks_vs_oracle"])
print("cpu", [(l["threads"], l["value"]) for l in j["cpu_baseline"]["by_threads"]])
print("e2e one", [(r["dst_layout"], r["host_memory"], r["gibs"]) for r in j["end_to_end"]["one_batch_at_a_time"]])
print("e2e inflight", j["end_to_end"]["batches_in_flight"])
print("inverse", j["detransform"]["value"], j["detransform"]["roofline"])
PY
