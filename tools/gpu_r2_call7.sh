#!/bin/bash
# round-2 GPU call 7: the driver's own bench line (--gpus 1 --steps 20 --warmup 5) with the 'sustained' leg, wall time of the whole command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c7; mkdir -p $O
S=$(date +%s)
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s"
tail -3 $O/bench.err | cut -c1-300
python - <<PY
import json
j = json.loads(open("$O/bench_driver_line.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"])
print("sustained", j["sustained"])
print("detransform", j["detransform"]["value"], "e2e", j["end_to_end"]["value"], "cpu", [(l["threads"], l["value"]) for l in j["cpu_baseline"]["by_threads"]])
PY
