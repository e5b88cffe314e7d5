#!/bin/bash
# One GPU-box call: parity tests, the inverse chain (timing, batches in flight, phase laps from libtsxform_prof2.so), the bench line, rocprofv3 kernel stats of the bench.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dec
rm -rf $O; mkdir -p $O
cd $R
timeout 170 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 120 python tools/detransform_bench.py 2048 libtsxform.so > $O/detransform_new.txt 2>&1; tail -4 $O/detransform_new.txt
[ -f tools/_libs/libtsxform_prof2.so ] && timeout 100 python tools/detransform_bench.py 2048 libtsxform_prof2.so > $O/detransform_prof2.txt 2>&1; tail -9 $O/detransform_prof2.txt
timeout 240 python bench.py --steps 9 --warmup 1 2> $O/bench_full.err | tail -1 > $O/bench_full.json; cat $O/bench_full.json | cut -c1-400
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $O/stats.log 2>&1
find $O/stats -name "*kernel_trace.csv" -delete; find $O/stats -name "*agent_info.csv" -delete; head -c 100000 $O/stats.log > $O/stats.log.head; rm -f $O/stats.log
find $O -name "*kernel_stats.csv" | head -2
du -sh $O
