#!/bin/bash
# round-2 GPU call 8: the GPU suite + smoke with the block-priority default, then the in-flight harness with and without it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c8; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
L=tiered-storage-for-apache-kafka_amd/libtsxform.so
echo -n "default (alternating priorities): "; timeout 60 python tools/sweep_libs.py $L 2>/dev/null | cut -c30-140
echo -n "no priorities:                    "; TSX_ZSTD_SCHED=4,32,0 timeout 60 python tools/sweep_libs.py $L 2>/dev/null | cut -c30-140
