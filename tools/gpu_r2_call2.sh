#!/bin/bash
# round-2 GPU call 2: GPU suite on the reworked front end, host-memory rates (staged pipeline, registered memory), bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
timeout 300 python tools/hostmem_bench.py 256 > $O/hostmem.txt 2>&1
timeout 400 python bench.py --steps 6 > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest_gpu.log; cat $O/hostmem.txt; cat $O/bench.json | cut -c1-600
