#!/bin/bash
# Short GPU-box call while iterating on the decoder: GPU parity tests, inverse-chain timing, phase laps (prof2 flavour).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/decq
rm -rf $O; mkdir -p $O
cd $R
timeout 170 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -2 $O/pytest_gpu.log
timeout 100 python tools/detransform_bench.py 2048 libtsxform.so 2>&1 | grep -v amdgpu.ids > $O/detransform_new.txt; head -3 $O/detransform_new.txt | tail -1; grep -c "round trip exact" $O/detransform_new.txt
[ -f tiered-storage-for-apache-kafka_amd/libtsxform_prof2.so ] && timeout 100 python tools/detransform_bench.py 2048 libtsxform_prof2.so 2>&1 | grep -v amdgpu.ids > $O/detransform_prof2.txt; tail -8 $O/detransform_prof2.txt
