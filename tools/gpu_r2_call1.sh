#!/bin/bash
# round-2 GPU call 1: libzstd probe, A/B of parser builds, GPU parity suite on the new parser, instruction/line counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c1; rm -rf $O; mkdir -p $O
( find / -xdev \( -name 'libzstd*' -o -name '*zstd*.jar' -o -name 'zstd-jni*' \) 2>/dev/null; python -c "import zstandard; print('zstandard', zstandard.ZSTD_VERSION)" 2>&1 | tail -1; which java javac 2>&1; nproc; grep -m1 "model name" /proc/cpuinfo ) > $O/zstd_probe.txt 2>&1
python tools/sweep_libs.py --steps 6 tools/_libs/libtsxform_base.so tools/_libs/libtsxform_v2.so tools/_libs/libtsxform_v2_k1.so tools/_libs/libtsxform_v2_k3.so tools/_libs/libtsxform_v2_k2_16.so tools/_libs/libtsxform_v2_k2_4.so tools/_libs/libtsxform_v2_w4.so tools/_libs/libtsxform_v2_w6.so > $O/sweep.txt 2> $O/sweep.err
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/prof_zstd.py --chunks 256 --lib libtsxform.so --data /tmp/k256.npy > /dev/null 2>&1
CMD="python $R/tools/prof_zstd.py --chunks 2048 --dist K --chain --lib libtsxform.so --data /tmp/k256.npy"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-include-regex zstd_compress -d $R/$O/pmc/p$i -o p$i --output-format csv -- $CMD > $R/$O/pmc_p$i.log 2>&1
done
cd $R; python tools/show_pmc.py $O/pmc > $O/pmc_summary.txt 2>&1
find $O -name "*agent_info.csv" -delete
tail -3 $O/pytest_gpu.log; cat $O/sweep.txt; cat $O/pmc_summary.txt
