#!/bin/bash
# two quick PMC passes (instruction mix + waits) for zstd_compress_kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_q
rm -rf $OUT; mkdir -p $OUT
python $R/tools/prof_zstd.py --chunks 256 --lib libtsxform.so --data /tmp/k256.npy > /dev/null 2>&1
CMD="python $R/tools/prof_zstd.py --chunks 2048 --dist K --lib libtsxform.so --data /tmp/k256.npy"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-include-regex zstd_compress -d $OUT/p$i -o p$i --output-format csv -- $CMD > $OUT/p$i.log 2>&1
done
