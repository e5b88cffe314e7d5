#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ic
mkdir -p $OUT
python $R/tools/prof_zstd.py --chunks 256 --lib libtsxform.so --data /tmp/k256.npy > /dev/null 2>&1
CMD="python $R/tools/prof_zstd.py --chunks 2048 --dist K --lib libtsxform.so --data /tmp/k256.npy"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ" "SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQC_ICACHE_BUSY_CYCLES SQC_TC_STALL" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-include-regex zstd_compress -d $OUT/p$i -o p$i --output-format csv -- $CMD > $OUT/p$i.log 2>&1
done
