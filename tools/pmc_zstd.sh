#!/bin/bash
# PMC passes for zstd_service_kernel (one launch = the 2048 chunks of one batch) (run on the GPU box through gpurun).  Counters in separate passes, no tracing domains.
# The synthetic chunks are generated once outside rocprofv3 (thousands of tiny generator kernels would each be serialized).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_DIR:-pmc}       # PMC_DIR=pmc_quad TSX_ZSTD_QUAD=1 bash tools/pmc_zstd.sh: the four-chunks-per-wave kernel
mkdir -p $OUT
python $R/tools/prof_zstd.py --chunks 256 --lib libtsxform.so --data /tmp/k256.npy > /dev/null 2>&1
CMD="python $R/tools/prof_zstd.py --chunks 2048 --dist K --chain --lib libtsxform.so --data /tmp/k256.npy"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-include-regex zstd_service -d $OUT/p$i -o p$i --output-format csv -- $CMD > $OUT/p$i.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head -20
