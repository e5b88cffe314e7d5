#!/bin/bash
# PMC passes over the frame decoder (zstd_decompress_kernel), one counter group per run (never together with a trace domain):
# instruction mix, wait cycles, LDS activity / bank conflicts, HBM traffic.  Run on the GPU box: bash tools/pmc_dec.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_dec
rm -rf $O; mkdir -p $O
CMD="python $R/tools/detransform_bench.py 2048 libtsxform.so"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-include-regex zstd_decompress -d $O/pmc$i -o p$i --output-format csv -- $CMD > $O/pmc$i.log 2>&1
done
find $O -name "*agent_info.csv" -delete
find $O -name "*counter_collection.csv" | head
