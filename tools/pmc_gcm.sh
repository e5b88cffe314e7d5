#!/bin/bash
# The GCM half of tools/pmc_small.sh alone (after a change to gcm.hip that leaves the CRC kernel's sources as they are): FETCH_SIZE / WRITE_SIZE
# and the SQ groups of gcm_ctr_ghash_kernel, one counter group per run, no trace domains.  tools/pmc_traffic.py at home then refreshes the record.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_small
mkdir -p $O; rm -rf $O/gcm_p*
python $R/tools/prof_small.py gcm_crc --data /tmp/k256_1g.npy --reps 1 > /dev/null 2>&1
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --kernel-include-regex "gcm_ctr_ghash" -d $O/gcm_p$i -o p$i --output-format csv -- python $R/tools/prof_small.py gcm_crc --data /tmp/k256_1g.npy > $O/gcm_p$i.log 2>&1
done
find $O -name "*agent_info.csv" -delete
find $O -name "*counter_collection.csv" | head -20
