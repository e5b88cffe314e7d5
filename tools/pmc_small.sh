#!/bin/bash
# PMC passes over the stand-alone CRC32C and AES-256-GCM kernels (BASELINE configs[1], [2]: one 1 GiB segment, device resident): HBM traffic
# and, for GCM, the LDS pipe (its design is 16 LDS look-ups per byte: T-table AES + 4-bit GHASH).  One counter group per run, no trace domains.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_small
rm -rf $O; mkdir -p $O
python $R/tools/prof_small.py crc --data /tmp/k256_1g.npy --reps 1 > /dev/null 2>&1
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --kernel-include-regex "crc32c_partial" -d $O/crc_p$i -o p$i --output-format csv -- python $R/tools/prof_small.py crc --data /tmp/k256_1g.npy > $O/crc_p$i.log 2>&1
  timeout 100 rocprofv3 --pmc $set --kernel-include-regex "gcm_ctr_ghash" -d $O/gcm_p$i -o p$i --output-format csv -- python $R/tools/prof_small.py gcm_crc --data /tmp/k256_1g.npy > $O/gcm_p$i.log 2>&1
done
find $O -name "*agent_info.csv" -delete
find $O -name "*counter_collection.csv" | head -20
