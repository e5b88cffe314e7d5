#!/bin/bash
# round-2 GPU call 3: the bench line with its new legs, rocprofv3 kernel stats of the bench command, PMC passes of the compressor
# (current build) and - for the first time - of the frame decoder, the line-rate wall (tools/ubench/mix) on the same box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final; rm -rf $O; mkdir -p $O
cd $R
timeout 500 python bench.py --steps 18 --warmup 1 2> $O/bench.err | tail -1 > $O/bench_full.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python $R/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-verify --no-end-to-end > $O/stats.log 2>&1
find $O/stats -name "*kernel_trace.csv" -delete; find $O/stats -name "*agent_info.csv" -delete; tail -c 3000 $O/stats.log > $O/stats.log.tail; rm -f $O/stats.log
python $R/tools/prof_zstd.py --chunks 256 --lib libtsxform.so --data /tmp/k256.npy > /dev/null 2>&1
CMD="python $R/tools/prof_zstd.py --chunks 2048 --dist K --chain --lib libtsxform.so --data /tmp/k256.npy"
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-include-regex zstd_compress -d $O/pmc/p$i -o p$i --output-format csv -- $CMD > $O/pmc_p$i.log 2>&1
done
CMD="python $R/tools/detransform_bench.py 2048 libtsxform.so"
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-include-regex zstd_decompress -d $O/pmc_dec/p$i -o p$i --output-format csv -- $CMD > $O/pmc_dec_p$i.log 2>&1
done
find $O -name "*agent_info.csv" -delete
cd $R
python tools/show_pmc.py $O/pmc > $O/pmc_compress_summary.txt 2>&1
python tools/show_pmc.py $O/pmc_dec > $O/pmc_decompress_summary.txt 2>&1
timeout 100 tools/ubench/mix > $O/ubench_mix_same_box.txt 2>&1
find $O -name "*counter_collection.csv" -size +2M -delete
cut -c1-1500 $O/bench_full.json; echo; cat $O/pmc_compress_summary.txt; echo ---; cat $O/pmc_decompress_summary.txt; echo ---; grep "5120\|4096" $O/ubench_mix_same_box.txt | head -8
