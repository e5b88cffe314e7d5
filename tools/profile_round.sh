#!/bin/bash
# Round-end evidence on the GPU box: bench lines, rocprofv3 kernel stats of the bench command, PMC traffic of the Zstd kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 12 --warmup 1 2>/dev/null | tail -1 > $O/bench_full.json
python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_full_inflight1.json
python bench.py --workload gcm_crc --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_gcm_crc.json
python bench.py --workload crc --steps 20 --warmup 2 2>/dev/null | tail -1 > $O/bench_crc.json
python tools/detransform_bench.py 2048 libtsxform.so 2>&1 | grep -v amdgpu.ids > $O/detransform.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-verify > $O/stats.log 2>&1
find $O/stats -name "*kernel_trace.csv" -delete; find $O/stats -name "*agent_info.csv" -delete; head -c 200000 $O/stats.log > $O/stats.log.head; rm -f $O/stats.log
python $R/tools/prof_zstd.py --chunks 256 --lib libtsxform.so --data /tmp/k256.npy > /dev/null 2>&1
CMD="python $R/tools/prof_zstd.py --chunks 2048 --dist K --chain --lib libtsxform.so --data /tmp/k256.npy"
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-include-regex zstd_compress -d $O/pmc$i -o p$i --output-format csv -- $CMD > $O/pmc$i.log 2>&1
done
find $O -name "*agent_info.csv" -delete
du -sh $O
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
