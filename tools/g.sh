#!/bin/bash
# build product + prof libs, run timing + phase profile + zstd parity on the GPU box; $1 = tag
cd /root/repo/tiered-storage-for-apache-kafka_amd/csrc && make 2>&1 | grep -E "error"; make prof 2>&1 | grep -E "error"
cd /root/repo && /usr/local/graft/bin/gpurun --timeout 400 -- "python tools/prof_zstd.py --chunks 2048 --lib libtsxform.so 2>&1 | grep -E 'zstd_ms_1'; python tools/prof_zstd.py --chunks 2048 --dist K --out gpurun_out/prof_K_2048_$1.json > gpurun_out/prof_K.log 2>&1; python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'zstd or chain' 2>&1 | tail -3" 2>&1 | tail -5
python tools/show_prof.py gpurun_out/prof_K_2048_$1.json
