#!/bin/bash
# one-off sweep on the GPU box: speculation width / occupancy / hw queues under batches in flight
cd /root/repo
run() { # lib inflight
  timeout 200 python tools/bench_with_lib.py $1 --steps 12 --warmup 1 --inflight $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'inflight', $2, j['value'], j['ms_per_step'], j['roofline']['stage_ms_per_step']['zstd'])"
}
for t in w4 w6 w12; do run tools/_libs/libtsxform_$t.so 3; done
run tools/_libs/libtsxform_occ4.so 2
run tools/_libs/libtsxform_occ4.so 3
export GPU_MAX_HW_QUEUES=8
run tiered-storage-for-apache-kafka_amd/libtsxform.so 4
run tiered-storage-for-apache-kafka_amd/libtsxform.so 6
