#!/bin/bash
# N = 8 share of the sharded schedule (one rank through RCCL, no wire) in both forms of the pipelined exchange, twice each on one box
# (kernel trace of the split form: STEPS=8 scripts/gpu_prof_cmd.sh split8 --shard-rehearsal --features 4096 --k 2)
set -o pipefail
mkdir -p gpurun_out
for v in 0 1 0 1; do
  CLT_AB=exchange_split=$v timeout -k 10 200 python3 bench.py --shard-rehearsal --features 4096 --k 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/share_$v.json || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/share_$v.json')); print('exchange_split=$v N=8 share', round(d['ms_per_step'],3), 'ms')"
done
