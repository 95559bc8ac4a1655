#!/bin/bash
# one rank's share of an N-way feature shard, N = 8 / 4 / 2 (sharded schedule through RCCL on one rank; no wire time)
set -o pipefail
for a in "8 4096 2" "4 8192 4" "2 16384 8"; do
  set -- $a
  timeout -k 10 200 python3 bench.py --shard-rehearsal --features $2 --k $3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/share.json || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/share.json')); print('N=$1 share', round(d['ms_per_step'],3), 'ms')"
done
