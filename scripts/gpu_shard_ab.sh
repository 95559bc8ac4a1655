#!/bin/bash
# N = 8 share of the sharded schedule (one rank through RCCL, no wire) in both forms of the pipelined exchange + kernel trace of the split form
set -o pipefail
mkdir -p gpurun_out
for v in 0 1 0 1; do
  CLT_AB=exchange_split=$v timeout -k 10 200 python3 bench.py --shard-rehearsal --features 4096 --k 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/share_$v.json || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/share_$v.json')); print('exchange_split=$v N=8 share', round(d['ms_per_step'],3), 'ms')"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_split
cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_split -o split -- python3 bench.py --shard-rehearsal --features 4096 --k 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/prof_split.log 2>&1 || exit 1
ls gpurun_out/prof_split
