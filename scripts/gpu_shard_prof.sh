#!/bin/bash
# one rank's share of an N-way feature shard (sharded schedule on one rank through RCCL): wall ms/step, kernel time per step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-8}
F=$((32768 / N)); K=$((16 / N))
mkdir -p gpurun_out; rm -rf gpurun_out/shardprof
timeout -k 10 200 python -u bench.py --shard-rehearsal --features $F --k $K --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/shard_plain.log 2>&1 || { tail -20 gpurun_out/shard_plain.log; exit 1; }
tail -1 gpurun_out/shard_plain.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('wall ms/step', round(l['ms_per_step'],3), l['config']['parallelism'], 'fallbacks', l['config']['select_fallbacks'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shardprof -- python3 bench.py --shard-rehearsal --features $F --k $K --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/shard_prof.log 2>&1 || { tail -20 gpurun_out/shard_prof.log; exit 1; }
f=$(find gpurun_out/shardprof -name '*kernel_stats.csv' | head -1)
python3 scripts/prof_summary.py $f "shard rehearsal N=$N (37 steps incl. warm-up + 2 instrumented)" | head -45
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(int(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernel ms per step", tot/1e6/37, "launches per step", calls/37)
PY
