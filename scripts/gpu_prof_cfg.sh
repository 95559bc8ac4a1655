#!/bin/bash
# rocprofv3 kernel stats of one bench configuration -> markdown on stdout: bash scripts/gpu_prof_cfg.sh <name> <bench args...>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
mkdir -p gpurun_out; rm -rf gpurun_out/$name
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/$name.log 2>&1 || { tail -30 gpurun_out/$name.log; exit 1; }
tail -1 gpurun_out/$name.log | cut -c1-700
f=$(find gpurun_out/$name -name '*kernel_stats.csv' | head -1)
python3 scripts/prof_summary.py $f "$name: bench.py $* (13 steps + 2 instrumented)" > gpurun_out/${name}_kernel_stats.md
head -34 gpurun_out/${name}_kernel_stats.md
