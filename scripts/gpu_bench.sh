#!/bin/bash
# smoke + bench + rocprofv3 kernel trace of the same bench command; each GPU step under its own time limit.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 180 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 400 python -u bench.py "$@" > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1 || { tail -30 gpurun_out/bench_prof.log; exit 1; }
tail -1 gpurun_out/bench_prof.log
find gpurun_out/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -25
