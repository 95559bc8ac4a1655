#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no trace domains
# besides the kernel trace), each pass its own short run under its own time limit.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_$c.log 2>&1 || { tail -20 gpurun_out/pmc_$c.log; exit 1; }
  tail -1 gpurun_out/pmc_$c.log | cut -c1-200
done
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name '*.csv' | xargs ls -la
