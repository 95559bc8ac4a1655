#!/bin/bash
# rocprofv3 kernel trace of one bench configuration: bash scripts/gpu_prof.sh <outdir-name> <bench args...>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
mkdir -p gpurun_out; rm -rf gpurun_out/$name
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$name.log 2>&1 || { tail -30 gpurun_out/$name.log; exit 1; }
tail -1 gpurun_out/$name.log | cut -c1-300
