#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
prof() { tag=$1; shift; rm -rf gpurun_out/prof_$tag; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }; }
prof head && prof pltj --nonlinearity jumprelu --decoder plt && prof topk --nonlinearity topk && prof aux --aux --nonlinearity topk && prof pltl --nonlinearity jumprelu --decoder plt --theta 0.26 && prof cltl --nonlinearity jumprelu --decoder clt --theta 0.26
