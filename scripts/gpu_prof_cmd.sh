#!/bin/bash
# rocprofv3 kernel statistics of one bench.py configuration: scripts/gpu_prof_cmd.sh <tag> <bench.py flags ...>
# (STEPS / WARMUP in the environment: default 5 / 2)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out
rm -rf gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -30 gpurun_out/prof_$tag.log; exit 1; }
tail -1 gpurun_out/prof_$tag.log
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 scripts/prof_summary.py "$f" "$tag: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} --no-cpu-baseline $*" > gpurun_out/prof_$tag.md
head -24 gpurun_out/prof_$tag.md
