#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/gpu_tests.sh || exit 1
for rep in 1 2; do
for lib in "" ab/libclt_hip_RP_NT_3.so ab/libclt_hip_RP_NT_0.so; do
    CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/rpnt.log 2>&1 || { tail -20 gpurun_out/rpnt.log; exit 1; }
    echo "${lib:-shipped} $(tail -1 gpurun_out/rpnt.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3), d["config"]["loss"])')"
done; done
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
