#!/bin/bash
# same-box A/B: a remainder of two list entries requested together in the row pass (RP_TAIL2)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_RP_TAIL2_1.so timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread -k "row_pass or training_matches or threshold" > gpurun_out/tail2_tests.log 2>&1 || { tail -30 gpurun_out/tail2_tests.log; exit 1; }
tail -2 gpurun_out/tail2_tests.log
for rep in 1 2 3; do
for v in 0 1; do
    CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_RP_TAIL2_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/tail2.log 2>&1 || { tail -20 gpurun_out/tail2.log; exit 1; }
    echo "RP_TAIL2=$v $(tail -1 gpurun_out/tail2.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3), d["config"]["loss"])')"
done; done
for v in 0 1; do
    CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_RP_TAIL2_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --nonlinearity jumprelu --decoder plt --theta 0.26 --sparse auto > gpurun_out/tail2.log 2>&1 || exit 1
    echo "RP_TAIL2=$v plt lists $(tail -1 gpurun_out/tail2.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
done
