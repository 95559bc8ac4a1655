#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
for lib in ab/libclt_hip_RP_HU_2.so ab/libclt_hip_RP_HU_3.so ab/libclt_hip_RP_HU_4.so; do
    CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/rphu.log 2>&1 || { tail -20 gpurun_out/rphu.log; exit 1; }
    echo "${lib:-shipped} $(tail -1 gpurun_out/rphu.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3))')"
done; done
