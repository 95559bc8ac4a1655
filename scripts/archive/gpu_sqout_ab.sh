#!/bin/bash
# same-box A/B of the headline step with the row pass's sq_out code compiled in (1) / out (0)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
for lib in ab/libclt_hip_RP_SQOUT_0.so ab/libclt_hip_RP_SQOUT_1.so; do
    CLT_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/sqout.log 2>&1 || { tail -20 gpurun_out/sqout.log; exit 1; }
    echo "$lib $(tail -1 gpurun_out/sqout.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3))')"
done; done
