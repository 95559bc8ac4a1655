#!/bin/bash
# same-box A/B of the gather forward's entries in flight per wave (FWD_U): headline step + the forward kernel's own time
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in 4 8 16; do
    CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_FWD_U_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/fwdu.log 2>&1 || { tail -20 gpurun_out/fwdu.log; exit 1; }
    echo "FWD_U=$v $(tail -1 gpurun_out/fwdu.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1))')"
done; done
