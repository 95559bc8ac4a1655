#!/bin/bash
# same-box A/B of non-temporal weight / moment streams in the row pass (RP_NT: bit 0 loads, bit 1 stores)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in 0 1 2 3; do
    CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_RP_NT_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/rpnt.log 2>&1 || { tail -20 gpurun_out/rpnt.log; exit 1; }
    echo "RP_NT=$v $(tail -1 gpurun_out/rpnt.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3), d["config"]["loss"])')"
done; done
