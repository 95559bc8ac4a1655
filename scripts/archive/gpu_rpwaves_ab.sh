#!/bin/bash
# same-box A/B: occupancy hint on the row pass (RP_WAVES = waves per SIMD; 5 and 6 spill 60 / 132 bytes of scratch)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "" ab/libclt_hip_RP_WAVES_5.so ab/libclt_hip_RP_WAVES_6.so; do   # (built from a tree that still had the RP_WAVES hook)
    CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/rpw.log 2>&1 || { tail -20 gpurun_out/rpw.log; exit 1; }
    echo "${lib:-shipped} $(tail -1 gpurun_out/rpw.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3))')"
done; done
