#!/bin/bash
# same-box A/B: non-temporal activation streams in the tile passes (TILE_NT: bit 0 loads, bit 1 stores), dense JumpReLU configurations
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in 0 1 2 3; do
  for cfg in "--nonlinearity jumprelu --decoder plt" "--nonlinearity jumprelu --decoder plt --theta 0.26 --sparse auto"; do
    CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_TILE_NT_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > gpurun_out/tnt.log 2>&1 || { tail -20 gpurun_out/tnt.log; exit 1; }
    echo "TILE_NT=$v $cfg $(tail -1 gpurun_out/tnt.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
  done
done; done
