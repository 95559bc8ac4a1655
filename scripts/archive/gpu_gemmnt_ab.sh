#!/bin/bash
# same-box A/B: non-temporal C stores in the NT GEMM epilogues (dense JumpReLU configurations + headline)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in 0 1; do
  for cfg in "--nonlinearity jumprelu --decoder plt" "--nonlinearity jumprelu --decoder clt --steps 5" "--aux --nonlinearity topk"; do
    CLT_HIP_LIBRARY=$PWD/ab/libclt_hip_GEMM_NT_STORE_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > gpurun_out/gnt.log 2>&1 || { tail -20 gpurun_out/gnt.log; exit 1; }
    echo "GEMM_NT_STORE=$v $cfg $(tail -1 gpurun_out/gnt.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
  done
done; done
