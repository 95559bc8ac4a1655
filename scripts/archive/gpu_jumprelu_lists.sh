#!/bin/bash
# JumpReLU module: activation-list regime (low L0) beside the dense GEMM regime, PLT (config 2) and CLT.
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/jumprelu_lists.jsonl; : > $out
for dec in plt clt; do
  for mode in "--theta 0.26 --sparse auto" "--theta 0.26 --sparse off" "--theta 0.03 --sparse auto"; do
    timeout -k 10 240 python -u bench.py --nonlinearity jumprelu --decoder $dec $mode --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/jl.log 2>&1 || { tail -20 gpurun_out/jl.log; exit 1; }
    tail -1 gpurun_out/jl.log >> $out
    tail -1 gpurun_out/jl.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$dec $mode', round(l['ms_per_step'],2), 'ms', l['config']['regime'], l['roofline'] and round(l['roofline']['achieved']))"
  done
done
