#!/bin/bash
# A/B of bf16 activations (pre-activations / feature gradient as bf16) in the dense bf16 JumpReLU regime.
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/act16.jsonl
for dec in plt clt; do
  for a in 0 1 0 1; do
    CLT_AB=dense_bf16_act=$a timeout -k 10 200 python bench.py --nonlinearity jumprelu --decoder $dec --sparse off --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/act16_one.log 2>&1 || { tail -20 gpurun_out/act16_one.log; exit 1; }
    tail -1 gpurun_out/act16_one.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dec act16=$a', d['ms_per_step'], d['value'])" | tee -a gpurun_out/act16.jsonl
  done
done
