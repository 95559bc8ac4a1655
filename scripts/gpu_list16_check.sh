#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_fullsize_configs_gpu.py tests/test_fp16_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "build or threshold or lists or sparse or jumprelu or fp16" > gpurun_out/list16_tests.log 2>&1 || { tail -40 gpurun_out/list16_tests.log; exit 1; }
tail -2 gpurun_out/list16_tests.log
for dec in plt clt; do for v in 0 1; do
  CLT_AB=list_bf16_act=$v timeout -k 10 240 python -u bench.py --nonlinearity jumprelu --decoder $dec --theta 0.26 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/jl.log 2>&1 || { tail -20 gpurun_out/jl.log; exit 1; }
  echo "$dec CLT_AB=list_bf16_act=$v $(tail -1 gpurun_out/jl.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["regime"], d["config"]["loss"])')"
done; done
