#!/bin/bash
# TopK-aux step: tests of the compacted pass, then bench A/B of the switches named on the command line (VAR=0 vs default)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "columns or compact or aux or mask_positive or weight_prep or topk_rows or selection" > gpurun_out/aux_tests.log 2>&1 || { tail -30 gpurun_out/aux_tests.log; exit 1; }
tail -2 gpurun_out/aux_tests.log
for var in "$@"; do
  for v in 0 1; do
    env $var=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --nonlinearity topk --aux > gpurun_out/aux_ab.log 2>&1 || { tail -20 gpurun_out/aux_ab.log; exit 1; }
    echo "$var=$v $(tail -1 gpurun_out/aux_ab.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
  done
done
