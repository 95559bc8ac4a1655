#!/bin/bash
# PerLayerTopK selection: tests, then the two list-regime benches that use it
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_fullsize_configs_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "select or topk or rows or aux or config5 or perlayer" > gpurun_out/topk_tests.log 2>&1 || { tail -30 gpurun_out/topk_tests.log; exit 1; }
tail -2 gpurun_out/topk_tests.log
for args in "--nonlinearity topk" "--nonlinearity topk --aux"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $args > gpurun_out/topk_ab.log 2>&1 || { tail -20 gpurun_out/topk_ab.log; exit 1; }
  echo "$args: $(tail -1 gpurun_out/topk_ab.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
done
