#!/bin/bash
# after the load-hoisting changes: kernel + step tests, weight_prep probe, benches of the configurations that use the touched kernels
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=.
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/hoist_tests.log 2>&1 || { tail -30 gpurun_out/hoist_tests.log; exit 1; }
tail -2 gpurun_out/hoist_tests.log
timeout -k 10 120 python3 scripts/prep_probe.py 2>&1 | tail -5 || exit 1
for args in "" "--nonlinearity topk" "--nonlinearity topk --aux" "--nonlinearity jumprelu --decoder plt" "--nonlinearity jumprelu --decoder clt --steps 5 --warmup 2"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $args > gpurun_out/hoist_bench.log 2>&1 || { tail -20 gpurun_out/hoist_bench.log; exit 1; }
  echo "[$args] $(tail -1 gpurun_out/hoist_bench.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
done
