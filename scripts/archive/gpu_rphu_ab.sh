#!/bin/bash
# row pass with RP_HU entries' gathers in flight (ab/libclt_hip_RP_HU_<n>.so): parity tests on each variant, then headline and
# JumpReLU list-regime benches
set -o pipefail
mkdir -p gpurun_out
for lib in "" ab/libclt_hip_RP_HU_2.so ab/libclt_hip_RP_HU_4.so; do
  CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "row_pass or sparse or lists or golden or step" > gpurun_out/rphu_tests.log 2>&1 || { tail -30 gpurun_out/rphu_tests.log; exit 1; }
  echo "${lib:-shipped}: $(tail -1 gpurun_out/rphu_tests.log)"
done
for lib in "" ab/libclt_hip_RP_HU_2.so ab/libclt_hip_RP_HU_4.so; do
  for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--nonlinearity jumprelu --decoder plt --theta 0.22 --sparse on --steps 6 --warmup 2" "--nonlinearity jumprelu --decoder plt --theta 0.18 --sparse on --steps 6 --warmup 2" "--nonlinearity jumprelu --decoder clt --theta 0.22 --sparse on --steps 5 --warmup 2"; do
    CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > gpurun_out/rphu.log 2>&1 || { tail -20 gpurun_out/rphu.log; exit 1; }
    echo "${lib:-shipped} [$args] $(tail -1 gpurun_out/rphu.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
  done
done
