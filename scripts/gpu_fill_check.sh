#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_fullsize_configs_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "build or threshold or lists or sparse or jumprelu" > gpurun_out/fill_tests.log 2>&1 || { tail -30 gpurun_out/fill_tests.log; exit 1; }
tail -2 gpurun_out/fill_tests.log
bash scripts/gpu_jumprelu_lists.sh
