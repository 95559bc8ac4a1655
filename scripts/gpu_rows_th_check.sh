#!/bin/bash
# four-rows-per-block threshold list build: kernel tests, the probe, then the list-regime steps
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread -k "build or threshold or list or sparse" > gpurun_out/rows_th_tests.log 2>&1 || { tail -30 gpurun_out/rows_th_tests.log; exit 1; }
tail -2 gpurun_out/rows_th_tests.log
PYTHONPATH=$PWD timeout -k 10 200 python -u scripts/build_probe.py || exit 1
for dec in plt clt; do
  bash scripts/gpu_prof_cfg.sh ra_$dec --nonlinearity jumprelu --decoder $dec --theta 0.26 --sparse auto > gpurun_out/ra_$dec.out 2>&1 || { tail -20 gpurun_out/ra_$dec.out; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/ra_$dec.log; grep "count_rows\|fill_rows" gpurun_out/ra_${dec}_kernel_stats.md
done
