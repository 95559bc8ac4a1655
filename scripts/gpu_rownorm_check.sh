#!/bin/bash
# kept row norms (row pass sq_out): kernel + engine tests, then the JumpReLU list-regime steps with and without them, then the headline
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread -k "row_pass or threshold or row_norm" > gpurun_out/rownorm_tests.log 2>&1 || { tail -30 gpurun_out/rownorm_tests.log; exit 1; }
tail -2 gpurun_out/rownorm_tests.log
: > gpurun_out/rownorm.jsonl
for dec in clt plt; do for keep in 1 0; do
  echo "== $dec keep=$keep"
  CLT_ROW_PASS_NORMS=$keep timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 3 --nonlinearity jumprelu --decoder $dec --theta 0.26 --sparse auto 2>&1 | tail -1 | tee -a gpurun_out/rownorm.jsonl | cut -c1-200 || exit 1
done; done
timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | tee -a gpurun_out/rownorm.jsonl | cut -c1-700
