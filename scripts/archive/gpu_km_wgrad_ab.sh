#!/bin/bash
# PLT dense JumpReLU steps with (1) / without (0) the k-major weight gradient (no transposed feature copy), three batch sizes, twice each; dense step tests first
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/km_wgrad_ab.jsonl
timeout -k 10 300 python -u -m pytest tests/test_step_gpu.py tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "dense_bf16_fast_path or fused_dense_jumprelu or adam_pass_that_also or reference_cycle" > gpurun_out/km_wgrad_test.log 2>&1 || { tail -30 gpurun_out/km_wgrad_test.log; exit 1; }
tail -n 2 gpurun_out/km_wgrad_test.log
run() {
  tag=$1; shift
  for rep in 1 2; do for on in 1 0; do
    CLT_AB=km_wgrad_plt=$on timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 20 --warmup 4 --nonlinearity jumprelu --decoder plt "$@" > gpurun_out/kw_run.log 2>&1 || { tail -20 gpurun_out/kw_run.log; exit 1; }
    tail -1 gpurun_out/kw_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$tag','km_wgrad_plt':$on,'ms_per_step':round(d['ms_per_step'],3),'loss':d['config']['loss'],'peak_gb':d['config']['hbm']['peak_allocated_gb']}))" | tee -a gpurun_out/km_wgrad_ab.jsonl
  done; done
}
run plt_jumprelu_b4096 && run plt_jumprelu_b8192 --batch 8192 && run plt_jumprelu_b1024 --batch 1024
