#!/bin/bash
# PLT dense JumpReLU steps with (1) / without (0) the tail split of the decoder forward GEMM, three batch sizes, twice each; and the new test
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/tail_split_ab.jsonl
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "tail_split or grouped" > gpurun_out/tail_split_test.log 2>&1 || { tail -30 gpurun_out/tail_split_test.log; exit 1; }
tail -n 2 gpurun_out/tail_split_test.log
run() {
  tag=$1; shift
  for rep in 1 2; do for on in 1 0; do
    CLT_AB=gemm_tail_split=$on timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 3 --nonlinearity jumprelu --decoder plt "$@" > gpurun_out/ts_run.log 2>&1 || { tail -20 gpurun_out/ts_run.log; exit 1; }
    tail -1 gpurun_out/ts_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$tag','gemm_tail_split':$on,'ms_per_step':round(d['ms_per_step'],3),'loss':d['config']['loss']}))" | tee -a gpurun_out/tail_split_ab.jsonl
  done; done
}
run plt_jumprelu_b8192 --batch 8192 && run plt_jumprelu_b2048 --batch 2048 && run plt_jumprelu_b4096
