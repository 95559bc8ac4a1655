#!/bin/bash
# JumpReLU list regime with the phased list build (1) or the one-call build (0): PLT at two densities, CLT; same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/thr_lists_ab.jsonl
run() {
  tag=$1; shift
  for rep in 1 2; do for on in 1 0; do
    CLT_AB=list_csc_stream=$on timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 3 --nonlinearity jumprelu --sparse ${SPARSE:-auto} "$@" > gpurun_out/thr_run.log 2>&1 || { tail -20 gpurun_out/thr_run.log; exit 1; }
    tail -1 gpurun_out/thr_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$tag','list_csc_stream':$on,'ms_per_step':round(d['ms_per_step'],3),'loss':d['config']['loss'],'density':d['config'].get('list_density'),'regime':d['config']['regime']}))" | tee -a gpurun_out/thr_lists_ab.jsonl
  done; done
}
run plt_theta0.26 --decoder plt --theta 0.26 && run plt_theta0.22 --decoder plt --theta 0.22 && run clt_theta0.26 --decoder clt --theta 0.26
