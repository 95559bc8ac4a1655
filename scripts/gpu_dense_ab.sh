#!/bin/bash
# dense-regime configurations (and the headline), one bench line each on one box; SWITCH=<CLT_AB name> SWITCH_VALUES="1 0" runs every
# configuration once per value: SWITCH=table_cache SWITCH_VALUES="1 0" bash scripts/gpu_dense_ab.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/dense.jsonl
run() {
  tag=$1; shift
  for on in ${SWITCH_VALUES:-1}; do
  CLT_AB=${SWITCH:+$SWITCH=$on} timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/dense_run.log 2>&1 || { tail -20 gpurun_out/dense_run.log; exit 1; }
  tail -1 gpurun_out/dense_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$tag','${SWITCH:-default}':'$on','ms_per_step':round(d['ms_per_step'],3),'loss':d['config']['loss']}))" | tee -a gpurun_out/dense.jsonl
  done
}
run clt_jumprelu --nonlinearity jumprelu --decoder clt && run plt_jumprelu --nonlinearity jumprelu --decoder plt && run default_yaml --nonlinearity jumprelu --decoder clt --features 10000 --batch 4000 && run topk_aux --aux --nonlinearity topk && run headline
