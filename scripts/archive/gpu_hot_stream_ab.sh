#!/bin/bash
# zipf / uniform / JumpReLU-list configurations with the hot rows' launches on their own stream (1) or behind the common launch (0); same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/hot_stream_ab.jsonl
run() {
  tag=$1; shift
  for rep in 1 2; do for on in 1 0; do
    CLT_AB=hot_rows_stream=$on timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > gpurun_out/hot_run.log 2>&1 || { tail -20 gpurun_out/hot_run.log; exit 1; }
    tail -1 gpurun_out/hot_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$tag','hot_rows_stream':$on,'ms_per_step':round(d['ms_per_step'],3),'loss':d['config']['loss'],'density':d['config'].get('list_density')}))" | tee -a gpurun_out/hot_stream_ab.jsonl
  done; done
}
run zipf --skew zipf && run uniform && run plt_jumprelu_lists_0.5pct --nonlinearity jumprelu --decoder plt --theta 0.2 --sparse on
