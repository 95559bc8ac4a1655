#!/bin/bash
# feature-major half of the list build on its own stream beside the gather forward (1) or on the launch stream (0); same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/csc_stream_ab.jsonl
run() {
  tag=$1; shift
  for rep in 1 2 3; do for on in 1 0; do
    CLT_AB=list_csc_stream=$on timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > gpurun_out/csc_run.log 2>&1 || { tail -20 gpurun_out/csc_run.log; exit 1; }
    tail -1 gpurun_out/csc_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$tag','list_csc_stream':$on,'ms_per_step':round(d['ms_per_step'],3),'loss':d['config']['loss'],'fallbacks':d['config']['select_fallbacks']}))" | tee -a gpurun_out/csc_stream_ab.jsonl
  done; done
}
run uniform && run zipf --skew zipf
