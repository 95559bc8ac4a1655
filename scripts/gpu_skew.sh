#!/bin/bash
# uniform vs heavy-tailed (zipf) feature frequencies on the headline configuration, hot-row form on / off; same box.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/skew.jsonl
for cfg in "--skew none" "--skew zipf" ; do
  for hot in 64 0; do
    CLT_AB=hot_min_len=$hot timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 $cfg "$@" > gpurun_out/skew_run.log 2>&1 || { tail -20 gpurun_out/skew_run.log; exit 1; }
    tail -1 gpurun_out/skew_run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'cfg':'$cfg','hot_min_len':$hot,'ms_per_step':round(d['ms_per_step'],3),'tokens_per_s':round(d['value']),'row_pass_us':round(d['roofline']['avg_launch_us'],1),'fallbacks':d['config']['select_fallbacks'],'skew':d['config'].get('skew')}))" | tee -a gpurun_out/skew.jsonl
  done
done
