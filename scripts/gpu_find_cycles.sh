#!/bin/bash
# scripts/find_cycles.py over the step's regimes at a small shape: allocation growth with the collector off, and the closures found in cycles
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/cycles.txt
run() {
  tag=$1; shift
  echo "## $tag: $*" >> gpurun_out/cycles.txt
  timeout -k 10 200 python3 -u scripts/find_cycles.py --features 2048 --batch 512 "$@" > gpurun_out/cycles_run.txt 2>&1 || { tail -20 gpurun_out/cycles_run.txt; exit 1; }
  grep "^step\|^collected\|^\[" gpurun_out/cycles_run.txt >> gpurun_out/cycles.txt
  grep "^function\|^object" gpurun_out/cycles_run.txt | grep -v memory_stats | sort | uniq -c >> gpurun_out/cycles.txt || true
}
run headline --k 4 && run plt_jumprelu_dense --nonlinearity jumprelu --decoder plt && run clt_jumprelu_dense --nonlinearity jumprelu --decoder clt && \
run clt_jumprelu_lists --nonlinearity jumprelu --decoder clt --theta 0.26 && run topk --nonlinearity topk --k 4 && run topk_aux --nonlinearity topk --aux --k 4 && \
run fp32 --precision 32 --k 4 && run relu --nonlinearity relu && run fp16 --precision 16-mixed --k 4 && run shard_rehearsal --shard-rehearsal --k 4 && run shard_rehearsal_jumprelu --shard-rehearsal --nonlinearity jumprelu && run zipf --skew zipf --k 4
cat gpurun_out/cycles.txt
