#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for n in 1 2; do
  for cfg in "--nonlinearity topk --aux --features 20000 --batch 2000"; do
    CLT_AB=kseg_dgrad_streams=$n timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > gpurun_out/kseg.log 2>&1 || { tail -20 gpurun_out/kseg.log; exit 1; }
    echo "streams=$n $cfg: $(tail -1 gpurun_out/kseg.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"].get("loss"))')"
  done
done; done
