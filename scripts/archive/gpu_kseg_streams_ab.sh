#!/bin/bash
# K-segmented feature-gradient GEMMs on one stream vs alternating over two (short last rounds): TopK + aux and default.yaml shapes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_step_gpu.py tests/test_kernels_gpu.py -x -q -m gpu --timeout 120 --timeout-method thread -k "aux or dgrad or dense_bf16 or training_matches" > gpurun_out/kseg_tests.log 2>&1 || { tail -30 gpurun_out/kseg_tests.log; exit 1; }
tail -2 gpurun_out/kseg_tests.log
for rep in 1 2; do for n in 1 2; do
  for cfg in "--aux --nonlinearity topk" "--nonlinearity jumprelu --features 10000 --batch 4000" "--nonlinearity topk --aux --features 20000 --batch 2000"; do
    CLT_AB=kseg_dgrad_streams=$n timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 --warmup 3 $cfg > gpurun_out/kseg.log 2>&1 || { tail -20 gpurun_out/kseg.log; exit 1; }
    echo "streams=$n $cfg: $(tail -1 gpurun_out/kseg.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"].get("loss"))')"
  done
done; done
