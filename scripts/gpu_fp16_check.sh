#!/bin/bash
# full GPU suite, then the headline in bf16 (regression check of the format-templated kernels) and in fp16 mode
set -o pipefail
mkdir -p gpurun_out
bash scripts/gpu_tests.sh || exit 1
for p in bf16-mixed 16-mixed bf16-mixed 16-mixed; do
  timeout -k 10 200 python bench.py --precision $p --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fp16_one.log 2>&1 || { tail -20 gpurun_out/fp16_one.log; exit 1; }
  tail -1 gpurun_out/fp16_one.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p', d['dtype'], round(d['ms_per_step'],3), round(d['value']), d['roofline']['achieved'], d['config'].get('select_fallbacks'))" | tee -a gpurun_out/fp16_bench.txt
done
