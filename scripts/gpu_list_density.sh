#!/bin/bash
# JumpReLU list regime: step time against the share of active entries (theta sweep, lists forced), PLT or CLT ($1)
set -o pipefail
mkdir -p gpurun_out
dec=${1:-plt}; out=gpurun_out/list_density_$dec.jsonl; : > $out
for th in 0.30 0.24 0.21 0.19 0.175 0.16; do
  timeout -k 10 240 python -u bench.py --nonlinearity jumprelu --decoder $dec --theta $th --sparse on --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ld.log 2>&1 || { tail -20 gpurun_out/ld.log; exit 1; }
  tail -1 gpurun_out/ld.log >> $out
  tail -1 gpurun_out/ld.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$dec theta $th:', round(l['ms_per_step'],2), 'ms', l['config']['regime'], 'active share', l['config'].get('list_density_first'), '->', l['config']['list_density'])"
done
