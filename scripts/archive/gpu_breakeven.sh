#!/bin/bash
# JumpReLU module: list regime forced (--sparse on) vs dense (--sparse off) over a range of thresholds = active densities
set -o pipefail
mkdir -p gpurun_out
for dec in plt clt; do for th in 0.22 0.18 0.15 0.12; do for sp in on off; do
  timeout -k 10 240 python -u bench.py --nonlinearity jumprelu --decoder $dec --theta $th --sparse $sp --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/be.log 2>&1 || { tail -20 gpurun_out/be.log; exit 1; }
  echo "$dec theta=$th sparse=$sp $(tail -1 gpurun_out/be.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), "ms density", d["config"]["list_density"])')"
done; done; done
