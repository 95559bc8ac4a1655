#!/bin/bash
# occupancy-hint variants (ab/libclt_hip_<MACRO>_<v>.so, scripts/build_variants.sh): weight_prep probe + PLT JumpReLU bench
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=.
for lib in "" ab/libclt_hip_WP_WAVES_3.so ab/libclt_hip_WP_WAVES_4.so; do
  echo "== weight_prep ${lib:-shipped}"
  CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 120 python3 scripts/prep_probe.py 2>&1 | tail -5 || exit 1
done
for lib in "" ab/libclt_hip_JL_WAVES_4.so ab/libclt_hip_WP_WAVES_4.so; do
  CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --nonlinearity jumprelu --decoder plt --sparse off > gpurun_out/waves_ab.log 2>&1 || { tail -20 gpurun_out/waves_ab.log; exit 1; }
  echo "PLT jumprelu ${lib:-shipped}: $(tail -1 gpurun_out/waves_ab.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
done
