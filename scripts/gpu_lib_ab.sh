#!/bin/bash
# bench.py with the shipped library and with each variant library given on the command line (ab/*.so); extra bench args via BENCH_ARGS
set -o pipefail
mkdir -p gpurun_out
for lib in "" "$@"; do
  for rep in 1 2; do
    CLT_HIP_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/lib_ab.log 2>&1 || { tail -20 gpurun_out/lib_ab.log; exit 1; }
    echo "${lib:-shipped}: $(tail -1 gpurun_out/lib_ab.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["config"]["loss"])')"
  done
done
