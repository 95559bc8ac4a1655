#!/bin/bash
# bench.py over the BASELINE.json configurations that fit one GPU (one JSON line each into gpurun_out/configs.jsonl)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; [ -n "$APPEND" ] || : > gpurun_out/configs.jsonl
run() { sleep 4; echo "== $*" ; timeout -k 10 400 python -u bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | tee -a gpurun_out/configs.jsonl | cut -c1-420 || exit 1; }
run --steps 20 --warmup 5
run --steps 10 --warmup 3 --aux --nonlinearity topk
run --steps 10 --warmup 3 --nonlinearity jumprelu --decoder plt
run --steps 10 --warmup 3 --nonlinearity jumprelu --decoder plt --batch 8192
run --steps 5 --warmup 2 --nonlinearity jumprelu --decoder clt
run --steps 5 --warmup 2 --nonlinearity jumprelu --decoder clt --layers 16 --d-acts 2048 --features 8192
run --steps 5 --warmup 2 --nonlinearity topk --aux --layers 32 --d-acts 4096 --features 8192 --distinct-batches 1
# the shapes of the reference's own YAML files (config/default.yaml: JumpReLU CLT, F=10000/layer, batch 4000;
# config/topk-clt.yaml: PerLayerTopK(16) + aux(512), F=20000/layer, batch 2000)
run --steps 10 --warmup 3 --nonlinearity jumprelu --features 10000 --batch 4000
run --steps 10 --warmup 3 --nonlinearity topk --aux --features 20000 --batch 2000
# other batch sizes of the headline and of config 2 (B in {1024, 4096, 8192})
run --steps 20 --warmup 5 --batch 1024
run --steps 20 --warmup 5 --batch 8192
run --steps 10 --warmup 3 --nonlinearity jumprelu --decoder plt --batch 1024
run --steps 10 --warmup 3 --nonlinearity topk
