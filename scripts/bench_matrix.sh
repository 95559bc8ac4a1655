#!/bin/bash
# bench.py over a matrix of less-travelled configurations (fp32 mode, ReLU, PLT sparse, sharded rehearsal of the dense and aux
# paths, tiny / odd shapes): a smoke check that every path still runs end to end, 3 steps each.
set -o pipefail
run() { echo "== $*"; timeout -k 10 300 python -u bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | cut -c1-160; }
run --precision 32 --features 4096 --batch 1024
run --precision 32 --features 4096 --batch 1024 --nonlinearity jumprelu
run --nonlinearity relu --features 8192 --batch 2048
run --decoder plt --nonlinearity batchtopk
run --decoder plt --nonlinearity topk --aux
run --shard-rehearsal --nonlinearity jumprelu --features 4096
run --shard-rehearsal --nonlinearity topk --aux --features 4096 --k 4
run --precision 32 --nonlinearity topk --aux --features 4096 --batch 1024
run --features 32 --d-acts 110 --layers 2 --batch 16 --k 4
run --features 1000 --d-acts 100 --layers 3 --batch 100 --nonlinearity jumprelu
