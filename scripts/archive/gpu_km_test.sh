#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "${KEXPR:-k_major or gemm or dense_contractions or grouped or tail_split}" > gpurun_out/km_test.log 2>&1; rc=$?
tail -n 40 gpurun_out/km_test.log
exit $rc
