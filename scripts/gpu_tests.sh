#!/bin/bash
# GPU test run: every step under its own time limit, chained so that a fault ends the call.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_gpu.log
exit $rc
