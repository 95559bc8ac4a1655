#!/bin/bash
# Round-2 evidence run: headline bench (both APIs), rocprofv3 kernel stats of the same command, PMC traffic passes,
# a 50-step uninstrumented run (step-time stability), and the configuration table.  Every GPU step under its own limit.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.log 2>&1 || { tail -20 gpurun_out/r02_bench.log; exit 1; }
tail -1 gpurun_out/r02_bench.log > gpurun_out/r02_bench.json; cut -c1-300 gpurun_out/r02_bench.json
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --api fused_train_step --no-cpu-baseline > gpurun_out/r02_bench_fts.log 2>&1 || exit 1
tail -1 gpurun_out/r02_bench_fts.log > gpurun_out/r02_bench_fused_train_step.json; cut -c1-300 gpurun_out/r02_bench_fused_train_step.json
timeout -k 10 200 python -u bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r02_bench50.log 2>&1 || exit 1
tail -1 gpurun_out/r02_bench50.log > gpurun_out/r02_bench_50steps.json; cut -c1-300 gpurun_out/r02_bench_50steps.json
rm -rf gpurun_out/r02prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02prof.log 2>&1 || { tail -20 gpurun_out/r02prof.log; exit 1; }
f=$(find gpurun_out/r02prof -name '*kernel_stats.csv' | head -1)
python3 scripts/prof_summary.py $f "r02 headline: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline (27 steps incl. warm-up and the 2-step roofline pass)" > gpurun_out/r02_kernel_stats.md
head -16 gpurun_out/r02_kernel_stats.md
bash scripts/gpu_pmc.sh > gpurun_out/r02_pmc.log 2>&1 || { tail -20 gpurun_out/r02_pmc.log; exit 1; }
python3 scripts/pmc_summary.py > gpurun_out/r02_pmc_traffic.md && cp profiles/pmc_traffic.json gpurun_out/r02_pmc_traffic.json
head -8 gpurun_out/r02_pmc_traffic.md
