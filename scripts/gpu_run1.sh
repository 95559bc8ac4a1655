set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 200 python -u bench.py --features 4096 --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1 || { tail -20 gpurun_out/bench_small.log; exit 1; }
tail -1 gpurun_out/bench_small.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dense -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dense.log 2>&1 || { tail -20 gpurun_out/bench_dense.log; exit 1; }
tail -1 gpurun_out/bench_dense.log
