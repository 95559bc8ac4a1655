set -o pipefail
bash scripts/gpu_skew.sh && cp gpurun_out/skew.jsonl gpurun_out/final_skew.jsonl && \
STEPS=20 WARMUP=5 bash scripts/gpu_prof_cmd.sh zipf --skew zipf > /dev/null && \
STEPS=10 WARMUP=3 bash scripts/gpu_prof_cmd.sh topk --nonlinearity topk > /dev/null && \
STEPS=10 WARMUP=3 bash scripts/gpu_prof_cmd.sh plt_jumprelu --nonlinearity jumprelu --decoder plt > /dev/null && \
STEPS=8 bash scripts/gpu_prof_cmd.sh split8 --shard-rehearsal --features 4096 --k 2 > /dev/null && \
bash scripts/gpu_list_density.sh plt && bash scripts/gpu_shard_shares.sh && CLT_AB=exchange_split=0 bash scripts/gpu_shard_shares.sh
