#!/usr/bin/env python3
"""Wall time of every training step of one bench configuration (synchronised per step): where do slow steps come from?
python scripts/step_times.py <steps> <bench.py flags ...>"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench  # noqa: E402
import crosslayer_transcoder_amd as cta  # noqa: E402

args = bench.parse_args() if hasattr(bench, "parse_args") else None
assert args is not None, "bench.parse_args missing"
cta.set_precision(getattr(args, "precision", "bf16-mixed"))
dev = torch.device("cuda:0")
mod = bench.build(args, dev)
mod.attach_optimizers()
A = bench.synth_target_map(args.layers, args.d_acts, dev)
batches = [bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=i, A=A) for i in range(8)]
torch.cuda.synchronize()
for t in range(steps):
    r0 = torch.cuda.memory_reserved()
    t0 = time.perf_counter()
    mod.training_step(batches[t % 8], t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"step {t:3d} {dt:9.2f} ms  reserved {torch.cuda.memory_reserved() / 2**30:7.2f} GiB (+{(torch.cuda.memory_reserved() - r0) / 2**30:.2f})", flush=True)
