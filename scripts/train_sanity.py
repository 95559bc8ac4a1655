#!/usr/bin/env python3
"""Longer-run sanity of the headline configuration through training_step: loss falls, no void fused attempts pile up, no
NaN, the fp16 loss scale stays put.  python scripts/train_sanity.py [bf16-mixed|16-mixed] [steps] [batchtopk|topk-aux]
(topk-aux: PerLayerTopK(16) + aux(512), a feature dead after 3 silent steps — the compacted aux pass at work)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import crosslayer_transcoder_amd as cta  # noqa: E402
from crosslayer_transcoder_amd import engine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16-mixed"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kind = sys.argv[3] if len(sys.argv) > 3 else "batchtopk"
args = argparse.Namespace(layers=12, d_acts=768, features=32768, k=16, nonlinearity="topk" if kind == "topk-aux" else "batchtopk",
                          decoder="clt", aux=kind == "topk-aux", batch=4096, dead_steps=3, shard_rehearsal=False, theta=0.03)
cta.set_precision(prec)
dev = torch.device("cuda:0")
mod = bench.build(args, dev)
mod.learning_rate = 1e-4
mod.attach_optimizers()
A = bench.synth_target_map(12, 768, dev)
batches = [bench.synth_batch(4096, 12, 768, dev, seed=i, A=A) for i in range(8)]
losses = []
for t in range(steps):
    loss = mod.training_step(batches[t % 8], t)
    if t % 25 == 0 or t == steps - 1:
        losses.append(float(loss))
        sc = getattr(mod, "loss_scaler", None)
        extra = mod.last_step["terms"]
        print(f"step {t:4d} loss {losses[-1]:.5f} dead {float(extra['dead_fraction']):.4f}" + (f" aux {float(extra['aux']):.5f}" if "aux" in extra else "")
              + f" fallbacks {engine.STATS['select_fallbacks']}/{engine.STATS['fused_select_steps']}"
              + (f" scale {sc.scale:.0f} skipped {sc.skipped_steps}" if sc is not None and prec.startswith("16") else ""), flush=True)
assert all(l == l for l in losses), "NaN loss"
assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])
print("ok")
