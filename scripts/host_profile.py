#!/usr/bin/env python3
"""Host-side cost of one step at the per-rank size of an 8-way feature shard (F=4096/layer, k=2): cProfile of 30 steps."""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, crosslayer_transcoder_amd as cta
args = argparse.Namespace(layers=12, d_acts=768, features=4096, k=2, nonlinearity="batchtopk", decoder="clt", aux=False, batch=4096,
                          dead_steps=1, shard_rehearsal=False)
cta.set_precision("bf16-mixed")
dev = torch.device("cuda:0")
mod = bench.build(args, dev)
opt = mod.configure_optimizers()
batch = bench.synth_batch(4096, 12, 768, dev)
for i in range(5):
    mod.fused_train_step(batch, i, opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(30):
    mod.fused_train_step(batch, i + 5, opt)
pr.disable()
torch.cuda.synchronize()
print("ms/step wall", (time.perf_counter() - t0) / 30 * 1e3)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
