"""Where does a logging step (batch_idx % log_metrics_every == 0) spend its time?  (headline configuration)"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import crosslayer_transcoder_amd as cta
from crosslayer_transcoder_amd import metrics as M

args = argparse.Namespace(layers=12, d_acts=768, features=32768, k=16, nonlinearity="batchtopk", decoder="clt", aux=False,
                          batch=4096, dead_steps=1, shard_rehearsal=False)
dev = torch.device("cuda", 0)
cta.set_precision("bf16-mixed")
mod = bench.build(args, dev)
opt = mod.configure_optimizers()
batch = bench.synth_batch(args.batch, args.layers, args.d_acts, dev)


def timed(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
    return r

for i in range(4):
    mod.fused_train_step(batch, i, opt)
timed("plain step", lambda: mod.fused_train_step(batch, 7, opt))
timed("logging step", lambda: mod.fused_train_step(batch, 50, opt))
timed("logging step (again)", lambda: mod.fused_train_step(batch, 100, opt))
out = mod.last_step
norms = timed("weight_norm_metrics", lambda: M.weight_norm_metrics(mod.model))
counts = timed("fire_counts", lambda: M.fire_counts(out))
m = timed("training_metrics", lambda: M.training_metrics(mod, out, batch, counts, norms))
timed("log() of every value", lambda: [mod.log(k, v) for k, v in m.items()])
if mod.dead_features is not None:
    timed("dead_features.update", lambda: mod.dead_features.update(counts=counts, tokens=batch.shape[0]))
    timed("dead_features.compute", lambda: mod.dead_features.compute())
