"""Per-step wall time of the headline configuration over many steps: does the fused selection keep succeeding?"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import crosslayer_transcoder_amd as cta

args = argparse.Namespace(layers=12, d_acts=768, features=32768, k=16, nonlinearity="batchtopk", decoder="clt", aux=False,
                          batch=4096, dead_steps=1, shard_rehearsal=False)
dev = torch.device("cuda", 0)
cta.set_precision("bf16-mixed")
mod = bench.build(args, dev)
opt = mod.configure_optimizers()
batch = bench.synth_batch(args.batch, args.layers, args.d_acts, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = mod.fused_train_step(batch, i, opt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    out = mod.last_step
    S = out.get("lists")
    fused = "pre_actvs" not in dict.keys(out)
    cs = S.col_start.view(torch.int32)
    colmax = int((cs[1:] - cs[:-1]).max())
    print(f"step {i:3d} {dt:7.2f} ms fused={int(fused)} loss={float(loss):.4f} total={int(S.total)} max_col={colmax} "
          f"thr={float(mod.model.nonlinearity.threshold):.4f}", flush=True)
