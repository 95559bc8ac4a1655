import sys, torch, argparse
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, crosslayer_transcoder_amd as cta
for ds in (1, 3):
    args = argparse.Namespace(layers=12, d_acts=768, features=32768, k=16, nonlinearity="topk", decoder="clt", aux=True, batch=4096, dead_steps=ds, shard_rehearsal=False)
    cta.set_precision("bf16-mixed")
    mod = bench.build(args, torch.device("cuda:0"))
    opt = mod.configure_optimizers()
    batch = bench.synth_batch(4096, 12, 768, "cuda:0")
    for i in range(5):
        mod.fused_train_step(batch, i, opt)
        t = mod.last_step["terms"]
        torch.cuda.synchronize()
        print(ds, i, "dead", float(t["dead_fraction"]), "aux", float(t.get("aux", 0)), "lists", int(mod.last_step["lists"].total))
    del mod, opt
