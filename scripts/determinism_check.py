#!/usr/bin/env python3
"""Run-to-run reproducibility of a bench configuration inside one process: the module is built twice from the same seed and stepped
over the same batches; after every step the loss, every gradient the step hands out and every parameter are compared bit for bit.
python scripts/determinism_check.py <steps> <bench.py flags ...>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench  # noqa: E402
import crosslayer_transcoder_amd as cta  # noqa: E402

args = bench.parse_args()
cta.set_precision(args.precision)
dev = torch.device("cuda:0")
A = bench.synth_target_map(args.layers, args.d_acts, dev)
batches = [bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=i, A=A) for i in range(4)]
sparse = {"auto": None, "on": True, "off": False}[args.sparse]


def run():
    mod = bench.build(args, dev)
    if sparse is not None:
        mod.sparse_regime = sparse
    mod.attach_optimizers()
    trace = []
    for t in range(steps):
        loss = mod.training_step(batches[t % 4], t)
        rec = {"loss": loss.detach().clone()}
        for k, v in mod.last_step["grads"].items():
            rec["grad " + k] = v.detach().clone()
        for k, p in mod.model.named_parameters():
            rec["param " + k] = p.detach().clone()
        trace.append(rec)
    torch.cuda.synchronize()
    return trace


a, b = run(), run()
bad = 0
for t, (ra, rb) in enumerate(zip(a, b)):
    for k in ra:
        if not torch.equal(ra[k], rb[k]):
            d = (ra[k].double() - rb[k].double()).abs()
            print(f"step {t}: {k} differs in {int((ra[k] != rb[k]).sum())} of {ra[k].numel()} elements, max |diff| {float(d.max()):.3g}")
            bad += 1
print("reproducible" if not bad else f"{bad} differences")
