#!/usr/bin/env python3
"""Which decoder rows differ between two runs of the same first step (JumpReLU list regime), and how long are their columns?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "--nonlinearity", "jumprelu", "--decoder", "plt", "--theta", "0.26", "--sparse", "on", "--features", "8192"] + sys.argv[1:]
import bench  # noqa: E402
import crosslayer_transcoder_amd as cta  # noqa: E402

args = bench.parse_args()
cta.set_precision(args.precision)
dev = torch.device("cuda:0")
A = bench.synth_target_map(args.layers, args.d_acts, dev)
batch = bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=0, A=A)
outs = []
for rep in range(2):
    mod = bench.build(args, dev)
    mod.sparse_regime = True
    mod.attach_optimizers()
    mod.training_step(batch, 0)
    S = mod.last_step["lists"]
    cs = S.col_start.long()
    outs.append((mod.model.decoder.W.detach().clone(), (cs[1:] - cs[:-1]).clone()))
    torch.cuda.synchronize()
(Wa, la), (Wb, lb) = outs
assert torch.equal(la, lb)
L, F, d = Wa.shape
rows = (Wa != Wb).any(-1).reshape(-1)
lens = la.reshape(-1)
print("rows that differ:", int(rows.sum()), "of", rows.numel(), "; hot columns (> 64 entries):", int((lens > 64).sum()))
print("differing rows that are hot:", int((rows & (lens > 64)).sum()), "; hot rows that do not differ:", int((~rows & (lens > 64)).sum()))
dl = lens[rows]
print("lengths of differing rows: min", int(dl.min()) if dl.numel() else None, "max", int(dl.max()) if dl.numel() else None)
per_row = (Wa != Wb).reshape(-1, d)[rows].sum(-1)
print("differing elements per differing row: min", int(per_row.min()), "median", int(per_row.median()), "max", int(per_row.max()))
