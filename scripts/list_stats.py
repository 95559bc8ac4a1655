#!/usr/bin/env python3
"""Row / column length distribution of the activation lists of one bench.py configuration (after a few steps).
    python scripts/list_stats.py --nonlinearity jumprelu --decoder plt --theta 0.19 --sparse on"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv += ["--no-cpu-baseline"]
    args = bench.parse_args()
    import crosslayer_transcoder_amd as cta
    cta.set_precision(args.precision)
    dev = torch.device("cuda:0")
    mod = bench.build(args, dev, 1)
    if args.skew == "zipf":
        bench.apply_skew(mod, args)
    opt, sched = mod.attach_optimizers()
    A = bench.synth_target_map(args.layers, args.d_acts, dev)
    sparse = {"auto": None, "on": True, "off": False}[args.sparse]
    if sparse is not None:
        mod.sparse_regime = sparse
    for t in range(3):
        mod.training_step(bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=t, A=A), t)
    S = mod.last_step["lists"]
    for name, start in (("rows (token, layer)", S.row_start), ("columns (layer, feature)", S.col_start)):
        n = (start[1:].long() - start[:-1].long()).float()
        q = torch.quantile(n[torch.randperm(n.numel(), device=n.device)[:1000000]], torch.tensor([0.5, 0.9, 0.99, 0.999], device=n.device))
        print(f"{name}: count {n.numel()}, entries {int(n.sum())}, mean {float(n.mean()):.2f}, median/p90/p99/p99.9 {[int(v) for v in q.tolist()]}, "
              f"max {int(n.max())}, > 24: {int((n > 24).sum())}, > 64: {int((n > 64).sum())}, > 1024: {int((n > 1024).sum())}")


if __name__ == "__main__":
    main()
