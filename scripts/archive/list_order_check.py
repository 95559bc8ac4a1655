#!/usr/bin/env python3
"""Are the threshold lists of the same pre-activations the same bits from build to build (one-call and phased form)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crosslayer_transcoder_amd as cta  # noqa: E402
from crosslayer_transcoder_amd import ops  # noqa: E402

cta.set_precision("bf16-mixed")
dev = torch.device("cuda:0")
B, L, F = 4096, 12, 8192
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, L, F, device=dev, generator=g).bfloat16()
theta = torch.full((1, L, F), float(sys.argv[1]) if len(sys.argv) > 1 else 2.7, device=dev)
bias = torch.zeros(L, F, device=dev)
ref = None
for rep in range(6):
    split = rep % 2 == 1
    S = ops.sparse_build_threshold(x, theta, B * L * F, bias, split=split)
    if split:
        S.join()
    torch.cuda.synchronize()
    n = int(S.total)
    cur = {k: getattr(S, k)[: (n if getattr(S, k).numel() >= n and k not in ("row_start", "col_start", "fired") else None)].clone()
           for k in ("row_start", "ent_col", "ent_val", "col_start", "csc_tok", "csc_val", "csc_ent", "fired")}
    if ref is None:
        ref = cur
        cs = cur["col_start"].long()
        lens = cs[1:] - cs[:-1]
        print("entries", n, "density %.4f %%" % (100.0 * n / (B * L * F)), "longest column", int(lens.max()), "columns > 64:", int((lens > 64).sum()))
    else:
        bad = [k for k in cur if not torch.equal(cur[k], ref[k])]
        print("build", rep, "phased" if split else "one call", "differs in" if bad else "same bits", bad)
