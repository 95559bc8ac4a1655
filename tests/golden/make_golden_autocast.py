"""Golden vectors of the 16-bit modes: the REFERENCE (autograd + torch.optim.Adam) run under
torch.autocast("cpu", dtype=torch.bfloat16) and under torch.autocast("cpu", dtype=torch.float16) with
torch.amp.GradScaler("cpu") — what Lightning's `precision: "bf16-mixed"` / `"16-mixed"` (config/default.yaml:13) wrap
around training_step / backward / optimizer.step.

    python tests/golden/make_golden_autocast.py        # writes tests/golden/ac_*_{bf16,fp16}.pt

Sizes where 16-bit rounding matters (B = 256, L = 3, d_acts = 128, d_features = 512).  Inputs and initial parameters come
from tests/golden_cases.py:autocast_case_inputs (seeded; checksums stored); what is stored per step is small: the loss and
its logged terms, the active set of step 0 as a bit mask, sampled pre-activation / reconstruction entries, per-parameter
gradient norms plus 2048 sampled gradient entries, dead-feature counters, thresholds, the scaler's state.  Data only: no
reference source is copied.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402  (imports the reference through _ref_import)
from tests.golden_cases import (AC_CASES, AC_DTYPES, AC_SHAPE, autocast_case_inputs, tensor_checksum)  # noqa: E402

N_SAMPLES = 2048


_IDX = {}


def sample(key, t, g):
    """Sampled entries of t at positions drawn once per key (stored in the fixture under "sample_idx") + the full norm."""
    flat = t.detach().float().reshape(-1)
    if key not in _IDX:
        n = min(N_SAMPLES, flat.numel())
        _IDX[key] = torch.randperm(flat.numel(), generator=g)[:n].sort().values.to(torch.int32)
    return {"val": flat[_IDX[key].long()].clone(), "norm": flat.double().norm().float()}


def run(name, mode):
    dtype = AC_DTYPES[mode]
    _IDX.clear()
    cfg, opt, batches, params, la0 = autocast_case_inputs(name)
    B, L, d, F = (AC_SHAPE[k] for k in "BLdF")
    torch.manual_seed(42)
    mod = MG.build_module(cfg, L, d, F, opt)
    max_steps = 10
    mod.trainer.max_steps = max_steps
    with torch.no_grad():
        for k, p in mod.model.named_parameters():
            p.copy_(params[k])
        mod.last_active.copy_(la0)
    conf = mod.configure_optimizers()
    optimizer = conf["optimizer"] if isinstance(conf, dict) else conf
    scaler = torch.amp.GradScaler("cpu") if mode == "fp16" else None          # default init_scale 65536, as Lightning's
    gs = torch.Generator().manual_seed(99)
    out = {"mode": mode, "max_steps": max_steps, "steps": [],
           "checksums": dict({k: tensor_checksum(v) for k, v in params.items()}, batches=tensor_checksum(batches))}
    for t in range(batches.shape[0]):
        batch = batches[t]
        mod.global_step = 5 + t           # lambda(t) = lambda * global_step / max_steps > 0
        mod.logged = {}
        optimizer.zero_grad()
        with torch.autocast("cpu", dtype=dtype):
            loss = mod.training_step(batch, t)
        if scaler is not None:
            scaler.scale(loss).backward()
            scale = float(scaler.get_scale())
            scaler.unscale_(optimizer)
        else:
            loss.backward()
            scale = 1.0
        rec = {"loss": loss.detach().float().clone(), "grad_scale": torch.tensor(scale),
               "logged": {k: (v.float() if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in mod.logged.items()
                          if k.startswith("training/") or k == "metrics/dead_features"},
               "grads": {k: sample(k, p.grad, gs) for k, p in mod.model.named_parameters()},
               "grads_finite": torch.tensor(all(bool(torch.isfinite(p.grad).all()) for p in mod.model.parameters())),
               "last_active": mod.last_active.clone()}
        if t == 0:
            nlm = mod.model.nonlinearity
            keep = nlm.threshold.clone() if hasattr(nlm, "threshold") else None
            with torch.no_grad(), torch.autocast("cpu", dtype=dtype):
                pre, feat, rn, rc = mod.forward(batch[:, 0])
            if keep is not None:
                nlm.threshold = keep
            rec["dtypes"] = {"pre": str(pre.dtype), "features": str(feat.dtype), "recons_norm": str(rn.dtype)}
            rec["active_bits"] = torch.from_numpy(np.packbits((feat > 0).numpy().reshape(-1)))
            rec["n_active"] = (feat > 0).sum()
            rec["pre"] = sample("pre", pre, gs)
            rec["features"] = sample("features", feat, gs)
            rec["recons_norm"] = sample("recons_norm", rn, gs)
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
            rec["scale_after"] = torch.tensor(float(scaler.get_scale()))
        else:
            optimizer.step()
        nlm = mod.model.nonlinearity
        if hasattr(nlm, "threshold"):
            rec["threshold"] = nlm.threshold.detach().float().clone()
        out["steps"].append(rec)
    out["final_params"] = {k: sample(k, p, gs) for k, p in mod.model.named_parameters()}
    out["sample_idx"] = dict(_IDX)
    path = os.path.join(HERE, f"{name}_{mode}.pt")
    torch.save(out, path)
    print(f"{name}_{mode}: {os.path.getsize(path) / 1024:.0f} KiB, losses", [float(s["loss"]) for s in out["steps"]],
          "active", int(out["steps"][0]["n_active"]), out["steps"][0]["dtypes"])


if __name__ == "__main__":
    for name in AC_CASES:
        for mode in AC_DTYPES:
            run(name, mode)
