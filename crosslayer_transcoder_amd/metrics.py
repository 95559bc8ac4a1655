"""Logging side outputs of the training step, computed on the device from what the step already has.

The reference's `log_training_metrics` (model/clt_lightning.py:131-280) copies the [B, L, F] feature tensor to the
host (`features.cpu() > 0`, :139) and reduces [B, L, d] tensors with a dozen torch ops every `log_metrics_every`
steps; `DeadFeatures.update` (metrics/dead_features.py:40-42) does the same copy every step.  Here the batch-sized
reductions are two HIP passes (`clt_recon_metrics_f32`: per-column fp64 sums over the batch; per-feature fire counts
from the CSC column offsets of the activation lists, or `clt_count_positive_f32` in the dense regime) and the rest is
arithmetic on [L, d] / [L, F] sized results.  Same metric names and values as the reference.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def fire_counts(out) -> torch.Tensor:
    """int64 [L, F]: in how many tokens of the batch each feature was active (features > 0)."""
    S = out.get("lists") if hasattr(out, "get") else None
    if S is not None:                                   # CSC column offsets: entries are the positive activations
        cs = S.col_start.view(torch.int32).to(torch.int64)
        return (cs[1:] - cs[:-1]).view(S.L, S.F)
    return ops.count_positive(out["features"]).to(torch.int64)


def weight_norm_metrics(model) -> Dict[str, torch.Tensor]:
    """Mean encoder / decoder direction magnitudes (clt_lightning.py:166-176): row_sqnorm passes over the weights.
    The reference logs them before the optimizer step; with Adam fused into the backward pass call this first."""
    enc, dec = model.encoder, model.decoder
    L, F = enc.n_layers, enc.d_features
    dev = enc.W.device
    sq = torch.zeros(L * F, dtype=torch.float32, device=dev)
    ops.row_sqnorm(enc.Wt.contiguous(), sq, 0, False)
    m = {"model/encoder_direction_magnitude": sq.sqrt().mean()}
    tot, cnt = torch.zeros((), dtype=torch.float64, device=dev), 0
    for w in dec.weights():
        rows = w.numel() // w.shape[-1]
        sqd = torch.zeros(rows, dtype=torch.float32, device=dev)
        ops.row_sqnorm(w.detach(), sqd, 0, False)
        tot, cnt = tot + sqd.sqrt().double().sum(), cnt + rows
    m["model/decoder_direction_magnitude"] = tot / cnt
    return m


def training_metrics(mod, out, batch: torch.Tensor, counts: Optional[torch.Tensor] = None,
                     weight_norms: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """The batch-dependent values of log_training_metrics (clt_lightning.py:141-199) as 0-dim device tensors."""
    model = mod.model
    ost = model.output_standardizer
    r = out["recons_norm"]
    Bn, L, d = r.shape
    s = ops.recon_metrics(r, batch[:, 1], ost.mean, ost.std)            # [L, d, 6] fp64 sums over the batch
    ss_err, s_y, s_yy, s_resc, s_r, s_rr = (s[..., q] for q in range(6))
    ss_total = s_yy - s_y * s_y / Bn
    ratio = ss_err / ss_total
    m = {"metrics/fraction_of_variance_unexplained": ratio.mean()}
    fvu_l, mse_l = ratio.mean(dim=1), ss_err.sum(dim=1) / (Bn * d)
    N = Bn * L * d
    m["metrics/train_mse"] = ss_err.sum() / N
    m["metrics/train_mse_rescaled"] = s_resc.sum() / N
    tot_r, tot_rr = s_r.sum(), s_rr.sum()
    m["metrics/recons_standardized_std"] = ((tot_rr - tot_r * tot_r / N) / (N - 1)).sqrt()
    m.update(weight_norms if weight_norms is not None else weight_norm_metrics(model))
    F = model.encoder.d_features
    if counts is None:
        counts = fire_counts(out)
    per_layer = counts.sum(dim=1).double()
    m["metrics/L0_avg_per_layer"] = per_layer.sum() / (Bn * L)
    for l in range(L):
        m[f"layers/fraction_of_variance_unexplained/layer_{l}"] = fvu_l[l]
        m[f"layers/train_mse/layer_{l}"] = mse_l[l]
        m[f"layers/L0/layer_{l}"] = per_layer[l] / Bn
    return {k: v.float() for k, v in m.items()}


class DeadFeatures:
    """Device-side equivalent of the reference's torchmetrics `DeadFeatures` (metrics/dead_features.py): same
    constructor arguments, `update` / `compute` / `reset`, but `update` also accepts per-feature fire counts so the
    [B, L, F] tensor never has to exist (activation-list regime) or leave the device."""

    def __init__(self, n_features: int, n_layers: int, return_per_layer=False, return_neuron_indices=False,
                 return_log_freqs=False, device=None, **kwargs):
        self.n_features, self.n_layers = n_features, n_layers
        self.return_per_layer, self.return_neuron_indices, self.return_log_freqs = return_per_layer, return_neuron_indices, return_log_freqs
        self.n_active = torch.zeros(n_layers, n_features, dtype=torch.long, device=device)
        self.n_total = torch.zeros(1, dtype=torch.long, device=device)

    def to(self, device):
        self.n_active, self.n_total = self.n_active.to(device), self.n_total.to(device)
        return self

    def update(self, features: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None, tokens: Optional[int] = None):
        if counts is None:
            counts = ops.count_positive(features.detach()).to(torch.int64) if features.is_cuda else (features.detach() > 0).sum(dim=0)
            tokens = features.shape[0]
        if self.n_active.device != counts.device:
            self.to(counts.device)
        self.n_active += counts
        self.n_total += int(tokens)

    def compute(self):
        dead = self.n_active == 0
        ret = {"mean": dead.float().mean()}
        if self.return_neuron_indices:
            ret["layer_indices"], ret["feature_indices"] = dead.nonzero(as_tuple=True)
        if self.return_per_layer:
            ret["per_layer"] = dead.float().mean(dim=1)
        if self.return_log_freqs:
            ret["log_freqs"] = torch.clamp(torch.log10(self.n_active / self.n_total), min=-10)
        return ret

    def reset(self):
        self.n_active.zero_()
        self.n_total.zero_()
