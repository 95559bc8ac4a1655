"""Fused training-step engine: loss and ALL parameter gradients of one step, computed by explicit calls
into the HIP kernels (no autograd graph), in the order the reference's autograd would produce them.

Mirrors, op for op, reference model/clt_lightning.py:290-312 (base), :420-475 (JumpReLU module) and
:519-567 (TopK module with auxiliary loss).  Gradient formulas are those validated in the CPU oracle
against the reference's autograd (oracle/clt_oracle.py:training_step — not imported here).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .model.clt import CrosslayerDecoder, Decoder
from .model.jumprelu import JumpReLU, ReLU
from .model.topk import BatchTopK, PerLayerBatchTopK, PerLayerTopK
from .precision import get_precision


def _op(t):
    return ops.cast_bf16(t.contiguous()) if get_precision() == "bf16" else t.contiguous()


def _select(nl, pre):
    """Training forward of a TopK-family module on dense pre-activations (updates its EMA buffer)."""
    assert nl.k <= pre.shape[-1]
    if isinstance(nl, PerLayerTopK):
        return ops.topk_rows(pre, nl.k)
    if nl.k > 0 and nl.e is None:
        raise TypeError("BatchTopK needs `e` (threshold EMA rate) for training")
    return ops.topk_segmented(pre, nl.k, isinstance(nl, PerLayerBatchTopK), nl.threshold if nl.k > 0 else None,
                              nl.e or 0.0)


class _Decoder:
    """Dense decoder passes for either decoder class, on operands in the active precision."""

    def __init__(self, dec):
        self.dec = dec
        self.clt = isinstance(dec, CrosslayerDecoder)
        self.L = dec.n_layers
        self.W32 = dec.weights()
        self.Wop = [_op(w) for w in self.W32]

    def fwd(self, fo):
        return ops.clt_decoder_fwd(fo, self.Wop, self.dec.b) if self.clt else ops.plt_decoder_fwd(fo, self.Wop[0], self.dec.b)

    def dgrad(self, go):
        return ops.clt_decoder_dgrad(go, self.Wop) if self.clt else ops.plt_decoder_dgrad(go, self.Wop[0])

    def wgrad(self, fo, go, grads: Dict[str, torch.Tensor]):
        """grads[name] (+)= wgrad; accumulates when the entry already exists."""
        if self.clt:
            for l in range(self.L):
                k = f"decoder.W_{l}"
                grads[k] = ops.clt_decoder_wgrad(fo, go, l, grads.get(k), accumulate=k in grads)
        else:
            grads["decoder.W"] = ops.plt_decoder_wgrad(fo, go, grads.get("decoder.W"), accumulate="decoder.W" in grads)

    def sqnorms(self):
        F = self.dec.d_features
        sq = torch.zeros(self.L * F, dtype=torch.float32, device=self.W32[0].device)
        for w in self.W32:                      # ascending l: the reference's accumulation order (:440-444)
            ops.row_sqnorm(w, sq, 0, True)
        return sq

    def add_norm_grads(self, dn, sq, grads):
        for l, w in enumerate(self.W32):
            ops.rowscale_add(w, dn, sq, grads[f"decoder.W_{l}" if self.clt else "decoder.W"])


def training_step(mod, batch: torch.Tensor, batch_idx: int) -> Dict[str, object]:
    """Returns {"loss": 0-dim fp32 tensor, "terms": {name: tensor}, "grads": {param_name: tensor},
    "dead": uint8 [L,F]}.  `mod` is a CrossLayerTranscoderModule (or subclass)."""
    model = mod.model
    enc, dec, nl = model.encoder, model.decoder, model.nonlinearity
    if not batch.is_cuda:
        raise RuntimeError("training_step needs a CUDA (HIP) batch; there is no CPU path")
    if batch.dtype != torch.float32:
        batch = batch.float()                   # the shm buffer hands over fp16 (datamodule contract)
    if batch_idx == 0:
        model.initialize_standardizers(batch)
    ist, ost = model.input_standardizer, model.output_standardizer
    if ist is None or ost is None:
        raise ValueError("training needs both standardizers")
    ist._require_init()
    resid, mlp_out = batch[:, 0], batch[:, 1]
    Bn, L, d = resid.shape
    F = enc.d_features
    bf16 = get_precision() == "bf16"
    xs32, xs16 = ops.standardize(resid, ist.mean, ist.std, not bf16, bf16)
    xo = xs16 if bf16 else xs32
    pre = ops.encoder_fwd(xo, _op(enc.W), enc.b)

    kind = mod._loss_kind
    theta = None
    if isinstance(nl, JumpReLU):
        theta = nl.theta
        feat = ops.jumprelu_fwd(pre, theta)
    elif isinstance(nl, ReLU):
        feat = ops.relu_fwd(pre)
    elif isinstance(nl, (PerLayerTopK, BatchTopK, PerLayerBatchTopK)):
        feat = _select(nl, pre)
    else:
        raise NotImplementedError(f"fused step: unsupported nonlinearity {type(nl).__name__}")

    D = _Decoder(dec)
    fo = _op(feat)
    r = D.fwd(fo)
    want_ys = kind == "topk"
    loss_sum, g, ys = ops.mse(r, mlp_out, ost.mean, ost.std, True, want_ys)
    N = Bn * L * d
    terms = {"mse": (loss_sum[0] / N).float()}
    grads: Dict[str, torch.Tensor] = {}

    fired = ops.column_any_positive(feat)
    dead, dead_cnt = ops.dead_update(fired, Bn, mod.last_active, mod.tokens_till_dead)
    terms["dead_fraction"] = dead_cnt[0].float() / (L * F)

    if kind == "jumprelu":
        if not isinstance(nl, (JumpReLU, ReLU)):
            raise NotImplementedError("JumpReLU loss module needs a JumpReLU/ReLU nonlinearity in the fused step")
        lam = mod._lambda * (mod.global_step / mod.trainer.max_steps)
        th = theta if theta is not None else torch.zeros(1, L, F, device=pre.device)
        bw = nl._bandwidth if theta is not None else 1.0
        sq = D.sqnorms()
        go = _op(g)
        dfeat = D.dgrad(go)
        sums, dpre, dth, dn = ops.jumprelu_loss_bwd(pre, th, sq, dfeat, bw, lam / Bn, mod.c, mod.use_tanh,
                                                    mod.pre_actv_loss / (Bn * L))
        D.wgrad(fo, go, grads)
        D.add_norm_grads(dn, sq, grads)
        grads["decoder.b"] = ops.column_sum(g)
        if theta is not None:
            grads["nonlinearity.theta"] = dth
        terms["sparsity"] = (lam * sums[0] / Bn).float()
        terms["pre_actv"] = (mod.pre_actv_loss * sums[1] / (Bn * L)).float()
        terms["weighted_features_mean"] = (sums[2] / (Bn * L * F)).float()
        terms["sparsity_penalty"] = torch.tensor(lam)
        loss = terms["mse"] + terms["sparsity"] + terms["pre_actv"]
    else:
        if kind == "topk":
            aux = mod.topk_aux
            dead_pre = ops.mask_dead(pre, dead)
            dead_feat = _select(aux, dead_pre)
            dfo = _op(dead_feat)
            aux_r = D.fwd(dfo)
            aux_sum, g_aux = ops.aux_mse(aux_r, ys, r, g, mod.aux_loss_scale)      # g += g_aux in place
            terms["aux"] = (aux_sum[0] / N).float()
            loss = terms["mse"] + mod.aux_loss_scale * terms["aux"]
            gao = _op(g_aux)
            D.wgrad(dfo, gao, grads)
            grads["decoder.b"] = ops.column_sum(g_aux)
        else:
            loss = terms["mse"]
        go = _op(g)
        D.wgrad(fo, go, grads)
        grads["decoder.b"] = ops.column_sum(g, grads.get("decoder.b"), accumulate="decoder.b" in grads)
        dfeat = D.dgrad(go)
        if theta is not None:
            dpre, grads["nonlinearity.theta"] = ops.jumprelu_bwd(pre, theta, nl._bandwidth, dfeat)
        elif isinstance(nl, ReLU):
            dpre = ops.relu_bwd(pre, dfeat)
        else:
            dpre = ops.mask_positive(feat, dfeat)
        if kind == "topk":
            ops.mask_positive(dead_feat, D.dgrad(gao), out=dpre, accumulate=True)   # aux path reaches dead entries only

    grads["encoder.W"] = ops.encoder_wgrad(xo, _op(dpre))
    grads["encoder.b"] = ops.column_sum(dpre)
    terms["loss"] = loss
    return {"loss": loss, "terms": terms, "grads": grads, "dead": dead, "features": feat, "recons_norm": r,
            "pre_actvs": pre}
