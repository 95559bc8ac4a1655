"""CPU oracle for the crosslayer-transcoder training step.  TEST INFRASTRUCTURE ONLY.

This is a plain restatement (torch CPU tensor ops, explicit hand-derived backward passes, no
autograd, no HIP) of the arithmetic the reference performs on its hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product package
`crosslayer_transcoder_amd` never does.

Parity pin: every function here is checked in `tests/test_oracle_golden.py` against golden
vectors produced by running the *reference itself* (autograd + torch.optim.Adam) in the build
container — see `tests/golden/make_golden.py` — plus the known-answer vectors of the reference's
own `tests/test_topk.py` and the algebraic identities of `tests/test_standardization_folding.py`.

All file:line citations are relative to the reference repository root.

Conventions
-----------
params : dict keyed by the reference's state-dict names
         "encoder.W" [L,d,F], "encoder.b" [L,F],
         CLT: "decoder.W_{l}" [(l+1),F,d] for l in 0..L-1 ; PLT: "decoder.W" [L,F,d],
         "decoder.b" [L,d], JumpReLU: "nonlinearity.theta" [1,L,F]
batch  : [B, 2, L, d]  (batch[:,0] = residual-stream input, batch[:,1] = MLP output target)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor

# Optional emulation of the build's bf16 mode (bf16 GEMM operands, fp32 accumulate and fp32 results): every
# dense-contraction operand passes through _q().  None = exact fp32/fp64 restatement of the reference.
_OPERAND_DTYPE = None
_GRAD_SCALE = 1.0
_ROUND_PRE = False
_AUTOCAST = False
_ROUND_ACC = True


class operand_rounding:
    """with operand_rounding(torch.bfloat16): ...  — round GEMM operands as the 16-bit MFMA path does.
    round_acc (default on): the ENCODER product is rounded to the operand format before the fp32 bias is added — the one
    output rounding of autocast that the build reproduces (CLT_GEMM_ROUND_ACC), because the sparsifier's comparisons and
    selections are discontinuous in it; every other contraction keeps its fp32 accumulator result in the build.
    grad_scale (fp16 mode, `precision: "16-mixed"`): gradient operands are rounded while multiplied by the loss scale, as
    they are under torch.amp.GradScaler (scale the loss, backward in fp16, unscale the fp32 gradients); an overflow shows
    as inf / NaN in the gradients, which the caller treats as a skipped step.
    round_pre: the encoder's pre-activations are rounded to the operand format as well — what the reference's autocast does
    (under torch.autocast the encoder einsum, model/clt.py:157-165, RETURNS a 16-bit tensor; the sparsifier thresholds those
    values).  Off: the build's fp32-accumulator pre-activations (selection modes, fp32 list regime).
    autocast: restate torch.autocast itself rather than the build's 16-bit mode — every contraction (forward AND its
    autograd backward) also RETURNS a 16-bit tensor: the encoder result is rounded and the fp32 bias added afterwards
    (model/clt.py:157-165: `pre_actvs + self.b.to(acts_norm.dtype)` promotes back to fp32), each target layer's
    reconstruction is rounded before the bias (clt.py:337-343), each weight gradient and each per-target-layer feature
    gradient is rounded (carrying the loss scale) before it is accumulated in fp32.  Pinned by the reference's own autocast
    runs (tests/golden/ac_*_{bf16,fp16}.pt, tests/test_oracle_golden.py:test_autocast_restatement_matches_reference)."""

    def __init__(self, dtype, grad_scale: float = 1.0, round_pre: bool = False, autocast: bool = False, round_acc: bool = True):
        self.dtype, self.grad_scale, self.round_pre = dtype, float(grad_scale), bool(round_pre)
        self.autocast, self.round_acc = bool(autocast), bool(round_acc)

    def __enter__(self):
        global _OPERAND_DTYPE, _GRAD_SCALE, _ROUND_PRE, _AUTOCAST, _ROUND_ACC
        self.old = (_OPERAND_DTYPE, _GRAD_SCALE, _ROUND_PRE, _AUTOCAST, _ROUND_ACC)
        _OPERAND_DTYPE, _GRAD_SCALE, _ROUND_PRE, _AUTOCAST, _ROUND_ACC = (self.dtype, self.grad_scale, self.round_pre,
                                                                          self.autocast, self.round_acc)

    def __exit__(self, *a):
        global _OPERAND_DTYPE, _GRAD_SCALE, _ROUND_PRE, _AUTOCAST, _ROUND_ACC
        _OPERAND_DTYPE, _GRAD_SCALE, _ROUND_PRE, _AUTOCAST, _ROUND_ACC = self.old


def _q(t: Tensor) -> Tensor:
    return t if _OPERAND_DTYPE is None else t.to(_OPERAND_DTYPE).to(t.dtype)


def _qg(t: Tensor) -> Tensor:
    """A gradient as a GEMM operand: rounded while it carries the loss scale."""
    if _OPERAND_DTYPE is None:
        return t
    return (t * _GRAD_SCALE).to(_OPERAND_DTYPE).to(t.dtype) / _GRAD_SCALE


def _qo(t: Tensor) -> Tensor:
    """A forward contraction's result: 16-bit under the autocast restatement, the fp32 accumulator otherwise."""
    return _q(t) if _AUTOCAST else t


def _qgo(t: Tensor) -> Tensor:
    """A backward contraction's result (a gradient, carrying the loss scale), same rule."""
    return _qg(t) if _AUTOCAST else t


# --------------------------------------------------------------------------------------------
# A1  standardizers — model/standardize.py:39-59 (input), :80-108 (output)
# --------------------------------------------------------------------------------------------
def standardizer_stats(x: Tensor) -> Tuple[Tensor, Tensor]:
    """mean / unbiased std over the batch axis, std clamped at 1e-8 (standardize.py:44-46, 83-85)."""
    mean = x.mean(dim=0)
    std = x.std(dim=0).clamp(min=1e-8)
    return mean, std


def standardizer_init(batch: Tensor) -> Dict[str, Tensor]:
    im, isd = standardizer_stats(batch[:, 0])
    om, osd = standardizer_stats(batch[:, 1])
    return {"in_mean": im, "in_std": isd, "out_mean": om, "out_std": osd}


def standardize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """(x - mean) / std  (standardize.py:57, :106)."""
    return (x - mean) / std


def destandardize(r: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """r * std + mean  (standardize.py:96)."""
    return r * std + mean


# --------------------------------------------------------------------------------------------
# A2  encoder — model/clt.py:147-167
# --------------------------------------------------------------------------------------------
def encoder_fwd(xs: Tensor, W: Tensor, b: Tensor) -> Tensor:
    """pre[b,l,:] = xs[b,l,:] @ W[l] + b[l]."""
    B, L, d = xs.shape
    pre = torch.empty(B, L, W.shape[2], dtype=xs.dtype)
    for l in range(L):
        prod = _q(xs[:, l]) @ _q(W[l])
        pre[:, l] = (_q(prod) if (_AUTOCAST or _ROUND_ACC) else prod) + b[l]
    return _q(pre) if _ROUND_PRE else pre


def encoder_fwd_layer(x: Tensor, W: Tensor, b: Tensor, layer: int) -> Tensor:
    """Inference path of ONE layer, model/clt.py:134-145: x [batch, seq, d_acts] -> [batch, seq, d_features]."""
    prod = _q(x) @ _q(W[layer])
    return (_q(prod) if (_AUTOCAST or _ROUND_ACC) else prod) + b[layer]


def encoder_bwd(xs: Tensor, dpre: Tensor) -> Tuple[Tensor, Tensor]:
    """dW[l] = xs[:,l]^T @ dpre[:,l];  db = sum_b dpre.  (No dx: the batch does not require grad.)"""
    L = xs.shape[1]
    dW = torch.stack([_qgo(_q(xs[:, l]).T @ _qg(dpre[:, l])) for l in range(L)])
    return dW, dpre.sum(dim=0)


# --------------------------------------------------------------------------------------------
# A3 / A4  ReLU and JumpReLU — model/jumprelu.py:17-22, 31-48
# --------------------------------------------------------------------------------------------
def relu_fwd(pre: Tensor) -> Tensor:
    return torch.clamp(pre, min=0)


def relu_bwd(pre: Tensor, g: Tensor) -> Tensor:
    """aten threshold_backward: gradient passes where pre > 0."""
    return g * (pre > 0)


def jumprelu_fwd(pre: Tensor, theta: Tensor) -> Tensor:
    """f = x * [(x > theta) and (x > 0)]   (jumprelu.py:36-37)."""
    return pre * ((pre > theta) & (pre > 0))


def _heavyside(x: Tensor) -> Tensor:
    return (x > 0).to(x.dtype)


def jumprelu_bwd(pre: Tensor, theta: Tensor, bandwidth: float, g: Tensor) -> Tuple[Tensor, Tensor]:
    """dx = g with dx[x<0] = 0 (so the gradient also passes for 0 <= x <= theta);
    dtheta = sum_b -(theta/bw) * rect((x-theta)/bw) * g, rect(u) = H(u+.5) - H(u-.5)
    (jumprelu.py:9-14, 41-48; autograd sum-reduces the broadcast [B,L,F] -> [1,L,F])."""
    dx = g.clone()
    dx[pre < 0] = 0
    u = (pre - theta) / bandwidth
    rect = _heavyside(u + 0.5) - _heavyside(u - 0.5)
    dtheta = (-(theta / bandwidth) * rect * g).sum(dim=0, keepdim=True)
    return dx, dtheta


# --------------------------------------------------------------------------------------------
# A5-A7  TopK family — model/topk.py:53-63, 76-96, 109-133
# Canonical selection rule (SURVEY.md §8c rule 2): larger value first, then LOWER index.
# --------------------------------------------------------------------------------------------
def select_topk_lastdim(x: Tensor, k: int) -> Tensor:
    """Indices [.., k] of the k largest entries along the last dim, ties to the lower index."""
    if k == 0:
        return torch.empty(*x.shape[:-1], 0, dtype=torch.long)
    order = torch.sort(x, dim=-1, descending=True, stable=True).indices
    return order[..., :k]


def _scatter_relu(x: Tensor, idx: Tensor) -> Tuple[Tensor, Tensor]:
    vals = torch.clamp(torch.gather(x, -1, idx), min=0)  # relu on the selected values (topk.py:60)
    out = torch.zeros_like(x)
    out.scatter_(-1, idx, vals)
    return out, vals


def per_layer_topk_fwd(pre: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """k of F per (b,l) row (topk.py:53-63).  Returns (features, idx[B,L,k])."""
    assert k <= pre.shape[-1]
    idx = select_topk_lastdim(pre, k)
    out, _ = _scatter_relu(pre, idx)
    return out, idx


def batch_topk_fwd(pre: Tensor, k: int) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """Global k*B*L of B*L*F (topk.py:76-96).  Returns (features, flat idx, min_k or None)."""
    B, L, F = pre.shape
    assert k <= F
    bk = k * B * L
    flat = pre.reshape(-1)
    if bk == 0:
        return torch.zeros_like(pre), torch.empty(0, dtype=torch.long), None
    idx = select_topk_lastdim(flat, bk)
    out, vals = _scatter_relu(flat, idx)
    return out.reshape(B, L, F), idx, vals.min()


def per_layer_batch_topk_fwd(pre: Tensor, k: int) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """Per layer, k*B of B*F (topk.py:109-133).  idx are positions in the (batch feature) axis."""
    B, L, F = pre.shape
    assert k <= F
    bk = k * B
    x = pre.permute(1, 0, 2).reshape(L, B * F)
    if bk == 0:
        return torch.zeros_like(pre), torch.empty(L, 0, dtype=torch.long), None
    idx = select_topk_lastdim(x, bk)
    out, vals = _scatter_relu(x, idx)
    return out.reshape(L, B, F).permute(1, 0, 2).contiguous(), idx, vals.min(dim=-1).values


def threshold_ema(threshold: Tensor, min_k: Tensor, e: float) -> Tensor:
    """threshold <- (1-e)*threshold + e*min_k   (topk.py:94-95, 131-132)."""
    return (1 - e) * threshold + e * min_k


def topk_bwd(features: Tensor, g: Tensor) -> Tensor:
    """Backward of scatter(relu(gather)): gradient passes at selected entries whose value > 0,
    i.e. exactly where the output feature is > 0."""
    return g * (features > 0)


def topk_inference(pre: Tensor, threshold: float) -> Tensor:
    """topk.py:71-74 / 104-107."""
    t = max(0.0, float(threshold))
    return torch.where(pre > t, pre, torch.zeros_like(pre))


# --------------------------------------------------------------------------------------------
# A8 / A9  decoders — model/clt.py:319-344 (cross-layer), :233-249 (per-layer)
# --------------------------------------------------------------------------------------------
def crosslayer_decoder_fwd(feat: Tensor, Ws: List[Tensor], b: Tensor) -> Tensor:
    """recons[:,l] = sum_{l'<=l} feat[:,l'] @ W_l[l'] ; + b."""
    B, L, F = feat.shape
    d = b.shape[1]
    r = torch.empty(B, L, d, dtype=feat.dtype)
    for l in range(L):
        r[:, l] = _qo(_q(feat[:, : l + 1].reshape(B, -1)) @ _q(Ws[l].reshape(-1, d)))
    return r + b


def crosslayer_decoder_bwd(feat: Tensor, Ws: List[Tensor], g: Tensor) -> Tuple[Tensor, List[Tensor], Tensor]:
    """dfeat[:,l'] = sum_{l>=l'} g[:,l] @ W_l[l']^T ; dW_l[l'] = feat[:,l']^T @ g[:,l] ; db = sum_b g."""
    B, L, F = feat.shape
    dfeat = torch.zeros_like(feat)
    dWs = []
    for l in range(L):
        dW = torch.empty_like(Ws[l])
        for lp in range(l + 1):
            dfeat[:, lp] += _qgo(_qg(g[:, l]) @ _q(Ws[l][lp]).T)
            dW[lp] = _qgo(_q(feat[:, lp]).T @ _qg(g[:, l]))
        dWs.append(dW)
    return dfeat, dWs, g.sum(dim=0)


def plt_decoder_fwd(feat: Tensor, W: Tensor, b: Tensor) -> Tensor:
    L = feat.shape[1]
    return torch.stack([_qo(_q(feat[:, l]) @ _q(W[l])) for l in range(L)], dim=1) + b


def plt_decoder_fwd_layer(feat: Tensor, W: Tensor, b: Tensor, layer: int) -> Tensor:
    """Inference path, model/clt.py:216-231: feat [batch, seq, F] (or [batch, seq, L, F], of which layer `layer` is
    taken) -> [batch, seq, d_acts]."""
    if feat.ndim == 4:
        feat = feat[:, :, layer, :]
    return _q(feat) @ _q(W[layer]) + b[layer]


def crosslayer_decoder_fwd_layer(feat: Tensor, Ws: List[Tensor], b: Tensor, layer: int) -> Tensor:
    """Inference path, model/clt.py:304-317: feat [batch, seq, from_layer = layer + 1, F] -> [batch, seq, d_acts]
    = sum over the source layers l' <= layer of feat[:, :, l'] @ W_layer[l'], + b[layer]."""
    d = b.shape[1]
    return _q(feat.flatten(-2)) @ _q(Ws[layer].reshape(-1, d)) + b[layer]


def plt_decoder_bwd(feat: Tensor, W: Tensor, g: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    L = feat.shape[1]
    dfeat = torch.stack([_qgo(_qg(g[:, l]) @ _q(W[l]).T) for l in range(L)], dim=1)
    dW = torch.stack([_qgo(_q(feat[:, l]).T @ _qg(g[:, l])) for l in range(L)])
    return dfeat, dW, g.sum(dim=0)


def decoder_weights(params: Dict[str, Tensor], kind: str, L: int):
    return [params[f"decoder.W_{l}"] for l in range(L)] if kind == "clt" else params["decoder.W"]


def decoder_fwd(feat, params, kind):
    L = feat.shape[1]
    W = decoder_weights(params, kind, L)
    b = params["decoder.b"]
    return crosslayer_decoder_fwd(feat, W, b) if kind == "clt" else plt_decoder_fwd(feat, W, b)


def decoder_bwd(feat, params, kind, g) -> Tuple[Tensor, Dict[str, Tensor]]:
    L = feat.shape[1]
    W = decoder_weights(params, kind, L)
    if kind == "clt":
        dfeat, dWs, db = crosslayer_decoder_bwd(feat, W, g)
        grads = {f"decoder.W_{l}": dWs[l] for l in range(L)}
    else:
        dfeat, dW, db = plt_decoder_bwd(feat, W, g)
        grads = {"decoder.W": dW}
    grads["decoder.b"] = db
    return dfeat, grads


def decoder_norms(params, kind: str, L: int) -> Tensor:
    """dec_norms[l',f] = sqrt(sum_{l>=l'} sum_d W_l[l',f,d]^2)  (clt_lightning.py:438-448). -> [1,L,F]"""
    if kind == "clt":
        F = params["decoder.W_0"].shape[1]
        sq = torch.zeros(L, F, dtype=params["decoder.W_0"].dtype)
        for l in range(L):
            sq[: l + 1] = sq[: l + 1] + (params[f"decoder.W_{l}"] ** 2).sum(dim=-1)
    else:
        sq = (params["decoder.W"] ** 2).sum(dim=-1)
    return sq.sqrt().unsqueeze(0)


# --------------------------------------------------------------------------------------------
# A14  dead-feature bookkeeping — model/clt_lightning.py:282-288
# --------------------------------------------------------------------------------------------
def update_dead_features(last_active: Tensor, feat: Tensor, tokens_till_dead: int) -> Tuple[Tensor, Tensor]:
    la = last_active.clone()
    la[feat.sum(dim=0) > 0.0] = 0
    la += feat.shape[0]
    return la, la > tokens_till_dead


# --------------------------------------------------------------------------------------------
# nonlinearity dispatch
# --------------------------------------------------------------------------------------------
def nonlinearity_fwd(pre: Tensor, nl: dict, params, buffers: dict):
    """nl = {"kind": relu|jumprelu|topk|batchtopk|perlayerbatchtopk, "k":, "e":, "bandwidth":}
    Returns (features, new_buffers)."""
    kind = nl["kind"]
    nb = dict(buffers)
    if kind == "relu":
        return relu_fwd(pre), nb
    if kind == "jumprelu":
        return jumprelu_fwd(pre, params["nonlinearity.theta"]), nb
    if kind == "topk":
        return per_layer_topk_fwd(pre, nl["k"])[0], nb
    if kind == "batchtopk":
        f, _, mk = batch_topk_fwd(pre, nl["k"])
        if mk is not None:
            nb["threshold"] = threshold_ema(buffers["threshold"], mk, nl["e"])
        return f, nb
    if kind == "perlayerbatchtopk":
        f, _, mk = per_layer_batch_topk_fwd(pre, nl["k"])
        if mk is not None:
            nb["threshold"] = threshold_ema(buffers["threshold"], mk, nl["e"])
        return f, nb
    if kind == "identity":
        return pre, nb
    raise ValueError(kind)


def nonlinearity_bwd(pre, feat, nl, params, dfeat) -> Tuple[Tensor, Dict[str, Tensor]]:
    kind = nl["kind"]
    if kind == "relu":
        return relu_bwd(pre, dfeat), {}
    if kind == "jumprelu":
        dx, dth = jumprelu_bwd(pre, params["nonlinearity.theta"], nl["bandwidth"], dfeat)
        return dx, {"nonlinearity.theta": dth}
    if kind in ("topk", "batchtopk", "perlayerbatchtopk"):
        return topk_bwd(feat, dfeat), {}
    if kind == "identity":
        return dfeat, {}
    raise ValueError(kind)


# --------------------------------------------------------------------------------------------
# A10  forward composition — model/clt.py:403-425
# --------------------------------------------------------------------------------------------
def transcoder_fwd(resid, params, std, nl, dec_kind, buffers):
    xs = standardize(resid, std["in_mean"], std["in_std"])
    pre = encoder_fwd(xs, params["encoder.W"], params["encoder.b"])
    feat, nb = nonlinearity_fwd(pre, nl, params, buffers)
    r = decoder_fwd(feat, params, dec_kind)
    recons = destandardize(r, std["out_mean"], std["out_std"])
    return xs, pre, feat, r, recons, nb


def _add_grads(dst: Dict[str, Tensor], src: Dict[str, Tensor]):
    for k, v in src.items():
        dst[k] = dst[k] + v if k in dst else v


# --------------------------------------------------------------------------------------------
# A11-A13  training steps (loss + all parameter gradients)
# --------------------------------------------------------------------------------------------
def training_step(batch: Tensor, params: Dict[str, Tensor], std: Dict[str, Tensor], cfg: dict,
                  buffers: dict) -> dict:
    """One loss/gradient evaluation.

    cfg = {"decoder": "clt"|"plt", "nonlinearity": {...},
           "loss": {"kind": "base"|"jumprelu"|"topk", ...}}
      jumprelu: lambda_sparsity, c_sparsity, use_tanh, pre_actv_loss, step_frac (= global_step/max_steps)
      topk    : topk_aux {nonlinearity dict}, aux_loss_scale, tokens_till_dead
    buffers = {"last_active": int64 [L,F], "threshold": ..., "aux_threshold": ...}
    Returns {"loss", "terms", "grads", "buffers", "pre", "features", "recons_norm", "recons"}.
    """
    nl, dk, lc = cfg["nonlinearity"], cfg["decoder"], cfg["loss"]
    resid, mlp_out = batch[:, 0], batch[:, 1]
    B, L, d = resid.shape
    xs, pre, feat, r, recons, nb = transcoder_fwd(resid, params, std, nl, dk, buffers)
    y = standardize(mlp_out, std["out_mean"], std["out_std"])
    N = B * L * d
    diff = r - y
    mse = (diff * diff).mean()                                   # clt_lightning.py:303-304
    terms = {"mse": mse}
    grads: Dict[str, Tensor] = {}
    ttd = lc.get("tokens_till_dead", 1_000_000)
    la, dead = update_dead_features(buffers["last_active"], feat, ttd)
    nb["last_active"] = la
    terms["dead_fraction"] = dead.to(pre.dtype).mean()

    g = 2.0 * diff / N                                           # dL/drecons_norm of the MSE term
    dpre_extra = None
    dfeat_extra = None

    if lc["kind"] == "base":
        loss = mse
    elif lc["kind"] == "jumprelu":                               # clt_lightning.py:420-475
        lam = lc["lambda_sparsity"] * lc["step_frac"]
        c = lc["c_sparsity"]
        n = decoder_norms(params, dk, L)                         # [1,L,F]
        w = feat * n * c
        t = torch.tanh(w) if lc["use_tanh"] else w
        sparsity = lam * t.sum(dim=[1, 2]).mean()
        pal = torch.clamp(-pre, min=0).sum(dim=-1).mean() * lc["pre_actv_loss"]
        loss = mse + sparsity + pal
        terms.update(sparsity=sparsity, pre_actv=pal, weighted_features_mean=w.mean())
        dt = (1 - t * t) if lc["use_tanh"] else torch.ones_like(t)
        dfeat_extra = (lam / B) * dt * c * n
        dn = (lam / B) * (dt * c * feat).sum(dim=0)              # [L,F]
        coef = dn / n[0]
        if dk == "clt":
            for l in range(L):
                grads[f"decoder.W_{l}"] = coef[: l + 1].unsqueeze(-1) * params[f"decoder.W_{l}"]
        else:
            grads["decoder.W"] = coef.unsqueeze(-1) * params["decoder.W"]
        dpre_extra = (-lc["pre_actv_loss"] / (B * L)) * (pre < 0).to(pre.dtype)
    elif lc["kind"] == "topk":                                   # clt_lightning.py:519-567
        aux_nl = lc["topk_aux"]
        dead_pre = torch.where(dead.unsqueeze(0).expand_as(pre), pre, torch.full_like(pre, -math.inf))
        aux_buf = {"threshold": buffers.get("aux_threshold")}
        dead_feat, aux_nb = nonlinearity_fwd(dead_pre, aux_nl, params, aux_buf)
        if aux_nb.get("threshold") is not None:
            nb["aux_threshold"] = aux_nb["threshold"]
        aux_recons = decoder_fwd(dead_feat, params, dk)
        e = aux_recons - (y - r)
        aux_raw = e * e
        finite = torch.isfinite(aux_raw)
        aux = torch.nan_to_num(aux_raw, 0.0)
        s = lc["aux_loss_scale"]
        loss = mse + s * aux.mean()
        terms["aux"] = aux.mean()
        g_aux = (s * 2.0 / N) * e * finite
        g = g + g_aux                                            # recons_err is not detached (:548)
        dfa, ga = decoder_bwd(dead_feat, params, dk, g_aux)
        _add_grads(grads, ga)
        dpa, _ = nonlinearity_bwd(dead_pre, dead_feat, aux_nl, params, dfa)
        dpre_extra = torch.where(dead.unsqueeze(0).expand_as(pre), dpa, torch.zeros_like(dpa))
    else:
        raise ValueError(lc["kind"])

    dfeat, gd = decoder_bwd(feat, params, dk, g)
    _add_grads(grads, gd)
    if dfeat_extra is not None:
        dfeat = dfeat + dfeat_extra
    dpre, gn = nonlinearity_bwd(pre, feat, nl, params, dfeat)
    _add_grads(grads, gn)
    if dpre_extra is not None:
        dpre = dpre + dpre_extra
    dW, db = encoder_bwd(xs, dpre)
    grads["encoder.W"] = dW
    grads["encoder.b"] = db
    terms["loss"] = loss
    return {"loss": loss, "terms": terms, "grads": grads, "buffers": nb, "pre": pre, "features": feat,
            "recons_norm": r, "recons": recons}


# --------------------------------------------------------------------------------------------
# A15  Adam / AdamW + LR schedules — clt_lightning.py:333-392 (torch.optim semantics, pinned torch 2.8)
# --------------------------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0, adamw: bool = False):
    """One torch.optim.Adam / AdamW update (step counts from 1).  Returns (p, m, v)."""
    if weight_decay != 0.0:
        if adamw:
            p = p * (1.0 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m = m + (g - m) * (1.0 - beta1)
    v = v * beta2 + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def lr_at(step: int, base_lr: float, warmup_steps: int = 0, lr_decay_step: int = 0,
          lr_decay_factor: float = 1.0) -> float:
    """Learning rate used by optimizer step number `step` (0-based count of completed steps).
    LinearLR(start_factor=1e-7, end_factor=1, total_iters=warmup) takes precedence over
    MultiStepLR([lr_decay_step], gamma) (clt_lightning.py:357-389)."""
    if warmup_steps > 0:
        t = min(step, warmup_steps)
        return base_lr * (1e-7 + (1.0 - 1e-7) * t / warmup_steps)
    if lr_decay_step > 0:
        return base_lr * (lr_decay_factor if step >= lr_decay_step else 1.0)
    return base_lr


def fit(batches: List[Tensor], params: Dict[str, Tensor], cfg: dict, buffers: dict, opt: dict,
        max_steps: int) -> dict:
    """The Lightning automatic-optimisation loop restated: for step t, training_step(batch, t)
    (standardizers initialised when t == 0), backward, optimizer.step(), scheduler.step()."""
    params = {k: v.clone() for k, v in params.items()}
    buffers = dict(buffers)
    mstate = {k: torch.zeros_like(v) for k, v in params.items()}
    vstate = {k: torch.zeros_like(v) for k, v in params.items()}
    std = None
    losses, term_log = [], []
    for t, batch in enumerate(batches):
        if t == 0:
            std = standardizer_init(batch)
        c = dict(cfg)
        c["loss"] = dict(cfg["loss"])
        c["loss"]["step_frac"] = t / max_steps
        out = training_step(batch, params, std, c, buffers)
        buffers = out["buffers"]
        lr = lr_at(t, opt["lr"], opt.get("warmup_steps", 0), opt.get("lr_decay_step", 0),
                   opt.get("lr_decay_factor", 1.0))
        adamw = opt.get("optimizer", "adam") == "adamw"
        for k in params:
            params[k], mstate[k], vstate[k] = adam_step(
                params[k], out["grads"][k], mstate[k], vstate[k], t + 1, lr,
                opt.get("beta1", 0.9), opt.get("beta2", 0.999), 1e-8,
                0.01 if adamw else 0.0, adamw)
        losses.append(out["loss"])
        term_log.append(out["terms"])
    return {"params": params, "buffers": buffers, "std": std, "losses": losses, "terms": term_log,
            "m": mstate, "v": vstate}


# --------------------------------------------------------------------------------------------
# standardizer folding — model/clt.py:169-185, 251-266, 346-360
# --------------------------------------------------------------------------------------------
def fold_encoder(W: Tensor, b: Tensor, mean: Tensor, std: Tensor) -> Tuple[Tensor, Tensor]:
    Wf = W / std.unsqueeze(-1)
    bf = b - torch.einsum("ld,ldf->lf", mean, Wf)
    return Wf, bf


def fold_decoder_plt(W: Tensor, b: Tensor, mean: Tensor, std: Tensor) -> Tuple[Tensor, Tensor]:
    return W * std.unsqueeze(1), b * std + mean


def fold_decoder_clt(Ws: List[Tensor], b: Tensor, mean: Tensor, std: Tensor) -> Tuple[List[Tensor], Tensor]:
    return [Ws[l] * std[l] for l in range(len(Ws))], b * std + mean


# ------------------------------------------------------------------------------------------------
# Logging side outputs: clt_lightning.py:131-199 (log_training_metrics), metrics/dead_features.py:40-60
# ------------------------------------------------------------------------------------------------
def training_metrics(features: Tensor, recons_norm: Tensor, mlp_out: Tensor, std: Dict[str, Tensor],
                     params: Dict[str, Tensor], dec_kind: str) -> Dict[str, Tensor]:
    """Every batch-dependent value log_training_metrics logs, under the reference's names (tokens_seen, the learning
    rate and the W&B tables are host bookkeeping, not restated)."""
    L = recons_norm.shape[1]
    yn = standardize(mlp_out, std["out_mean"], std["out_std"])                  # :135
    recons = destandardize(recons_norm, std["out_mean"], std["out_std"])
    mse = (recons_norm - yn) ** 2                                               # :142
    ss_err = ((yn - recons_norm) ** 2).sum(dim=0)                               # :144-145
    ss_total = ((yn - yn.mean(dim=0, keepdim=True)) ** 2).sum(dim=0)            # :146-148
    out = {"metrics/fraction_of_variance_unexplained": (ss_err / ss_total).mean()}      # :149
    fvu_l = (ss_err / ss_total).mean(dim=-1)                                    # :151
    for l in range(L):
        out[f"layers/fraction_of_variance_unexplained/layer_{l}"] = fvu_l[l]
    out["model/encoder_direction_magnitude"] = params["encoder.W"].norm(p=2, dim=1).mean()           # :166-167
    if dec_kind == "clt":                                                        # :168-176
        lens = torch.cat([params[f"decoder.W_{l}"].norm(p=2, dim=-1).flatten() for l in range(L)])
    else:
        lens = params["decoder.W"].norm(p=2, dim=-1).flatten()
    out["model/decoder_direction_magnitude"] = lens.mean()
    mse_l = mse.mean(dim=(0, 2))                                                # :179
    for l in range(L):
        out[f"layers/train_mse/layer_{l}"] = mse_l[l]
    out["metrics/train_mse"] = mse.mean()                                       # :184
    out["metrics/train_mse_rescaled"] = ((recons - mlp_out) ** 2).mean()        # :185
    out["metrics/recons_standardized_std"] = recons_norm.std()                  # :189
    active = features > 0                                                       # :139
    out["metrics/L0_avg_per_layer"] = torch.count_nonzero(active) / (features.shape[0] * features.shape[1])   # :190-194
    l0 = torch.count_nonzero(active, dim=(0, 2)) / features.shape[0]            # :197
    for l in range(L):
        out[f"layers/L0/layer_{l}"] = l0[l]
    return out


def dead_features_update(n_active: Tensor, n_total: Tensor, features: Tensor) -> Tuple[Tensor, Tensor]:
    """DeadFeatures.update (metrics/dead_features.py:40-42)."""
    return n_active + (features > 0.0).sum(dim=0), n_total + features.shape[0]


def dead_features_compute(n_active: Tensor, n_total: Tensor) -> Dict[str, Tensor]:
    """DeadFeatures.compute with every return_* flag on (metrics/dead_features.py:44-60)."""
    dead = n_active == 0
    li, fi = dead.nonzero(as_tuple=True)
    return {"mean": dead.float().mean(), "layer_indices": li, "feature_indices": fi, "per_layer": dead.float().mean(dim=1),
            "log_freqs": torch.clamp(torch.log10(n_active / n_total), min=-10)}
