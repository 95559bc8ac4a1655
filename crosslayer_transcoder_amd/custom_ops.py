"""torch.library registration of the model's forward-path operators (namespace `clt_hip`): the "thin torch custom-op
layer" over the C ABI.  Every operator `CrossLayerTranscoder.forward` runs (reference model/clt.py:403-425 — standardize,
encoder, sparsifier, decoder, de-standardize) is a dispatcher op with a fake-tensor (shape) rule and an autograd rule whose
backward is again made of registered ops, so the module is traceable by `torch.compile` / `torch.export` as a graph of
opaque HIP calls (the reference's `compile: True` flag, config/default.yaml:89) and visible to the dispatcher like any
ATen op.  The bodies only cast operands for the active precision and call `ops` (ctypes -> libclt_hip.so); nothing here
computes in torch.  The fused training step (`engine.training_step`) does not go through these: it calls `ops` directly."""
from typing import List, Optional

import torch
from torch import Tensor

from . import ops
from .precision import get_precision

_DEV = "cuda"


def _operand(t: Tensor) -> Tensor:
    """fp32 tensor -> GEMM operand in the active precision."""
    t = t.contiguous()
    if get_precision() == "32":
        return t
    ops.sync_operand_format()                   # (fp16 / bf16 operands as the active precision says)
    return ops.cast_bf16(t)


def _feature_major(W: Tensor) -> Tensor:
    """[L,d,F] encoder weight -> its [L,F,d] storage view (a free permute when the layout is the native one)."""
    Wt = W.permute(0, 2, 1)
    return Wt if Wt.is_contiguous() else Wt.contiguous()


def _none():
    return torch.empty(0, device=_DEV)


# ---- encoder (reference model/clt.py:157-165) ------------------------------------------------------------------------
@torch.library.custom_op("clt_hip::encoder", mutates_args=(), device_types=_DEV)
def encoder(xs: Tensor, W: Tensor, b: Optional[Tensor]) -> Tensor:
    return ops.encoder_fwd(_operand(xs), _operand(_feature_major(W)), b)


@encoder.register_fake
def _(xs, W, b):
    return xs.new_empty(xs.shape[0], xs.shape[1], W.shape[2], dtype=torch.float32)


@torch.library.custom_op("clt_hip::encoder_bwd", mutates_args=(), device_types=_DEV)
def encoder_bwd(dpre: Tensor, xs: Tensor, W: Tensor, need_dx: bool) -> List[Tensor]:
    """-> [dx (empty unless need_dx), dW [L,d,F], db [L,F]]"""
    xo, Wo, do = _operand(xs), _operand(_feature_major(W)), _operand(dpre)
    dW = ops.encoder_wgrad(xo, do).permute(0, 2, 1).contiguous()
    db = ops.column_sum(dpre.contiguous())
    dx = ops.encoder_dgrad(do, Wo) if need_dx else _none()
    return [dx, dW, db]


@encoder_bwd.register_fake
def _(dpre, xs, W, need_dx):
    return [torch.empty_like(xs) if need_dx else xs.new_empty(0), W.new_empty(W.shape), dpre.new_empty(dpre.shape[1:])]


def _encoder_setup(ctx, inputs, output):
    xs, W, b = inputs
    ctx.save_for_backward(xs, W)
    ctx.has_bias = b is not None


def _encoder_backward(ctx, dpre):
    xs, W = ctx.saved_tensors
    dx, dW, db = torch.ops.clt_hip.encoder_bwd(dpre, xs, W, ctx.needs_input_grad[0])
    return (dx if ctx.needs_input_grad[0] else None, dW if ctx.needs_input_grad[1] else None,
            db if ctx.has_bias and ctx.needs_input_grad[2] else None)


encoder.register_autograd(_encoder_backward, setup_context=_encoder_setup)


# ---- per-layer decoder (model/clt.py:241-248) ---------------------------------------------------------------------------
@torch.library.custom_op("clt_hip::plt_decoder", mutates_args=(), device_types=_DEV)
def plt_decoder(feat: Tensor, W: Tensor, b: Tensor) -> Tensor:
    return ops.plt_decoder_fwd(_operand(feat), _operand(W), b)


@plt_decoder.register_fake
def _(feat, W, b):
    return feat.new_empty(feat.shape[0], feat.shape[1], W.shape[2], dtype=torch.float32)


@torch.library.custom_op("clt_hip::plt_decoder_bwd", mutates_args=(), device_types=_DEV)
def plt_decoder_bwd(g: Tensor, feat: Tensor, W: Tensor) -> List[Tensor]:
    """-> [dfeat [B,L,F], dW [L,F,d], db [L,d]]"""
    go = _operand(g)
    return [ops.plt_decoder_dgrad(go, _operand(W)), ops.plt_decoder_wgrad(_operand(feat), go), ops.column_sum(g.contiguous())]


@plt_decoder_bwd.register_fake
def _(g, feat, W):
    return [torch.empty_like(feat), torch.empty_like(W), g.new_empty(g.shape[1:])]


def _plt_setup(ctx, inputs, output):
    feat, W, b = inputs
    ctx.save_for_backward(feat, W)


def _plt_backward(ctx, g):
    feat, W = ctx.saved_tensors
    df, dW, db = torch.ops.clt_hip.plt_decoder_bwd(g, feat, W)
    need = ctx.needs_input_grad
    return df if need[0] else None, dW if need[1] else None, db if need[2] else None


plt_decoder.register_autograd(_plt_backward, setup_context=_plt_setup)


# ---- cross-layer decoder (model/clt.py:334-343) -------------------------------------------------------------------------
@torch.library.custom_op("clt_hip::clt_decoder", mutates_args=(), device_types=_DEV)
def clt_decoder(feat: Tensor, b: Tensor, Ws: List[Tensor]) -> Tensor:
    return ops.clt_decoder_fwd(_operand(feat), [_operand(w) for w in Ws], b)


@clt_decoder.register_fake
def _(feat, b, Ws):
    return feat.new_empty(feat.shape[0], feat.shape[1], Ws[0].shape[2], dtype=torch.float32)


@torch.library.custom_op("clt_hip::clt_decoder_bwd", mutates_args=(), device_types=_DEV)
def clt_decoder_bwd(g: Tensor, feat: Tensor, Ws: List[Tensor]) -> List[Tensor]:
    """-> [dfeat, db, dW_0, ..., dW_{L-1}]"""
    go, fo = _operand(g), _operand(feat)
    Wo = [_operand(w) for w in Ws]
    return [ops.clt_decoder_dgrad(go, Wo), ops.column_sum(g.contiguous())] + [ops.clt_decoder_wgrad(fo, go, l) for l in range(len(Wo))]


@clt_decoder_bwd.register_fake
def _(g, feat, Ws):
    return [torch.empty_like(feat), g.new_empty(g.shape[1:])] + [torch.empty_like(w) for w in Ws]


def _clt_setup(ctx, inputs, output):
    feat, b, Ws = inputs
    ctx.save_for_backward(feat, *Ws)


def _clt_backward(ctx, g):
    feat, *Ws = ctx.saved_tensors
    out = torch.ops.clt_hip.clt_decoder_bwd(g, feat, Ws)
    need = ctx.needs_input_grad
    return out[0] if need[0] else None, out[1] if need[1] else None, list(out[2:])


clt_decoder.register_autograd(_clt_backward, setup_context=_clt_setup)


# ---- ReLU / JumpReLU with the rectangle-window STE (model/jumprelu.py:17-73) --------------------------------------------
@torch.library.custom_op("clt_hip::relu", mutates_args=(), device_types=_DEV)
def relu(x: Tensor) -> Tensor:
    return ops.relu_fwd(x.contiguous())


@relu.register_fake
def _(x):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op("clt_hip::relu_bwd", mutates_args=(), device_types=_DEV)
def relu_bwd(x: Tensor, g: Tensor) -> Tensor:
    return ops.relu_bwd(x.contiguous(), g.contiguous())


@relu_bwd.register_fake
def _(x, g):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


relu.register_autograd(lambda ctx, g: torch.ops.clt_hip.relu_bwd(ctx.saved_tensors[0], g),
                       setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


@torch.library.custom_op("clt_hip::jumprelu", mutates_args=(), device_types=_DEV)
def jumprelu(x: Tensor, theta: Tensor, bandwidth: float) -> Tensor:
    return ops.jumprelu_fwd(x.contiguous(), theta)


@jumprelu.register_fake
def _(x, theta, bandwidth):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op("clt_hip::jumprelu_bwd", mutates_args=(), device_types=_DEV)
def jumprelu_bwd(x: Tensor, theta: Tensor, bandwidth: float, g: Tensor) -> List[Tensor]:
    dx, dth = ops.jumprelu_bwd(x.contiguous(), theta, bandwidth, g.contiguous())
    return [dx, dth.reshape(theta.shape)]


@jumprelu_bwd.register_fake
def _(x, theta, bandwidth, g):
    return [torch.empty_like(x, memory_format=torch.contiguous_format), torch.empty_like(theta)]


def _jump_setup(ctx, inputs, output):
    x, theta, bandwidth = inputs
    ctx.save_for_backward(x, theta)
    ctx.bandwidth = bandwidth


def _jump_backward(ctx, g):
    x, theta = ctx.saved_tensors
    dx, dth = torch.ops.clt_hip.jumprelu_bwd(x, theta, ctx.bandwidth, g)
    return dx, dth, None


jumprelu.register_autograd(_jump_backward, setup_context=_jump_setup)


# ---- TopK family (model/topk.py:48-133): dense features from the select kernels; backward = g * [features > 0] ----------
@torch.library.custom_op("clt_hip::topk_rows", mutates_args=(), device_types=_DEV)
def topk_rows(x: Tensor, k: int) -> Tensor:
    return ops.topk_rows(x.contiguous(), k)


@topk_rows.register_fake
def _(x, k):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op("clt_hip::batch_topk", mutates_args=(), device_types=_DEV)
def batch_topk(x: Tensor, k: int, per_layer: bool, threshold: Optional[Tensor], e: float) -> List[Tensor]:
    """BatchTopK / PerLayerBatchTopK training forward -> [features, updated EMA threshold] (topk.py:84-95, 117-132).
    Functional: the kernel updates a copy of the threshold, which the module writes back to its buffer."""
    thr = threshold.clone() if threshold is not None else None
    feat = ops.topk_segmented(x.contiguous(), k, per_layer, thr, e)
    return [feat, thr if thr is not None else x.new_empty(0)]


@batch_topk.register_fake
def _(x, k, per_layer, threshold, e):
    return [torch.empty_like(x, memory_format=torch.contiguous_format), torch.empty_like(threshold) if threshold is not None else x.new_empty(0)]


@torch.library.custom_op("clt_hip::mask_positive", mutates_args=(), device_types=_DEV)
def mask_positive(feat: Tensor, g: Tensor) -> Tensor:
    return ops.mask_positive(feat, g.contiguous())


@mask_positive.register_fake
def _(feat, g):
    return torch.empty_like(feat)


topk_rows.register_autograd(lambda ctx, g: (torch.ops.clt_hip.mask_positive(ctx.saved_tensors[0], g), None),
                            setup_context=lambda ctx, inputs, output: ctx.save_for_backward(output))
batch_topk.register_autograd(lambda ctx, grads: (torch.ops.clt_hip.mask_positive(ctx.saved_tensors[0], grads[0]), None, None, None, None),
                             setup_context=lambda ctx, inputs, output: ctx.save_for_backward(output[0]))


@torch.library.custom_op("clt_hip::threshold", mutates_args=(), device_types=_DEV)
def threshold(x: Tensor, thr: Tensor) -> Tensor:
    """Inference forward of the EMA sparsifiers: x * [x > threshold] (topk.py:71-74, 104-107)."""
    return ops.threshold_fwd(x.contiguous(), thr, False, 1)


@threshold.register_fake
def _(x, thr):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


# ---- standardizers (model/standardize.py:30-117); gradient w.r.t. the activations only (mean / std are buffers) -----------
@torch.library.custom_op("clt_hip::standardize", mutates_args=(), device_types=_DEV)
def standardize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    return ops.standardize(x, mean, std)[0]


@standardize.register_fake
def _(x, mean, std):
    return x.new_empty(x.shape, dtype=torch.float32)


@torch.library.custom_op("clt_hip::destandardize", mutates_args=(), device_types=_DEV)
def destandardize(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    return ops.destandardize(x.contiguous(), mean, std)


@destandardize.register_fake
def _(x, mean, std):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


def _std_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1], inputs[2])


# d/dx (x - mean) / std = g / std  and  d/dx (x * std + mean) = g * std: the same kernels with a zero mean
standardize.register_autograd(
    lambda ctx, g: (torch.ops.clt_hip.standardize(g, torch.zeros_like(ctx.saved_tensors[0]), ctx.saved_tensors[1]), None, None),
    setup_context=_std_setup)
destandardize.register_autograd(
    lambda ctx, g: (torch.ops.clt_hip.destandardize(g, torch.zeros_like(ctx.saved_tensors[0]), ctx.saved_tensors[1]), None, None),
    setup_context=_std_setup)


#: name -> OpOverloadPacket, for tests and introspection
REGISTERED = ("encoder", "encoder_bwd", "plt_decoder", "plt_decoder_bwd", "clt_decoder", "clt_decoder_bwd", "relu", "relu_bwd",
              "jumprelu", "jumprelu_bwd", "topk_rows", "batch_topk", "mask_positive", "threshold", "standardize", "destandardize")
