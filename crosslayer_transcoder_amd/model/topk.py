"""TopK / BatchTopK / PerLayerBatchTopK sparsifiers (reference model/topk.py) on the HIP radix-select
kernels.  Selection rule: larger value first, ties to the lower index."""
from typing import Union

import torch

from .serializable_module import SerializableModule


class _SelectFn(torch.autograd.Function):
    """forward = dense features from one of the select kernels; backward = g * [features > 0]
    (gradient of scatter(relu(gather)))."""

    @staticmethod
    def forward(ctx, x, mode, k, threshold, e):
        from .. import ops
        x = x.contiguous()
        if mode == "rows":
            feat = ops.topk_rows(x, k)
        else:
            feat = ops.topk_segmented(x, k, mode == "layer", threshold, e)
        ctx.save_for_backward(feat)
        return feat

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        (feat,) = ctx.saved_tensors
        return ops.mask_positive(feat, g.contiguous()), None, None, None, None


class TopK(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.topk"

    def __init__(self, k: int, e=None, n_layers=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.k = k
        self.e = e
        self.n_layers = n_layers

    def _init_args(self):
        return {"k": self.k, "e": self.e, "n_layers": self.n_layers}

    def forward(self, features, layer: Union[int, str] = "all"):
        if layer == "all":
            return self._training_forward(features)
        return self._inference_forward(features, layer)

    def _check(self, features):
        if features.dim() != 3:
            raise ValueError("expected [batch, n_layers, d_features]")
        assert self.k <= features.shape[-1]          # reference: AssertionError when k > d_features


class PerLayerTopK(TopK):
    def _training_forward(self, features):
        self._check(features)
        return _SelectFn.apply(features, "rows", self.k, None, 0.0)

    @torch.no_grad()
    def _inference_forward(self, features, layer):
        shp = features.shape
        out = _SelectFn.apply(features.reshape(-1, 1, shp[-1]), "rows", self.k, None, 0.0)
        return out.reshape(shp)


class _EmaTopK(TopK):
    _per_layer = False

    def _training_forward(self, features):
        self._check(features)
        if self.k > 0 and self.e is None:
            raise TypeError("BatchTopK needs `e` (threshold EMA rate) for training")
        # the reference rebinds `self.threshold` to a fresh tensor each step (topk.py:95,132); the kernel
        # updates the registered buffer in place, which is state-dict-equivalent and allocation-free
        return _SelectFn.apply(features, "layer" if self._per_layer else "batch", self.k,
                               self.threshold if self.k > 0 else None, self.e or 0.0)

    @torch.no_grad()
    def _inference_forward(self, features, layer):
        from .. import ops
        thr = self.threshold[layer:layer + 1].contiguous() if self._per_layer else self.threshold
        return ops.threshold_fwd(features.contiguous(), thr, False, 1)


class BatchTopK(_EmaTopK):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("threshold", torch.zeros(1))


class PerLayerBatchTopK(_EmaTopK):
    _per_layer = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("threshold", torch.zeros(self.n_layers))
