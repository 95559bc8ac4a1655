"""TopK / BatchTopK / PerLayerBatchTopK sparsifiers (reference model/topk.py) on the HIP radix-select
kernels.  Selection rule: larger value first, ties to the lower index."""
from typing import Union

import torch

from .. import custom_ops  # noqa: F401  (registers torch.ops.clt_hip.*)
from .serializable_module import SerializableModule


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
        return torch.ops.clt_hip.topk_rows(features, self.k)       # backward = g * [features > 0] (scatter(relu(gather)))

    @torch.no_grad()
    def _inference_forward(self, features, layer):
        shp = features.shape
        out = torch.ops.clt_hip.topk_rows(features.reshape(-1, 1, shp[-1]), self.k)
        return out.reshape(shp)


class _EmaTopK(TopK):
    _per_layer = False

    def _training_forward(self, features):
        self._check(features)
        if self.k > 0 and self.e is None:
            raise TypeError("BatchTopK needs `e` (threshold EMA rate) for training")
        # the reference rebinds `self.threshold` to a fresh tensor each step (topk.py:95,132); here the registered buffer is
        # overwritten with the op's updated copy, which is state-dict-equivalent
        feat, thr = torch.ops.clt_hip.batch_topk(features, self.k, self._per_layer, self.threshold if self.k > 0 else None,
                                                 float(self.e or 0.0))
        if self.k > 0:
            with torch.no_grad():
                self.threshold.copy_(thr)
        return feat

    @torch.no_grad()
    def _inference_forward(self, features, layer):
        thr = self.threshold[layer:layer + 1].contiguous() if self._per_layer else self.threshold
        return torch.ops.clt_hip.threshold(features, thr)


class BatchTopK(_EmaTopK):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("threshold", torch.zeros(1))


class PerLayerBatchTopK(_EmaTopK):
    _per_layer = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("threshold", torch.zeros(self.n_layers))
