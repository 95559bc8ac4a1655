"""Standardizers (reference model/standardize.py).  The dimension-wise pair — the one every reference
config uses — runs on HIP kernels (clt_column_stats_f32 / clt_standardize_f32 / clt_destandardize_f32)."""
import torch

from .. import custom_ops  # noqa: F401  (registers torch.ops.clt_hip.*)
from .serializable_module import SerializableModule


class Standardizer(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.standardize"

    def __init__(self, **kwargs):
        super().__init__()

    def initialize_from_batch(self, batch):
        pass

    def forward(self, batch, layer="all"):
        return batch

    def standardize(self, batch, layer="all"):
        return batch


class _Dimensionwise(Standardizer):
    _io = 0

    def __init__(self, n_layers, activation_dim):
        super().__init__()
        self.n_layers = n_layers
        self.activation_dim = activation_dim
        self.register_buffer("mean", torch.empty(n_layers, activation_dim))
        self.register_buffer("std", torch.empty(n_layers, activation_dim))
        self.is_initialized = False

    def _init_args(self):
        return {"n_layers": self.n_layers, "activation_dim": self.activation_dim}

    @torch.no_grad()
    def initialize_from_batch(self, batch):
        """mean / unbiased std (clamped at 1e-8) over the batch axis of batch[:, io]
        (standardize.py:39-47, 80-86)."""
        from .. import ops
        mean, std = ops.column_stats(batch[:, self._io].float())
        self.mean.data, self.std.data = mean, std
        self.is_initialized = True

    def _require_init(self):
        if not self.is_initialized:
            raise ValueError("Standardizer not initialized")

    def _sel(self, layer):
        if layer == "all":
            return self.mean, self.std
        return self.mean[layer].contiguous(), self.std[layer].contiguous()

    def _norm(self, x, layer):
        mean, std = self._sel(layer)
        if layer != "all":                 # [..., d] rows against one layer's statistics
            flat = x.reshape(-1, x.shape[-1])
            return torch.ops.clt_hip.standardize(flat, mean, std).reshape(x.shape)
        return torch.ops.clt_hip.standardize(x, mean, std)


class DimensionwiseInputStandardizer(_Dimensionwise):
    _io = 0

    def forward(self, batch, layer="all"):
        self._require_init()
        return self._norm(batch, layer)


class DimensionwiseOutputStandardizer(_Dimensionwise):
    _io = 1

    def forward(self, batch, layer="all"):
        """De-standardize: batch * std + mean (standardize.py:88-98)."""
        self._require_init()
        mean, std = self._sel(layer)
        if layer != "all":
            flat = batch.reshape(-1, batch.shape[-1])
            return torch.ops.clt_hip.destandardize(flat, mean, std).reshape(batch.shape)
        return torch.ops.clt_hip.destandardize(batch, mean, std)

    def standardize(self, mlp_out, layer="all"):
        return self._norm(mlp_out, layer)
