"""Encoder / Decoder / CrosslayerDecoder / CrossLayerTranscoder with the reference's constructor
arguments, forward signatures, parameter names and shapes (reference model/clt.py:115-467), computing on
the HIP GEMM kernels.  State-dict keys: encoder.W [L,d,F], encoder.b [L,F], decoder.W_{l} [(l+1),F,d]
(cross-layer) or decoder.W [L,F,d] (per-layer), decoder.b [L,d]."""
from pathlib import Path
from typing import Optional, Union

import torch
import torch.nn as nn
import yaml
from safetensors.torch import save_file

from .. import custom_ops  # noqa: F401  (registers torch.ops.clt_hip.*)
from ..precision import get_precision
from .serializable_module import SerializableModule
from .standardize import Standardizer


def _feature_major(W: torch.Tensor) -> torch.Tensor:
    """[L,d,F] encoder weight -> its [L,F,d] storage view (a free permute when the layout is the native one)."""
    Wt = W.permute(0, 2, 1)
    return Wt if Wt.is_contiguous() else Wt.contiguous()


def _operand(t: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> GEMM operand in the active precision."""
    from .. import ops
    t = t.contiguous()
    if get_precision() == "32":
        return t
    ops.sync_operand_format()                   # (fp16 / bf16 operands as the active precision says)
    return ops.cast_bf16(t)


class Encoder(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.clt"

    def __init__(self, d_acts: int, d_features: int, n_layers: int):
        super().__init__()
        self.d_acts, self.d_features, self.n_layers = d_acts, d_features, n_layers
        # Same name and logical shape [L,d,F] as the reference (state-dict compatible), but stored
        # feature-major ([L,F,d] in memory): one feature's d weights are contiguous, which is what the MFMA
        # B-operand (k-contiguous) and the sparse per-feature gradient/Adam row pass both want.
        self.W = nn.Parameter(torch.empty(n_layers, d_features, d_acts).permute(0, 2, 1))
        self.bias = True
        self.b = nn.Parameter(torch.empty(n_layers, d_features))
        self._is_folded = False
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / self.d_features ** 0.5                # clt.py:128-132
        with torch.no_grad():
            self.W.uniform_(-bound, bound)
            self.b.zero_()

    @torch.no_grad()
    def forward_layer(self, acts_norm, layer: int):
        """Inference: [batch, seq, d_acts] of ONE layer -> [batch, seq, d_features] (clt.py:134-145)."""
        from .. import ops
        shp = acts_norm.shape
        x = _operand(acts_norm.reshape(-1, 1, shp[-1]))
        W = _operand(self.Wt[layer:layer + 1])
        return ops.encoder_fwd(x, W, self.b[layer:layer + 1].contiguous()).reshape(*shp[:-1], self.d_features)

    @property
    def Wt(self) -> torch.Tensor:
        """Feature-major [L,F,d] view of W (contiguous; re-laid out once if a load/assignment changed it)."""
        Wt = self.W.permute(0, 2, 1)
        if not Wt.is_contiguous():
            with torch.no_grad():
                self.W.data = Wt.contiguous().permute(0, 2, 1)
            Wt = self.W.permute(0, 2, 1)
        return Wt

    def forward(self, acts_norm, layer="all"):
        if layer != "all":
            return self.forward_layer(acts_norm, layer)
        return torch.ops.clt_hip.encoder(acts_norm, self.W, self.b)          # custom_ops.py: dispatcher op + autograd rule

    @torch.no_grad()
    def fold(self, input_standardizer: Standardizer):
        """W <- W / std[..., None]; b <- b - mean @ W_folded (clt.py:169-185).  One-off export step.  On the GPU the
        contraction goes through the library's exact-fp32 GEMM like every other contraction of the package (no torch GPU
        GEMM anywhere); a module held on the CPU (export tooling) uses plain tensor algebra."""
        if self._is_folded:
            return self
        self.W.data = self.W / input_standardizer.std.unsqueeze(-1)
        if self.W.is_cuda:
            from .. import ops
            mw = ops.encoder_fwd(input_standardizer.mean.float().unsqueeze(0).contiguous(), self.Wt.contiguous(), None)[0]
        else:
            mw = torch.einsum("ld,ldf->lf", input_standardizer.mean, self.W)
        self.b.data = self.b - mw
        self._is_folded = True
        return self

    def _init_args(self):
        return {"d_acts": self.d_acts, "d_features": self.d_features, "n_layers": self.n_layers}


class _DecoderBase(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.clt"

    def _init_args(self):
        return {"d_acts": self.d_acts, "d_features": self.d_features, "n_layers": self.n_layers}

    def _bound(self):
        return 1.0 / (self.d_acts * self.n_layers) ** 0.5   # clt.py:211-214, 293-302


class Decoder(_DecoderBase):
    """Per-layer transcoder decoder: recons[:, l] = f[:, l] @ W[l] + b[l]."""

    def __init__(self, d_acts: int, d_features: int, n_layers: int):
        super().__init__()
        self.d_acts, self.d_features, self.n_layers = d_acts, d_features, n_layers
        self.W = nn.Parameter(torch.empty(n_layers, d_features, d_acts))
        self.b = nn.Parameter(torch.empty(n_layers, d_acts))
        self._is_folded = False
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            self.W.uniform_(-self._bound(), self._bound())
            self.b.zero_()

    def weights(self):
        return [self.W]

    @torch.no_grad()
    def forward_layer(self, features, layer: int):
        from .. import ops
        if features.ndim == 4:
            features = features[:, :, layer, :]
        shp = features.shape
        f = _operand(features.reshape(-1, 1, shp[-1]))
        r = ops.plt_decoder_fwd(f, _operand(self.W[layer:layer + 1]), self.b[layer:layer + 1].contiguous())
        return r.reshape(*shp[:-1], self.d_acts)

    def forward(self, features, layer="all"):
        if layer != "all":
            return self.forward_layer(features, layer)
        return torch.ops.clt_hip.plt_decoder(features, self.W, self.b)

    @torch.no_grad()
    def fold(self, output_standardizer: Standardizer):
        if self._is_folded:
            return self
        self.W.data = self.W * output_standardizer.std.unsqueeze(1)          # clt.py:251-266
        self.b.data = self.b * output_standardizer.std + output_standardizer.mean
        self._is_folded = True
        return self


class CrosslayerDecoder(_DecoderBase):
    """Causal cross-layer decoder: recons[:, l] = sum_{l' <= l} f[:, l'] @ W_l[l'] + b[l]."""

    def __init__(self, d_acts: int, d_features: int, n_layers: int):
        super().__init__()
        self.d_acts, self.d_features, self.n_layers = d_acts, d_features, n_layers
        for l in range(n_layers):
            self.register_parameter(f"W_{l}", nn.Parameter(torch.empty(l + 1, d_features, d_acts)))
        self.b = nn.Parameter(torch.empty(n_layers, d_acts))
        self._is_folded = False
        self.reset_parameters()

    def weights(self):
        return [getattr(self, f"W_{l}") for l in range(self.n_layers)]

    def reset_parameters(self):
        with torch.no_grad():
            for w in self.weights():
                w.uniform_(-self._bound(), self._bound())
            self.b.zero_()

    @torch.no_grad()
    def forward_layer(self, features, layer: int):
        """Inference: features [batch, seq, from_layer (= layer+1), F] -> [batch, seq, d] (clt.py:304-317)."""
        from .. import ops
        shp = features.shape
        K = (layer + 1) * self.d_features
        f = _operand(features.reshape(-1, 1, K))
        W = _operand(getattr(self, f"W_{layer}").reshape(1, K, self.d_acts))
        r = ops.plt_decoder_fwd(f, W, self.b[layer:layer + 1].contiguous())
        return r.reshape(*shp[:-2], self.d_acts)

    def forward(self, features, layer="all"):
        if layer != "all":
            return self.forward_layer(features, layer)
        return torch.ops.clt_hip.clt_decoder(features, self.b, self.weights())

    @torch.no_grad()
    def fold(self, output_standardizer: Standardizer):
        if self._is_folded:
            return self
        self.b.data = self.b * output_standardizer.std + output_standardizer.mean   # clt.py:346-360
        for l, w in enumerate(self.weights()):
            w.data = w * output_standardizer.std[l]
        self._is_folded = True
        return self


class CrossLayerTranscoder(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.clt"

    def __init__(self, nonlinearity, encoder: Encoder, decoder: Union[Decoder, CrosslayerDecoder],
                 input_standardizer: Optional[Standardizer] = None,
                 output_standardizer: Optional[Standardizer] = None):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.nonlinearity = nonlinearity
        self.input_standardizer = input_standardizer
        self.output_standardizer = output_standardizer
        self._is_folded = False
        self.reset_parameters()

    def reset_parameters(self):
        self.encoder.reset_parameters()
        self.decoder.reset_parameters()

    def initialize_standardizers(self, batch):
        self.input_standardizer.initialize_from_batch(batch)
        self.output_standardizer.initialize_from_batch(batch)

    def forward(self, acts):
        """-> (pre_actvs [B,L,F], features [B,L,F], recons_norm [B,L,d], recons [B,L,d])  (clt.py:403-425)."""
        if self.input_standardizer is not None:
            acts = self.input_standardizer(acts)
        pre_actvs = self.encoder(acts)
        features = self.nonlinearity(pre_actvs)
        recons_norm = self.decoder(features)
        recons = self.output_standardizer(recons_norm) if self.output_standardizer is not None else recons_norm
        return pre_actvs, features, recons_norm, recons

    def _init_args(self):
        cfg = lambda m: m.to_config() if m is not None else None  # noqa: E731
        return {"nonlinearity": cfg(self.nonlinearity), "encoder": cfg(self.encoder), "decoder": cfg(self.decoder),
                "input_standardizer": cfg(self.input_standardizer),
                "output_standardizer": cfg(self.output_standardizer)}

    def fold(self):
        if self._is_folded:
            return
        if self.input_standardizer is not None:
            self.encoder.fold(self.input_standardizer)
        if self.output_standardizer is not None:
            self.decoder.fold(self.output_standardizer)
        self._is_folded = True
        self.input_standardizer = None
        self.output_standardizer = None

    def save_pretrained(self, directory: Path, fold_standardizers: bool = True):
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        if fold_standardizers:
            self.fold()
        config = self.to_config()
        config["is_folded"] = self._is_folded
        with open(directory / "config.yaml", "w") as fh:
            yaml.dump({"model": config}, fh)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, directory / "checkpoint.safetensors")
