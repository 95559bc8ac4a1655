"""ReLU / JumpReLU sparsifiers (reference model/jumprelu.py) on HIP kernels."""
import torch
import torch.nn as nn

from .. import custom_ops  # noqa: F401  (registers torch.ops.clt_hip.*)
from .serializable_module import SerializableModule


class ReLU(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.jumprelu"

    def forward(self, input, layer=None):
        return torch.ops.clt_hip.relu(input)


class JumpReLU(SerializableModule):
    _ref_module = "crosslayer_transcoder.model.jumprelu"

    def __init__(self, theta=0.0, bandwidth=1.0, n_layers=12, d_features=768 * 8):
        super().__init__()
        self.theta = nn.Parameter(torch.full((1, n_layers, d_features), float(theta)))
        self.register_buffer("bandwidth", torch.tensor(float(bandwidth)))
        self._init_theta = theta
        self._bandwidth = float(bandwidth)
        self.n_layers = n_layers
        self.d_features = d_features

    def forward(self, input):
        # fwd: x*[(x>theta)&(x>0)]; bwd: dx = g (0 where x<0), dtheta via the rectangle-window STE (jumprelu.py:31-48),
        # one fused pass + deterministic column reduction (custom_ops.jumprelu / jumprelu_bwd)
        return torch.ops.clt_hip.jumprelu(input, self.theta, self._bandwidth)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._bandwidth = float(self.bandwidth)          # host copy: no device sync on the hot path

    def _init_args(self):
        return {"theta": self._init_theta, "bandwidth": self._bandwidth, "n_layers": self.n_layers,
                "d_features": self.d_features}
