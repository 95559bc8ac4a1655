"""crosslayer_transcoder_amd — MI355X-native crosslayer-transcoder training step.

Host side mirrors the reference's module / YAML contract; all arithmetic runs in
hand-written HIP kernels (csrc/) reached through the C ABI in include/clt_hip.h.
"""
from .precision import get_precision, precision, set_precision  # noqa: F401

REF_PREFIX = "crosslayer_transcoder."
OUR_PREFIX = "crosslayer_transcoder_amd."

__version__ = "0.1.0"
