"""Compute precision of the dense contractions: "32" (exact-fp32 MFMA, the parity mode of
`precision: "32"`) or "bf16" (bf16 operands / fp32 accumulate, `precision: "bf16-mixed"`).
A surrounding torch.autocast(dtype=bfloat16) region also selects bf16, as Lightning would set up."""
import contextlib

import torch

_PRECISION = "32"


def set_precision(p: str):
    global _PRECISION
    p = {"32": "32", "32-true": "32", "fp32": "32", "bf16": "bf16", "bf16-mixed": "bf16"}.get(str(p))
    if p is None:
        raise ValueError("precision must be '32' or 'bf16-mixed' (fp16 + GradScaler is not implemented)")
    _PRECISION = p


def get_precision() -> str:
    if torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
        return "bf16"
    return _PRECISION


@contextlib.contextmanager
def precision(p: str):
    old = _PRECISION
    set_precision(p)
    try:
        yield
    finally:
        set_precision(old)
