"""Compute precision of the dense contractions: "32" (exact-fp32 MFMA, the parity mode of `precision: "32"`), "bf16"
(bf16 operands / fp32 accumulate, `precision: "bf16-mixed"`) or "fp16" (fp16 operands / fp32 accumulate + dynamic loss
scaling, `precision: "16-mixed"` — the setting of the reference's production YAMLs, config/default.yaml:13).
A surrounding torch.autocast region selects the matching 16-bit mode, as Lightning would set up."""
import contextlib

import torch

_PRECISION = "32"
_NAMES = {"32": "32", "32-true": "32", "fp32": "32", "bf16": "bf16", "bf16-mixed": "bf16", "16-mixed": "fp16", "16": "fp16",
          "fp16": "fp16"}


def set_precision(p: str):
    global _PRECISION
    q = _NAMES.get(str(p))
    if q is None:
        raise ValueError("precision must be '32', 'bf16-mixed' or '16-mixed'")
    _PRECISION = q


def get_precision() -> str:
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_gpu_dtype()
        if dt == torch.bfloat16:
            return "bf16"
        if dt == torch.float16:
            return "fp16"
    return _PRECISION


def mixed16() -> bool:
    """16-bit GEMM operands with fp32 accumulation (either format)."""
    return get_precision() != "32"


def operand_dtype() -> torch.dtype:
    """The 16-bit operand format of the mixed modes (what the library's conversion kernels write)."""
    return torch.float16 if get_precision() == "fp16" else torch.bfloat16


@contextlib.contextmanager
def precision(p: str):
    old = _PRECISION
    set_precision(p)
    try:
        yield
    finally:
        set_precision(old)


class LossScaler:
    """Dynamic loss scaling of the fp16 mode with torch.amp.GradScaler's rule and defaults (init 2**16, x2 after 2000 clean
    steps, x0.5 after an overflow; Lightning's "16-mixed" plugin creates exactly that scaler).  The overflow flag lives on
    the device: the conversion kernels raise it, the Adam kernels honour it (a skipped step changes nothing), the host
    reads it once at the end of the step to move the scale."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale = float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.growth_tracker = 0
        self.skipped_steps = 0

    def update(self, found_inf: bool):
        if found_inf:
            self.scale *= self.backoff_factor
            self.growth_tracker = 0
            self.skipped_steps += 1
        else:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0

    def state_dict(self):
        return {"scale": self.scale, "growth_tracker": self.growth_tracker}

    def load_state_dict(self, sd):
        self.scale, self.growth_tracker = float(sd["scale"]), int(sd["growth_tracker"])


# ---- A/B switches -------------------------------------------------------------------------------------------------
def _parse_ab():
    """CLT_AB="name=value,name=value": the ONE environment variable the package reads for same-box A/B runs of code paths
    whose default is settled (scripts/gpu_*_ab.sh).  Unknown names raise at import, so a typo cannot silently measure the
    default twice."""
    import os
    out = {}
    for item in filter(None, (t.strip() for t in os.environ.get("CLT_AB", "").split(","))):
        name, _, val = item.partition("=")
        out[name.strip()] = val.strip() or "1"
    return out


_AB = _parse_ab()
_AB_KNOWN = set()


def ab_switch(name: str, default: int) -> int:
    """Integer value of switch `name` from CLT_AB, else `default`."""
    _AB_KNOWN.add(name)
    return int(_AB.get(name, default))


def ab_check():
    unknown = set(_AB) - _AB_KNOWN
    if unknown:
        raise ValueError(f"CLT_AB names unknown switches {sorted(unknown)}; known: {sorted(_AB_KNOWN)}")
