"""Import the reference's model math (read-only, from /root/reference) for golden generation.

Used ONLY by tests/golden/make_golden.py, in the build container.  Nothing here travels to the
GPU box at run time and no test imports it.  The reference package's __init__ files pull in
lightning / nnsight / wandb / torchmetrics / jaxtyping, none of which is installed here; those
are annotation- or logging-only on this path, so they are replaced by inert placeholder modules
and the five model files are loaded by path under their real dotted names.
No reference file is modified or copied.
"""
import importlib.util
import sys
import types
import typing

import torch
import torch.nn as nn
import typing_extensions

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(dotted, relpath):
    spec = importlib.util.spec_from_file_location(dotted, f"{REF}/{relpath}")
    m = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = m
    spec.loader.exec_module(m)
    return m


class _Sub:
    def __class_getitem__(cls, item):
        return torch.Tensor


class _LightningModule(nn.Module):
    """Just enough of lightning.LightningModule for training_step(): log / global_step / trainer."""

    def __init__(self, *a, **k):
        super().__init__()
        self.logged = {}
        self.global_step = 0
        self.trainer = types.SimpleNamespace(max_steps=1, num_devices=1, optimizers=[])
        self.logger = None

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, name, value, *a, **k):
        self.logged[name] = value.detach().clone() if torch.is_tensor(value) else value

    def lr_schedulers(self):
        return None


def load_reference():
    if "crosslayer_transcoder.model.clt_lightning" in sys.modules:
        return sys.modules
    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    _mod("jaxtyping", Float=_Sub, Int=_Sub, Bool=_Sub)
    L = _mod("lightning", LightningModule=_LightningModule, LightningDataModule=object)
    lp = _mod("lightning.pytorch")
    lpl = _mod("lightning.pytorch.loggers", WandbLogger=type("WandbLogger", (), {}))
    L.pytorch = lp
    lp.loggers = lpl
    wb = _mod("wandb")
    wb.wandb_run = types.SimpleNamespace(Run=type("Run", (), {}))
    _mod("torchmetrics", Metric=nn.Module)
    _mod("nnsight")
    _mod("psutil") if "psutil" not in sys.modules else None
    for pkg in ("crosslayer_transcoder", "crosslayer_transcoder.model", "crosslayer_transcoder.metrics",
                "crosslayer_transcoder.utils"):
        m = _mod(pkg)
        m.__path__ = []
    _mod("crosslayer_transcoder.utils.utils", get_webtext_dataloader=None)
    base = "crosslayer_transcoder"
    for name in ("serializable_module", "standardize", "jumprelu", "topk", "clt"):
        _load(f"{base}.model.{name}", f"{base}/model/{name}.py")
    _mod(f"{base}.metrics.dead_features", DeadFeatures=type("DeadFeatures", (), {}))
    _mod(f"{base}.metrics.replacement_model_accuracy",
         ReplacementModelAccuracy=type("ReplacementModelAccuracy", (), {}))
    _load(f"{base}.model.clt_lightning", f"{base}/model/clt_lightning.py")
    return sys.modules
