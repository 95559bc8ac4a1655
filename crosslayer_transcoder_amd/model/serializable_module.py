"""Config (de)serialisation contract of the reference (model/serializable_module.py:16-69):
`to_config()` -> {"class_path", "init_args"}; `from_config()` rebuilds nested modules by class_path;
`save_pretrained` / `from_pretrained` use config.yaml + checkpoint.safetensors.

class_path strings are written with the REFERENCE package prefix (`crosslayer_transcoder.`) so that
checkpoints stay interchangeable (pinned by the reference's tests/test_save_to_pretrained.py:64-91);
on load, both that prefix and this package's own are accepted and resolve to the classes here.
"""
import importlib
import re
from pathlib import Path
from typing import Any, Dict, Union

import yaml
from safetensors.torch import load_file, save_file
from torch import nn

REF_PREFIX = "crosslayer_transcoder."
OUR_PREFIX = "crosslayer_transcoder_amd."


def resolve_class(class_path: str):
    """Import `pkg.mod.Class`; reference-prefixed paths map onto this package."""
    mod_name, cls_name = class_path.rsplit(".", 1)
    if mod_name.startswith(REF_PREFIX):
        mod_name = OUR_PREFIX + mod_name[len(REF_PREFIX):]
    return getattr(importlib.import_module(mod_name), cls_name)


#: PyYAML (YAML 1.1) reads exponent-only numbers such as `1e-3` — the spelling of the reference's configs
#: (config/debug-cpu.yaml: `learning_rate: 1e-3`, `pre_actv_loss: 1e-6`) — as strings; jsonargparse turns them into
#: floats from the constructor's annotations.  Same effect here.
_SCI_NUMBER = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def build_from_config(node: Any):
    """Instantiate a {"class_path", "init_args"} tree (the LightningCLI / jsonargparse subclass schema)."""
    if isinstance(node, str) and _SCI_NUMBER.match(node):
        return float(node)
    if isinstance(node, dict) and "class_path" in node:
        cls = resolve_class(node["class_path"])
        kwargs = {k: build_from_config(v) for k, v in (node.get("init_args") or {}).items()}
        return cls(**kwargs)
    if isinstance(node, list):
        return [build_from_config(v) for v in node]
    return node


class SerializableModule(nn.Module):
    #: module path the reference uses for this class; to_config() reports it
    _ref_module = "crosslayer_transcoder.model"

    def _class_path(self) -> str:
        return f"{self._ref_module}.{type(self).__name__}"

    def _init_args(self) -> Dict[str, Any]:
        return {}

    def to_config(self) -> Dict[str, Any]:
        return {"class_path": self._class_path(), "init_args": self._init_args()}

    @classmethod
    def from_config(cls, config: Dict[str, Any]):
        kwargs = {k: build_from_config(v) for k, v in (config.get("init_args") or {}).items()}
        return cls(**kwargs)

    def save_pretrained(self, directory: Path) -> None:
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        with open(directory / "config.yaml", "w") as fh:
            yaml.dump({"model": self.to_config()}, fh)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, directory / "checkpoint.safetensors")

    @classmethod
    def from_pretrained(cls, directory: Union[Path, str]):
        directory = Path(directory)
        with open(directory / "config.yaml") as fh:
            full = yaml.safe_load(fh)
        cfg = full.get("model")
        if cfg is None:
            raise ValueError("Model config not found in config.yaml", full)
        model = cls.from_config(cfg)
        model.load_state_dict(load_file(directory / "checkpoint.safetensors"))
        model._is_folded = cfg.get("is_folded", False)
        return model
