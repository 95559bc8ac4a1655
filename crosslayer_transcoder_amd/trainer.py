"""Minimal YAML -> objects loader and fit loop for the reference's LightningCLI configs
(reference main.py:12-48, config/*.yaml) when Lightning itself is not installed: the `model:` tree is
instantiated from its class_path/init_args nodes (reference package prefix accepted), the `data:` node likewise
(TensorDataModule), `trainer.precision` / `trainer.max_steps` / `seed_everything` are honoured, and `fit` drives
`training_step` the way Lightning's loop would: with `fused_step` (default, manual optimisation) the module's
training_step runs forward + backward + Adam itself; with `fused_step=False` the automatic-optimisation order
training_step -> backward -> optimizer.step -> scheduler.step is run here."""
import itertools
from typing import Iterable, Optional

import torch
import yaml

from .model.serializable_module import build_from_config
from .precision import set_precision


def load_config(path: str) -> dict:
    with open(path) as fh:
        return yaml.safe_load(fh)


def build_module(cfg: dict, drop_eval_metrics: bool = True):
    node = dict(cfg["model"])
    ia = dict(node.get("init_args") or {})
    if drop_eval_metrics:                      # nnsight / torchmetrics objects: eval-time, out of scope
        ia["replacement_model"] = None
        ia["dead_features"] = None
    node["init_args"] = ia
    return build_from_config(node)


def build_datamodule(cfg: dict):
    """The `data:` node of a LightningCLI config (reference config/debug-cpu.yaml:79-86).  Only the file-fed
    TensorDataModule is in scope; the streaming ActivationDataModule needs nnsight + network."""
    node = cfg.get("data")
    if not node:
        return None
    if not node["class_path"].endswith("TensorDataModule"):
        raise NotImplementedError(f"data module {node['class_path']} is out of scope (streaming activation generation)")
    from .data.tensor_datamodule import TensorDataModule
    return TensorDataModule(**(node.get("init_args") or {}))


def apply_trainer_precision(cfg: dict):
    """`trainer.precision`: "32", "bf16-mixed" and "16-mixed" (fp16 operands + dynamic loss scaling, the setting of the
    reference's production YAMLs, config/default.yaml:13) map one to one onto precision.set_precision."""
    set_precision(str((cfg.get("trainer") or {}).get("precision", "32")))


def fit(module, batches: Iterable[torch.Tensor], max_steps: int, device="cuda:0", fused: Optional[bool] = None,
        on_step=None):
    """Run `max_steps` optimizer steps through `module.training_step`.  fused: override module.fused_step."""
    module.to(device)
    if fused is not None:
        module.fused_step = bool(fused)
        module.automatic_optimization = not module.fused_step
    module.trainer.max_steps = max_steps
    module.configure_model()
    opt, sched = module.attach_optimizers()
    losses = []
    from .data.prefetch import prefetch
    for t, batch in enumerate(prefetch(itertools.islice(batches, max_steps), device)):   # pinned, double-buffered H2D
        if module.fused_step:
            loss = module.training_step(batch, t)
        else:
            opt.zero_grad(set_to_none=True)
            loss = module.training_step(batch, t)
            loss.backward()
            opt.step()
            if sched is not None:
                sched.step()
            module.global_step += 1
        losses.append(loss.detach())
        if on_step is not None:
            on_step(t, loss)
    return losses, opt


def run_config(cfg, device="cuda:0", max_steps: Optional[int] = None, fused: Optional[bool] = None, tensor_file=None,
               init_state: Optional[dict] = None):
    """`python main.py fit --config X.yaml` without LightningCLI: seed, precision, model tree, data module, fit.
    cfg: path or loaded dict.  tensor_file overrides data.init_args.tensor_file; init_state: a state dict loaded into
    module.model before training (non-strict: e.g. the parameters of a checkpoint).  Returns (module, losses)."""
    if isinstance(cfg, str):
        cfg = load_config(cfg)
    if cfg.get("seed_everything") is not None:
        torch.manual_seed(int(cfg["seed_everything"]))
    apply_trainer_precision(cfg)
    module = build_module(cfg)
    if init_state is not None:
        module.model.load_state_dict(init_state, strict=False)
    dm = build_datamodule(cfg)
    if dm is None:
        raise ValueError("config has no data: section")
    if tensor_file is not None:
        dm.tensor_file = tensor_file
    dm.setup("fit")
    steps = int((cfg.get("trainer") or {}).get("max_steps", 1)) if max_steps is None else max_steps
    if int((cfg.get("trainer") or {}).get("accumulate_grad_batches", 1)) != 1:
        raise NotImplementedError("accumulate_grad_batches != 1")
    losses, _ = fit(module, dm.train_dataloader(), steps, device, fused)
    return module, losses
