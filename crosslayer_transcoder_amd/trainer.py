"""Minimal YAML -> objects loader and fit loop for the reference's LightningCLI configs
(reference main.py:12-48, config/*.yaml) when Lightning itself is not installed: the `model:` tree is
instantiated from its class_path/init_args nodes (reference package prefix accepted), and `fit` runs the
automatic-optimisation order: training_step -> backward -> optimizer.step -> scheduler.step."""
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


def apply_trainer_precision(cfg: dict):
    prec = str((cfg.get("trainer") or {}).get("precision", "32"))
    set_precision({"16-mixed": "bf16-mixed"}.get(prec, prec))      # fp16+GradScaler -> bf16 (documented)


def fit(module, batches: Iterable[torch.Tensor], max_steps: int, device="cuda:0", fused: bool = True,
        on_step=None):
    """Run `max_steps` optimizer steps.  `fused=False` goes through training_step/backward/step exactly
    as Lightning's loop would; `fused=True` uses module.fused_train_step (same arithmetic, no autograd)."""
    module.to(device)
    module.trainer.max_steps = max_steps
    conf = module.configure_optimizers()
    if isinstance(conf, dict):
        opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    else:
        opt, sched = conf, None
    losses = []
    from .data.prefetch import prefetch
    import itertools
    for t, batch in enumerate(prefetch(itertools.islice(batches, max_steps), device)):   # pinned, double-buffered H2D
        if fused:
            loss = module.fused_train_step(batch, t, opt, sched)
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
