"""Golden vectors for the logging side outputs (reference clt_lightning.py:131-280 `log_training_metrics`,
metrics/dead_features.py): every value the reference logs at step 0 of three golden configurations, plus the
DeadFeatures statistics after two updates.  Run in the build container only:

    python tests/golden/make_golden_metrics.py      # writes tests/golden/metrics.pt

Inputs / parameters are those of the corresponding training fixtures (same seeds, same builders).
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference modules through _ref_import)


class _Metric(nn.Module):          # torchmetrics.Metric stand-in: add_state only
    def __init__(self, **kw):
        super().__init__()

    def add_state(self, name, default, dist_reduce_fx=None):
        setattr(self, name, default.clone())


def _dead_features_cls():
    sys.modules["torchmetrics"] = types.ModuleType("torchmetrics")
    sys.modules["torchmetrics"].Metric = _Metric
    spec = importlib.util.spec_from_file_location("ref_dead_features", "/root/reference/crosslayer_transcoder/metrics/dead_features.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.DeadFeatures


def first_step(name):
    case = torch.load(os.path.join(HERE, f"{name}.pt"), weights_only=True)
    cfg, opt = case["cfg"], case["opt"]
    B, _, L, d = case["batches"][0].shape
    torch.manual_seed(42)
    mod = mg.build_module(cfg, L, d, cfg["d_features"], opt)
    mod.trainer.max_steps = int(case["max_steps"])
    mod.last_active.copy_(case["last_active0"])
    with torch.no_grad():
        for k, p in mod.model.named_parameters():
            assert torch.equal(p, case["init_params"][k]), k        # same init as the training fixture
    mod.log_metrics_every = 50
    mod.logged = {}
    mod.global_step = 0
    mod.training_step(case["batches"][0], 0)
    logged = {k: (v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in mod.logged.items()}
    DF = _dead_features_cls()
    df = DF(n_features=cfg["d_features"], n_layers=L, return_per_layer=True, return_neuron_indices=True, return_log_freqs=True)
    with torch.no_grad():
        for t in range(2):
            _, feat, _, _ = mod.forward(case["batches"][t][:, 0])
            df.update(feat)
    stats = {k: v.clone() for k, v in df.compute().items()}
    return {"logged": logged, "dead_features": stats, "n_active": df.n_active.clone(), "n_total": df.n_total.clone()}


def main():
    out = {name: first_step(name) for name in ("dbg_jumprelu_clt", "jumprelu_plt_notanh", "topk_clt_aux", "batchtopk_clt_base")}
    torch.save(out, os.path.join(HERE, "metrics.pt"))
    for n, r in out.items():
        print(n, len(r["logged"]), "logged values;", sorted(r["logged"])[:4], "...")


if __name__ == "__main__":
    main()
