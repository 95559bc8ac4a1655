"""Generate golden vectors by running the REFERENCE (autograd + torch.optim) in the build container.

    python tests/golden/make_golden.py          # writes tests/golden/*.pt

Only data (inputs, initial parameters, expected outputs) is written; no reference source is copied.
The fixtures are loaded with torch.load(weights_only=True).  `tests/debug_actvs.pt` of the
reference (a data file, [16,2,2,110] fp32) is embedded in the `dbg_jumprelu_clt` fixture as the
input batch because BASELINE.json configs[0] is defined on it.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference  # noqa: E402

mods = load_reference()
clt = mods["crosslayer_transcoder.model.clt"]
jr = mods["crosslayer_transcoder.model.jumprelu"]
tk = mods["crosslayer_transcoder.model.topk"]
sd = mods["crosslayer_transcoder.model.standardize"]
cl = mods["crosslayer_transcoder.model.clt_lightning"]


def build_nl(nl, L, F):
    k = nl["kind"]
    if k == "relu":
        return jr.ReLU()
    if k == "jumprelu":
        return jr.JumpReLU(theta=nl["theta"], bandwidth=nl["bandwidth"], n_layers=L, d_features=F)
    if k == "topk":
        return tk.PerLayerTopK(k=nl["k"])
    if k == "batchtopk":
        return tk.BatchTopK(k=nl["k"], e=nl["e"])
    if k == "perlayerbatchtopk":
        return tk.PerLayerBatchTopK(k=nl["k"], e=nl["e"], n_layers=L)
    raise ValueError(k)


def build_module(cfg, L, d, F, opt):
    dec_cls = clt.CrosslayerDecoder if cfg["decoder"] == "clt" else clt.Decoder
    model = clt.CrossLayerTranscoder(
        nonlinearity=build_nl(cfg["nonlinearity"], L, F),
        encoder=clt.Encoder(d, F, L),
        decoder=dec_cls(d, F, L),
        input_standardizer=sd.DimensionwiseInputStandardizer(L, d),
        output_standardizer=sd.DimensionwiseOutputStandardizer(L, d),
    )
    lc = cfg["loss"]
    common = dict(model=model, learning_rate=opt["lr"], warmup_steps=opt.get("warmup_steps", 0),
                  lr_decay_step=opt.get("lr_decay_step", 0), lr_decay_factor=opt.get("lr_decay_factor", 1.0),
                  optimizer=opt.get("optimizer", "adam"), beta1=opt.get("beta1"), beta2=opt.get("beta2"),
                  log_metrics_every=10 ** 9)
    if lc["kind"] == "base":
        return cl.CrossLayerTranscoderModule(tokens_till_dead=lc.get("tokens_till_dead", 1_000_000), **common)
    if lc["kind"] == "jumprelu":
        return cl.JumpReLUCrossLayerTranscoderModule(
            lambda_sparsity=lc["lambda_sparsity"], c_sparsity=lc["c_sparsity"], use_tanh=lc["use_tanh"],
            pre_actv_loss=lc["pre_actv_loss"], **common)
    if lc["kind"] == "topk":
        return cl.TopKCrossLayerTranscoderModule(
            topk_aux=build_nl(lc["topk_aux"], L, F), tokens_till_dead=lc["tokens_till_dead"],
            aux_loss_scale=lc["aux_loss_scale"], **common)
    raise ValueError(lc["kind"])


def run_case(name, cfg, opt, batches, max_steps, last_active0=None, seed=42):
    torch.manual_seed(seed)
    B, _, L, d = batches[0].shape
    F = cfg["d_features"]
    mod = build_module(cfg, L, d, F, opt)
    mod.trainer.max_steps = max_steps
    if last_active0 is not None:
        mod.last_active.copy_(last_active0)
    out = {"cfg": cfg, "opt": opt, "max_steps": max_steps, "batches": torch.stack(batches),
           "init_params": {k: v.detach().clone() for k, v in mod.model.named_parameters()},
           "last_active0": mod.last_active.clone(), "steps": []}
    conf = mod.configure_optimizers()
    if isinstance(conf, dict):
        optimizer, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    else:
        optimizer, sched = conf, None
    for t, batch in enumerate(batches):
        mod.global_step = t
        mod.logged = {}
        # batch_idx == 0 also triggers log_training_metrics (0 % n == 0); skip it with a huge odd idx
        # after the first step, exactly as the loop would for batch_idx % log_metrics_every != 0.
        optimizer.zero_grad()
        loss = mod.training_step(batch, t)
        loss.backward()
        if t == 0:   # intermediates, from a second forward that must not advance the EMA buffer
            nlm = mod.model.nonlinearity
            keep = nlm.threshold.clone() if hasattr(nlm, "threshold") else None
            with torch.no_grad():
                pre, feat, rn, rc = mod.forward(batch[:, 0])
            if keep is not None:
                nlm.threshold = keep
        rec = {"loss": loss.detach().clone(),
               "logged": {k: (v if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in mod.logged.items()
                          if not k.startswith(("layers/", "model/encoder", "model/decoder", "metrics/train",
                                               "metrics/fraction", "metrics/recons", "metrics/L0"))},
               "lr": torch.tensor(optimizer.param_groups[0]["lr"]),
               "grads": {k: p.grad.detach().clone() for k, p in mod.model.named_parameters()},
               "last_active": mod.last_active.clone()}
        if t == 0:
            rec["pre"] = pre.detach().clone()
            rec["features"] = feat.detach().clone()
            rec["recons_norm"] = rn.detach().clone()
            rec["recons"] = rc.detach().clone()
            rec["std"] = {"in_mean": mod.model.input_standardizer.mean.clone(),
                          "in_std": mod.model.input_standardizer.std.clone(),
                          "out_mean": mod.model.output_standardizer.mean.clone(),
                          "out_std": mod.model.output_standardizer.std.clone()}
        optimizer.step()
        if sched is not None:
            sched.step()
        nlm = mod.model.nonlinearity
        if hasattr(nlm, "threshold"):
            rec["threshold"] = nlm.threshold.detach().clone()
        if hasattr(mod, "topk_aux") and hasattr(mod.topk_aux, "threshold"):
            rec["aux_threshold"] = mod.topk_aux.threshold.detach().clone()
        out["steps"].append(rec)
    out["final_params"] = {k: v.detach().clone() for k, v in mod.model.named_parameters()}
    path = os.path.join(HERE, f"{name}.pt")
    torch.save(out, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, losses", [float(s['loss']) for s in out['steps']])


def synth(B, L, d, steps, seed=7):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(L, d, generator=g)
    bs = []
    for _ in range(steps):
        x = torch.randn(B, L, d, generator=g) * 1.5 + 0.3
        y = 0.5 * x * A + 0.1 * torch.randn(B, L, d, generator=g)
        bs.append(torch.stack([x, y], dim=1))
    return bs


def main():
    # ---- configs[0]: config/debug-cpu.yaml on tests/debug_actvs.pt ---------------------------
    acts = torch.load("/root/reference/tests/debug_actvs.pt", weights_only=True)
    dbg = {"decoder": "clt", "d_features": 32,
           "nonlinearity": {"kind": "jumprelu", "theta": 0.03, "bandwidth": 0.01},
           "loss": {"kind": "jumprelu", "lambda_sparsity": 0.05, "c_sparsity": 1, "use_tanh": True,
                    "pre_actv_loss": 1e-6}}
    run_case("dbg_jumprelu_clt", dbg, {"lr": 1e-3, "lr_decay_step": 2, "lr_decay_factor": 0.1},
             [acts, acts.flip(0), acts.roll(3, 0), acts], max_steps=100)

    L, d, F, B = 3, 12, 24, 6
    jp = {"decoder": "plt", "d_features": F,
          "nonlinearity": {"kind": "jumprelu", "theta": 0.05, "bandwidth": 0.2},
          "loss": {"kind": "jumprelu", "lambda_sparsity": 0.3, "c_sparsity": 0.7, "use_tanh": False,
                   "pre_actv_loss": 1e-2}}
    run_case("jumprelu_plt_notanh", jp, {"lr": 2e-3, "warmup_steps": 3, "optimizer": "adamw", "beta1": 0.8,
                                         "beta2": 0.95},
             synth(B, L, d, 3), max_steps=4)
    rb = {"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "relu"}, "loss": {"kind": "base"}}
    run_case("relu_clt_base", rb, {"lr": 1e-3}, synth(B, L, d, 2, seed=8), max_steps=10)

    # ---- TopK family with aux loss ---------------------------------------------------------
    L, d, F, B = 3, 10, 32, 8
    la0 = torch.zeros(L, F, dtype=torch.long)
    la0[:, ::3] = 50          # every third feature starts "dead" unless it fires
    la0[1, 5:20] = 50
    for kind, nl in (("topk", {"kind": "topk", "k": 3}),
                     ("batchtopk", {"kind": "batchtopk", "k": 3, "e": 0.1}),
                     ("perlayerbatchtopk", {"kind": "perlayerbatchtopk", "k": 3, "e": 0.05})):
        cfg = {"decoder": "clt", "d_features": F, "nonlinearity": nl,
               "loss": {"kind": "topk", "topk_aux": {"kind": "topk", "k": 6}, "tokens_till_dead": 40,
                        "aux_loss_scale": 1 / 32}}
        run_case(f"{kind}_clt_aux", cfg, {"lr": 1e-3}, synth(B, L, d, 3, seed=11), max_steps=10,
                 last_active0=la0)
    # aux k larger than the number of dead features (exercises the -inf entries), PLT decoder
    la1 = torch.zeros(L, F, dtype=torch.long)
    la1[:, :4] = 50
    cfg = {"decoder": "plt", "d_features": F, "nonlinearity": {"kind": "topk", "k": 4},
           "loss": {"kind": "topk", "topk_aux": {"kind": "topk", "k": 9}, "tokens_till_dead": 40,
                    "aux_loss_scale": 0.5}}
    run_case("topk_plt_aux_fewdead", cfg, {"lr": 1e-3}, synth(B, L, d, 2, seed=12), max_steps=10,
             last_active0=la1)
    cfg = {"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "batchtopk", "k": 2, "e": 0.1},
           "loss": {"kind": "base"}}
    run_case("batchtopk_clt_base", cfg, {"lr": 1e-3}, synth(B, L, d, 2, seed=13), max_steps=10)

    # ---- selection-only vectors (tie-free by construction: a permutation of distinct values) ----
    g = torch.Generator().manual_seed(3)
    sel = {}
    for (Bs, Ls, Fs, k) in ((4, 3, 64, 5), (2, 12, 768, 16), (5, 2, 100, 100), (3, 4, 37, 1)):
        n = Bs * Ls * Fs
        vals = (torch.randperm(n, generator=g).float() - n / 3) / 17.0
        x = vals.reshape(Bs, Ls, Fs)
        key = f"{Bs}x{Ls}x{Fs}_k{k}"
        sel[key] = {"x": x, "k": torch.tensor(k),
                    "topk": tk.PerLayerTopK(k=k)(x),
                    "batchtopk": tk.BatchTopK(k=k, e=0.1)(x),
                    "perlayerbatchtopk": tk.PerLayerBatchTopK(k=k, e=0.1, n_layers=Ls)(x)}
        m = tk.BatchTopK(k=k, e=0.1)
        m(x)
        sel[key]["batchtopk_threshold"] = m.threshold.clone()
        m = tk.PerLayerBatchTopK(k=k, e=0.1, n_layers=Ls)
        m(x)
        sel[key]["perlayerbatchtopk_threshold"] = m.threshold.clone()
    torch.save(sel, os.path.join(HERE, "selection.pt"))
    print("selection:", os.path.getsize(os.path.join(HERE, "selection.pt")) // 1024, "KiB")

    # ---- JumpReLU function alone (fwd + STE bwd), incl. points inside the rectangle window ------
    torch.manual_seed(5)
    x = (torch.randn(7, 3, 16) * 0.1).requires_grad_()
    m = jr.JumpReLU(theta=0.05, bandwidth=0.04, n_layers=3, d_features=16)
    with torch.no_grad():
        m.theta.add_(torch.randn_like(m.theta) * 0.02)
    go = torch.randn(7, 3, 16)
    out = m(x)
    out.backward(go)
    torch.save({"x": x.detach(), "theta": m.theta.detach(), "bandwidth": m.bandwidth.clone(), "grad_out": go,
                "out": out.detach(), "dx": x.grad.clone(), "dtheta": m.theta.grad.clone()},
               os.path.join(HERE, "jumprelu_fn.pt"))


if __name__ == "__main__":
    main()
