"""Shared helpers: replay a golden case (tests/golden/*.pt) through a step implementation."""
import torch

TRAIN_CASES = ["dbg_jumprelu_clt", "jumprelu_plt_notanh", "relu_clt_base", "topk_clt_aux", "batchtopk_clt_aux",
               "perlayerbatchtopk_clt_aux", "topk_plt_aux_fewdead", "batchtopk_clt_base"]


def init_buffers(case):
    cfg = case["cfg"]
    L = case["batches"].shape[3]
    buf = {"last_active": case["last_active0"].clone()}
    kind = cfg["nonlinearity"]["kind"]
    if kind == "batchtopk":
        buf["threshold"] = torch.zeros(1)
    elif kind == "perlayerbatchtopk":
        buf["threshold"] = torch.zeros(L)
    return buf


def assert_close(a, b, rtol, atol, what=""):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.tensor(float(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.tensor(float(b)).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    if not bool((err <= tol).all()):
        i = int((err - tol).argmax())
        raise AssertionError(f"{what}: max violation at flat {i}: got {a.reshape(-1)[i].item():.9g} want "
                             f"{b.reshape(-1)[i].item():.9g} (|err| {err.reshape(-1)[i].item():.3g}, "
                             f"max|want| {b.abs().max().item():.3g})")


def check_first_adam_step(name, opt, params, case, rec, sl=None):
    """After the FIRST optimizer step — wherever it ran: inside the backward row pass (weight matrices) or in
    clt_adam_f32 — the moments and the parameters against oracle.adam_step on the REFERENCE's gradient.  exp_avg is
    linear and exp_avg_sq quadratic in the gradient, so they pin the fused gradient itself (rtol 1e-5 + the gradient's
    absolute floor 2e-6 * max|g|: fp32 reduction-order noise); the parameter pins bias correction, eps placement,
    weight decay and lr wherever the update is well conditioned (|g| >> eps), and is within one lr elsewhere.
    sl(name, tensor): this rank's slice of a full-size tensor (feature-sharded runs)."""
    from oracle import clt_oracle as O
    o = case["opt"]
    b1 = 0.9 if o.get("beta1") is None else o["beta1"]
    b2 = 0.999 if o.get("beta2") is None else o["beta2"]
    adamw = o.get("optimizer", "adam") == "adamw"
    lr = float(rec["lr"])
    for k, p in params.items():
        g, p0 = rec["grads"][k], case["init_params"][k]
        gmax = float(g.abs().max())
        want_p, want_m, want_v = O.adam_step(p0, g, torch.zeros_like(p0), torch.zeros_like(p0), 1, lr, b1, b2, 1e-8,
                                             0.01 if adamw else 0.0, adamw)
        if sl is not None:
            g, want_p, want_m, want_v = (sl(k, t) for t in (g, want_p, want_m, want_v))
        st = opt.state[p]
        assert int(st["step"]) == 1, f"{name} {k}: optimizer step count {st['step']}"
        assert_close(st["exp_avg"], want_m, 1e-5, 2e-6 * (1 - b1) * gmax + 1e-30, f"{name} exp_avg {k}")
        assert_close(st["exp_avg_sq"], want_v, 2e-5, 4e-6 * (1 - b2) * gmax * gmax + 1e-36, f"{name} exp_avg_sq {k}")
        well = g.abs() > 1e-3 * gmax
        got = p.detach().cpu()
        assert_close(got[well], want_p[well], 1e-6, 2e-3 * lr, f"{name} {k} after Adam step 1 (well-conditioned entries)")
        assert_close(got, want_p, 1e-6, 1.01 * lr, f"{name} {k} after Adam step 1")
