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


def assert_close_rounded(a, b, ulp, rtol, atol, frac=2e-3, what=""):
    """Values that passed through a 16-bit rounding of an fp32 accumulator (the encoder product of the 16-bit modes): two
    correct evaluations differ in summation order, so a small share of the elements lands on the neighbouring 16-bit
    value.  Every element within one unit of the 16-bit format (relative `ulp`) of the reference, and all but `frac` of them
    within the tight (rtol, atol)."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    assert bool((err <= atol + ulp * b.abs()).all()), f"{what}: off by more than one 16-bit unit (max |err| {float(err.max()):.3g})"
    loose = float((err > atol + rtol * b.abs()).double().mean())
    assert loose <= frac, f"{what}: {loose:.3g} of the elements differ beyond fp32 noise (allowed {frac:.3g} rounding flips)"


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


# ------------------------------------------------------------------------------------------------------------------
# Autocast fixtures (tests/golden/make_golden_autocast.py): the REFERENCE run under torch.autocast("cpu", bf16 / fp16)
# (+ torch.amp.GradScaler("cpu") for fp16) at a size where 16-bit rounding matters.  Inputs and initial parameters are
# too large to commit (> 1 MB per case), so both the generator and the tests build them here from seeded CPU generators;
# every fixture carries checksums of what the generator saw, and `autocast_case` refuses to hand out inputs that differ.
# ------------------------------------------------------------------------------------------------------------------
AC_SHAPE = dict(B=256, L=3, d=128, F=512)
AC_CASES = ["ac_jumprelu_clt", "ac_batchtopk_clt_base", "ac_topk_clt_aux", "ac_jumprelu_plt_notanh"]
AC_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}
AC_STEPS = 2


def _ac_cfg(name):
    F = AC_SHAPE["F"]
    if name == "ac_jumprelu_clt":          # config/debug-cpu.yaml's loss settings (tanh, lambda 0.05, c 1, pre-act 1e-6 -> 1e-3)
        return ({"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "jumprelu", "theta": 0.05, "bandwidth": 0.1},
                 "loss": {"kind": "jumprelu", "lambda_sparsity": 0.05, "c_sparsity": 1.0, "use_tanh": True,
                          "pre_actv_loss": 1e-3}}, {"lr": 1e-3})
    if name == "ac_jumprelu_plt_notanh":
        return ({"decoder": "plt", "d_features": F, "nonlinearity": {"kind": "jumprelu", "theta": 0.1, "bandwidth": 0.05},
                 "loss": {"kind": "jumprelu", "lambda_sparsity": 0.02, "c_sparsity": 0.7, "use_tanh": False,
                          "pre_actv_loss": 1e-2}}, {"lr": 2e-3, "optimizer": "adamw", "beta1": 0.8, "beta2": 0.95})
    if name == "ac_batchtopk_clt_base":
        return ({"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "batchtopk", "k": 16, "e": 0.1},
                 "loss": {"kind": "base"}}, {"lr": 1e-3})
    if name == "ac_topk_clt_aux":
        return ({"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "topk", "k": 16},
                 "loss": {"kind": "topk", "topk_aux": {"kind": "topk", "k": 32}, "tokens_till_dead": 1000,
                          "aux_loss_scale": 1 / 32}}, {"lr": 1e-3})
    raise KeyError(name)


def autocast_case_inputs(name):
    """cfg, opt, batches [steps, B, 2, L, d], init_params (reference state-dict names), last_active0 — all from seeds."""
    cfg, opt = _ac_cfg(name)
    B, L, d, F = (AC_SHAPE[k] for k in "BLdF")
    g = torch.Generator().manual_seed(1000 + AC_CASES.index(name))
    A = torch.randn(L, d, generator=g)
    bs = []
    for _ in range(AC_STEPS):
        x = torch.randn(B, L, d, generator=g) * 1.5 + 0.3
        y = 0.5 * x * A + 0.1 * torch.randn(B, L, d, generator=g)
        bs.append(torch.stack([x, y], dim=1))
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5,
              "encoder.b": torch.randn(L, F, generator=g) * 0.01,
              "decoder.b": torch.randn(L, d, generator=g) * 0.01}
    bd = (d * L) ** -0.5
    if cfg["decoder"] == "clt":
        for l in range(L):
            params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * bd
    else:
        params["decoder.W"] = (torch.rand(L, F, d, generator=g) * 2 - 1) * bd
    if cfg["nonlinearity"]["kind"] == "jumprelu":
        th = cfg["nonlinearity"]["theta"]
        params["nonlinearity.theta"] = th * (1 + 0.2 * (torch.rand(1, L, F, generator=g) * 2 - 1))
    la0 = torch.zeros(L, F, dtype=torch.long)
    if cfg["loss"]["kind"] == "topk":
        # every fifth feature cannot fire (bias five sigma below the row's top 16) and has been silent for 5000 tokens: dead
        params["encoder.b"][:, ::5] = -1.5
        la0[:, ::5] = 5000
        la0[:, 1::7] = 5000          # these fire in the batch: alive again after update_dead_features
    return cfg, opt, torch.stack(bs), params, la0


def tensor_checksum(t):
    t = t.detach().double().reshape(-1)
    w = torch.arange(1, t.numel() + 1, dtype=torch.float64) % 1009
    return torch.stack([t.sum(), (t * w).sum(), t.abs().max()])


def autocast_case(name, mode, load_golden):
    """The fixture of (case, "bf16" | "fp16") completed with the regenerated inputs, in the layout of the fp32 goldens
    (so build_module / init_buffers take it)."""
    fx = load_golden(f"{name}_{mode}")
    cfg, opt, batches, params, la0 = autocast_case_inputs(name)
    # (fp64 sums: their last bits depend on the host's vector width, hence a tolerance far below any real difference)
    same = lambda a, b: torch.allclose(a, b, rtol=1e-10, atol=1e-7)  # noqa: E731
    assert same(tensor_checksum(batches), fx["checksums"]["batches"]), "regenerated batches differ from the generator's"
    for k, p in params.items():
        assert same(tensor_checksum(p), fx["checksums"][k]), f"regenerated {k} differs from the generator's"
    return dict(fx, cfg=cfg, opt=opt, batches=batches, init_params=params, last_active0=la0, max_steps=fx["max_steps"])


def sampled(fx, key, t):
    """The entries of t at the positions the fixture sampled for `key`."""
    return t.detach().cpu().double().reshape(-1)[fx["sample_idx"][key].long()]


def sample_rel_l2(fx, key, got, rec):
    """Relative l2 error of `got` on the fixture's sampled entries of `key` (rec = {"val", "norm"})."""
    w = rec["val"].double()
    return float((sampled(fx, key, got) - w).norm() / (w.norm() + 1e-30))
