"""`precision: "16-mixed"` (the reference's production setting, config/default.yaml:13 = torch.autocast(float16) +
torch.amp.GradScaler): fp16 GEMM operands with fp32 accumulation and the library's own dynamic loss scaling.
Checked against the oracle run with fp16 operand rounding and the same loss scale on the gradient operands
(oracle.operand_rounding(torch.float16, grad_scale)); the reference itself cannot run its fp16 path on this CPU-only
oracle side, so this mode's parity is pinned to the oracle's restatement of autocast's rounding points, not to a
reference-generated fixture ("parity unpinned" for fp16 — the fp32 / bf16 modes are pinned)."""
import pytest
import torch

from tests.golden_cases import assert_close

pytestmark = pytest.mark.gpu

B, L, d, F = 512, 3, 128, 512


def _case(dec_kind, nl_kind, loss_kind):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, L, d, generator=g)
    y = 0.5 * x * torch.randn(L, d, generator=g) + 0.1 * torch.randn(B, L, d, generator=g)
    batch = torch.stack([x, y], 1)
    nl = {"relu": {"kind": "relu"}, "jumprelu": {"kind": "jumprelu", "theta": 0.05, "bandwidth": 0.1},
          "batchtopk": {"kind": "batchtopk", "k": 8, "e": 0.1}}[nl_kind]
    lc = {"kind": "base"} if loss_kind == "base" else {"kind": "jumprelu", "lambda_sparsity": 0.05, "c_sparsity": 1.0,
                                                       "use_tanh": True, "pre_actv_loss": 1e-3}
    cfg = {"decoder": dec_kind, "d_features": F, "nonlinearity": nl, "loss": lc}
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.randn(L, F, generator=g) * 0.01,
              "decoder.b": torch.randn(L, d, generator=g) * 0.01}
    bd = (d * L) ** -0.5
    if dec_kind == "clt":
        for l in range(L):
            params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * bd
    else:
        params["decoder.W"] = (torch.rand(L, F, d, generator=g) * 2 - 1) * bd
    if nl_kind == "jumprelu":
        params["nonlinearity.theta"] = torch.full((1, L, F), 0.05)
    return {"cfg": cfg, "opt": {"lr": 1e-3}, "batches": batch[None], "init_params": params,
            "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}


def _oracle(case, scale, step_frac=0.5):
    from oracle import clt_oracle as O
    batch = case["batches"][0]
    c = dict(case["cfg"])
    c["loss"] = dict(c["loss"], step_frac=step_frac) if c["loss"]["kind"] == "jumprelu" else c["loss"]
    buf = {"last_active": torch.zeros(L, F, dtype=torch.long), "threshold": torch.zeros(1)}
    with O.operand_rounding(torch.float16, grad_scale=scale):
        return O.training_step(batch, case["init_params"], O.standardizer_init(batch), c, buf)


@pytest.mark.parametrize("dec_kind,nl_kind,loss_kind,sparse", [("clt", "batchtopk", "base", True), ("clt", "batchtopk", "base", False),
                                                               ("clt", "jumprelu", "jumprelu", False), ("plt", "relu", "base", None),
                                                               ("plt", "jumprelu", "jumprelu", False)])
def test_fp16_step_loss_and_gradients_vs_oracle(dec_kind, nl_kind, loss_kind, sparse):
    """Loss and every gradient (handed out carrying the loss scale, `out["grad_scale"]`) of one step in fp16 mode."""
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    case = _case(dec_kind, nl_kind, loss_kind)
    mod = build_module(case)
    mod.global_step = 5
    mod.log_metrics_every = 0
    with precision("16-mixed"):
        out = mod.compute_step(case["batches"][0].cuda(), 0, sparse=sparse)
    S = out["grad_scale"]
    assert S == 65536.0 == mod.loss_scaler.scale
    ref = _oracle(case, S)
    assert_close(out["loss"], ref["loss"], 1e-3, 1e-6, "fp16 loss")
    for k, gref in ref["grads"].items():
        gg = out["grads"][k].reshape(gref.shape).cpu() / S
        rel = (gg - gref).norm() / (gref.norm() + 1e-12)
        assert rel < 4e-3, f"{dec_kind}/{nl_kind} fp16 grad {k}: rel l2 err {rel:.3g}"
    # what the scale is for: rounded unscaled, dL/dr (~1e-5 at this size, ~1e-8 at B = 4096, L = 12, d = 768) sits in fp16's
    # subnormals, where it keeps only a few bits
    if loss_kind == "base":
        lost = _oracle(case, 1.0)["grads"]["encoder.W"]
        assert (lost - ref["grads"]["encoder.W"]).norm() > 2e-3 * ref["grads"]["encoder.W"].norm()


@pytest.mark.parametrize("dec_kind,nl_kind,loss_kind", [("clt", "batchtopk", "base"), ("clt", "jumprelu", "jumprelu")])
def test_fp16_fused_training_step_scaler_and_skip(dec_kind, nl_kind, loss_kind):
    """training_step in fp16 mode: Adam applied inside the kernels on the unscaled gradient (moments vs oracle.adam_step);
    an overflowing step changes nothing, gives the step counts back and halves the scale; the next step trains again."""
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import precision
    from crosslayer_transcoder_amd.precision import LossScaler
    from oracle import clt_oracle as O
    from tests.build_cases import build_module
    case = _case(dec_kind, nl_kind, loss_kind)
    batch = case["batches"][0].cuda()
    with precision("16-mixed"):
        mod = build_module(case)
        mod.log_metrics_every = 0
        opt, _ = mod.attach_optimizers()
        params = dict(mod.model.named_parameters())
        # ---- an overflowing first step: scale so large that the scaled dL/dr leaves fp16's range
        mod.loss_scaler = LossScaler(init_scale=2.0 ** 40)
        before = {k: p.detach().clone() for k, p in params.items()}
        mod.training_step(batch, 0)
        assert mod.loss_scaler.scale == 2.0 ** 39 and mod.loss_scaler.skipped_steps == 1
        for k, p in params.items():
            assert torch.equal(p.detach(), before[k]), f"skipped step changed {k}"
            st = opt.state.get(p, {})
            assert int(st.get("step", 0)) == 0, k
            if "exp_avg" in st:
                assert not st["exp_avg"].any() and not st["exp_avg_sq"].any(), k
        # ---- a clean step at the default scale
        mod.loss_scaler = LossScaler()
        mod.global_step = 0
        loss = mod.training_step(batch, 0)
        assert mod.loss_scaler.scale == 65536.0 and mod.loss_scaler.growth_tracker == 1 and torch.isfinite(loss)
    ref = _oracle(case, 65536.0, step_frac=0.0)
    assert_close(loss, ref["loss"], 1e-3, 1e-6, "fp16 fused loss")
    for k, p in params.items():
        g, p0 = ref["grads"][k], case["init_params"][k]
        _, want_m, want_v = O.adam_step(p0, g, torch.zeros_like(p0), torch.zeros_like(p0), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, False)
        st = opt.state[p]
        assert int(st["step"]) == 1, k
        rel_m = float((st["exp_avg"].cpu().reshape(want_m.shape) - want_m).norm() / (want_m.norm() + 1e-30))
        rel_v = float((st["exp_avg_sq"].cpu().reshape(want_v.shape) - want_v).norm() / (want_v.norm() + 1e-30))
        # (theta: its gradient counts the entries inside the rectangle window around theta; with the pre-activations kept as
        # fp16 — as autocast keeps them — entries at the window's edges switch sides)
        tol = 1.5e-2 if k == "nonlinearity.theta" else 4e-3
        assert rel_m < tol and rel_v < 2 * tol, f"{k}: exp_avg {rel_m:.3g} exp_avg_sq {rel_v:.3g}"
        step = (p.detach().cpu() - p0).abs().max()
        assert 0 < float(step) <= 1.01e-3, f"{k}: first Adam step moved by {float(step):.3g}"
    # the 16-bit operand shadows the weight passes refreshed are fp16 copies of the new weights
    enc = mod.model.encoder
    if getattr(enc, "_bf16_shadow", None) is not None and loss_kind == "base":
        assert enc._bf16_shadow.dtype == torch.float16
        assert torch.equal(enc._bf16_shadow, enc.Wt.detach().half())
    cta.set_precision("32")


def test_bf16_mode_unaffected_after_fp16():
    """Switching back restores bf16 operands (library state follows the precision setting)."""
    from crosslayer_transcoder_amd import ops, precision
    x = torch.randn(64, 64).cuda() * 1e-6
    with precision("16-mixed"):
        ops.sync_operand_format()
        h = ops.cast_bf16(x)
        assert h.dtype == torch.float16 and torch.equal(h, x.half())
    with precision("bf16-mixed"):
        ops.sync_operand_format()
        h = ops.cast_bf16(x)
        assert h.dtype == torch.bfloat16 and torch.equal(h, x.bfloat16())


def test_debug_yaml_in_16_mixed_follows_the_fp32_curve(tmp_path, golden):
    """config/debug-cpu.yaml with `trainer.precision: "16-mixed"` (what the reference's production YAMLs set): 100 fused
    steps from the reference's initial parameters stay on the reference's fp32 loss curve at the 16-bit tolerance (d_acts =
    110 is not a multiple of 8: the contractions go through the generic fp16 MFMA kernel), the loss scale never backs off."""
    import os
    from crosslayer_transcoder_amd import get_precision, set_precision, trainer
    cur = golden("dbg_yaml_curve")
    tfile = str(tmp_path / "debug_actvs.pt")
    torch.save(golden("dbg_jumprelu_clt")["batches"][0], tfile)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = trainer.load_config(os.path.join(root, "config", "debug-cpu.yaml"))
    cfg["trainer"]["precision"] = "16-mixed"
    try:
        mod, losses = trainer.run_config(cfg, tensor_file=tfile, init_state=cur["init_params"], fused=True)
        assert get_precision() == "fp16" and len(losses) == 100
        got = torch.stack([x.float().cpu() for x in losses])
        assert torch.isfinite(got).all()
        assert_close(got, cur["losses"], 2e-2, 1e-6, "100-step curve in fp16 mode vs the reference's fp32 curve")
        assert mod.loss_scaler.skipped_steps == 0 and mod.loss_scaler.scale == 65536.0
    finally:
        set_precision("32")


def test_fp16_gemm_epilogues_raise_the_overflow_flag():
    """The GEMM translation units hold the loss scaler's state too: an fp16 store epilogue (clt_gemm_bf16_nt_out16) or the
    rounded encoder product (CLT_GEMM_ROUND_ACC, fast and generic kernels) that overflows fp16 raises the flag."""
    from crosslayer_transcoder_amd import ops, precision
    g = torch.Generator().manual_seed(3)
    Bn, L, d, F = 256, 2, 64, 256
    xs = torch.randn(Bn, L, d, generator=g).cuda()
    W = torch.randn(L, F, d, generator=g).cuda()
    b = torch.zeros(L, F).cuda()
    with precision("16-mixed"):
        flag = ops.inf_flag(xs.device)
        for scale, want in ((1.0, 0), (300.0, 1)):          # |x| * 300 * |w| * 300 summed over 64 terms >> 65504
            x16, w16 = ops.cast_bf16(xs * scale), ops.cast_bf16(W * scale)
            for call in (lambda: ops.encoder_fwd_out16(x16, w16, b), lambda: ops.encoder_fwd(x16, w16, b),
                         lambda: ops.encoder_fwd(x16[:100].contiguous(), w16[:, :100].contiguous(), b[:, :100].contiguous())):   # (generic kernel)
                flag.zero_()
                ops.set_loss_scaling(True, 65536.0, flag)
                assert call() is not None
                assert int(flag.item()) == want, f"scale {scale}: flag {int(flag.item())}"
    ops.set_loss_scaling(False)


def test_fp16_overflowing_pre_activations_skip_the_step():
    """A JumpReLU module whose 16-bit pre-activation tensor overflows fp16 in the encoder GEMM's store epilogue: the step is
    skipped as a whole (nothing changes, step counts given back) and the loss scale halves."""
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    case = _case("plt", "jumprelu", "jumprelu")
    case["init_params"]["encoder.W"] = case["init_params"]["encoder.W"] * 3e5        # pre ~ N(0, (0.3 * 3e5)^2): far beyond 65504
    batch = case["batches"][0].cuda()
    try:
        with precision("16-mixed"):
            mod = build_module(case)
            mod.log_metrics_every = 0
            opt, _ = mod.attach_optimizers()
            before = {k: p.detach().clone() for k, p in mod.model.named_parameters()}
            mod.training_step(batch, 0)
            assert mod.loss_scaler.skipped_steps == 1 and mod.loss_scaler.scale == 32768.0
            for k, p in mod.model.named_parameters():
                assert torch.equal(p.detach(), before[k]), f"skipped step changed {k}"
                assert int(opt.state.get(p, {}).get("step", 0)) == 0, k
    finally:
        cta.set_precision("32")
