"""GPU parity of whole training steps (fwd + bwd + Adam through the drop-in module classes and the C ABI)
against the golden vectors generated from the reference, fp32 mode: losses rtol 1e-5, gradients rtol 1e-4 with
an absolute floor of 1e-6 x max|grad| (fp32 reduction-order noise), selected feature sets bit-exact."""
import pytest
import torch

from tests.golden_cases import TRAIN_CASES, assert_close, check_first_adam_step

pytestmark = pytest.mark.gpu


def _opt_and_sched(mod):
    return mod.attach_optimizers()


@pytest.mark.parametrize("fused", ["training_step", False, True, "dense", "compact", "compact-grads", "lists", "lists-grads"],
                         ids=["training_step", "autograd-loop", "fused-step", "fused-dense-gemm", "fused-step-compact-aux",
                              "compact-aux-grads", "threshold-lists", "threshold-lists-grads"])
@pytest.mark.parametrize("name", TRAIN_CASES)
def test_training_matches_reference(name, fused, golden, monkeypatch):
    """Modes: "training_step" = the Lightning-facing contract alone (manual optimisation: forward + backward + Adam
    inside training_step, with the optimizer / scheduler objects of configure_optimizers()); "autograd-loop" =
    fused_step=False under the automatic-optimisation order; the others call fused_train_step with options.
    "threshold-lists": JumpReLU / ReLU forced into the activation-list regime (sparse=True: decoder forward by row
    gather, decoder weight gradient + dec_norms term + Adam in the fused row pass); "-grads": the same with
    materialised gradients."""
    from tests.build_cases import build_module
    lists = isinstance(fused, str) and fused.startswith("lists")
    if lists and name not in ("dbg_jumprelu_clt", "jumprelu_plt_notanh", "relu_clt_base"):
        pytest.skip("thresholding sparsifiers only")
    compact = isinstance(fused, str) and fused.startswith("compact")
    if compact:
        # the TopK-aux pass as a dense sub-problem on the compacted dead columns (engine.COMPACT_AUX): forced here —
        # at the golden sizes (F = 32) the padded dead-column count exceeds the share at which the step would choose it
        if "aux" not in name:
            pytest.skip("no aux loss in this case")
        from crosslayer_transcoder_amd import engine
        monkeypatch.setattr(engine, "COMPACT_AUX_MAX_SHARE", 1e9)
    case = golden(name)
    mod = build_module(case)
    if fused is False:
        mod.fused_step, mod.automatic_optimization = False, True
    opt, sched = _opt_and_sched(mod)
    params = dict(mod.model.named_parameters())
    for t, rec in enumerate(case["steps"]):
        batch = case["batches"][t].cuda()
        assert_close(torch.tensor(opt.param_groups[0]["lr"]), rec["lr"], 1e-6, 0, f"{name} lr {t}")
        if fused == "training_step":
            assert mod.fused_step and not mod.automatic_optimization
            loss = mod.training_step(batch, t)
            assert not loss.requires_grad
            grads = {k: mod.last_step["grads"][k].reshape(p.shape) for k, p in params.items() if k in mod.last_step["grads"]}
        elif lists:
            loss = mod.fused_train_step(batch, t, opt, sched, fuse_adam=(fused == "lists"), sparse=True)
            assert mod.last_step.get("lists") is not None
            grads = {k: mod.last_step["grads"][k].reshape(p.shape) for k, p in params.items() if k in mod.last_step["grads"]}
            assert (fused == "lists") != any(k.startswith("decoder.W") for k in grads)
        elif fused == "dense":                      # TopK family through the dense MFMA GEMM regime
            loss = mod.fused_train_step(batch, t, opt, sched, fuse_adam=False, sparse=False)
            grads = {k: mod.last_step["grads"][k].reshape(p.shape) for k, p in params.items()}
        elif fused == "compact-grads":
            loss = mod.fused_train_step(batch, t, opt, sched, fuse_adam=False)
            grads = {k: mod.last_step["grads"][k].reshape(p.shape) for k, p in params.items()}
        elif fused:
            loss = mod.fused_train_step(batch, t, opt, sched)
            # weight matrices updated inside the sparse row pass have no materialised gradient
            grads = {k: mod.last_step["grads"][k].reshape(p.shape) for k, p in params.items() if k in mod.last_step["grads"]}
        else:
            opt.zero_grad(set_to_none=True)
            loss = mod.training_step(batch, t)
            loss.backward()
            grads = {k: p.grad.clone() for k, p in params.items()}
            opt.step()
            if sched is not None:
                sched.step()
            mod.global_step += 1
        assert_close(loss, rec["loss"], 2e-5, 1e-7, f"{name} loss step {t}")
        out = mod.last_step
        if t == 0:
            assert torch.equal(out["features"].cpu() > 0, rec["features"] > 0), f"{name}: active set differs"
            assert_close(out["features"], rec["features"], 1e-5, 1e-6, "features")
            assert_close(out["recons_norm"], rec["recons_norm"], 1e-5, 2e-6, "recons_norm")
            assert_close(mod.model.input_standardizer.std, rec["std"]["in_std"], 1e-5, 1e-7, "in_std")
            for k, gref in rec["grads"].items():
                if k in grads:
                    assert_close(grads[k], gref, 1e-4, 2e-6 * float(gref.abs().max()) + 1e-12, f"{name} grad {k}")
            check_first_adam_step(name, opt, params, case, rec)
        assert torch.equal(mod.last_active.cpu(), rec["last_active"]), f"{name} last_active {t}"
        assert mod.global_step == t + 1
        lg = rec["logged"]
        if "training/aux_loss" in lg:
            assert_close(out["terms"]["aux"], lg["training/aux_loss"], 5e-5, 1e-7, "aux")
        if "training/sparsity_loss" in lg:
            assert_close(out["terms"]["sparsity"], lg["training/sparsity_loss"], 5e-5, 1e-8, "sparsity")
            assert_close(out["terms"]["pre_actv"], lg["training/pre_actv_loss"], 5e-5, 1e-10, "pre_actv")
        assert_close(out["terms"]["dead_fraction"], lg["metrics/dead_features"], 1e-6, 1e-7, "dead")
        if "threshold" in rec:
            assert_close(mod.model.nonlinearity.threshold, rec["threshold"], 1e-5, 1e-7, "threshold")
    # after 2-4 Adam steps parameters follow the reference trajectory (Adam's sign-like first steps amplify
    # gradient noise on near-zero gradients, hence the absolute term of ~lr/5; the tight check of the update
    # arithmetic is check_first_adam_step above)
    lr = case["opt"]["lr"]
    for k, p in params.items():
        assert_close(p, case["final_params"][k], 1e-4, 0.3 * lr, f"{name} final {k}")


def test_loss_backward_scales_handed_out_gradients(golden):
    """fused_step=False (automatic optimisation): the gradients that loss.backward() hands out follow the incoming
    gradient — a GradScaler or accumulate_grad_batches multiplies the loss (reference: plain autograd)."""
    from tests.build_cases import build_module
    case = golden("batchtopk_clt_base")
    grads = []
    for scale in (1.0, 2.0, 1.0 / 4):
        mod = build_module(case)
        mod.fused_step, mod.automatic_optimization = False, True
        loss = mod.training_step(case["batches"][0].cuda(), 0)
        (loss * scale).backward()
        grads.append({k: p.grad.clone() for k, p in mod.model.named_parameters()})
    for k, g in grads[0].items():
        assert float(g.abs().max()) > 0
        assert torch.equal(grads[1][k], 2.0 * g) and torch.equal(grads[2][k], 0.25 * g), k


def test_module_level_autograd_matches_reference(golden):
    """The un-fused drop-in path: CrossLayerTranscoder.forward + torch autograd through the per-op Functions."""
    from tests.build_cases import build_module
    for name in ("relu_clt_base", "batchtopk_clt_base", "jumprelu_plt_notanh"):
        case = golden(name)
        mod = build_module(case)
        batch = case["batches"][0].cuda()
        mod.model.initialize_standardizers(batch)
        pre, feat, rn, rc = mod.model(batch[:, 0])
        rec = case["steps"][0]
        assert_close(pre, rec["pre"], 1e-5, 1e-6, "pre"); assert_close(feat, rec["features"], 1e-5, 1e-6, "feat")
        assert_close(rn, rec["recons_norm"], 1e-5, 2e-6, "rn"); assert_close(rc, rec["recons"], 1e-5, 2e-6, "rc")
        if case["cfg"]["loss"]["kind"] == "base":
            y = mod.model.output_standardizer.standardize(batch[:, 1])
            # MSE through the HIP kernel so that no arithmetic here is torch's
            from crosslayer_transcoder_amd import ops
            _, g, _ = ops.mse(rn.detach(), batch[:, 1], mod.model.output_standardizer.mean, mod.model.output_standardizer.std)
            rn.backward(g)
            for k, p in mod.model.named_parameters():
                gref = rec["grads"][k]
                assert_close(p.grad, gref, 1e-4, 2e-6 * float(gref.abs().max()) + 1e-12, f"{name} autograd grad {k}")


def test_bf16_mode_tracks_fp32(golden):
    """bf16 operands / fp32 accumulate.  Loss within rtol 2e-2 of the fp32 reference golden (north_star
    tolerance); gradients within 2e-2 (relative l2) of the oracle run with the same operand rounding — masks
    such as [pre < 0] or the JumpReLU window flip on a few entries when pre-activations move by a bf16 ulp, so
    the fp32 golden gradient is not the right yardstick for them at B = 16."""
    from crosslayer_transcoder_amd import precision
    from oracle import clt_oracle as O
    from tests.build_cases import build_module
    from tests.golden_cases import init_buffers
    for name in ("dbg_jumprelu_clt", "relu_clt_base", "batchtopk_clt_base"):
        case = golden(name)
        for sparse in ((None, False) if name.startswith("batchtopk") else (None,)):
            mod = build_module(case)
            with precision("bf16-mixed"):
                out = mod.compute_step(case["batches"][0].cuda(), 0, sparse=sparse)
            rec = case["steps"][0]
            assert_close(out["loss"], rec["loss"], 2e-2, 1e-4, f"{name} bf16 loss")
            cfg = dict(case["cfg"]); cfg["loss"] = dict(cfg["loss"], step_frac=0.0)
            batch = case["batches"][0]
            with O.operand_rounding(torch.bfloat16):
                ref = O.training_step(batch, case["init_params"], O.standardizer_init(batch), cfg, init_buffers(case))
            assert_close(out["loss"], ref["loss"], 1e-3, 1e-6, f"{name} bf16 loss vs rounded oracle")
            for k, gref in ref["grads"].items():
                g = out["grads"][k].reshape(gref.shape).cpu()
                rel = (g - gref).norm() / (gref.norm() + 1e-12)
                assert rel < 2e-2, f"{name} bf16 grad {k} (sparse={sparse}): rel l2 err {rel:.3g}"


def test_pretrained_roundtrip_and_folding(tmp_path, golden):
    """reference tests/test_load_from_pretrained.py:52-54 + test_standardization_folding.py (fp32 tolerance)."""
    from crosslayer_transcoder_amd.model import CrossLayerTranscoder
    from tests.build_cases import build_module
    case = golden("topk_clt_aux")
    mod = build_module(case)
    batch = case["batches"][0].cuda()
    mod.model.initialize_standardizers(batch)
    with torch.no_grad():
        _, _, _, recons = mod.model(batch[:, 0])
    mod.model.save_pretrained(tmp_path, fold_standardizers=True)
    loaded = CrossLayerTranscoder.from_pretrained(tmp_path).cuda()
    assert loaded._is_folded and loaded.input_standardizer is None
    for (n1, p1), (n2, p2) in zip(mod.model.named_parameters(), loaded.named_parameters()):
        assert n1 == n2 and p1.shape == p2.shape
    with torch.no_grad():
        _, _, _, recons_folded = loaded(batch[:, 0])
    assert_close(recons_folded, recons, 1e-4, 1e-4 * float(recons.abs().max()), "folded recons")


@pytest.mark.parametrize("dec_kind,nl_kind,loss_kind,act16", [("clt", "jumprelu", "jumprelu", False), ("plt", "jumprelu", "jumprelu", False),
                                                              ("clt", "jumprelu", "jumprelu", True), ("plt", "jumprelu", "jumprelu", True),
                                                              ("clt", "relu", "base", False), ("clt", "batchtopk", "base", False)])
def test_dense_bf16_fast_path_step(dec_kind, nl_kind, loss_kind, act16, monkeypatch):
    """Shapes large enough for the glds NT GEMM kernel (K % 64 == 0, M*N >= 256^2) in every dense contraction
    (decoder forward with transposed weights + slab sum, dgrad, token-contracted weight gradients): loss and
    gradients vs the oracle run with the same bf16 operand rounding (rtol 2e-2, relative l2 for gradients).
    act16: the pre-activations and the feature gradient leave their GEMMs as bf16 (engine.DENSE_BF16_ACTIVATIONS)."""
    from crosslayer_transcoder_amd import engine, precision
    monkeypatch.setattr(engine, "DENSE_BF16_ACTIVATIONS", act16)
    from oracle import clt_oracle as O
    from tests.build_cases import build_module
    B, L, d, F = 512, 3, 128, 512
    g = torch.Generator().manual_seed(7)
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
    case = {"cfg": cfg, "opt": {"lr": 1e-3}, "batches": batch[None], "init_params": params,
            "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}
    mod = build_module(case)
    mod.global_step = 5
    mod.log_metrics_every = 0                     # (logging at step 0 would materialise the lazy fp32 features)
    with precision("bf16-mixed"):
        out = mod.compute_step(batch.cuda(), 0, sparse=False)
    c = dict(cfg); c["loss"] = dict(lc, step_frac=0.5)
    buf = {"last_active": torch.zeros(L, F, dtype=torch.long), "threshold": torch.zeros(1)}
    with O.operand_rounding(torch.bfloat16):
        ref = O.training_step(batch, params, O.standardizer_init(batch), c, buf)
    assert_close(out["loss"], ref["loss"], 2e-3, 1e-6, "loss")
    assert_close(out["recons_norm"], ref["recons_norm"], 2e-2, 2e-2 * float(ref["recons_norm"].abs().max()), "recons_norm")
    for k, gref in ref["grads"].items():
        gg = out["grads"][k].reshape(gref.shape).cpu()
        rel = (gg - gref).norm() / (gref.norm() + 1e-12)
        assert rel < 2e-2, f"{dec_kind}/{nl_kind} bf16 grad {k}: rel l2 err {rel:.3g}"
    if loss_kind == "jumprelu" and act16:
        assert "pre_actvs" not in dict.keys(out) and out["pre_actvs"].dtype == torch.float32      # (lazy fp32 view of the bf16 tensor)
        pre_ref = ref["pre"]
        assert_close(out["pre_actvs"], pre_ref, 1e-2, 1e-2 * float(pre_ref.abs().max()), "bf16 pre-activations")
        # (the lazy features come from the lazy pre-activations, through a weak reference to the output that holds both)
        assert "features" not in dict.keys(out)
        assert torch.equal(out["features"], ops_mod().jumprelu_fwd(out["pre_actvs"], mod.model.nonlinearity.theta.detach()))
    elif loss_kind == "jumprelu":
        # the fused tile passes (bf16 operands written directly) against the separate fp32 passes + casts: the forward GEMM
        # operands are bit-identical; the backward pass takes tanh from the hardware exp2 / rcp (about 1e-6 absolute), which can
        # move a d(pre) entry across a bf16 rounding boundary, and its column-sum / loss-sum reduction trees differ
        assert "features" not in dict.keys(out), "the fused dense path should not materialise fp32 features"
        mod2 = build_module(case)
        mod2.global_step = 5
        mod2.log_metrics_every = 0
        engine.FUSE_DENSE_JUMPRELU = False
        try:
            with precision("bf16-mixed"):
                out2 = mod2.compute_step(batch.cuda(), 0, sparse=False)
        finally:
            engine.FUSE_DENSE_JUMPRELU = True
        assert "features" in dict.keys(out2)
        assert torch.equal(out["features"], out2["features"])
        assert torch.equal(out["recons_norm"], out2["recons_norm"])
        assert torch.equal(out["dead"], out2["dead"])
        assert_close(out["loss"], out2["loss"], 2e-6, 0, "fused vs unfused loss")
        for k, g2 in out2["grads"].items():
            rel = float((out["grads"][k] - g2).norm() / (g2.norm() + 1e-12))
            assert rel < 1e-3, f"fused vs unfused grad {k}: rel l2 err {rel:.3g}"


def ops_mod():
    from crosslayer_transcoder_amd import ops
    return ops


@pytest.mark.parametrize("dec_kind,nl_kind,loss_kind", [("plt", "jumprelu", "jumprelu"), ("clt", "jumprelu", "jumprelu"), ("clt", "batchtopk", "base")])
def test_steps_leave_no_reference_cycle_behind(dec_kind, nl_kind, loss_kind):
    """With Python's cyclic collector off, the device memory in use does not grow from step to step: a step's output must not
    sit in a reference cycle (one through its lazy `features` closure kept every step's gradients and 16-bit pre-activations
    alive until the collector ran: +6 GB every few steps at B = 8192)."""
    import gc
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    B, L, d, F = 512, 3, 128, 512
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, L, d, generator=g)
    batch = torch.stack([x, 0.5 * x + 0.1 * torch.randn(B, L, d, generator=g)], 1).cuda()
    nl = {"jumprelu": {"kind": "jumprelu", "theta": 0.05, "bandwidth": 0.1}, "batchtopk": {"kind": "batchtopk", "k": 8, "e": 0.1}}[nl_kind]
    lc = {"kind": "base"} if loss_kind == "base" else {"kind": "jumprelu", "lambda_sparsity": 0.05, "c_sparsity": 1.0, "use_tanh": True,
                                                       "pre_actv_loss": 1e-3}
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F), "decoder.b": torch.zeros(L, d)}
    if dec_kind == "clt":
        for l in range(L):
            params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    else:
        params["decoder.W"] = (torch.rand(L, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    if nl_kind == "jumprelu":
        params["nonlinearity.theta"] = torch.full((1, L, F), 0.05)
    case = {"cfg": {"decoder": dec_kind, "d_features": F, "nonlinearity": nl, "loss": lc}, "opt": {"lr": 1e-3}, "batches": batch[None].cpu(),
            "init_params": params, "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 100}
    mod = build_module(case)
    mod.log_metrics_every = 0
    mod.attach_optimizers()
    with precision("bf16-mixed"):
        for t in range(3):
            mod.training_step(batch, t)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        try:
            used = []
            for t in range(3, 7):
                mod.training_step(batch, t)
                torch.cuda.synchronize()
                used.append(torch.cuda.memory_allocated())
        finally:
            gc.enable()
    # (a few hundred bytes may come and go — a pointer table of new addresses, a counter —; one step's tensors are megabytes here)
    assert max(used) - min(used) < (1 << 16), [u - used[0] for u in used]


@pytest.mark.parametrize("nl_kind", ["batchtopk", "perlayerbatchtopk"])
def test_fused_encoder_select_equals_unfused(nl_kind):
    """From the second step on the encoder GEMM appends the selection candidates in its epilogue (bound = 0.9 x last
    step's k-th value) and no pre-activation tensor exists.  Same MFMA main loop => bit-identical pre-activations =>
    the whole trajectory must equal the unfused path exactly; a void attempt (bound too high) falls back."""
    import crosslayer_transcoder_amd.engine as E
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    B, L, d, F = 512, 3, 128, 512
    g = torch.Generator().manual_seed(11)
    batches = [torch.randn(B, 2, L, d, generator=g) for _ in range(4)]
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d)}
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    nl = {"kind": nl_kind, "k": 8, "e": 0.1}
    case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": nl, "loss": {"kind": "base"}}, "opt": {"lr": 1e-3},
            "batches": torch.stack(batches), "init_params": params, "last_active0": torch.zeros(L, F, dtype=torch.long),
            "max_steps": 10}
    results = {}
    for fuse in (True, False):
        E.FUSE_ENCODER_SELECT = fuse
        try:
            mod = build_module(case)
            opt = mod.configure_optimizers()
            losses, fused_steps = [], 0
            with precision("bf16-mixed"):
                for t, b in enumerate(batches):
                    if fuse and t == 3:                               # poison the bound: the attempt must be void, not wrong
                        mod.model.nonlinearity._kth = torch.full_like(mod.model.nonlinearity._kth, 0x4E6E6B28 - 2 ** 31)
                    losses.append(float(mod.fused_train_step(b.cuda(), t, opt)))
                    fused_steps += "pre_actvs" not in dict.keys(mod.last_step)
            results[fuse] = (losses, {k: p.detach().clone() for k, p in mod.model.named_parameters()},
                             mod.model.nonlinearity.threshold.clone(), fused_steps)
        finally:
            E.FUSE_ENCODER_SELECT = True
    assert results[True][3] == 2 and results[False][3] == 0, "fused epilogue expected on steps 1 and 2 only"
    assert results[True][0] == results[False][0]
    assert torch.equal(results[True][2], results[False][2])
    for k, p in results[True][1].items():
        assert torch.equal(p, results[False][1][k]), f"{k} differs between the fused and the unfused path"


def test_fused_encoder_rows_select_equals_unfused():
    """PerLayerTopK, bf16, base loss: from the second step on the encoder GEMM carries the per-row selection in its epilogue (no
    pre-activation tensor); the trajectory must equal the unfused path exactly, and a void attempt (bounds that cannot hold)
    reruns the step unfused with the same result."""
    import crosslayer_transcoder_amd.engine as E
    from crosslayer_transcoder_amd import ops, precision
    from tests.build_cases import build_module
    B, L, d, F = 512, 3, 128, 512
    g = torch.Generator().manual_seed(12)
    batches = [torch.randn(B, 2, L, d, generator=g) for _ in range(4)]
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d)}
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "topk", "k": 8}, "loss": {"kind": "base"}},
            "opt": {"lr": 1e-3}, "batches": torch.stack(batches), "init_params": params,
            "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}
    results = {}
    odds = ops.ROWS_FAIL_ODDS
    for fuse in (True, False):
        E.FUSE_ENCODER_ROWS = fuse
        E.STATS["select_fallbacks"] = 0
        try:
            mod = build_module(case)
            opt = mod.configure_optimizers()
            losses, fused_steps = [], 0
            with precision("bf16-mixed"):
                for t, b in enumerate(batches):
                    ops.ROWS_FAIL_ODDS = 0.999 if (fuse and t == 2) else odds        # step 2: a void attempt, then the unfused rerun
                    losses.append(float(mod.fused_train_step(b.cuda(), t, opt)))
                    fused_steps += "pre_actvs" not in dict.keys(mod.last_step)
            results[fuse] = (losses, {k: p.detach().clone() for k, p in mod.model.named_parameters()}, fused_steps,
                             E.STATS["select_fallbacks"])
        finally:
            E.FUSE_ENCODER_ROWS = True
            ops.ROWS_FAIL_ODDS = odds
    # (step 0 runs unfused: the HBM headroom is asked once the optimizer state exists; step 2 is the void attempt)
    assert results[True][2] == 2 and results[True][3] == 1 and results[False][2] == 0, results[True][2:]
    assert results[True][0] == results[False][0]
    for k, p in results[True][1].items():
        assert torch.equal(p, results[False][1][k]), f"{k} differs between the fused and the unfused path"


@pytest.mark.parametrize("prec", ["bf16-mixed", "16-mixed"])
@pytest.mark.parametrize("dec_kind", ["clt", "plt"])
def test_adam_pass_that_also_prepares_the_next_steps_operands(dec_kind, prec):
    """Dense 16-bit regime: the optimizer's update of a decoder matrix also writes the updated matrix's 16-bit copy, transposed
    copy and squared row norms (clt_adam_prep_f32), and the encoder's update its 16-bit shadow — the next step then reads no
    fp32 weight for its operands.  Same bits as preparing them from the updated weights: four training steps (fp16: the first
    ones are skipped by the loss scaler, whose registrations must not leak) equal the unfused path exactly."""
    import crosslayer_transcoder_amd.engine as E
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    B, L, d, F = 256, 3, 128, 512
    g = torch.Generator().manual_seed(31)
    batches = [torch.randn(B, 2, L, d, generator=g) for _ in range(4)]
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d), "nonlinearity.theta": torch.full((1, L, F), 0.03)}
    if dec_kind == "clt":
        for l in range(L):
            params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    else:
        params["decoder.W"] = (torch.rand(L, F, d, generator=g) * 2 - 1) * d ** -0.5
    cfg = {"decoder": dec_kind, "d_features": F, "nonlinearity": {"kind": "jumprelu", "theta": 0.03, "bandwidth": 1.0},
           "loss": {"kind": "jumprelu", "lambda_sparsity": 2e-4, "c_sparsity": 0.1, "use_tanh": True, "pre_actv_loss": 1e-8}}
    case = {"cfg": cfg, "opt": {"lr": 1e-3}, "batches": torch.stack(batches), "init_params": params,
            "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}
    results = {}
    for fuse in (True, False):
        E.FUSE_ADAM_PREP = fuse
        try:
            mod = build_module(case)
            mod.global_step = 5
            mod.log_metrics_every = 0
            opt = mod.configure_optimizers()
            losses, tagged = [], 0
            with precision(prec):
                for t, b in enumerate(batches):
                    losses.append(float(mod.fused_train_step(b.cuda(), t, opt, sparse=False)))
                    tagged += all(hasattr(w, "_clt_prep") for w in mod.model.decoder.weights())
            results[fuse] = (losses, {k: p.detach().clone() for k, p in mod.model.named_parameters()}, tagged)
        finally:
            E.FUSE_ADAM_PREP = True
    assert results[True][2] >= 2 and results[False][2] == 0, (results[True][2], results[False][2])
    assert results[True][0] == results[False][0], (results[True][0], results[False][0])
    for k, p in results[True][1].items():
        assert torch.equal(p, results[False][1][k]), f"{k} differs between the fused and the separate preparation"


def test_pinned_prefetcher_and_fit_loop(golden):
    """Batch ingest edge: double-buffered pinned H2D yields exactly the host batches (pinned and pageable sources, fp16
    and fp32), and trainer.fit on the debug-cpu golden reproduces the reference loss curve through it."""
    from crosslayer_transcoder_amd import trainer
    from crosslayer_transcoder_amd.data.prefetch import PinnedPrefetcher
    from tests.build_cases import build_module
    g = torch.Generator().manual_seed(0)
    for dtype, pin in ((torch.float16, True), (torch.float32, False)):
        host = [torch.randn(64, 2, 3, 40, generator=g).to(dtype) for _ in range(7)]
        src = [h.pin_memory() if pin else h for h in host]
        got = []
        for b in PinnedPrefetcher(src, "cuda:0", depth=2):
            got.append((b * 1).cpu())                    # consume on the compute stream before the slot is reused
        assert len(got) == 7 and all(torch.equal(a, h) for a, h in zip(got, host))
    case = golden("dbg_jumprelu_clt")
    mod = build_module(case)
    losses, _ = trainer.fit(mod, list(case["batches"]), case["max_steps"])      # fewer batches than max_steps: stops early
    for t, rec in enumerate(case["steps"]):
        assert_close(losses[t], rec["loss"], 2e-5, 1e-7, f"fit loss {t}")
    mod = build_module(case)                      # the automatic-optimisation order (fused_step=False)
    losses, _ = trainer.fit(mod, list(case["batches"]), case["max_steps"], fused=False)
    for t, rec in enumerate(case["steps"]):
        assert_close(losses[t], rec["loss"], 2e-5, 1e-7, f"fit (autograd) loss {t}")


def test_fp16_batches_through_a_logging_step(golden):
    """The shared-memory buffer hands batches over as fp16 (data/shared_memory.py:168-214): the step — including a step
    that logs (batch_idx % log_metrics_every == 0 reads the batch again) — equals the step on the widened batch."""
    from tests.build_cases import build_module
    case = golden("batchtopk_clt_base")
    b16 = case["batches"][0].to(torch.float16).cuda()
    res = []
    for batch in (b16, b16.float()):
        mod = build_module(case)
        mod.log_metrics_every = 1
        opt = mod.configure_optimizers()
        res.append([float(mod.fused_train_step(batch, t, opt)) for t in range(2)])
    assert res[0] == res[1]


@pytest.mark.parametrize("fused", [False, True], ids=["training_step", "fused-adam-step"])
@pytest.mark.parametrize("name", ["dbg_jumprelu_clt", "jumprelu_plt_notanh", "topk_clt_aux", "batchtopk_clt_base"])
def test_logging_side_outputs_match_reference(name, fused, golden):
    """Everything log_training_metrics logs at step 0 (clt_lightning.py:131-199) and the DeadFeatures statistics
    (metrics/dead_features.py), computed on the device without the [B,L,F] host copy, vs the values the reference
    logged (tests/golden/metrics.pt).  With Adam fused into the backward pass the weight norms still come from the
    pre-update weights."""
    import os
    from crosslayer_transcoder_amd.metrics import DeadFeatures
    from tests.build_cases import build_module
    gm = torch.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.pt"), weights_only=True)[name]
    case = golden(name)
    mod = build_module(case)
    L, F = case["last_active0"].shape
    mod.log_metrics_every = 50
    mod.compute_dead_features, mod.compute_dead_features_every = True, 10 ** 9
    mod.dead_features = DeadFeatures(F, L, return_per_layer=True, return_neuron_indices=True, return_log_freqs=True)
    mod.dead_features.reset = lambda: None        # (global_step 0 is a compute-and-reset step, as in the reference :246-280)
    opt, sched = _opt_and_sched(mod)
    batch = case["batches"][0].cuda()
    if fused:
        mod.fused_train_step(batch, 0, opt, sched)
    else:
        mod.fused_step, mod.automatic_optimization = False, True
        mod.training_step(batch, 0)
    checked = 0
    for k, want in gm["logged"].items():
        if k.startswith(("layers/", "model/", "metrics/")) or k == "training/tokens_seen":
            got = mod.logged[k]
            got = got.float().cpu() if torch.is_tensor(got) else torch.tensor(float(got))
            assert torch.allclose(got, want.float(), rtol=2e-5, atol=1e-7), (k, float(got), float(want))
            checked += 1
    assert checked >= 14
    # DeadFeatures: the module updated it from the step's fire counts (batch 0); add batch 1 from a plain forward
    df = mod.dead_features
    mod2 = build_module(case)
    mod2.model.initialize_standardizers(batch)
    with torch.no_grad():
        feat1 = mod2.model(case["batches"][1][:, 0].cuda())[1]
    df.update(feat1)
    assert torch.equal(df.n_active.cpu(), gm["n_active"]) and torch.equal(df.n_total.cpu(), gm["n_total"])
    stats = df.compute()
    for k, want in gm["dead_features"].items():
        assert torch.allclose(stats[k].float().cpu(), want.float(), rtol=1e-6, atol=0), k


def _threshold_case(dec_kind, nl_kind, theta, B=512, L=3, d=128, F=512, seed=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, d, generator=g)
    y = 0.5 * x * torch.randn(L, d, generator=g) + 0.1 * torch.randn(B, L, d, generator=g)
    batch = torch.stack([x, y], 1)
    nl = {"kind": "relu"} if nl_kind == "relu" else {"kind": "jumprelu", "theta": theta, "bandwidth": 0.1}
    lc = {"kind": "jumprelu", "lambda_sparsity": 0.05, "c_sparsity": 1.0, "use_tanh": True, "pre_actv_loss": 1e-3}
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.randn(L, F, generator=g) * 0.01,
              "decoder.b": torch.randn(L, d, generator=g) * 0.01}
    bd = (d * L) ** -0.5
    if dec_kind == "clt":
        for l in range(L):
            params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * bd
    else:
        params["decoder.W"] = (torch.rand(L, F, d, generator=g) * 2 - 1) * bd
    if nl_kind == "jumprelu":
        params["nonlinearity.theta"] = torch.full((1, L, F), theta)
    cfg = {"decoder": dec_kind, "d_features": F, "nonlinearity": nl, "loss": lc}
    return {"cfg": cfg, "opt": {"lr": 1e-3}, "batches": batch[None], "init_params": params,
            "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}


@pytest.mark.parametrize("dec_kind,nl_kind", [("clt", "jumprelu"), ("plt", "jumprelu"), ("clt", "relu")])
def test_threshold_lists_bf16_step(dec_kind, nl_kind):
    """JumpReLU / ReLU in the activation-list regime, bf16 mode, at shapes that take the fast GEMM kernels: loss and
    every gradient vs the oracle with the same operand rounding (rtol 2e-2 / relative l2), and — Adam fused into the row
    pass — the first-step moments vs oracle.adam_step on the list regime's own materialised gradients."""
    from crosslayer_transcoder_amd import precision
    from oracle import clt_oracle as O
    from tests.build_cases import build_module
    case = _threshold_case(dec_kind, nl_kind, 0.6)          # pre ~ N(0, 0.29^2): about 2 % (JumpReLU) / 50 % (ReLU) active
    batch = case["batches"][0]
    mod = build_module(case)
    mod.global_step, mod.log_metrics_every = 5, 0
    with precision("bf16-mixed"):
        out = mod.compute_step(batch.cuda(), 0, sparse=True)
    assert out.get("lists") is not None
    if nl_kind == "jumprelu":
        assert 0 < int(out["lists"].total) < 0.05 * batch.shape[0] * 3 * 512
    c = dict(case["cfg"]); c["loss"] = dict(c["loss"], step_frac=0.5)
    buf = {"last_active": case["last_active0"].clone()}
    # (the pre-activations are the encoder product rounded to bf16 plus the fp32 bias — autocast's order, the oracle's
    # default round_acc — whether they pass through the 16-bit tensor of the JumpReLU module's regime or stay fp32)
    with O.operand_rounding(torch.bfloat16):
        ref = O.training_step(batch, case["init_params"], O.standardizer_init(batch), c, buf)
    assert_close(out["loss"], ref["loss"], 2e-3, 1e-6, "loss")
    assert torch.equal(out["features"].cpu() > 0, ref["features"] > 0) or \
        float(((out["features"].cpu() > 0) != (ref["features"] > 0)).float().mean()) < 1e-4      # bf16 ulps at the threshold
    for k, gref in ref["grads"].items():
        gg = out["grads"][k].reshape(gref.shape).cpu()
        rel = (gg - gref).norm() / (gref.norm() + 1e-12)
        assert rel < 2e-2, f"{dec_kind}/{nl_kind} lists bf16 grad {k}: rel l2 err {rel:.3g}"
    # fused: same gradients must arrive in the moments (exp_avg = (1 - beta1) * g after the first step)
    mod2 = build_module(case)
    mod2.global_step, mod2.log_metrics_every = 5, 0
    opt, _ = mod2.attach_optimizers()
    with precision("bf16-mixed"):
        mod2.fused_train_step(batch.cuda(), 0, opt, sparse=True)
    # (with Adam fused the gather forward reads the bf16 shadow rows instead of the fp32 rows: the reconstruction moves
    # by bf16 rounding of the weights, hence a relative-l2 comparison)
    for k, p in mod2.model.named_parameters():
        g = out["grads"][k].reshape(p.shape).double()
        m = opt.state[p]["exp_avg"].double()
        rel = float((m - 0.1 * g).norm() / (0.1 * g.norm() + 1e-30))
        assert rel < 1e-2, f"fused exp_avg {k}: rel l2 {rel:.3g}"


@pytest.mark.parametrize("dec_kind", ["clt", "plt"])
def test_threshold_lists_kept_row_norms_equal_recomputed(dec_kind, monkeypatch):
    """List regime with Adam fused into the row pass: the squared row lengths the row passes leave behind
    (engine.ROW_PASS_KEEPS_NORMS) are, bit for bit, clt_row_sqnorm_f32 of the updated matrices, so next step's dec_norms are
    the ones a recomputation would give; an in-place change of a weight from outside moves the version key and drops them.
    Three steps with and without: same losses (the two row-pass instantiations contract their fp32 Adam arithmetic
    differently, so parameters agree to rounding, not bit for bit)."""
    from crosslayer_transcoder_amd import engine, ops, precision
    from tests.build_cases import build_module
    case = _threshold_case(dec_kind, "jumprelu", 0.6)
    batch = case["batches"][0].cuda()
    res = []
    for keep in (True, False):
        monkeypatch.setattr(engine, "ROW_PASS_KEEPS_NORMS", keep)
        mod = build_module(case)
        mod.global_step, mod.log_metrics_every = 5, 0
        opt, _ = mod.attach_optimizers()
        losses = []
        with precision("bf16-mixed"):
            for i in range(3):
                losses.append(float(mod.fused_train_step(batch, i, opt, sparse=True)))
                dec = mod.model.decoder
                kept = getattr(dec, "_row_sq_parts", None)
                assert (kept is not None) == keep
                if keep:
                    D = engine._Decoder(dec)
                    assert kept[0] == D._version_key()
                    for w, part in zip(D.W32, kept[1]):
                        ref = torch.zeros_like(part)
                        ops.row_sqnorm(w, ref, 0, False)
                        assert torch.equal(part, ref)
                    if i == 1:                  # a change from outside: the kept norms are not used for the next step
                        with torch.no_grad():
                            D.W32[0].mul_(1.01)
                        assert kept[0] != D._version_key()
        res.append((losses, {k: p.detach().clone() for k, p in mod.model.named_parameters()}))
    # (the keep=False run did not scale W32[0] after step 2: compare the first two losses only)
    assert res[0][0][:2] == pytest.approx(res[1][0][:2], rel=1e-4)


def test_threshold_list_gate_chooses_the_regime():
    """sparse=None on a JumpReLU module: the device-side count picks the list regime below the break-even density and the
    dense GEMM regime above it (then backs off: no attempt on the next steps); both give the same step (fp32 mode)."""
    from crosslayer_transcoder_amd import engine
    from tests.build_cases import build_module
    for theta, want_lists in ((0.9, True), (0.0, False)):
        case = _threshold_case("clt", "jumprelu", theta)
        batch = case["batches"][0].cuda()
        mod = build_module(case)
        mod.global_step, mod.log_metrics_every = 5, 0
        dense0 = engine.STATS.get("list_gate_dense", 0)
        out = mod.compute_step(batch, 0)
        density = float((out["features"] > 0).float().mean())
        assert (out.get("lists") is not None) == want_lists, f"theta {theta}: density {density:.4f}"
        assert (density <= engine.LIST_MAX_DENSITY) == want_lists
        if not want_lists:
            assert engine.STATS["list_gate_dense"] == dense0 + 1 and mod._list_skip >= engine.LIST_RETRY_FIRST
            mod.compute_step(batch, 1)
            assert engine.STATS["list_gate_dense"] == dense0 + 1      # backed off: no second attempt
        ref = build_module(case)
        ref.global_step, ref.log_metrics_every = 5, 0
        out2 = ref.compute_step(batch, 0, sparse=not want_lists) if want_lists else ref.compute_step(batch, 0, sparse=True)
        assert_close(out["loss"], out2["loss"], 1e-5, 1e-8, "loss: list regime vs dense regime")
        for k, g2 in out2["grads"].items():
            assert_close(out["grads"][k].reshape(g2.shape), g2, 1e-4, 2e-6 * float(g2.abs().max()) + 1e-12, f"grad {k}")


def test_debug_yaml_config_end_to_end(tmp_path, golden):
    """BASELINE.json configs[0] through the YAML itself: config/debug-cpu.yaml (the reference's class_path / init_args
    trees) -> seed, precision "32", model tree, TensorDataModule on the reference's fixture tensor, 100 fused training
    steps.  From the reference's initial parameters the loss curve must follow the curve the reference produced
    (tests/golden/dbg_yaml_curve.pt); every batch is the whole 16-row tensor in a shuffled order, so the data module's
    own permutations change nothing but the summation order."""
    import os
    from crosslayer_transcoder_amd import get_precision, trainer
    cur = golden("dbg_yaml_curve")
    acts = golden("dbg_jumprelu_clt")["batches"][0]
    tfile = str(tmp_path / "debug_actvs.pt")
    torch.save(acts, tfile)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fused in (True, False):
        mod, losses = trainer.run_config(os.path.join(root, "config", "debug-cpu.yaml"), tensor_file=tfile,
                                         init_state=cur["init_params"], fused=fused)
        assert get_precision() == "32" and len(losses) == 100 and mod.global_step == 100
        got = torch.stack([x.float().cpu() for x in losses])
        assert_close(got[:5], cur["losses"][:5], 2e-5, 1e-7, "first steps")
        assert_close(got, cur["losses"], 5e-4, 1e-7, f"100-step curve (fused={fused})")
        for k, p in mod.model.named_parameters():
            assert_close(p, cur["final_params"][k], 1e-3, 0.05, f"final {k}")


@pytest.mark.parametrize("prec", ["32", "bf16-mixed"])
@pytest.mark.parametrize("kind", ["batchtopk", "topk"])
def test_second_streams_give_the_same_bits(kind, prec, monkeypatch):
    """The step's second streams — engine.HOT_ROWS_STREAM (the hot rows' launches of every weight pass, hot_phase 1 / 2 of
    clt_sparse_row_pass_f32) and ops.LIST_CSC_STREAM (the feature-major half of the fused selections' list build, stage 7 / 3 of
    the ext entry points) — only reorder independent work: eight fused Adam steps over two batches leave every parameter and
    moment bit-identical to the one-stream order, with heavy-tailed lists (hot columns present) and the fused selections active."""
    from crosslayer_transcoder_amd import engine, ops, precision
    from tests.build_cases import build_module
    B, L, d, F, k = 768, 3, 128, 512, 8
    g = torch.Generator().manual_seed(29)
    batches = []
    A = torch.randn(L, d, generator=g)
    for _ in range(2):
        x = torch.randn(B, L, d, generator=g)
        batches.append(torch.stack([x, 0.5 * x * A + 0.1 * torch.randn(B, L, d, generator=g)], 1))
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.randn(L, d, generator=g) * 0.01}
    params["encoder.b"][:, 5::97] = 0.9
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    nl = {"kind": "batchtopk", "k": k, "e": 0.1} if kind == "batchtopk" else {"kind": "topk", "k": k}
    case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": nl, "loss": {"kind": "base"}},
            "opt": {"lr": 1e-3}, "batches": torch.stack(batches), "init_params": params, "last_active0": torch.zeros(L, F, dtype=torch.long),
            "max_steps": 10}
    res = {}
    for on in (True, False):
        monkeypatch.setattr(engine, "HOT_ROWS_STREAM", on)
        monkeypatch.setattr(ops, "LIST_CSC_STREAM", on)
        fused0 = engine.STATS["fused_select_steps"]
        mod = build_module(case)
        mod.log_metrics_every = 0
        opt, _ = mod.attach_optimizers()
        with precision(prec):
            for t in range(8):
                mod.fused_train_step(batches[t % 2].cuda(), t, opt)
        S = mod.last_step["lists"]
        assert ops.hot_columns(S) is not None and int((S.col_start[1:] - S.col_start[:-1]).max()) > 256
        if prec != "32" and kind == "batchtopk":      # (the fused encoder + selection runs in the 16-bit modes, from the second step on)
            assert engine.STATS["fused_select_steps"] - fused0 == 7
        torch.cuda.synchronize()
        res[on] = {k_: (p.detach().clone(), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone())
                   for k_, p in mod.model.named_parameters()}
    for k_ in res[True]:
        for a, b in zip(res[True][k_], res[False][k_]):
            assert torch.equal(a, b), k_


@pytest.mark.parametrize("prec", ["32", "bf16-mixed"])
@pytest.mark.parametrize("kind", ["batchtopk", "topk-aux-lists"])
def test_heavy_tailed_feature_frequencies_hot_rows(kind, prec, monkeypatch):
    """Real dictionaries are heavy-tailed: here a few features of every layer fire in about half of the tokens (encoder-bias
    offsets), so their lists hold hundreds of entries where the median column holds a handful.  Those rows go through the
    multi-wave launch of the weight pass (ops.HOT_MIN_LEN: segments of a list per wave, combined in wave order) and the
    per-entry dot reduction; loss, every gradient and the first fused Adam step against the oracle — and the same step with
    the hot-row form switched off (one wave per row) agrees to summation-order rounding."""
    from crosslayer_transcoder_amd import engine, ops, precision
    from oracle import clt_oracle as O
    from tests.build_cases import build_module
    B, L, d, F, k = 768, 3, 128, 512, 8
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, L, d, generator=g)
    y = 0.5 * x * torch.randn(L, d, generator=g) + 0.1 * torch.randn(B, L, d, generator=g)
    batch = torch.stack([x, y], 1)
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.randn(L, d, generator=g) * 0.01}
    params["encoder.b"][:, 5::97] = 0.9              # ~6 features per layer far above the selection threshold half of the time
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    aux = kind != "batchtopk"
    nl = {"kind": "topk", "k": k} if aux else {"kind": "batchtopk", "k": k, "e": 0.1}
    loss = {"kind": "topk", "topk_aux": {"kind": "topk", "k": 24}, "tokens_till_dead": 100, "aux_loss_scale": 1 / 32} if aux else {"kind": "base"}
    la0 = torch.zeros(L, F, dtype=torch.long)
    if aux:
        la0[:, ::4] = 5000
        params["encoder.b"][:, ::4] = -1.5
        params["encoder.b"][:, 8::100] = -0.25       # dead features that dominate the aux selection: hot columns of the SECOND list
        la0[:, 8::100] = 5000
        monkeypatch.setattr(engine, "COMPACT_AUX", False)     # the aux pass on activation lists (second list of the row pass)
    case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": nl, "loss": loss}, "opt": {"lr": 1e-3},
            "batches": batch[None], "init_params": params, "last_active0": la0, "max_steps": 10}
    bf16 = prec != "32"
    outs = {}
    hot0 = ops.HOT_MIN_LEN
    assert hot0 > 0
    for hot in (hot0, 0):
        monkeypatch.setattr(ops, "HOT_MIN_LEN", hot)
        mod = build_module(case)
        mod.log_metrics_every = 0
        with precision(prec):
            outs[hot] = mod.compute_step(batch.cuda(), 0)
    out = outs[hot0]
    S = out["lists"]
    lens = (S.col_start[1:] - S.col_start[:-1]).long()
    assert int(lens.max()) > 256 and int((lens > hot0).sum()) >= 10 and float(lens.float().median()) < 20, "the case must be heavy-tailed (and hold columns long enough to be walked in segments)"
    buf = {"last_active": la0.clone(), "threshold": torch.zeros(1)}
    import contextlib
    with (O.operand_rounding(torch.bfloat16) if bf16 else contextlib.nullcontext()):
        ref = O.training_step(batch, params, O.standardizer_init(batch), case["cfg"], buf)
    assert_close(out["loss"], ref["loss"], 1e-3 if bf16 else 2e-5, 1e-7, "loss")
    if not bf16:
        assert torch.equal(out["features"].cpu() > 0, ref["features"] > 0)
    for k_, gref in ref["grads"].items():
        got = out["grads"][k_].reshape(gref.shape).cpu()
        if bf16:
            rel = float((got - gref).norm() / (gref.norm() + 1e-12))
            assert rel < 2e-2, f"{kind} {prec} grad {k_}: rel l2 {rel:.3g}"
        else:
            assert_close(got, gref, 1e-4, 4e-6 * float(gref.abs().max()) + 1e-12, f"{kind} grad {k_}")
        other = outs[0]["grads"][k_].reshape(gref.shape).cpu()
        assert_close(got, other, 1e-4, 4e-6 * float(gref.abs().max()) + 1e-12, f"{kind} {prec} grad {k_}: hot-row form vs one wave per row")
    # Adam fused into the (hot-row) passes: first-step moments against the materialised gradients of the same regime
    monkeypatch.setattr(ops, "HOT_MIN_LEN", hot0)
    mod2 = build_module(case)
    mod2.log_metrics_every = 0
    opt, _ = mod2.attach_optimizers()
    with precision(prec):
        mod2.fused_train_step(batch.cuda(), 0, opt)
    for k_, p in mod2.model.named_parameters():
        gfull = out["grads"][k_].reshape(p.shape).double()
        m = opt.state[p]["exp_avg"].double()
        rel = float((m - 0.1 * gfull).norm() / (0.1 * gfull.norm() + 1e-30))
        assert rel < (1e-2 if bf16 else 1e-4), f"{kind} {prec} fused exp_avg {k_}: rel l2 {rel:.3g}"
        assert float((p.detach().double() - params[k_].double().to(p.device)).abs().max()) <= 1.01e-3
    # A segment table too small for every hot column (which columns get its slots depends on the order atomics arrive in): the
    # rows without slots are walked by one wave in the segmented order, so the step's bits do not depend on the table's size
    monkeypatch.setattr(ops, "HOT_SEG_SLOTS", 6)
    mod3 = build_module(case)
    mod3.log_metrics_every = 0
    opt3, _ = mod3.attach_optimizers()
    with precision(prec):
        mod3.fused_train_step(batch.cuda(), 0, opt3)
    cnt = mod3.last_step["lists"].hot[1]
    assert int(cnt[1]) > 6, "the case must ask for more segments than the table holds"
    for (k_, p), (_, p3) in zip(mod2.model.named_parameters(), mod3.model.named_parameters()):
        assert torch.equal(p.detach(), p3.detach()), f"{kind} {prec} {k_}: segment table size changes the result"
        assert torch.equal(opt.state[p]["exp_avg_sq"], opt3.state[p3]["exp_avg_sq"])
