"""16-bit modes of the HIP path against the REFERENCE's own autocast runs (tests/golden/ac_*_{bf16,fp16}.pt, made by
tests/golden/make_golden_autocast.py from the imported reference under torch.autocast("cpu", bf16 / fp16) (+ GradScaler)):
north_star's 16-bit bar — losses and gradients within 2e-2 — at B = 256, L = 3, d = 128, F = 512, where 16-bit rounding
matters.  The build keeps fp32 accumulators where autocast rounds continuous quantities (reconstruction, gradients) and
reproduces autocast's one DIScontinuous rounding point, the encoder product in front of the sparsifier
(CLT_GEMM_ROUND_ACC), so the selected / thresholded sets are the reference's up to accumulation-order noise."""
import numpy as np
import pytest
import torch

from tests.golden_cases import AC_CASES, assert_close, autocast_case, sample_rel_l2, sampled

pytestmark = pytest.mark.gpu

PREC = {"bf16": "bf16-mixed", "fp16": "16-mixed"}


def _regimes(name):
    if "jumprelu" in name:
        return [("default", None), ("lists", True)]          # dense 16-bit regime (the gate refuses lists at this density); forced lists
    return [("lists", None), ("dense", False)]               # TopK family: activation lists; dense MFMA regime


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("name", AC_CASES)
def test_step_matches_reference_autocast_run(name, mode, golden):
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    fx = autocast_case(name, mode, golden)
    rec = fx["steps"][0]
    batch = fx["batches"][0].cuda()
    act_ref = torch.from_numpy(np.unpackbits(rec["active_bits"].numpy())[:batch.shape[0] * batch.shape[2] * fx["cfg"]["d_features"]].astype(bool))
    for label, sparse in _regimes(name):
        mod = build_module(fx)
        mod.global_step, mod.log_metrics_every = 5, 0
        with precision(PREC[mode]):
            out = mod.compute_step(batch, 0, sparse=sparse)
        S = float(out["grad_scale"])
        assert S == float(rec["grad_scale"])
        assert_close(out["loss"], rec["loss"], 2e-2, 1e-6, f"{name} {mode} {label} loss")
        rel_loss = abs(float(out["loss"]) - float(rec["loss"])) / float(rec["loss"])
        lg = rec["logged"]
        if "training/sparsity_loss" in lg:
            assert_close(out["terms"]["sparsity"], lg["training/sparsity_loss"], 2e-2, 1e-8, "sparsity")
            assert_close(out["terms"]["pre_actv"], lg["training/pre_actv_loss"], 2e-2, 1e-10, "pre_actv")
        if "training/aux_loss" in lg:
            assert_close(out["terms"]["aux"], lg["training/aux_loss"], 2e-2, 1e-8, "aux")
        assert_close(out["terms"]["dead_fraction"], lg["metrics/dead_features"], 1e-6, 1e-7, "dead")
        act = (out["features"].cpu() > 0).reshape(-1)
        mism = int((act != act_ref).sum())
        assert mism <= 5e-4 * int(rec["n_active"]) + 2, f"{name} {mode} {label}: {mism} of {int(rec['n_active'])} active entries differ"
        assert sample_rel_l2(fx, "recons_norm", out["recons_norm"], rec["recons_norm"]) < 2e-2
        worst = 0.0
        for k, g in rec["grads"].items():
            got = out["grads"][k].reshape(fx["init_params"][k].shape) / S
            rel = sample_rel_l2(fx, k, got, g)
            worst = max(worst, rel)
            assert rel < 2e-2, f"{name} {mode} {label} grad {k}: rel l2 {rel:.3g}"
        print(f"[autocast parity] {name} {mode} {label}: loss rel {rel_loss:.2e}, active-set mismatches {mism}/{int(rec['n_active'])}, "
              f"worst gradient rel l2 {worst:.2e}")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("name", AC_CASES)
def test_two_fused_training_steps_follow_reference_autocast_run(name, mode, golden):
    """training_step (forward + backward + Adam inside the kernels) twice: both losses at 2e-2, the dead-feature counters
    exactly, the threshold EMA, the loss scale, and the parameter UPDATE (p - p0 on the fixture's sampled entries) against
    what torch.optim.Adam did to the reference's parameters under autocast."""
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import precision
    from tests.build_cases import build_module
    fx = autocast_case(name, mode, golden)
    try:
        with precision(PREC[mode]):
            mod = build_module(fx)
            mod.log_metrics_every = 0
            mod.attach_optimizers()
            for t, rec in enumerate(fx["steps"]):
                mod.global_step = 5 + t
                loss = mod.training_step(fx["batches"][t].cuda(), t)
                assert_close(loss, rec["loss"], 2e-2, 1e-6, f"{name} {mode} fused loss step {t}")
                assert torch.equal(mod.last_active.cpu(), rec["last_active"]), f"last_active step {t}"
                if "threshold" in rec:
                    assert_close(mod.model.nonlinearity.threshold, rec["threshold"], 2e-2, 1e-6, "threshold")
                if mode == "fp16":
                    assert mod.loss_scaler.scale == float(rec["scale_after"])
        for k, p in mod.model.named_parameters():
            p0 = sampled(fx, k, fx["init_params"][k])
            want = fx["final_params"][k]["val"].double() - p0
            got = sampled(fx, k, p.reshape(fx["init_params"][k].shape)) - p0
            moved = want.abs() > 0
            assert torch.equal(moved, got.abs() > 0) or float((moved != (got.abs() > 0)).float().mean()) < 2e-2, f"{k}: different entries moved"
            rel = float((got - want).norm() / (want.norm() + 1e-30))
            # Adam's first steps are sign-like (|update| ~ lr whatever |g|): entries whose tiny gradient changes sign under
            # 16-bit noise move the other way, so the update is compared in relative l2 at a loose bar, the losses tightly
            assert rel < 0.15, f"{name} {mode} update of {k}: rel l2 {rel:.3g}"
    finally:
        cta.set_precision("32")
