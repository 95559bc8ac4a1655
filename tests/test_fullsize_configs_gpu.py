"""Full-shape checks of the other BASELINE.json configurations, through size-independent properties (the oracle cannot
run these sizes in seconds), in the style of tests/test_fullsize_gpu.py:

* config 2 — PLT JumpReLU, GPT-2 small, L=12 d=768 F=32768/layer, B=4096, bf16;
* config 4, one rank's share of the 8-way feature shard — CLT JumpReLU + pre-activation loss, Llama-3.2-1B shape
  L=16 d=2048, F=8192/8 = 1024 features per layer and rank — and the whole F=8192 dictionary on one GPU;
* config 5, one rank's share — CLT PerLayerTopK(16) + aux PerLayerTopK(512), Llama-3-8B shape L=32 d=4096, F = 1024 per rank.

Properties: (1) every loss term equals a plain recomputation from the step's own pre-activations / reconstruction;
(2) the reconstruction equals an fp64 evaluation of sum(feature * W_row) + bias on sampled (token, layer) pairs;
(3) the two independent implementations of the step — dense MFMA GEMM regime and activation-list regime — agree on the
loss and on every gradient (relative l2; they differ by bf16 operand rounding only); (4) per-row TopK selects exactly
torch.topk's index set; (5) fused Adam: the loss falls over the first steps, rows without gradient do not move.
"""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(**kw):
    import bench
    import crosslayer_transcoder_amd as cta
    base = dict(layers=12, d_acts=768, features=32768, k=16, nonlinearity="jumprelu", decoder="plt", aux=False, batch=4096,
                dead_steps=1, shard_rehearsal=False, theta=0.03, shard_of=0)
    base.update(kw)
    args = argparse.Namespace(**base)
    cta.set_precision("bf16-mixed")
    dev = torch.device("cuda:0")
    mod = bench.build(args, dev)
    mod.log_metrics_every = 0
    if base.get("shard_of"):                  # one rank's share: the init bound of the encoder follows the GLOBAL feature count
        with torch.no_grad():
            mod.model.encoder.W.mul_(base["shard_of"] ** -0.5)
    A = bench.synth_target_map(args.layers, args.d_acts, dev)
    return mod, [bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=i, A=A) for i in range(2)], args


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


@torch.no_grad()
def _jumprelu_terms_from_outputs(mod, out, batch):
    """Loss terms of clt_lightning.py:434-468 recomputed with plain tensor algebra from the step's own outputs."""
    model = mod.model
    pre, r = out["pre_actvs"], out["recons_norm"]
    theta = model.nonlinearity.theta
    B, L, F = pre.shape
    ys = (batch[:, 1] - model.output_standardizer.mean) / model.output_standardizer.std
    mse = ((r - ys).double() ** 2).mean()
    lam = mod._lambda * (mod.global_step / mod.trainer.max_steps)
    sq = torch.zeros(L, F, dtype=torch.float64, device=pre.device)
    for l, w in enumerate(model.decoder.weights()):
        if w.shape[0] == L and not hasattr(model.decoder, "W_0"):      # per-layer decoder [L, F, d]
            sq += (w.double() ** 2).sum(-1)
        else:                                                          # cross-layer W_l [(l+1), F, d]
            sq[: w.shape[0]] += (w.double() ** 2).sum(-1)
    sparsity = pre_act = 0.0
    for l in range(L):                                                 # layer by layer: bounded temporaries
        p = pre[:, l]
        f = torch.where((p > theta[0, l]) & (p > 0), p, torch.zeros((), device=p.device)).double()
        wf = f * sq[l].sqrt() * mod.c
        sparsity = sparsity + (torch.tanh(wf) if mod.use_tanh else wf).sum()
        pre_act = pre_act + torch.relu(-p.double()).sum()
    return mse, lam * sparsity / B, mod.pre_actv_loss * pre_act / (B * L)


@pytest.mark.parametrize("cfg", ["config2_plt_jumprelu", "config4_shard_clt_jumprelu", "config4_full_clt_jumprelu"])
def test_jumprelu_configs_full_shape(cfg, monkeypatch):
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import engine
    # (config 4 twice: one rank's share of the 8-way feature shard, and the whole dictionary on one GPU — 80 GB at the peak)
    shape = (dict(decoder="plt") if cfg.startswith("config2") else
             dict(decoder="clt", layers=16, d_acts=2048, features=1024, shard_of=8) if "shard" in cfg else
             dict(decoder="clt", layers=16, d_acts=2048, features=8192))
    try:
        # ---- dense regime at the configuration's own theta = 0.03 (about 45 % of the entries active) -------------------------
        mod, batches, a = _build(**shape)
        batch = batches[0]
        out = mod.compute_step(batch, 0, sparse=False)
        assert out.get("lists") is None
        mse, sp, pa = _jumprelu_terms_from_outputs(mod, out, batch)
        assert abs(float(out["terms"]["mse"]) - float(mse)) <= 1e-4 * float(mse)
        assert abs(float(out["terms"]["sparsity"]) - float(sp)) <= 2e-3 * float(sp)          # (bf16 featurexnorm products)
        assert abs(float(out["terms"]["pre_actv"]) - float(pa)) <= 1e-4 * float(pa)
        # reconstruction on sampled (token, layer) pairs: sum over active features of bf16(feature) * bf16(W row) + bias
        pre, r, theta = out["pre_actvs"], out["recons_norm"], mod.model.nonlinearity.theta.detach()
        Ws = mod.model.decoder.weights()
        for b, l in ((0, 0), (1337, a.layers // 2), (a.batch - 1, a.layers - 1)):
            lo = 0 if a.decoder == "clt" else l
            p = pre[b, lo:l + 1]
            f = torch.where((p > theta[0, lo:l + 1]) & (p > 0), p, torch.zeros((), device=p.device)).bfloat16().double()
            W = (Ws[l] if a.decoder == "clt" else Ws[0][l:l + 1]).bfloat16().double()
            ref = torch.einsum("sf,sfd->d", f, W.detach()) + mod.model.decoder.b[l].detach().double()
            assert torch.allclose(r[b, l].double(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max())), (cfg, b, l)
        del mod, out, pre, r
        torch.cuda.empty_cache()
        # ---- the two regimes against each other at low L0 (theta raised: ~0.15 % active) --------------------------------------
        # (fp32 pre-activations in both: JumpReLU is discontinuous at theta, so with the dense regime's bf16 pre-activations
        # the ~2 % of the active entries within a bf16 ulp of theta switch — as they do under the reference's autocast, whose
        # encoder einsum returns bf16 — and the comparison would measure that, not the two code paths)
        monkeypatch.setattr(engine, "DENSE_BF16_ACTIVATIONS", False)
        res = {}
        for sparse in (False, True):
            mod, batches, a = _build(theta=0.26 if cfg.startswith("config2") else 0.9, **shape)
            out = mod.compute_step(batches[0], 0, sparse=sparse)
            assert (out.get("lists") is not None) == sparse
            res[sparse] = (float(out["loss"]), {k: v.clone() for k, v in out["grads"].items()}, out["recons_norm"].clone(),
                           float((out["features"] > 0).float().mean()) if sparse else None)
            del mod, out
            torch.cuda.empty_cache()
        assert 0 < res[True][3] < 0.01, res[True][3]
        assert abs(res[True][0] - res[False][0]) <= 2e-3 * abs(res[False][0])
        assert _rel(res[True][2], res[False][2]) < 2e-2
        for k, g in res[False][1].items():
            assert _rel(res[True][1][k].reshape(g.shape), g) < 2e-2, f"{cfg} grad {k}: {_rel(res[True][1][k].reshape(g.shape), g):.3g}"
        del res
        torch.cuda.empty_cache()
        monkeypatch.undo()
        # ---- fused Adam through training_step, auto regime (the device-side gate picks the lists at this density) -----
        mod, batches, a = _build(theta=0.26 if cfg.startswith("config2") else 0.9, **shape)
        mod.attach_optimizers()
        mses = []
        for t in range(4):                                   # the same batch every step: its reconstruction error must fall
            mod.training_step(batches[0], t)
            mses.append(float(mod.last_step["terms"]["mse"]))
        assert mod.last_step.get("lists") is not None, "low L0: the gate should choose the list regime"
        assert mses[3] < mses[1] < mses[0], mses
    finally:
        cta.set_precision("32")
        torch.cuda.empty_cache()


def test_topk_aux_config5_shard_full_shape():
    import crosslayer_transcoder_amd as cta
    shape = dict(nonlinearity="topk", decoder="clt", aux=True, layers=32, d_acts=4096, features=1024, k=16, shard_of=8)
    try:
        res = {}
        for sparse in (True, False):
            mod, batches, a = _build(**shape)
            out = mod.compute_step(batches[0], 0, sparse=sparse)
            if sparse:
                S = out["lists"]
                B, L, F, k = a.batch, a.layers, a.features, a.k
                assert int(S.total) == k * B * L and int(S.overflow) == 0
                # per-row TopK == torch.topk's index set on the same pre-activations (continuous values: no ties)
                pre, feat = out["pre_actvs"], out["features"]
                idx = pre.topk(k, dim=-1).indices
                want = torch.zeros_like(pre, dtype=torch.bool).scatter_(-1, idx, True) & (pre > 0)
                assert torch.equal(feat > 0, want)
                assert torch.equal(feat[want], pre[want])
                del pre, feat, idx, want
            res[sparse] = (float(out["loss"]), {k_: v.clone() for k_, v in out["grads"].items()}, out["recons_norm"].clone())
            del mod, out
            torch.cuda.empty_cache()
        assert abs(res[True][0] - res[False][0]) <= 2e-3 * abs(res[False][0])
        assert _rel(res[True][2], res[False][2]) < 2e-2
        for k_, g in res[False][1].items():
            assert _rel(res[True][1][k_].reshape(g.shape), g) < 2e-2, f"config 5 shard grad {k_}"
        del res
        torch.cuda.empty_cache()
        # training_step with fused Adam; from step 1 on features that never fired are dead and the aux pass is active
        # (one rank's share of k = 16 over 8 shards is about 2 per row; 100 features per layer get weak encoder rows, so they
        # stay positive half of the time yet never reach a row's top 2: dead from step 1 on, and picked up by the aux pass)
        mod, batches, a = _build(**dict(shape, k=2))
        with torch.no_grad():
            mod.model.encoder.W[:, :, :100].mul_(0.2)
        opt, _ = mod.attach_optimizers()
        W0 = mod.model.decoder.W_0.detach().clone()
        losses, aux = [], []
        for t in range(3):
            losses.append(float(mod.training_step(batches[t % 2], t)))
            aux.append(float(mod.last_step["terms"]["aux"]))
        # (no dead features at step 0: the aux reconstruction is the bias alone, so the aux term is the squared residual itself)
        assert all(torch.isfinite(torch.tensor(losses + aux))) and aux[0] > 0.0 and aux[1] != aux[0]
        dead = mod.last_step["dead"].bool()
        assert int(dead[:, :100].sum()) >= 0.9 * 100 * a.layers and int(dead[:, 100:].sum()) <= 0.02 * dead[:, 100:].numel()
        assert float(mod.last_step["terms"]["mse"]) < losses[0]
        # layer-0 features that are neither selected nor dead-and-chosen by the aux pass have zero gradient: rows unchanged
        moved = (mod.model.decoder.W_0.detach() != W0).any(dim=2)[0]
        m = opt.state[mod.model.decoder.W_0]["exp_avg"][0]
        assert torch.equal(moved, (m != 0).any(dim=1)), "exactly the rows with a gradient move"
        assert 0 < int(moved.sum()) <= a.features
    finally:
        cta.set_precision("32")
        torch.cuda.empty_cache()
