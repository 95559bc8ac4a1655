"""Full-size checks (BASELINE.json headline configuration: GPT-2-small CLT, L=12, d=768, F=32768/layer, B=4096,
BatchTopK k=16, bf16 operands) through size-independent properties — the oracle cannot run this size in seconds:

* the selected set is exactly {pre > T} plus the LOWEST flat indices of {pre == T}, T the smallest selected value (in the
  16-bit modes the pre-activations are the 16-bit encoder product + fp32 bias, as under the reference's autocast, so with
  the initial zero bias tens of thousands of entries tie at T), has exactly k*B*L entries, and the values are relu(pre)
  at those positions (recomputed from the pre-activations of the same encoder GEMM);
* the step whose encoder GEMM is fused with the candidate filter selects the identical set as the un-fused step;
* the gather forward equals an fp32 torch evaluation of sum(val * W_row) + bias on a sample of tokens;
* weight rows of features that did not fire (zero gradient, zero Adam moments) are bit-identical after the fused
  Adam row pass, rows that fired moved; the loss decreases over the first steps.
"""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(fuse):
    import bench
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import engine
    args = argparse.Namespace(layers=12, d_acts=768, features=32768, k=16, nonlinearity="batchtopk", decoder="clt", aux=False,
                              batch=4096, dead_steps=1, shard_rehearsal=False)
    cta.set_precision("bf16-mixed")
    engine.FUSE_ENCODER_SELECT = fuse
    dev = torch.device("cuda:0")
    mod = bench.build(args, dev)
    mod.log_metrics_every = 0
    return mod, mod.configure_optimizers(), bench.synth_batch(args.batch, args.layers, args.d_acts, dev), args


def test_headline_configuration_properties():
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import engine, ops
    try:
        mod, opt, batch, a = _build(True)
        B, L, F, d, k = a.batch, a.layers, a.features, a.d_acts, a.k
        enc_before = mod.model.encoder.Wt.detach().clone()
        dec_before = [w.detach().clone() for w in mod.model.decoder.weights()]
        dec_bias = mod.model.decoder.b.detach().clone()
        losses, sets, fired = [], [], None
        for t in range(3):
            loss = mod.fused_train_step(batch, t, opt)
            out = mod.last_step
            S = out["lists"]
            losses.append(float(loss))
            sets.append((S.col_start.clone(), S.csc_tok[: k * B * L].clone()))
            assert int(S.total) == k * B * L and int(S.overflow) == 0
            fired = S.fired.bool().clone() if fired is None else fired | S.fired.bool()
            if t == 1:
                assert "pre_actvs" not in dict.keys(out), "step 1 should run the fused encoder GEMM + filter"
            if t == 0:
                # selected set == {pre >= T}; values == relu(pre) there (pre from the same bf16 GEMM)
                pre = out["pre_actvs"]
                feat = out["features"]
                T = feat[feat > 0].min()
                want = (pre > T).view(-1)
                need = k * B * L - int(want.sum())
                ties = (pre == T).view(-1).nonzero().view(-1)              # ascending flat index
                assert 0 < need <= ties.numel(), (need, ties.numel())
                want[ties[:need]] = True                                   # canonical tie rule: lower flat index first
                print(f"[headline selection] {ties.numel()} entries tie at the k-th value, {need} of them selected")
                assert torch.equal(feat, torch.where(want.view_as(pre), pre.clamp_min(0.0), torch.zeros((), device=pre.device)))
                del want, ties
                # gather forward vs fp32 torch on 4 tokens (bf16 shadow weights, as the kernel reads them)
                r = out["recons_norm"]
                for b in (0, 1337, 4095):
                    for l in (0, 5, 11):
                        ref = dec_bias[l].double().clone()
                        f_bl = feat[b, : l + 1]                                   # [(l+1), F]
                        for lp, i in f_bl.nonzero().tolist():
                            ref += float(f_bl[lp, i]) * dec_before[l][lp, i].bfloat16().double()
                        assert torch.allclose(r[b, l].double(), ref, rtol=1e-4, atol=1e-5), (b, l)
                del pre, feat, dec_before
        assert losses[2] < losses[1] < losses[0], losses
        # rows that never fired in 3 identical batches: zero gradient, zero moments -> bit-identical weights
        enc_after = mod.model.encoder.Wt.detach()
        same = (enc_after == enc_before).all(dim=2)
        assert bool(same[~fired].all()), "weights of features without gradient must not move"
        assert bool((~same[fired]).all()), "weights of features that fired must move"
        m = opt.state[mod.model.encoder.W]["exp_avg"].permute(0, 2, 1)       # logical [L, d, F] -> [L, F, d]
        assert float(m[~fired].abs().max()) == 0.0
        del mod, opt, enc_before, enc_after, same, m, out, S
        torch.cuda.empty_cache()
        # the same three steps without the fused encoder GEMM + filter: identical selections, same losses
        mod2, opt2, batch2, _ = _build(False)
        for t in range(3):
            loss2 = mod2.fused_train_step(batch2, t, opt2)
            S2 = mod2.last_step["lists"]
            assert torch.equal(S2.col_start, sets[t][0]) and torch.equal(S2.csc_tok[: k * B * L], sets[t][1]), f"step {t}: selection differs"
            assert abs(float(loss2) - losses[t]) <= 1e-5 * abs(losses[t])
    finally:
        engine.FUSE_ENCODER_SELECT = True
        cta.set_precision("32")
        torch.cuda.empty_cache()


def test_perlayer_topk_streaming_selection_full_size():
    """PerLayerTopK straight to lists at the headline shape (49152 rows x 32768, k = 16): the two streaming passes (chunk top
    k, row merge) select, row by row, exactly what the dense-output per-row kernel selects (itself oracle-checked at small
    sizes); every row holds k entries, in ascending feature order, and the column counts add up."""
    from crosslayer_transcoder_amd import ops
    B, L, F, k = 4096, 12, 32768, 16
    g = torch.Generator(device="cuda:0").manual_seed(5)
    x = torch.randn(B, L, F, device="cuda:0", generator=g)
    x[7, 3] = 0.5                                        # a constant row: ties resolved by the lowest indices
    x[9, 1] = -1.0                                       # nothing positive: an empty row
    S = ops.select_lists(x, k, ops.SELECT_ROWS)
    dense = ops.topk_rows(x, k)
    assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val), dense)
    rs = S.row_start.long()
    cnt = (rs[1:] - rs[:-1]).view(B, L)
    assert int(cnt[9, 1]) == 0 and int(cnt[7, 3]) == k
    cnt[9, 1] = k
    assert bool((cnt == k).all())
    assert torch.equal(S.ent_col[int(rs[7 * L + 3]):int(rs[7 * L + 3]) + k].cpu() - 3 * F, torch.arange(k, dtype=torch.int32))
    del dense
    torch.cuda.empty_cache()


def test_perlayer_topk_fused_with_encoder_full_size():
    """PerLayerTopK at the headline shape with the selection in the encoder GEMM's epilogue (49152 rows, bounds from a 1/16
    feature sample, about 160 candidates a row): every list array equals the unfused path's (encoder GEMM, 6.4 GB of
    pre-activations, streaming rows selection), for two sample offsets; the slot blocks stay far from their capacity."""
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import ops
    from crosslayer_transcoder_amd._lib import lib
    B, L, d, F, k = 4096, 12, 768, 32768, 16
    try:
        cta.set_precision("bf16-mixed")
        ops.sync_operand_format()
        g = torch.Generator(device="cuda:0").manual_seed(6)
        xs = torch.randn(B, L, d, device="cuda:0", generator=g).bfloat16()
        Wt = ((torch.rand(L, F, d, device="cuda:0", generator=g) * 2 - 1) * F ** -0.5).bfloat16()
        b = torch.randn(L, F, device="cuda:0", generator=g) * 0.02
        want = ops.select_lists(ops.encoder_fwd(xs, Wt, b), k, ops.SELECT_ROWS)
        torch.cuda.empty_cache()
        j, slots = ops._rows_plan(k)
        dims = (B, L, F, k, slots, F // ops.ROWS_SAMPLE_STRIDE, j)
        for offset in (0, 7):
            got = ops.encoder_select_rows_lists(xs, Wt, b, k, offset=offset)
            assert got.confirm()
            n = int(want.total)
            assert n == int(got.total) == k * B * L
            for name in ("row_start", "col_start", "fired"):
                assert torch.equal(getattr(got, name), getattr(want, name)), name
            for name in ("ent_col", "ent_val", "csc_tok", "csc_val", "csc_ent"):
                assert torch.equal(getattr(got, name)[:n], getattr(want, name)[:n]), name
            ws = ops.workspace(lib.clt_select_rows_ext_workspace_bytes(*dims), xs.device)
            o = lib.clt_select_rows_ext_info(*dims, 1)
            cnt = ws[o:o + 4 * B * L].view(torch.int32)
            print(f"[fused rows, offset {offset}] candidates per row: mean {float(cnt.float().mean()):.1f}, max {int(cnt.max())} of {slots} slots")
            assert int(cnt.max()) <= slots // 2
    finally:
        cta.set_precision("32")
        ops.sync_operand_format()
        torch.cuda.empty_cache()
