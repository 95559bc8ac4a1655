"""GPU parity of each C-ABI entry point (through crosslayer_transcoder_amd.ops -> libclt_hip.so) against the
CPU oracle on the same seeded inputs.  fp32 tolerance: rtol 1e-5 (+ atol scaled to the output magnitude for
sums); selection: bit-exact."""
import pytest
import torch

from oracle import clt_oracle as O
from tests.golden_cases import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from crosslayer_transcoder_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True)
def _raw_gemm_results(request, monkeypatch):
    """The kernel tests of this file check the GEMMs' fp32 accumulators themselves; the encoder's autocast-order rounding of
    the 16-bit modes (ops.AUTOCAST_PRE -> CLT_GEMM_ROUND_ACC) is switched off for them and has its own test below."""
    from crosslayer_transcoder_amd import ops as _ops
    if "autocast_order" not in request.node.name:
        monkeypatch.setattr(_ops, "AUTOCAST_PRE", False)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def tol(ref, rtol=1e-5):
    return dict(rtol=rtol, atol=rtol * float(ref.abs().max()) + 1e-12)


@pytest.mark.parametrize("B,L,d", [(16, 2, 110), (7, 3, 13), (300, 4, 64)])
def test_standardizer(ops, B, L, d):
    batch = _rand(B, 2, L, d, seed=1) * 2 + 0.5
    ref = O.standardizer_init(batch)
    bd = batch.to(DEV)
    m, s = ops.column_stats(bd[:, 0])
    assert_close(m, ref["in_mean"], 1e-5, 1e-6, "mean"); assert_close(s, ref["in_std"], 1e-5, 1e-7, "std")
    m1, s1 = ops.column_stats(bd[:, 1])
    assert_close(m1, ref["out_mean"], 1e-5, 1e-6); assert_close(s1, ref["out_std"], 1e-5, 1e-7)
    xs32, xs16 = ops.standardize(bd[:, 0], ref["in_mean"].to(DEV), ref["in_std"].to(DEV), True, True)
    want = O.standardize(batch[:, 0], ref["in_mean"], ref["in_std"])
    assert torch.equal(xs32.cpu(), want)                          # same IEEE sub + div
    assert torch.equal(xs16.cpu(), want.to(torch.bfloat16))
    r = _rand(B, L, d, seed=2)
    assert_close(ops.destandardize(r.to(DEV), m1, s1), O.destandardize(r, ref["out_mean"], ref["out_std"]), 1e-6, 1e-6)


@pytest.mark.parametrize("B,L,d,F", [(16, 2, 110, 32), (5, 3, 7, 9), (130, 2, 96, 200), (64, 12, 768, 256)])
def test_dense_contractions_f32(ops, B, L, d, F):
    xs, W, b = _rand(B, L, d, seed=1), _rand(L, d, F, seed=2, scale=F ** -0.5), _rand(L, F, seed=3, scale=0.1)
    Wt = W.transpose(1, 2).contiguous()                       # feature-major storage [L,F,d]
    pre = ops.encoder_fwd(xs.to(DEV), Wt.to(DEV), b.to(DEV))
    ref = O.encoder_fwd(xs.double(), W.double(), b.double())
    assert_close(pre, ref, **tol(ref), what="encoder fwd")
    dpre = _rand(B, L, F, seed=4)
    dWt = ops.encoder_wgrad(xs.to(DEV), dpre.to(DEV))
    rW, _ = O.encoder_bwd(xs.double(), dpre.double())
    assert_close(dWt.permute(0, 2, 1), rW, **tol(rW), what="encoder wgrad")
    rdx = torch.stack([dpre[:, l].double() @ W[l].double().T for l in range(L)], 1)
    assert_close(ops.encoder_dgrad(dpre.to(DEV), Wt.to(DEV)), rdx, **tol(rdx), what="encoder dgrad")
    feat = torch.relu(_rand(B, L, F, seed=5))
    Ws = [_rand(l + 1, F, d, seed=10 + l, scale=0.05) for l in range(L)]
    bd = _rand(L, d, seed=6, scale=0.1)
    r = ops.clt_decoder_fwd(feat.to(DEV), [w.to(DEV) for w in Ws], bd.to(DEV))
    rr = O.crosslayer_decoder_fwd(feat.double(), [w.double() for w in Ws], bd.double())
    assert_close(r, rr, **tol(rr), what="clt decoder fwd")
    g = _rand(B, L, d, seed=7)
    rdf, rdWs, _ = O.crosslayer_decoder_bwd(feat.double(), [w.double() for w in Ws], g.double())
    df = ops.clt_decoder_dgrad(g.to(DEV), [w.to(DEV) for w in Ws])
    assert_close(df, rdf, **tol(rdf), what="clt decoder dgrad")
    for l in range(L):
        assert_close(ops.clt_decoder_wgrad(feat.to(DEV), g.to(DEV), l), rdWs[l], **tol(rdWs[l]), what=f"wgrad {l}")
    Wp = _rand(L, F, d, seed=8, scale=0.05)
    assert_close(ops.plt_decoder_fwd(feat.to(DEV), Wp.to(DEV), bd.to(DEV)),
                 O.plt_decoder_fwd(feat.double(), Wp.double(), bd.double()), **tol(rr), what="plt fwd")
    pdf, pdW, _ = O.plt_decoder_bwd(feat.double(), Wp.double(), g.double())
    assert_close(ops.plt_decoder_dgrad(g.to(DEV), Wp.to(DEV)), pdf, **tol(pdf), what="plt dgrad")
    assert_close(ops.plt_decoder_wgrad(feat.to(DEV), g.to(DEV)), pdW, **tol(pdW), what="plt wgrad")
    assert_close(ops.column_sum(g.to(DEV)), g.double().sum(0), **tol(g.double().sum(0)), what="colsum")


def test_gemm_layout_asymmetric(ops):
    """A = I with an asymmetric B catches a transposed C write (cdna guide §3)."""
    n = 96
    Bm = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    eye = torch.eye(n)
    out = ops.plt_decoder_fwd(eye.reshape(n, 1, n).to(DEV), Bm.reshape(1, n, n).to(DEV), None)
    assert torch.equal(out.cpu().reshape(n, n), Bm)


@pytest.mark.parametrize("B,L,d,F", [(64, 2, 128, 256), (130, 3, 72, 200)])
def test_dense_contractions_bf16(ops, B, L, d, F):
    """bf16 operands, fp32 accumulate: compared with the fp64 product of the SAME bf16-rounded operands."""
    bf = torch.bfloat16
    xs, W = _rand(B, L, d, seed=1).to(bf), _rand(L, d, F, seed=2, scale=F ** -0.5).to(bf)
    b = _rand(L, F, seed=3, scale=0.1)
    ref = O.encoder_fwd(xs.double(), W.double(), b.double())
    Wt = W.transpose(1, 2).contiguous()
    assert_close(ops.encoder_fwd(xs.to(DEV), Wt.to(DEV), b.to(DEV)), ref, **tol(ref, 2e-6), what="enc bf16 NT")
    dpre = _rand(B, L, F, seed=4).to(bf)
    rW, _ = O.encoder_bwd(xs.double(), dpre.double())
    assert_close(ops.encoder_wgrad(xs.to(DEV), dpre.to(DEV)).permute(0, 2, 1), rW, **tol(rW, 2e-6), what="enc wgrad bf16")
    feat = torch.relu(_rand(B, L, F, seed=5)).to(bf)
    Ws = [_rand(l + 1, F, d, seed=10 + l, scale=0.05).to(bf) for l in range(L)]
    g = _rand(B, L, d, seed=7).to(bf)
    rr = O.crosslayer_decoder_fwd(feat.double(), [w.double() for w in Ws], torch.zeros(L, d).double())
    assert_close(ops.clt_decoder_fwd(feat.to(DEV), [w.to(DEV) for w in Ws], None), rr, **tol(rr, 2e-6), what="dec bf16")
    rdf, rdWs, _ = O.crosslayer_decoder_bwd(feat.double(), [w.double() for w in Ws], g.double())
    assert_close(ops.clt_decoder_dgrad(g.to(DEV), [w.to(DEV) for w in Ws]), rdf, **tol(rdf, 2e-6), what="dgrad bf16")
    assert_close(ops.clt_decoder_wgrad(feat.to(DEV), g.to(DEV), L - 1), rdWs[L - 1], **tol(rdWs[L - 1], 2e-6))
    x32 = _rand(3, 70, 130, seed=9)
    assert torch.equal(ops.cast_transpose_bf16(x32.to(DEV)).cpu(), x32.to(bf).transpose(1, 2).contiguous())
    assert torch.equal(ops.cast_bf16(x32.to(DEV)).cpu(), x32.to(bf))


def test_relu_jumprelu(ops, golden):
    c = golden("jumprelu_fn")
    x, th, go = c["x"].to(DEV), c["theta"].to(DEV), c["grad_out"].to(DEV)
    assert torch.equal(ops.jumprelu_fwd(x, th).cpu(), c["out"])
    dx, dth = ops.jumprelu_bwd(x, th, float(c["bandwidth"]), go)
    assert torch.equal(dx.cpu(), c["dx"])
    assert_close(dth, c["dtheta"], 1e-5, 1e-6, "dtheta")
    xx = _rand(33, 5, 17, seed=3)
    gg = _rand(33, 5, 17, seed=4)
    assert torch.equal(ops.relu_fwd(xx.to(DEV)).cpu(), O.relu_fwd(xx))
    assert torch.equal(ops.relu_bwd(xx.to(DEV), gg.to(DEV)).cpu(), O.relu_bwd(xx, gg))
    # larger, with many points inside the rectangle window
    x = _rand(513, 3, 300, seed=5, scale=0.1); th = torch.full((1, 3, 300), 0.05) + _rand(1, 3, 300, seed=6, scale=0.01)
    go = _rand(513, 3, 300, seed=7)
    dxr, dthr = O.jumprelu_bwd(x, th, 0.04, go)
    dx, dth = ops.jumprelu_bwd(x.to(DEV), th.to(DEV), 0.04, go.to(DEV))
    assert torch.equal(dx.cpu(), dxr)
    assert_close(dth, dthr.double(), 1e-4, 1e-4 * float(dthr.abs().max()), "dtheta big")


def test_selection_golden(ops, golden):
    sel = golden("selection")
    for key, c in sel.items():
        x, k = c["x"].to(DEV), int(c["k"])
        L = x.shape[1]
        assert torch.equal(ops.topk_rows(x, k).cpu(), c["topk"]), key
        thr = torch.zeros(1, device=DEV)
        assert torch.equal(ops.topk_segmented(x, k, False, thr, 0.1).cpu(), c["batchtopk"]), key
        assert_close(thr, c["batchtopk_threshold"], 1e-6, 0)
        thr = torch.zeros(L, device=DEV)
        assert torch.equal(ops.topk_segmented(x, k, True, thr, 0.1).cpu(), c["perlayerbatchtopk"]), key
        assert_close(thr, c["perlayerbatchtopk_threshold"], 1e-6, 0)


@pytest.mark.parametrize("B,L,F,k", [(3, 4, 5, 3), (1, 1, 5, 3), (2, 8, 10, 3), (1, 12, 768, 3), (10, 1, 3, 3),
                                     (64, 3, 4096, 16), (9, 2, 70000, 33), (300, 12, 1024, 8),
                                     (8, 2, 10000, 16), (8, 2, 20000, 16), (6, 2, 24000, 70), (5, 1, 32768, 300),
                                     (40, 2, 2500, 16), (40, 2, 3072, 200), (33, 1, 2052, 700), (20, 3, 4100, 16)])
def test_selection_vs_oracle(ops, B, L, F, k):
    x = _rand(B, L, F, seed=B + F)
    xd = x.to(DEV)
    f, idx, val = ops.topk_rows(xd, k, want_compact=True)
    ref, ridx = O.per_layer_topk_fwd(x, k)
    assert torch.equal(f.cpu(), ref)
    ridx_sorted = ridx.sort(-1).values.reshape(B * L, k)
    rval = torch.gather(ref.reshape(B * L, F), 1, ridx_sorted)
    pos = rval > 0
    got_idx, got_val = idx.cpu(), val.cpu()
    for r in range(0, B * L, max(1, B * L // 50)):     # positive entries, ascending feature id, then (-1, 0) padding
        n = int(pos[r].sum())
        assert torch.equal(got_idx[r, :n].long(), ridx_sorted[r][pos[r]]) and torch.equal(got_val[r, :n], rval[r][pos[r]])
        assert (got_idx[r, n:] == -1).all() and (got_val[r, n:] == 0).all()
    thr = torch.full((1,), 0.25, device=DEV)
    fb = ops.topk_segmented(xd, k, False, thr, 0.1)
    rb, _, mk = O.batch_topk_fwd(x, k)
    assert torch.equal(fb.cpu(), rb)
    assert_close(thr, O.threshold_ema(torch.full((1,), 0.25), mk, 0.1), 1e-6, 1e-8)
    thr = torch.zeros(L, device=DEV)
    fl = ops.topk_segmented(xd, k, True, thr, 0.05)
    rl, _, mk = O.per_layer_batch_topk_fwd(x, k)
    assert torch.equal(fl.cpu(), rl)
    assert_close(thr, O.threshold_ema(torch.zeros(L), mk, 0.05), 1e-6, 1e-8)


def test_selection_edges_and_ties(ops):
    x = (torch.rand(3, 4, 5) + 0.1).to(DEV)                        # reference tests/test_topk.py:137-180
    for fn in (lambda k: ops.topk_rows(x, k), lambda k: ops.topk_segmented(x, k, False, None, 0.0),
               lambda k: ops.topk_segmented(x, k, True, None, 0.0)):
        assert torch.equal(fn(0), torch.zeros_like(x))
        assert torch.equal(fn(5), x)
        with pytest.raises(AssertionError):
            fn(6)
    # heavy positive ties: canonical rule = larger value, then lower index — bit-exact with the oracle
    g = torch.Generator().manual_seed(0)
    t = torch.randint(0, 6, (40, 3, 512), generator=g).float() - 1.0
    td = t.to(DEV)
    for k in (1, 7, 100):
        assert torch.equal(ops.topk_rows(td, k).cpu(), O.per_layer_topk_fwd(t, k)[0])
        assert torch.equal(ops.topk_segmented(td, k, False, None, 0.0).cpu(), O.batch_topk_fwd(t, k)[0])
        assert torch.equal(ops.topk_segmented(td, k, True, None, 0.0).cpu(), O.per_layer_batch_topk_fwd(t, k)[0])
    # aux path: -inf masked rows with fewer finite entries than k
    m = _rand(8, 3, 64, seed=2)
    dead = (torch.rand(3, 64, generator=g) < 0.1)
    md = ops.mask_dead(m.to(DEV), dead.to(torch.uint8).to(DEV))
    mref = torch.where(dead.expand_as(m), m, torch.full_like(m, -float("inf")))
    assert torch.equal(md.cpu(), mref)
    assert torch.equal(ops.topk_rows(md, 20).cpu(), O.per_layer_topk_fwd(mref, 20)[0])
    # known answers (reference tests/test_topk.py:77-113)
    ka = torch.tensor([[[1, 2, 3, 4, 5], [6, 7, 8, 9, 10]]], dtype=torch.float32, device=DEV)
    assert torch.equal(ops.topk_rows(ka, 3).cpu(), torch.tensor([[[0., 0, 3, 4, 5], [0, 0, 8, 9, 10]]]))
    two = torch.tensor([[[1, 2, 3, 4, 5], [6, 7, 8, 9, 10]], [[11, 12, 13, 14, 15], [16, 17, 18, 19, 20]]],
                       dtype=torch.float32, device=DEV)
    fb = ops.topk_segmented(two, 3, False, None, 0.0).cpu()
    assert (fb != 0).sum() == 12 and fb[1, 1, 4] == 20 and fb[1, 1, 3] == 19
    fl = ops.topk_segmented(two, 3, True, None, 0.0).cpu()
    assert (fl[:, 0] != 0).sum() == 6 and (fl[:, 1] != 0).sum() == 6
    # inference thresholds (topk.py:71-74, 104-107)
    thr = torch.tensor([0.3, -1.0, 0.7], device=DEV)
    xi = _rand(5, 3, 33, seed=9)
    out = ops.threshold_fwd(xi.to(DEV), thr, True, 3).cpu()
    for l in range(3):
        assert torch.equal(out[:, l], O.topk_inference(xi[:, l], float(thr[l])))
    assert torch.equal(ops.threshold_fwd(xi.to(DEV), thr[:1], False, 3).cpu(), O.topk_inference(xi, 0.3))
    assert torch.equal(ops.mask_positive(fb.to(DEV), two).cpu(), O.topk_bwd(fb, two.cpu()))


def test_losses_and_bookkeeping(ops):
    B, L, d, F = 37, 3, 20, 48
    r, y = _rand(B, L, d, seed=1), _rand(B, 2, L, d, seed=2)
    mean, std = _rand(L, d, seed=3), torch.rand(L, d) + 0.5
    ls, g, ys = ops.mse(r.to(DEV), y.to(DEV)[:, 1], mean.to(DEV), std.to(DEV), True, True)
    ysr = O.standardize(y[:, 1], mean, std)
    diff = (r - ysr).double()
    assert_close(ls[0] / diff.numel(), (diff ** 2).mean(), 1e-6, 0, "mse")
    assert_close(g, 2 * diff / diff.numel(), 1e-5, 1e-9, "g"); assert torch.equal(ys.cpu(), ysr)
    ar = _rand(B, L, d, seed=4); ar[0, 0, 0] = float("nan"); ar[1, 1, 1] = float("inf")
    gm = g.clone()
    s, ga = ops.aux_mse(ar.to(DEV), ys, r.to(DEV), gm, 0.25)
    e = ar - (ysr - r)
    aux = torch.nan_to_num(e * e, 0.0)
    assert_close(s[0] / e.numel(), aux.double().mean(), 1e-6, 0, "aux")
    gref = (0.25 * 2 / e.numel()) * e * torch.isfinite(e * e); gref[~torch.isfinite(e * e)] = 0
    assert_close(ga, gref, 1e-5, 1e-9, "g_aux"); assert_close(gm, g.cpu() + gref, 1e-5, 1e-9, "g_main")
    # JumpReLU module loss terms + fused backward vs the oracle formulas
    pre = _rand(B, L, F, seed=5, scale=0.2); theta = torch.full((1, L, F), 0.05) + _rand(1, L, F, seed=6, scale=0.02)
    Ws = [_rand(l + 1, F, d, seed=20 + l, scale=0.3) for l in range(L)]
    sq = torch.zeros(L * F, device=DEV)
    for l in range(L):
        ops.row_sqnorm(Ws[l].to(DEV), sq, 0, True)
    n = O.decoder_norms({f"decoder.W_{l}": Ws[l] for l in range(L)}, "clt", L)
    assert_close(sq.sqrt().reshape(1, L, F), n, 1e-5, 1e-7, "dec_norms")
    dfm = _rand(B, L, F, seed=7, scale=0.01)
    lam, c, coef, bw = 0.3 / B, 0.7, 1e-2 / (B * L), 0.04
    for use_tanh in (True, False):
        sums, dpre, dth, dn = ops.jumprelu_loss_bwd(pre.to(DEV), theta.to(DEV), sq, dfm.to(DEV), bw, lam, c, use_tanh, coef)
        f = O.jumprelu_fwd(pre, theta); w = f * n * c; t = torch.tanh(w) if use_tanh else w
        dt = 1 - t * t if use_tanh else torch.ones_like(t)
        df = dfm + lam * dt * c * n
        dxr, dthr = O.jumprelu_bwd(pre, theta, bw, df)
        assert_close(sums[0], t.double().sum(), 1e-6, 1e-9); assert_close(sums[1], torch.relu(-pre).double().sum(), 1e-6, 0)
        assert_close(sums[2], w.double().sum(), 1e-6, 1e-9)
        assert_close(dpre, dxr - coef * (pre < 0), 1e-5, 1e-6 * float(dxr.abs().max()), "dpre")
        assert_close(dth, dthr, 1e-4, 1e-5 * float(dthr.abs().max()), "dtheta")
        assert_close(dn.reshape(L, F), (lam * dt * c * f).sum(0), 1e-4, 1e-8, "dnorm")
    dW = _rand(L, F, d, seed=9).to(DEV); dW0 = dW.clone().cpu()
    ops.rowscale_add(Ws[L - 1].to(DEV), dn, sq, dW)
    assert_close(dW, dW0 + (dn.cpu() / sq.cpu().sqrt()).reshape(L, F, 1) * Ws[L - 1], 1e-5, 1e-7, "rowscale")
    # dead features
    feat = torch.relu(_rand(B, L, F, seed=11) - 1.5)
    la = torch.randint(0, 100, (L, F)); lad = la.clone().to(DEV)
    dead, cnt = ops.dead_update(ops.column_any_positive(feat.to(DEV)), B, lad, 60)
    lar, deadr = O.update_dead_features(la, feat, 60)
    assert torch.equal(lad.cpu(), lar) and torch.equal(dead.cpu().bool(), deadr) and int(cnt) == int(deadr.sum())


def test_adam(ops):
    n = 1000 * 37 + 3
    p, g = _rand(n, seed=1), _rand(n, seed=2, scale=0.01)
    for adamw, wd in ((False, 0.0), (True, 0.01)):
        pd, m, v = p.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        pr, mr, vr = p.clone(), torch.zeros(n), torch.zeros(n)
        sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        for step in range(1, 4):
            gs = g * step
            ops.adam(pd, gs.to(DEV), m, v, step, 1e-3, 0.9, 0.999, 1e-8, wd, adamw, sh)
            pr, mr, vr = O.adam_step(pr, gs, mr, vr, step, 1e-3, 0.9, 0.999, 1e-8, wd, adamw)
        assert_close(pd, pr, 1e-6, 1e-7, "p"); assert_close(m, mr, 1e-5, 1e-9, "m"); assert_close(v, vr, 1e-5, 1e-12, "v")
        assert torch.equal(sh.cpu(), pd.cpu().to(torch.bfloat16))


# ------------------------------------------------------------------------------------ sparse (list) path
def _sparse_features(B, L, F, k, seed):
    pre = _rand(B, L, F, seed=seed)
    feat, _ = O.per_layer_topk_fwd(pre, k)
    return feat


@pytest.mark.parametrize("B,L,F,d,k", [(16, 2, 32, 110, 4), (33, 3, 100, 20, 7), (8, 12, 512, 768, 16), (5, 2, 40, 7, 40),
                                       (7, 3, 40, 1536, 4), (6, 2, 48, 4096, 5)])
def test_sparse_lists_and_decoder(ops, B, L, F, d, k):
    """CSR/CSC construction is exact; gather forward and the fused row pass (dW, dots, Adam) match the dense
    oracle formulas."""
    feat = _sparse_features(B, L, F, k, seed=B + F)
    if k == F:
        feat[0] = 0                                           # an empty token row
    fd = feat.to(DEV)
    cap = k * B * L
    S = ops.sparse_build(fd, cap)
    nz = feat.reshape(B * L, F) > 0
    n = int(nz.sum())
    assert int(S.total) == n and int(S.overflow) == 0
    cnt = nz.sum(1)
    assert torch.equal(S.row_start.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]))
    rows, cols = nz.nonzero(as_tuple=True)                     # row-major order == CSR order
    layer = rows % L
    assert torch.equal(S.ent_col.cpu()[:n].long(), layer * F + cols)
    assert torch.equal(S.ent_val.cpu()[:n], feat.reshape(B * L, F)[nz])
    colmat = feat.permute(1, 2, 0).reshape(L * F, B) > 0       # [column, token]
    assert torch.equal(S.col_start.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), colmat.sum(1).cumsum(0)]))
    cc, tt = colmat.nonzero(as_tuple=True)
    assert torch.equal(S.csc_tok.cpu()[:n].long(), tt)
    assert torch.equal(S.csc_val.cpu()[:n], feat.permute(1, 2, 0).reshape(L * F, B)[colmat])
    assert torch.equal(S.fired.cpu().bool().reshape(-1), colmat.any(1))
    assert torch.equal(S.ent_val.cpu()[S.csc_ent.cpu()[:n].long()], S.csc_val.cpu()[:n])
    assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), feat)
    # overflow is flagged, not fatal
    if n > 1:
        assert int(ops.sparse_build(fd, n - 1).overflow) == 1

    g = _rand(B, L, d, seed=7)
    gd = g.to(DEV)
    bias = _rand(L, d, seed=8, scale=0.1)
    for cross in (True, False):
        if cross:
            Ws = [_rand(l + 1, F, d, seed=10 + l, scale=0.05) for l in range(L)]
            Wd = [w.to(DEV) for w in Ws]
            wptr = ops.pointer_table(Wd)
            ref = O.crosslayer_decoder_fwd(feat.double(), [w.double() for w in Ws], bias.double())
            rdf, rdW, _ = O.crosslayer_decoder_bwd(feat.double(), [w.double() for w in Ws], g.double())
        else:
            Wp = _rand(L, F, d, seed=9, scale=0.05)
            Ws, Wd = [Wp], [Wp.to(DEV)]
            wptr = ops.pointer_table([Wd[0][l] for l in range(L)])
            ref = O.plt_decoder_fwd(feat.double(), Wp.double(), bias.double())
            rdf, rdWp, _ = O.plt_decoder_bwd(feat.double(), Wp.double(), g.double())
            rdW = [rdWp]
        out = ops.sparse_decoder_fwd(S, wptr, bias.to(DEV), d, cross)
        assert_close(out, ref, **tol(ref), what=f"sparse fwd cross={cross}")
        slots = L if cross else 1
        dots = torch.zeros(slots, cap, device=DEV)
        for l, w in enumerate(Wd):
            dW = torch.empty_like(w)
            ops.sparse_row_pass(w, S, S.csc_val, gd, 0, l if cross else -1, grad_out=dW, dots=dots,
                                dot_slot=-1 if cross else 0)
            assert_close(dW, rdW[l], **tol(rdW[l]), what=f"sparse wgrad {l} cross={cross}")
        dval = ops.sparse_reduce_dots(dots, S, cross)
        dfeat = ops.sparse_scatter_dense(S, dval).cpu()
        want = (rdf * (feat > 0)).float()
        assert_close(dfeat, want, **tol(want), what=f"sparse dfeat cross={cross}")
        # bf16 gather source == fp32 source holding the bf16-rounded values, bit for bit
        gh = g.to(torch.bfloat16)
        dWa, dWb = torch.empty_like(Wd[-1]), torch.empty_like(Wd[-1])
        da, db_ = torch.zeros(slots, cap, device=DEV), torch.zeros(slots, cap, device=DEV)
        lsrc = (len(Wd) - 1) if cross else -1
        ops.sparse_row_pass(Wd[-1], S, S.csc_val, gh.to(DEV), 0, lsrc, grad_out=dWa, dots=da, dot_slot=-1 if cross else 0)
        ops.sparse_row_pass(Wd[-1], S, S.csc_val, gh.float().to(DEV), 0, lsrc, grad_out=dWb, dots=db_, dot_slot=-1 if cross else 0)
        assert torch.equal(dWa, dWb) and torch.equal(da, db_)
        # fused Adam == gradient + clt_adam_f32 (same per-element arithmetic up to FMA contraction)
        w = Wd[-1]
        wa, wb = w.clone(), w.clone()
        m1, v1 = torch.rand_like(w) * 1e-3, torch.rand_like(w) * 1e-6
        m2, v2 = m1.clone(), v1.clone()
        hp = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, adamw=True, step=3)
        sh = torch.empty(w.shape, dtype=torch.bfloat16, device=DEV)
        ops.sparse_row_pass(wa, S, S.csc_val, gd, 0, (len(Wd) - 1) if cross else -1, adam=hp, m=m1, v=v1, shadow=sh)
        dW = torch.empty_like(w)
        ops.sparse_row_pass(wb, S, S.csc_val, gd, 0, (len(Wd) - 1) if cross else -1, grad_out=dW)
        ops.adam(wb, dW, m2, v2, 3, 1e-3, 0.9, 0.999, 1e-8, 0.01, True)
        assert_close(wa, wb, 1e-6, 1e-8, "fused adam p"); assert_close(m1, m2, 1e-6, 1e-10, "m"); assert_close(v1, v2, 1e-6, 1e-13, "v")
        assert torch.equal(sh.cpu(), wa.cpu().to(torch.bfloat16))


def test_sparse_row_pass_two_lists_and_encoder(ops):
    """Second (aux) list, per-column coefficient sums and the layer-from-column addressing used by the encoder."""
    B, L, F, d = 21, 3, 64, 24
    f1, f2 = _sparse_features(B, L, F, 5, seed=1), _sparse_features(B, L, F, 9, seed=2)
    S1, S2 = ops.sparse_build(f1.to(DEV), 5 * B * L), ops.sparse_build(f2.to(DEV), 9 * B * L)
    xs = _rand(B, L, d, seed=3)
    c1, c2 = _rand(S1.cap, seed=4).to(DEV), _rand(S2.cap, seed=5).to(DEV)
    dp1, dp2 = ops.sparse_scatter_dense(S1, c1).cpu().double(), ops.sparse_scatter_dense(S2, c2).cpu().double()
    Wt = _rand(L, F, d, seed=6).to(DEV)
    dWt = torch.empty_like(Wt)
    db = torch.empty(L, F, device=DEV)
    dots2 = torch.zeros(1, S2.cap, device=DEV)
    ops.sparse_row_pass(Wt, S1, c1, xs.to(DEV), 0, -1, grad_out=dWt, colsum=db, second=(S2, c2, xs.to(DEV)), dots2=dots2)
    dpre = dp1 + dp2
    rW, rb = O.encoder_bwd(xs.double(), dpre)
    assert_close(dWt.permute(0, 2, 1), rW, **tol(rW), what="encoder wgrad from lists")
    assert_close(db, rb, **tol(rb), what="encoder db from lists")
    want = torch.stack([xs[:, l].double() @ Wt[l].cpu().double().T for l in range(L)], 1) * (f2 > 0)
    got = ops.sparse_scatter_dense(S2, ops.sparse_reduce_dots(dots2, S2, False)).cpu()
    assert_close(got, want.float(), **tol(want), what="dots of the 2nd list")


def _lists_equal(A, Bl, what):
    n = int(A.total)
    assert n == int(Bl.total), f"{what}: total {n} vs {int(Bl.total)}"
    assert int(A.overflow) == 0 and int(Bl.overflow) == 0
    for name in ("row_start", "col_start", "fired"):
        assert torch.equal(getattr(A, name), getattr(Bl, name)), f"{what}: {name}"
    for name in ("ent_col", "ent_val", "csc_tok", "csc_val", "csc_ent"):
        assert torch.equal(getattr(A, name)[:n], getattr(Bl, name)[:n]), f"{what}: {name}"


def test_list_columns_and_rows_are_ordered_at_every_length(ops):
    """The feature-major lists hold every column's entries in ascending token order and the token-major lists every row's in
    ascending column order, whatever the length: up to 24 entries one thread orders them, up to 1024 one wave (rank counting),
    beyond that a block (bitonic) — checked entry by entry against the dense tensor the lists were built from."""
    B, L, F = 3000, 2, 96
    g = torch.Generator().manual_seed(3)
    dens = torch.tensor([0.0, 0.002, 0.007, 0.02, 0.1, 0.3, 0.34, 0.5, 0.9, 1.0]).repeat_interleave(10)[:F]      # column lengths 0 .. 3000
    x = (torch.rand(B, L, F, generator=g) < dens).float() * (torch.rand(B, L, F, generator=g) + 0.1)
    x[:, 1, :48] = 0.0
    x[7] = torch.rand(L, F, generator=g) + 0.1                                 # a full row (96 entries per layer)
    xd = x.to(DEV)
    S = ops.sparse_build(xd, int((x > 0).sum()))
    n = int(S.total)
    assert n == int((x > 0).sum()) and int(S.overflow) == 0
    cs, tok, val = S.col_start.cpu().long(), S.csc_tok.cpu().long()[:n], S.csc_val.cpu()[:n]
    lens = cs[1:] - cs[:-1]
    assert int(lens.max()) == B and int(((lens > 24) & (lens <= 1024)).sum()) > 10 and int((lens > 1024).sum()) > 10
    col = torch.repeat_interleave(torch.arange(L * F), lens)
    same_col = col[1:] == col[:-1]
    assert bool((tok[1:][same_col] > tok[:-1][same_col]).all()), "tokens of a column must ascend"
    assert torch.equal(val, x.reshape(B, L * F)[tok, col])
    ent = S.csc_ent.cpu().long()[:n]
    assert torch.equal(S.ent_val.cpu()[:n][ent], val) and torch.equal(S.ent_col.cpu().long()[:n][ent], col)
    rs, ecol = S.row_start.cpu().long(), S.ent_col.cpu().long()[:n]
    rlen = rs[1:] - rs[:-1]
    row = torch.repeat_interleave(torch.arange(B * L), rlen)
    same_row = row[1:] == row[:-1]
    assert bool((ecol[1:][same_row] > ecol[:-1][same_row]).all()), "columns of a row must ascend"
    assert torch.equal(S.ent_val.cpu()[:n], x.reshape(B, L * F)[row // L, ecol])


@pytest.mark.parametrize("x16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("B,L,F", [(37, 3, 264), (8, 2, 4096), (5, 1, 1028), (2, 4, 32768), (9, 2, 30)])
def test_threshold_lists_equal_lists_of_the_thresholded_tensor(ops, B, L, F, x16):
    """clt_sparse_build_threshold_* (JumpReLU lists straight from the pre-activations; with aligned operands four rows of a layer
    per block, the layer's thresholds read once) == clt_sparse_build_f32 of x * [(x > theta) & (x > 0)] (jumprelu.py:36-37), bit
    for bit, in every list array.  Token counts that are no multiple of four, rows shorter than a wave's segment, F % 4 != 0
    (scalar form), 16-bit and fp32 pre-activations."""
    g = torch.Generator().manual_seed(B * 7 + F)
    x = torch.randn(B, L, F, generator=g) * 0.5
    theta = torch.rand(1, L, F, generator=g) * 0.8
    x = x.to(DEV)
    if x16:
        if F % 4:
            pytest.skip("16-bit pre-activations need F % 4 == 0")
        x = x.bfloat16()
    theta = theta.to(DEV)
    xf = x.float()
    dense = xf * ((xf > theta) & (xf > 0))
    cap = B * L * F
    _lists_equal(ops.sparse_build_threshold(x, theta, cap), ops.sparse_build(dense, cap), "threshold lists")
    relu = xf * (xf > 0)
    _lists_equal(ops.sparse_build_threshold(x, None, cap), ops.sparse_build(relu, cap), "relu lists")
    # the build in phases (gate outcome read early, feature-major lists on the second stream): the same lists after join()
    S = ops.sparse_build_threshold(x, theta, cap, split=True)
    assert not S.overflowed()
    S.join()
    _lists_equal(S, ops.sparse_build(dense, cap), "threshold lists, phased build")
    S = ops.sparse_build_threshold(x, theta, max(1, int((dense > 0).sum()) // 2), split=True)      # capacity exceeded: gated off
    assert S.overflowed() and int(S.total) == int((dense > 0).sum())
    S.join()


@pytest.mark.parametrize("B,L,F,k", [(3, 4, 5, 3), (16, 2, 32, 4), (64, 3, 4096, 16), (9, 2, 70000, 33), (300, 12, 1024, 8),
                                     (2, 2, 64, 64), (512, 4, 8192, 16), (16, 2, 20000, 16), (8, 2, 12000, 100)])
def test_select_lists_matches_dense_selection(ops, B, L, F, k):
    """Selection straight to lists == dense (oracle-checked) selection followed by the list build, bit for bit,
    in every mode, with and without the dead-feature mask; EMA thresholds agree."""
    x = _rand(B, L, F, seed=3 * B + F).to(DEV)
    g = torch.Generator().manual_seed(1)
    mask = (torch.rand(L, F, generator=g) < 0.3).to(torch.uint8).to(DEV)
    cap = k * B * L
    for cm in (None, mask):
        xm = x if cm is None else ops.mask_dead(x, cm)
        want = ops.sparse_build(ops.topk_rows(xm, k), cap)
        got = ops.select_lists(x, k, ops.SELECT_ROWS, col_mask=cm)
        _lists_equal(got, want, f"rows mask={cm is not None}")
        for mode, per_layer, S in ((ops.SELECT_BATCH, False, 1), (ops.SELECT_LAYER, True, L)):
            t1, t2 = torch.full((S,), 0.25, device=DEV), torch.full((S,), 0.25, device=DEV)
            want = ops.sparse_build(ops.topk_segmented(xm, k, per_layer, t1, 0.1), cap)
            got = ops.select_lists(x, k, mode, t2, 0.1, cm)
            _lists_equal(got, want, f"mode {mode} mask={cm is not None}")
            assert torch.equal(t1, t2), "threshold EMA"
    # the oracle, directly, for the unmasked batch mode
    ref, _, _ = O.batch_topk_fwd(x.cpu(), k)
    got = ops.select_lists(x, k, ops.SELECT_BATCH, torch.zeros(1, device=DEV), 0.1)
    assert torch.equal(ops.sparse_scatter_dense(got, got.csc_val).cpu(), ref)


def test_select_lists_ties_fallback_and_edges(ops):
    """Heavy ties: a few or ten thousand entries tied at the k-th value are ranked by flat index with a radix select on the
    index over the candidate list (one GPU: always; no exhaustive fallback, whatever the number of ties).  Must equal the
    oracle's rule (larger value, then lower index)."""
    g = torch.Generator().manual_seed(0)
    t = torch.randint(0, 6, (40, 3, 512), generator=g).float() - 1.0          # ~10k ties at the threshold value
    few = _rand(40, 3, 512, seed=4)
    few[3, 1, 7] = few[5, 2, 9] = few[1, 0, 100] = few[30, 2, 500] = 0.75      # 4 equal values straddling a cut
    for x, expect_fb in ((t, True), (few, False)):
        xd = x.to(DEV)
        for k in (1, 7, 100):
            got = ops.select_lists(xd, k, ops.SELECT_BATCH, None, 0.0)
            assert torch.equal(ops.sparse_scatter_dense(got, got.csc_val).cpu(), O.batch_topk_fwd(x, k)[0])
            assert int(got.used_fallback) == 0, "ties must not send the selection to the exhaustive path"
            got = ops.select_lists(xd, k, ops.SELECT_LAYER, None, 0.0)
            assert torch.equal(ops.sparse_scatter_dense(got, got.csc_val).cpu(), O.per_layer_batch_topk_fwd(x, k)[0])
            got = ops.select_lists(xd, k, ops.SELECT_ROWS)
            assert torch.equal(ops.sparse_scatter_dense(got, got.csc_val).cpu(), O.per_layer_topk_fwd(x, k)[0])
    # exact cut through the 4 equal values: choose k so that the k-th value is 0.75
    flat = few.reshape(-1)
    rank = int((flat > 0.75).sum())
    for kk in (rank + 1, rank + 2, rank + 3):
        if kk % (40 * 3) == 0:
            k = kk // 120
            got = ops.select_lists(few.to(DEV), k, ops.SELECT_BATCH, None, 0.0)
            assert torch.equal(ops.sparse_scatter_dense(got, got.csc_val).cpu(), O.batch_topk_fwd(few, k)[0])
    x = (torch.rand(3, 4, 5) + 0.1).to(DEV)                                      # k = 0, k = F, k > F
    for mode in (ops.SELECT_ROWS, ops.SELECT_BATCH, ops.SELECT_LAYER):
        z = ops.select_lists(x, 0, mode)
        assert int(z.total) == 0 and torch.equal(ops.sparse_scatter_dense(z, z.csc_val), torch.zeros_like(x))
        full = ops.select_lists(x, 5, mode)
        assert torch.equal(ops.sparse_scatter_dense(full, full.csc_val), x)
        with pytest.raises(AssertionError):
            ops.select_lists(x, 6, mode)


@pytest.mark.parametrize("Bn,L,d,F,k,fp16", [(300, 2, 128, 512, 8, False), (256, 1, 64, 1024, 16, False), (1000, 3, 192, 2048, 16, False),
                                             (513, 2, 768, 4096, 32, False), (260, 2, 72, 1024, 4, True), (64, 12, 768, 2048, 64, False)])
def test_encoder_fused_with_per_row_selection_autocast_order(ops, Bn, L, d, F, k, fp16):
    """PerLayerTopK lists from the encoder GEMM's row-select epilogue (row bounds from a 1/16 feature sample, candidates filed
    per row, exact selection on those) == the lists of the unfused path (encoder GEMM -> clt_select_lists_f32 rows mode),
    array by array: including rows with nothing positive, rows of equal values (ties by lowest index), a sample offset that
    rotates, edge tiles; and an attempt whose bounds cannot hold must say so and emit nothing."""
    from crosslayer_transcoder_amd import precision
    with precision("16-mixed" if fp16 else "bf16-mixed"):
        ops.sync_operand_format()
        dt = ops.op16()
        xs = _rand(Bn, L, d, seed=1)
        Wt = _rand(L, F, d, seed=2, scale=d ** -0.5)
        b = _rand(L, F, seed=3, scale=0.1)
        xs[5] = 0.0                                   # pre = bias: ordinary row
        Wt[0, 64:128] = 0.0; b[0, 64:128] = 0.7       # 64 equal values in every layer-0 row: ties at or above the k-th value
        xs16, Wt16 = xs.to(DEV).to(dt), Wt.to(DEV).to(dt)
        for trial, (bias, offset) in enumerate(((b, 0), (b, 5), (b - 3.0, 11), (None, 3))):      # b - 3: most rows have < k positives
            bd = bias.to(DEV) if bias is not None else None
            assert ops.encoder_select_rows_applicable(Bn, L, d, F, k)
            want = ops.select_lists(ops.encoder_fwd(xs16, Wt16, bd), k, ops.SELECT_ROWS)
            got = ops.encoder_select_rows_lists(xs16, Wt16, bd, k, offset=offset)
            assert got.confirm(), f"trial {trial}: the attempt should hold"
            _lists_equal(got, want, f"fused rows trial {trial}")
        odds = ops.ROWS_FAIL_ODDS
        try:                                          # bound = the sample's maximum: above the k-th value in about a third of the rows
            ops.ROWS_FAIL_ODDS = 0.999
            got = ops.encoder_select_rows_lists(xs16, Wt16, b.to(DEV), k, offset=0)
            if k > 1:
                assert not got.confirm() and int(got.total) == 0 and int(got.row_start[-1]) == 0
        finally:
            ops.ROWS_FAIL_ODDS = odds
    ops.sync_operand_format()


@pytest.mark.parametrize("Bn,L,d,F", [(300, 2, 128, 520), (256, 1, 64, 256), (1000, 3, 192, 1028), (513, 2, 768, 260),
                                      (300, 2, 72, 520), (260, 1, 200, 300), (256, 2, 104, 256), (270, 1, 8, 256)])
def test_gemm_nt_fast_path(ops, Bn, L, d, F):
    """The LDS-DMA / 16x16x32 NT kernel (K % 8 == 0 — a k tail shorter than the 64-wide k-tile is zero-filled by
    out-of-range DMA offsets —, M*N >= 256^2): encoder forward with bias, edge tiles in both directions, batched; plus
    accumulate and the A = I asymmetric-B layout check."""
    bf = torch.bfloat16
    xs, Wt = _rand(Bn, L, d, seed=1).to(bf), _rand(L, F, d, seed=2, scale=d ** -0.5).to(bf)
    b = _rand(L, F, seed=3, scale=0.1)
    ref = torch.stack([xs[:, l].double() @ Wt[l].double().T + b[l].double() for l in range(L)], 1)
    got = ops.encoder_fwd(xs.to(DEV), Wt.to(DEV), b.to(DEV))
    assert_close(got, ref, **tol(ref, 2e-6), what="NT fast path")
    # dgrad-shaped call (accumulate into an existing C): dx += dpre @ W^T uses transB with K = F
    if F % 64 == 0:
        dpre = _rand(Bn, L, F, seed=4).to(bf)
        Wd = _rand(L, d, F, seed=5, scale=0.05).to(bf)                 # [N = d][K = F] rows
        out = torch.zeros(Bn, L, d, device=DEV)
        for rep in range(2):
            ops._gemm("clt_gemm_bf16", dpre.to(DEV), 0, Wd.to(DEV), 0, out, 0, Bn, d, F, L * F, F, L * d, False, True, batch=L,
                      sA=F, sB=d * F, sC=d, accumulate=rep == 1)
        want = 2 * torch.stack([dpre[:, l].double() @ Wd[l].double().T for l in range(L)], 1)
        assert_close(out, want, **tol(want, 2e-6), what="NT accumulate")
    n = 256
    Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(bf)   # exactly representable, asymmetric
    eye = torch.eye(n).to(bf)
    out = ops.encoder_fwd(eye.reshape(n, 1, n).to(DEV), Bm.reshape(1, n, n).to(DEV), None)
    assert torch.equal(out.cpu().reshape(n, n), Bm.float().T)


@pytest.mark.parametrize("M,N,K,batch", [(512, 256, 512, 1), (264, 520, 200, 2), (768, 256, 4096, 1), (256, 264, 72, 3),
                                          (1032, 768, 130 * 8, 1)])
def test_gemm_k_major_operands(ops, M, N, K, batch):
    """The LDS-DMA kernel with k-major operands (A stored [K][M] and / or B stored [K][N]: transposed LDS reads instead of
    transposed copies) — every combination against the fp64 product of the same bf16 operands: edge tiles in M and N, a k tail,
    batches, bias, accumulate, and the A = I layout check."""
    bf = torch.bfloat16
    A = _rand(batch, M, K, seed=1, scale=K ** -0.5).to(bf)
    B = _rand(batch, K, N, seed=2).to(bf)
    bias = _rand(batch, N, seed=3, scale=0.1)
    want = torch.bmm(A.double(), B.double()) + bias.double()[:, None]
    for tA in (False, True):
        for tB in (False, True):
            a = (A.transpose(1, 2) if tA else A).contiguous().to(DEV)           # tA: stored [K][M]
            b = (B.transpose(1, 2) if tB else B).contiguous().to(DEV)           # tB: stored [N][K]
            out = torch.full((batch, M, N), float("nan"), device=DEV)
            for rep in range(2):
                ops._gemm("clt_gemm_bf16", a, 0, b, 0, out, 0, M, N, K, M if tA else K, K if tB else N, N, tA, tB, batch=batch,
                          sA=M * K, sB=K * N, sC=M * N, bias=bias.to(DEV), sBias=N, accumulate=rep == 1)
            assert_close(out, 2 * want, **tol(want, 4e-6), what=f"transA={tA} transB={tB}")
    n = 256
    Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(bf)   # exactly representable, asymmetric
    eye = torch.eye(n).to(bf).to(DEV)
    for tA, tB, want in ((False, False, Bm.float()), (True, False, Bm.float()), (True, True, Bm.float().T)):
        out = torch.empty(n, n, device=DEV)
        ops._gemm("clt_gemm_bf16", eye, 0, Bm.to(DEV), 0, out, 0, n, n, n, n, n, n, tA, tB)
        assert torch.equal(out.cpu(), want), (tA, tB)
        out = torch.empty(n, n, device=DEV)                                    # ... and with the asymmetric matrix as A
        ops._gemm("clt_gemm_bf16", Bm.to(DEV), 0, eye, 0, out, 0, n, n, n, n, n, n, tA, tB)
        assert torch.equal(out.cpu(), Bm.float().T if tA else Bm.float()), (tA, tB, "A")


@pytest.mark.parametrize("prec", ["bf16-mixed", "16-mixed"])
@pytest.mark.parametrize("Bn,L,F,d", [(512, 3, 512, 128), (264, 2, 1032, 72)])
def test_plt_weight_gradient_from_features_as_they_lie(ops, Bn, L, F, d, prec):
    """ops.wgrad_ta (the 16-bit features [B, L, F] as the k-major A operand of the per-layer decoder's weight-gradient GEMM)
    gives the bits of ops.wgrad_t on the transposed copy, also accumulating; both against fp64; bf16 and fp16 operands."""
    from crosslayer_transcoder_amd import precision
    gen = torch.Generator().manual_seed(5)
    try:
        with precision(prec):
            ops.sync_operand_format()
            feat = ops.cast_bf16(torch.relu(torch.randn(Bn, L, F, generator=gen)).cuda())
            assert feat.dtype == (torch.float16 if prec == "16-mixed" else torch.bfloat16)
            g = torch.randn(Bn, L, d, generator=gen).cuda()
            gT = ops.transpose_cast_2d(g).view(L, d, Bn)
            featT = feat.permute(1, 2, 0).contiguous()
            assert ops.wgrad_ta_ok(Bn, F, d)
            want = torch.einsum("blf,bld->lfd", feat.double().cpu(), gT.permute(2, 0, 1).double().cpu())
            a = ops.wgrad_ta(feat, gT, torch.empty(L, F, d, device=DEV))
            b = ops.wgrad_t(featT, gT, torch.empty(L, F, d, device=DEV), L, F * Bn, d * Bn)
            assert torch.equal(a, b)
            assert torch.allclose(a.double().cpu(), want, rtol=1e-4, atol=1e-3)
            assert torch.equal(ops.wgrad_ta(feat, gT, a.clone(), True), ops.wgrad_t(featT, gT, b.clone(), L, F * Bn, d * Bn, True))
    finally:
        ops.sync_operand_format()


@pytest.mark.parametrize("rows,L,F", [(64, 1, 256), (72, 2, 150), (200, 3, 64), (8, 1, 4), (136, 2, 1100)])
def test_fused_dense_jumprelu_tile_passes(ops, rows, L, F):
    """clt_jumprelu_fwd_bf16 / clt_jumprelu_loss_bwd_bf16 against the separate fp32 passes + casts (ragged tiles:
    rows not a multiple of 64, columns not a multiple of 256)."""
    g = torch.Generator().manual_seed(rows + F)
    x = torch.randn(rows, L, F, generator=g).cuda()
    theta = (torch.rand(1, L, F, generator=g) * 0.5).cuda()
    feat = ops.jumprelu_fwd(x, theta)
    fb, fT, fired = ops.jumprelu_fwd_bf16(x, theta)
    assert torch.equal(fb, ops.cast_bf16(feat))
    assert torch.equal(fT, ops.transpose_cast_2d(feat).view(L, F, rows))
    assert torch.equal(fired, ops.column_any_positive(feat))
    # no transposed copy: the same row-major copy and flags, fp32 and 16-bit x, with a bias
    fb1, fT1, fired1 = ops.jumprelu_fwd_bf16(x, theta, want_T=False)
    assert fT1 is None and torch.equal(fb1, fb) and torch.equal(fired1, fired)
    bias = torch.randn(L, F, generator=g).cuda()
    x16 = ops.cast_bf16(x)
    for xin in (x, x16):
        a, b = ops.jumprelu_fwd_bf16(xin, theta, bias), ops.jumprelu_fwd_bf16(xin, theta, bias, want_T=False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    fb0, fT0, fired0 = ops.jumprelu_fwd_bf16(x, None)
    relu = ops.relu_fwd(x)
    assert torch.equal(fb0, ops.cast_bf16(relu)) and torch.equal(fT0, ops.transpose_cast_2d(relu).view(L, F, rows))
    assert torch.equal(fired0, ops.column_any_positive(relu))
    sq = (torch.rand(L * F, generator=g) + 0.1).cuda()
    df = torch.randn(rows, L, F, generator=g).cuda()
    for tanh in (True, False):
        args = (x, theta, sq, df, 0.3, 0.01, 1.5, tanh, 1e-3)
        s1, dpre, dth1, dn1 = ops.jumprelu_loss_bwd(*args)
        s2, dT, db, dth2, dn2 = ops.jumprelu_loss_bwd_bf16(*args)
        want = ops.transpose_cast_2d(dpre).view(L, F, rows)
        if tanh:        # (hardware exp2 / rcp tanh: about 1e-6 absolute, which may move an entry across a bf16 rounding boundary)
            assert (dT != want).float().mean() < 1e-3
            assert torch.allclose(dT.float(), want.float(), rtol=2 ** -7, atol=1e-6)
        else:
            assert torch.equal(dT, want)
        for a, b, nm in ((s2, s1, "sums"), (dth2, dth1, "dtheta"), (dn2, dn1, "dnorm"), (db, ops.column_sum(dpre), "dbias")):
            a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max())), nm


def test_clt_dgrad_k_segmented_equals_accumulating_chain(ops, monkeypatch):
    """clt_gemm_bf16_nt_kseg (feature gradient of the cross-layer decoder as one GEMM per source layer, K-segmented
    over the per-target-layer weight tensors) vs the chain of accumulating GEMMs and vs an fp64 reference."""
    B, L, d, F = 512, 3, 128, 512
    gen = torch.Generator().manual_seed(21)
    g = torch.randn(B, L, d, generator=gen).cuda()
    Ws = [(torch.rand(l + 1, F, d, generator=gen) * 2 - 1).cuda() for l in range(L)]
    go, Wo = ops.cast_bf16(g), [ops.cast_bf16(w) for w in Ws]
    df = ops.clt_decoder_dgrad_kseg(go, Wo)
    assert df is not None
    ref = ops.clt_decoder_dgrad(go, Wo)
    assert torch.allclose(df, ref, rtol=1e-5, atol=1e-4)
    gd, want = go.double().cpu(), torch.zeros(B, L, F, dtype=torch.float64)
    for l in range(L):
        for lp in range(l + 1):
            want[:, lp] += gd[:, l] @ Wo[l][lp].double().cpu().T
    assert torch.allclose(df.double().cpu(), want, rtol=1e-4, atol=1e-3)
    # this shape (2 tiles per launch) alternates its launches over two streams: same bits on one stream, fp32 and 16-bit results
    for out16 in (False, True):
        monkeypatch.setattr(ops, "KSEG_DGRAD_STREAMS", 2)
        a2 = ops.clt_decoder_dgrad_kseg(go, Wo, out16)
        monkeypatch.setattr(ops, "KSEG_DGRAD_STREAMS", 1)
        a1 = ops.clt_decoder_dgrad_kseg(go, Wo, out16)
        assert torch.equal(a1, a2) and (out16 or torch.equal(a1, df))


def test_clt_decoder_fwd_grouped_launches(ops, monkeypatch):
    """Dense cross-layer decoder forward through the grouped NT GEMM: one launch for all (target, source) pairs, several
    launches under a small slab budget, both against an fp64 reference."""
    B, L, d, F = 512, 4, 128, 256
    gen = torch.Generator().manual_seed(33)
    feat = torch.relu(torch.randn(B, L, F, generator=gen)).cuda()
    Ws = [(torch.rand(l + 1, F, d, generator=gen) * 2 - 1).cuda() for l in range(L)]
    bias = torch.randn(L, d, generator=gen).cuda()
    fo = ops.cast_bf16(feat)
    WTs = [ops.cast_transpose_bf16(w.contiguous()) for w in Ws]
    one = ops.clt_decoder_fwd_t(fo, WTs, bias)
    monkeypatch.setattr(ops, "GROUPED_SLAB_BYTES", 3 * 4 * B * d)      # at most 3 pairs per launch
    many = ops.clt_decoder_fwd_t(fo, WTs, bias)
    assert torch.equal(one, many)
    want = torch.zeros(B, L, d, dtype=torch.float64)
    fd = fo.double().cpu()
    for l in range(L):
        for lp in range(l + 1):
            want[:, l] += fd[:, lp] @ WTs[l][lp].double().cpu().T
        want[:, l] += bias[l].double().cpu()
    assert torch.allclose(one.double().cpu(), want, rtol=1e-4, atol=1e-3)
    # the same from the weights as they lie ([F][d]: k-major B operand of the grouped launch): same bits
    Wops = [ops.cast_bf16(w.contiguous()) for w in Ws]
    assert torch.equal(ops.clt_decoder_fwd_t(fo, Wops, bias, k_major=True), many)
    monkeypatch.setattr(ops, "GROUPED_SLAB_BYTES", 4 << 30)
    assert torch.equal(ops.clt_decoder_fwd_t(fo, Wops, bias, k_major=True), one)


def test_plt_decoder_fwd_tail_split(ops, monkeypatch):
    """Per-layer decoder forward (K = features): the short last round of the persistent batched GEMM as a grouped launch over
    pieces of the contraction + slab sums — against the one-launch form and an fp64 reference; the split it picks at the
    benchmark's shapes."""
    assert ops._tail_split(12, 48, 32768, 256) == (2, 8)            # B 4096, d 768: 576 tiles = 2.25 rounds -> 2 + 3/8
    assert ops._tail_split(12, 24, 32768, 256) == (2, 16)           # B 2048: 1.125 rounds -> 1 + 3/16
    assert ops._tail_split(12, 96, 32768, 256) == (0, 1)            # B 8192: 4.5 rounds of 5, estimate within 10 %
    assert ops._tail_split(12, 128, 32768, 256) == (0, 1)           # whole rounds: nothing to gain
    assert ops._tail_split(12, 48, 768, 256) == (0, 1)              # short contraction: no pieces
    B, L, d, F = 512, 3, 256, 8192                                  # 2 tiles per layer
    gen = torch.Generator().manual_seed(34)
    feat = torch.relu(torch.randn(B, L, F, generator=gen)).cuda()
    W = (torch.rand(L, F, d, generator=gen) * 2 - 1).cuda()
    bias = torch.randn(L, d, generator=gen).cuda()
    fo, WT = ops.cast_bf16(feat), ops.cast_transpose_bf16(W.contiguous())
    picks = []
    real = ops._tail_split
    monkeypatch.setattr(ops, "_tail_split", lambda *a: picks.append(real(*a)) or picks[-1])
    split = ops.plt_decoder_fwd_t(fo, WT, bias)
    assert picks[-1][0] > 0 and picks[-1][1] > 1                    # (6 tiles on 256 CUs: every layer is cut)
    monkeypatch.setattr(ops, "_tail_split", lambda *a: (1, 2))      # a mixed launch: two layers batched, one cut in two
    mixed = ops.plt_decoder_fwd_t(fo, WT, bias)
    monkeypatch.setattr(ops, "GEMM_TAIL_SPLIT", False)
    one = ops.plt_decoder_fwd_t(fo, WT, bias)
    want = torch.einsum("blf,ldf->bld", fo.double().cpu(), WT.double().cpu()) + bias.double().cpu()
    for got in (split, mixed, one):
        assert torch.allclose(got.double().cpu(), want, rtol=1e-4, atol=2e-3)
    assert torch.equal(mixed[:, :2], one[:, :2])
    # without a bias
    monkeypatch.setattr(ops, "GEMM_TAIL_SPLIT", True)
    assert torch.allclose(ops.plt_decoder_fwd_t(fo, WT, None).double().cpu(), want - bias.double().cpu(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("nb,rows,n", [(3, 200, 768), (1, 64, 256), (2, 1000, 108)])
def test_weight_prep_one_pass(ops, nb, rows, n):
    """clt_weight_prep_bf16 == cast_bf16 + cast_transpose_bf16 (bit-exact) + row_sqnorm (accumulating) of the same tensor."""
    w = _rand(nb, rows, n, seed=31).to(DEV)
    sq0 = torch.rand(nb * rows + 5, generator=torch.Generator().manual_seed(3)).to(DEV)
    sq = sq0.clone()
    op, opT = ops.weight_prep_bf16(w, sq, True)
    assert torch.equal(op, ops.cast_bf16(w)) and torch.equal(opT, ops.cast_transpose_bf16(w))
    ref = sq0.clone()
    ref[: nb * rows] += (w.double() ** 2).sum(-1).reshape(-1).float()
    assert torch.allclose(sq, ref, rtol=1e-6, atol=0)
    assert torch.equal(sq[nb * rows:], sq0[nb * rows:])
    sq2 = torch.full_like(sq0, 7.0)
    ops.weight_prep_bf16(w, sq2, False)
    assert torch.allclose(sq2[: nb * rows], (w.double() ** 2).sum(-1).reshape(-1).float(), rtol=1e-6, atol=0)
    assert ops.weight_prep_bf16(_rand(1, 12, 768, seed=1).to(DEV)) is None          # rows % 8 != 0: caller falls back


@pytest.mark.parametrize("B,L,d", [(130, 3, 72), (64, 2, 128), (257, 1, 260)])
def test_mse_bf16_colsum_one_pass(ops, B, L, d):
    """clt_mse_bf16_colsum_f32 == clt_mse_f32 + cast_bf16 + column_sum: bf16 gradient rows bit-exact, sums to fp32 rounding."""
    batch = _rand(B, 2, L, d, seed=41).to(DEV)
    r = _rand(B, L, d, seed=42).to(DEV)
    mean, std = _rand(L, d, seed=43).to(DEV), (_rand(L, d, seed=44).abs() + 0.5).to(DEV)
    y = batch[:, 1]                                            # strided rows, as the step hands them over
    loss, g, _ = ops.mse(r, y, mean, std, True, False)
    loss2, g16, colsum = ops.mse_bf16_colsum(r, y, mean, std)
    assert torch.equal(g16, ops.cast_bf16(g))
    assert torch.allclose(loss2, loss, rtol=1e-12, atol=0)
    ref = g.double().sum(0).float()
    assert torch.allclose(colsum, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("slab,bf16,hot", [(False, False, False), (True, True, False), (False, True, False), (False, True, True),
                                          (True, False, True)])
def test_sparse_row_pass_group_equals_per_matrix_launches(ops, slab, bf16, hot):
    """clt_sparse_row_pass_group_f32 (all matrices of a cross-layer decoder in one launch) against one
    clt_sparse_row_pass_f32 launch per matrix: same arithmetic per row, so weights, moments, shadows and dots are bit-identical.
    hot: three features of every layer fire in every token (columns of 300 entries: walked in segments by the hot-row
    launches of both forms, the group form with a scratch region per matrix); both forms also agree with the one-wave-per-row
    form to summation-order rounding."""
    B, L, F, d = (300, 4, 96, 40) if hot else (37, 4, 96, 40)
    feat = _sparse_features(B, L, F, 6, seed=3)
    if hot:
        feat[:, :, 10:13] = torch.rand(B, L, 3, generator=torch.Generator().manual_seed(2)) + 0.5
    S = ops.sparse_build(feat.to(DEV), int((feat > 0).sum()))
    if hot:
        assert ops.hot_columns(S) is not None
        lens = (S.col_start[1:] - S.col_start[:-1])
        assert int(lens.max()) == B and int((lens > ops.HOT_MIN_LEN).sum()) == 3 * L
    g = _rand(B, L, d, seed=5).to(DEV)
    src = ops.cast_bf16(g) if bf16 else g
    g_lm = src.permute(1, 0, 2).contiguous()                     # layer-major slabs [L, B, d]
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, adamw=False, step=3)
    res = []
    for grouped in (False, True):
        Ws = [_rand(l + 1, F, d, seed=20 + l, scale=0.05).to(DEV) for l in range(L)]
        ms = [_rand(l + 1, F, d, seed=30 + l, scale=0.01).to(DEV) for l in range(L)]
        vs = [_rand(l + 1, F, d, seed=40 + l, scale=0.01).to(DEV).abs() for l in range(L)]
        shs = [torch.zeros(l + 1, F, d, dtype=torch.bfloat16, device=DEV) for l in range(L)]
        dots = torch.zeros(L, S.cap, device=DEV)
        ls = list(range(L - 1, -1, -1))
        if grouped:
            srcs = [g_lm[l] for l in ls] if slab else [src] * L
            assert ops.sparse_row_pass_group([Ws[l] for l in ls], S, S.csc_val, srcs, ls, adam=hp, ms=[ms[l] for l in ls],
                                             vs=[vs[l] for l in ls], shadows=[shs[l] for l in ls], dots=dots, src_slab=slab)
        else:
            for l in ls:
                ops.sparse_row_pass(Ws[l], S, S.csc_val, g_lm[l] if slab else src, 0, l, adam=hp, m=ms[l], v=vs[l], shadow=shs[l],
                                    dots=dots, dot_slot=-1, src_slab=slab)
        res.append((Ws, ms, vs, shs, dots))
    for a, b in zip(res[0][:4], res[1][:4]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(res[0][4], res[1][4])
    assert not torch.equal(res[0][0][L - 1], _rand(L, F, d, seed=20 + L - 1, scale=0.05).to(DEV))     # (the pass did move the weights)
    if hot:                                              # the same passes with one wave per row, whatever its length
        hot0 = ops.HOT_MIN_LEN
        try:
            ops.HOT_MIN_LEN = 0
            S1 = ops.sparse_build(feat.to(DEV), int((feat > 0).sum()))
            Ws = [_rand(l + 1, F, d, seed=20 + l, scale=0.05).to(DEV) for l in range(L)]
            ms = [_rand(l + 1, F, d, seed=30 + l, scale=0.01).to(DEV) for l in range(L)]
            vs = [_rand(l + 1, F, d, seed=40 + l, scale=0.01).to(DEV).abs() for l in range(L)]
            dots1 = torch.zeros(L, S1.cap, device=DEV)
            for l in range(L - 1, -1, -1):
                ops.sparse_row_pass(Ws[l], S1, S1.csc_val, g_lm[l] if slab else src, 0, l, adam=hp, m=ms[l], v=vs[l], dots=dots1,
                                    dot_slot=-1, src_slab=slab)
        finally:
            ops.HOT_MIN_LEN = hot0
        assert torch.equal(dots1, res[0][4])             # (a dot is one entry's: no summation order involved)
        for l in range(L):
            assert_close(res[0][1][l], ms[l], 1e-4, 1e-6 * float(ms[l].abs().max()), what=f"exp_avg of W_{l}, hot rows vs one wave per row")


@pytest.mark.parametrize("d,bf16", [(40, True), (768, True), (70, False), (1024, False)])
def test_sparse_row_pass_leaves_updated_row_norms(ops, d, bf16):
    """sq_out of clt_sparse_row_pass_f32 = clt_row_sqnorm_f32 of the UPDATED matrix, bit for bit (one shared operation order),
    and asking for it leaves the update what it was, bit for bit (the kernel with the extra output is its own instantiation).  A matrix with d > 1024 (several waves per row) refuses it."""
    B, L, F = 37, 3, 96
    feat = _sparse_features(B, L, F, 6, seed=3)
    S = ops.sparse_build(feat.to(DEV), 6 * B * L)
    g = _rand(B, L, d, seed=5).to(DEV)
    src = ops.cast_bf16(g) if bf16 else g
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, adamw=False, step=3)
    dn, sq_in = _rand(L * F, seed=8).to(DEV), (_rand(L * F, seed=9).abs() + 0.1).to(DEV)
    res = []
    for keep in (False, True):
        W, m = _rand(L, F, d, seed=20, scale=0.05).to(DEV), _rand(L, F, d, seed=30, scale=0.01).to(DEV)
        v = _rand(L, F, d, seed=40, scale=0.01).to(DEV).abs()
        sq_out = torch.full((L * F,), -1.0, device=DEV) if keep else None
        ops.sparse_row_pass(W, S, S.csc_val, src, 0, L - 1, adam=hp, m=m, v=v, row_term=(dn, sq_in), sq_out=sq_out)
        res.append((W, m, v, sq_out))
    for x, y in zip(res[0][:3], res[1][:3]):       # (its own instantiation; the Adam arithmetic is explicit FMAs, so: same bits)
        assert torch.equal(x, y)
    ref = torch.zeros(L * F, device=DEV)
    ops.row_sqnorm(res[1][0], ref, 0, False)
    assert torch.equal(res[1][3], ref)
    assert torch.allclose(ref, res[1][0].double().pow(2).sum(-1).view(-1).float(), rtol=1e-5)
    Wb = torch.zeros(2, 1028, device=DEV)
    with pytest.raises(Exception):
        ops.sparse_row_pass(Wb, S, S.csc_val, torch.zeros(B, L, 1028, device=DEV), 0, 0, adam=hp, m=torch.zeros_like(Wb),
                            v=torch.zeros_like(Wb), sq_out=torch.zeros(2, device=DEV))


@pytest.mark.parametrize("rows,L,F", [(64, 2, 256), (72, 3, 520)])
def test_fused_dense_jumprelu_tile_passes_bf16_inputs(ops, rows, L, F):
    """The same tile passes fed with bf16 pre-activations / feature gradients (what clt_gemm_bf16_nt_out16 leaves) are the
    fp32-input passes on the widened values, bit for bit."""
    g = torch.Generator().manual_seed(rows * 3 + F)
    x16 = torch.randn(rows, L, F, generator=g).cuda().bfloat16()
    df16 = torch.randn(rows, L, F, generator=g).cuda().bfloat16()
    theta = (torch.rand(1, L, F, generator=g) * 0.5).cuda()
    sq = (torch.rand(L * F, generator=g) + 0.1).cuda()
    a, b = ops.jumprelu_fwd_bf16(x16, theta), ops.jumprelu_fwd_bf16(x16.float(), theta)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    for xin, din in ((x16, df16), (x16, df16.float()), (x16.float(), df16)):
        got = ops.jumprelu_loss_bwd_bf16(xin, theta, sq, din, 0.3, 0.01, 1.5, True, 1e-3)
        ref = ops.jumprelu_loss_bwd_bf16(x16.float(), theta, sq, df16.float(), 0.3, 0.01, 1.5, True, 1e-3)
        for u, v in zip(got, ref):              # (the fp64 loss sums: each instantiation's own fp32 contraction of the tile sums)
            assert torch.equal(u, v) if u.dtype != torch.float64 else torch.allclose(u, v, rtol=1e-6, atol=0)


@pytest.mark.parametrize("Bn,L,d,F", [(512, 2, 128, 512), (300, 3, 72, 264)])
def test_gemm_nt_bf16_output(ops, Bn, L, d, F):
    """clt_gemm_bf16_nt_out16 / _kseg_out16: the bf16 result is the rounding of the fp32 result of the same kernel (same main
    loop, same accumulation order), ragged tiles included."""
    g = torch.Generator().manual_seed(Bn + F)
    xs = ops.cast_bf16(torch.randn(Bn, L, d, generator=g).cuda())
    Wt = ops.cast_bf16((torch.randn(L, F, d, generator=g) * 0.1).cuda())
    b = (torch.randn(L, F, generator=g) * 0.1).cuda()
    pre32 = ops.encoder_fwd(xs, Wt, b)
    pre16 = ops.encoder_fwd_out16(xs, Wt, b)
    assert pre16 is not None and pre16.dtype == torch.bfloat16
    assert torch.equal(pre16, pre32.bfloat16())
    gg = ops.cast_bf16(torch.randn(Bn, L, d, generator=g).cuda())
    Wp = ops.cast_bf16((torch.randn(L, F, d, generator=g) * 0.1).cuda())
    assert torch.equal(ops.plt_decoder_dgrad_out16(gg, Wp), ops.plt_decoder_dgrad(gg, Wp).bfloat16())
    if d % 64 == 0:
        Ws = [ops.cast_bf16((torch.randn(l + 1, F, d, generator=g) * 0.1).cuda()) for l in range(L)]
        d32, d16 = ops.clt_decoder_dgrad_kseg(gg, Ws), ops.clt_decoder_dgrad_kseg(gg, Ws, True)
        assert d32 is not None and d16 is not None and torch.equal(d16, d32.bfloat16())


@pytest.mark.parametrize("Bn,L,d,F,f16", [(512, 2, 128, 512, False), (300, 3, 72, 264, False), (40, 2, 24, 100, False), (512, 2, 128, 512, True)])
def test_encoder_product_rounding_autocast_order(ops, Bn, L, d, F, f16):
    """CLT_GEMM_ROUND_ACC (fast NT kernel, generic kernel, fp16 / bf16): pre = float(round16(x W)) + b, bit for bit — the same
    kernel without the flag and without a bias gives the fp32 accumulators it rounds (model/clt.py:157-165 under autocast).
    The candidate-filter epilogue (clt_gemm_bf16_select) applies the same rounding: checked by the fused-vs-unfused step
    tests, whose unfused path selects from these values."""
    from crosslayer_transcoder_amd import precision
    g = torch.Generator().manual_seed(Bn + F)
    with precision("16-mixed" if f16 else "bf16-mixed"):
        ops.sync_operand_format()
        xs = ops.cast_bf16(torch.randn(Bn, L, d, generator=g).cuda())
        Wt = ops.cast_bf16((torch.randn(L, F, d, generator=g) * 0.1).cuda())
        b = (torch.randn(L, F, generator=g) * 0.1).cuda()
        assert ops.AUTOCAST_PRE
        pre = ops.encoder_fwd(xs, Wt, b)
        ops.AUTOCAST_PRE = False
        try:
            acc = ops.encoder_fwd(xs, Wt, None)
        finally:
            ops.AUTOCAST_PRE = True
        want = acc.to(ops.op16()).float() + b
        assert torch.equal(pre, want)
        assert not torch.equal(pre, acc + b)                 # (the rounding is not a no-op on this data)
        p16 = ops.encoder_fwd_out16(xs, Wt, None)
        if p16 is not None:                                  # the 16-bit tensor + bias in the consumers = the same values
            assert torch.equal(p16.float() + b, want)
    ops.sync_operand_format()


def test_jumprelu_tile_passes_and_list_build_with_split_bias_autocast_order(ops):
    """The consumers of the 16-bit encoder product add the fp32 bias themselves (clt_jumprelu_fwd_bf16,
    clt_jumprelu_loss_bwd_bf16, clt_sparse_build_threshold_x16 with `bias`): identical to running them on the fp32 tensor
    float(x16) + b."""
    g = torch.Generator().manual_seed(12)
    Bn, L, F = 72, 3, 520
    x16 = (torch.randn(Bn, L, F, generator=g) * 0.3).cuda().bfloat16()
    b = (torch.randn(L, F, generator=g) * 0.05).cuda()
    theta = (torch.rand(1, L, F, generator=g) * 0.2).cuda()
    sq = (torch.rand(L * F, generator=g) + 0.5).cuda()
    df = torch.randn(Bn, L, F, generator=g).cuda()
    x32 = x16.float() + b
    for a, w in zip(ops.jumprelu_fwd_bf16(x16, theta, b), ops.jumprelu_fwd_bf16(x32, theta)):
        assert torch.equal(a, w)
    got = ops.jumprelu_loss_bwd_bf16(x16, theta, sq, df, 0.3, 0.01, 1.5, True, 1e-3, bias=b)
    ref = ops.jumprelu_loss_bwd_bf16(x32, theta, sq, df, 0.3, 0.01, 1.5, True, 1e-3)
    for i, (a, w) in enumerate(zip(got, ref)):       # (sums: each instantiation contracts its fp32 tile sums its own way)
        assert torch.allclose(a, w, rtol=1e-6, atol=0) if i == 0 else torch.equal(a, w), i
    cap = Bn * L * F
    for th in (theta, None):
        _lists_equal(ops.sparse_build_threshold(x16, th, cap, bias=b), ops.sparse_build_threshold(x32, th, cap), "split-bias lists")


def test_sparse_decoder_fwd_layer_ranges(ops):
    """The gather forward restricted to target-layer ranges (what the feature-sharded step launches group by group) fills
    the same result as one launch over all layers, in both output layouts."""
    B, L, F, d = 19, 7, 64, 48
    feat = _sparse_features(B, L, F, 5, seed=9)
    S = ops.sparse_build(feat.to(DEV), 5 * B * L)
    Wd = [_rand(l + 1, F, d, seed=50 + l, scale=0.05).to(DEV) for l in range(L)]
    wptr = ops.pointer_table(Wd)
    bias = _rand(L, d, seed=8, scale=0.1).to(DEV)
    for lm in (False, True):
        full = ops.sparse_decoder_fwd(S, wptr, bias, d, True, layer_major=lm)
        out = None
        for lo, hi in ((6, 7), (4, 6), (1, 4), (0, 1)):
            out = ops.sparse_decoder_fwd(S, wptr, bias, d, True, layer_major=lm, out=out if out is not None else torch.full_like(full, float("nan")),
                                         layers=(lo, hi))
        assert torch.equal(out, full)


@pytest.mark.parametrize("Bn,L,d,F,share", [(1024, 3, 768, 8192, 0.14), (4096, 2, 256, 32768, 0.1), (512, 2, 128, 1028, 0.4)])
def test_encoder_fwd_columns_is_gather_of_full_gemm(ops, Bn, L, d, F, share):
    """The compacted TopK-aux pass takes the dead columns' pre-activations from the encoder GEMM on the gathered weight rows:
    bit for bit the columns of the full GEMM (same kernel, k order independent of the tile), -inf at the pad columns."""
    g = torch.Generator().manual_seed(Bn + F)
    xs = ops.cast_bf16(torch.randn(Bn, L, d, generator=g).cuda())
    Wt32 = (torch.randn(L, F, d, generator=g) * 0.05).cuda()
    b = (torch.randn(L, F, generator=g) * 0.1).cuda()
    dead = (torch.rand(L, F, generator=g) < share).to(torch.uint8).cuda()
    dead[1] = 0
    dead[1, ::7] = 1                                             # (layers with different counts: pad columns)
    n_pad, cidx, _ = ops.compact_columns(dead)
    assert Bn * n_pad >= 256 * 256
    pre = ops.encoder_fwd(xs, ops.cast_bf16(Wt32), b)
    ref = ops.gather_cols(pre, cidx, float("-inf"))
    got = ops.encoder_fwd_columns(xs, Wt32, b, cidx)
    assert torch.isinf(ref).any() and torch.equal(got, ref)


@pytest.mark.parametrize("Bn,L,n", [(256, 2, 512), (72, 3, 132), (1000, 1, 260)])
def test_mask_positive_transposed_one_pass(ops, Bn, L, n):
    """clt_mask_positive_t_f32 = mask_positive, then the transposing 16-bit cast, then column sums — in one pass: the operand
    bit for bit, the column sums to fp32 summation-order tolerance."""
    g = torch.Generator().manual_seed(Bn + n)
    feat = torch.relu(torch.randn(Bn, L, n, generator=g) - 0.8).cuda()
    dfeat = torch.randn(Bn, L, n, generator=g).cuda()
    dpre = ops.mask_positive(feat, dfeat)
    refT = ops.transpose_cast_2d(dpre).view(L, n, Bn)
    refc = ops.column_sum(dpre)
    gotT, gotc = ops.mask_positive_t(feat, dfeat)
    assert torch.equal(gotT.view(torch.int16), refT.view(torch.int16))
    assert_close(gotc.cpu(), dpre.cpu().double().sum(0).float(), **tol(refc.cpu()))
    assert_close(gotc.cpu(), refc.cpu(), **tol(refc.cpu()))


def test_weight_prep_gather_equals_prep_of_gathered_rows(ops):
    """clt_weight_prep_gather_bf16: both 16-bit operand forms of the indexed rows, bit for bit what weight_prep_bf16 makes of
    gather_rows' fp32 copy (zero rows at the pads)."""
    g = torch.Generator().manual_seed(11)
    w = torch.randn(3, 500, 132, generator=g).cuda()
    dead = (torch.rand(4, 500, generator=g) < 0.3).to(torch.uint8).cuda()
    dead[2, 100:] = 0
    n_pad, cidx, _ = ops.compact_columns(dead)
    ref_op, ref_T = ops.weight_prep_bf16(ops.gather_rows(w, cidx))
    op, opT = ops.weight_prep_gather_bf16(w, cidx)
    assert (cidx < 0).any()
    assert torch.equal(op.view(torch.int16), ref_op.view(torch.int16)) and torch.equal(opT.view(torch.int16), ref_T.view(torch.int16))


@pytest.mark.parametrize("F,k", [(1100, 16), (2048, 100), (2500, 512), (4100, 16), (5000, 512), (6144, 1000)])
def test_topk_rows_wave_two_phase_search(ops, F, k):
    """Short-row per-row TopK (one wave per row): the k-th key is searched over the registers for the top 12 bits and over the
    compacted prefix matches for the rest; rows with more matches than the list holds (quantised values: heavy ties) carry on
    over the registers.  Bit-exact with the oracle either way, -inf masked entries and list output (F <= 2048) included."""
    g = torch.Generator().manual_seed(F + k)
    x = torch.randn(37, 2, F, generator=g)
    x[3] = torch.randint(0, 4, (2, F), generator=g).float() - 1.0          # 4 levels: ~F/4 ties at the k-th value
    x[5] = 0.37                                                            # one value: every key shares the prefix
    x[7] = torch.where(torch.rand(2, F, generator=g) < 0.9, torch.full((2, F), -float("inf")), x[7])
    x[9] = (x[9] * 64).round() / 64                                        # many small tie groups
    ref, ridx = O.per_layer_topk_fwd(x, k)
    xd = x.to(DEV)
    assert torch.equal(ops.topk_rows(xd, k).cpu(), ref)
    if F <= 2048:
        f, idx, val = ops.topk_rows(xd, k, want_compact=True)
        assert torch.equal(f.cpu(), ref)
        dense = torch.zeros(37 * 2, F)
        ii, vv = idx.cpu().long(), val.cpu()
        for r in range(37 * 2):
            m = ii[r] >= 0
            dense[r, ii[r][m]] = vv[r][m]
        assert torch.equal(dense.view(37, 2, F), ref)


@pytest.mark.parametrize("F,k", [(4100, 16), (8192, 64), (20000, 16), (32768, 16), (32768, 31), (2052, 5)])
def test_select_rows_streaming_chunks(ops, F, k):
    """PerLayerTopK straight to lists on long rows: one wave per 2048-element chunk keeps the chunk's own top k, one wave per
    row merges them.  Bit-exact with the oracle's rule (larger value, then lower index) on random rows, rows of a few levels
    (every chunk overflows its one-per-lane candidate list: the extract-max path), a constant row, rows with fewer than k
    positive values, mostly -inf rows, ties cut across chunks, and under the dead-feature mask."""
    g = torch.Generator().manual_seed(F + k)
    x = torch.randn(21, 2, F, generator=g)
    x[2] = torch.randint(0, 4, (2, F), generator=g).float() - 1.0
    x[4] = 0.37
    x[6] = -x[6].abs()
    x[6, 0, 5] = 1.5; x[6, 0, F - 1] = 0.25; x[6, 1, 2049 % F] = 2.0
    x[8] = torch.where(torch.rand(2, F, generator=g) < 0.98, torch.full((2, F), -float("inf")), x[8])
    x[10] = (x[10] * 8).round() / 8
    x[12, :, :] = 0.0
    x[12, 0, ::700] = 3.0                                                   # equal values spread over the chunks
    mask = (torch.rand(2, F, generator=g) < 0.3).to(torch.uint8)
    xd = x.to(DEV)
    for cm in (None, mask):
        xm = x if cm is None else torch.where(cm.bool().expand_as(x), x, torch.full_like(x, -float("inf")))
        got = ops.select_lists(xd, k, ops.SELECT_ROWS, col_mask=None if cm is None else cm.to(DEV))
        assert torch.equal(ops.sparse_scatter_dense(got, got.csc_val).cpu(), O.per_layer_topk_fwd(xm, k)[0])
