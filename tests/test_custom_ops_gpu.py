"""The torch.library layer (crosslayer_transcoder_amd/custom_ops.py): every registered operator passes torch's own
operator checks (schema, fake-tensor rule, autograd registration, AOT dispatch) on the device, and the module forward
captured by torch.compile (graph of clt_hip ops) equals the eager forward and backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu

L, d, F, B = 3, 64, 256, 32


def _inputs():
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).cuda()          # noqa: E731
    xs = r(B, L, d)
    W = (r(L, F, d) * 0.1).permute(0, 2, 1).requires_grad_()     # [L, d, F], feature-major storage
    b = (r(L, F) * 0.1).requires_grad_()
    feat = torch.relu(r(B, L, F)).requires_grad_()
    Wd = (r(L, F, d) * 0.1).requires_grad_()
    bd = (r(L, d) * 0.1).requires_grad_()
    Ws = [(r(l + 1, F, d) * 0.1).requires_grad_() for l in range(L)]
    theta = torch.full((1, L, F), 0.05).cuda().requires_grad_()
    pre = r(B, L, F).requires_grad_()
    return dict(xs=xs, W=W, b=b, feat=feat, Wd=Wd, bd=bd, Ws=Ws, theta=theta, pre=pre, g_pre=r(B, L, F), g_r=r(B, L, d),
                mean=r(L, d), std=r(L, d).abs() + 0.5, thr=torch.full((L,), 0.1).cuda())


def test_every_registered_op_passes_opcheck():
    from crosslayer_transcoder_amd import custom_ops as C
    t = _inputs()
    cases = {
        "encoder": (t["xs"], t["W"], t["b"]),
        "encoder_bwd": (t["g_pre"], t["xs"], t["W"].detach(), False),
        "plt_decoder": (t["feat"], t["Wd"], t["bd"]),
        "plt_decoder_bwd": (t["g_r"], t["feat"].detach(), t["Wd"].detach()),
        "clt_decoder": (t["feat"], t["bd"], t["Ws"]),
        "clt_decoder_bwd": (t["g_r"], t["feat"].detach(), [w.detach() for w in t["Ws"]]),
        "relu": (t["pre"],),
        "relu_bwd": (t["pre"].detach(), t["g_pre"]),
        "jumprelu": (t["pre"], t["theta"], 0.1),
        "jumprelu_bwd": (t["pre"].detach(), t["theta"].detach(), 0.1, t["g_pre"]),
        "topk_rows": (t["pre"], 8),
        "batch_topk": (t["pre"], 8, True, t["thr"], 0.1),
        "mask_positive": (t["feat"].detach(), t["g_pre"]),
        "threshold": (t["pre"].detach(), t["thr"][:1].contiguous()),
        "standardize": (t["xs"], t["mean"], t["std"]),
        "destandardize": (t["g_r"], t["mean"], t["std"]),
    }
    assert set(cases) == set(C.REGISTERED)
    for name, args in cases.items():
        op = getattr(torch.ops.clt_hip, name).default
        # (test_autograd_registration / test_schema / test_faketensor / test_aot_dispatch_dynamic)
        torch.library.opcheck(op, args)


@pytest.mark.parametrize("kind", ["clt_jumprelu", "plt_relu", "clt_batchtopk"])
def test_compiled_module_forward_and_backward_equal_eager(kind):
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd.model.clt import CrosslayerDecoder, CrossLayerTranscoder, Decoder, Encoder
    from crosslayer_transcoder_amd.model.clt_lightning import CrossLayerTranscoderModule
    from crosslayer_transcoder_amd.model.jumprelu import JumpReLU, ReLU
    from crosslayer_transcoder_amd.model.standardize import DimensionwiseInputStandardizer, DimensionwiseOutputStandardizer
    from crosslayer_transcoder_amd.model.topk import PerLayerBatchTopK
    torch.manual_seed(5)
    nl = {"clt_jumprelu": lambda: JumpReLU(theta=0.02, bandwidth=0.1, n_layers=L, d_features=F), "plt_relu": ReLU,
          "clt_batchtopk": lambda: PerLayerBatchTopK(k=8, e=0.1, n_layers=L)}[kind]()
    dec = (Decoder if kind.startswith("plt") else CrosslayerDecoder)(d, F, L)
    model = CrossLayerTranscoder(nl, Encoder(d, F, L), dec, DimensionwiseInputStandardizer(L, d), DimensionwiseOutputStandardizer(L, d))
    mod = CrossLayerTranscoderModule(model, compile=True).cuda()
    batch = torch.randn(B, 2, L, d).cuda()
    model.initialize_standardizers(batch)
    acts = batch[:, 0].contiguous()

    def run(fn):
        for p in model.parameters():
            p.grad = None
        if hasattr(nl, "threshold"):
            nl.threshold.zero_()
        pre, feat, rn, rec = fn(acts)
        (rn.square().mean() + 1e-3 * feat.sum()).backward()
        return [pre, feat, rn, rec], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    outs_e, grads_e = run(model)
    outs_c, grads_c = run(mod)                       # CrossLayerTranscoderModule.forward -> torch.compile(model, "aot_eager")
    assert mod._compiled_model is not None and "_compiled_model" not in dict(mod.named_modules())
    assert not any(k.startswith("_compiled") or "_orig_mod" in k for k in mod.state_dict())
    for a, b in zip(outs_e, outs_c):
        assert torch.equal(a, b)
    assert grads_e.keys() == grads_c.keys() and len(grads_e) >= 3
    for k in grads_e:
        assert torch.equal(grads_e[k], grads_c[k]), k
    assert cta is not None
