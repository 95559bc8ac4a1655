"""Inference `layer=int` paths of the drop-in modules (used by the reference's replacement-model metric,
metrics/replacement_model_accuracy.py:28-56) against the CPU oracle: Encoder.forward_layer (clt.py:134-145),
Decoder.forward_layer (:216-231), CrosslayerDecoder.forward_layer (:304-317), and the sparsifiers'
_inference_forward (topk.py:49-51 per-row top-k; :71-74 / :104-107 threshold compare)."""
import pytest
import torch

from tests.golden_cases import assert_close, assert_close_rounded

pytestmark = pytest.mark.gpu


def _modules(L, d, F, g):
    from crosslayer_transcoder_amd.model import CrosslayerDecoder, Decoder, Encoder
    enc, plt, clt = Encoder(d, F, L), Decoder(d, F, L), CrosslayerDecoder(d, F, L)
    with torch.no_grad():
        for m in (enc, plt, clt):
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    return enc.cuda(), plt.cuda(), clt.cuda()


@pytest.mark.parametrize("shape", [(2, 5, 3, 12, 24), (1, 7, 2, 110, 32), (4, 64, 3, 128, 512)])
def test_forward_layer_paths_fp32(shape):
    from oracle import clt_oracle as O
    B, S, L, d, F = shape
    g = torch.Generator().manual_seed(3)
    enc, plt, clt = _modules(L, d, F, g)
    W, b = enc.W.detach().cpu(), enc.b.detach().cpu()
    Ws, bc = [w.detach().cpu() for w in clt.weights()], clt.b.detach().cpu()
    for layer in range(L):
        x = torch.randn(B, S, d, generator=g)
        want = O.encoder_fwd_layer(x.double(), W.double(), b.double(), layer)
        tol = 2e-6 * float(want.abs().max())
        assert_close(enc(x.cuda(), layer=layer), want, 1e-5, tol, f"encoder layer {layer}")
        assert_close(enc.forward_layer(x.cuda(), layer), want, 1e-5, tol, f"encoder.forward_layer {layer}")
        f3 = torch.relu(torch.randn(B, S, F, generator=g))
        f4 = torch.relu(torch.randn(B, S, L, F, generator=g))
        want = O.plt_decoder_fwd_layer(f3.double(), plt.W.detach().cpu().double(), plt.b.detach().cpu().double(), layer)
        assert_close(plt(f3.cuda(), layer=layer), want, 1e-5, 2e-6 * float(want.abs().max()), f"decoder layer {layer}")
        want = O.plt_decoder_fwd_layer(f4.double(), plt.W.detach().cpu().double(), plt.b.detach().cpu().double(), layer)
        assert_close(plt.forward_layer(f4.cuda(), layer), want, 1e-5, 2e-6 * float(want.abs().max()), f"decoder 4-dim layer {layer}")
        fc = f4[:, :, :layer + 1].contiguous()
        want = O.crosslayer_decoder_fwd_layer(fc.double(), [w.double() for w in Ws], bc.double(), layer)
        assert_close(clt(fc.cuda(), layer=layer), want, 1e-5, 2e-6 * float(want.abs().max()), f"crosslayer decoder layer {layer}")


def test_forward_layer_paths_bf16():
    """bf16 mode: operands rounded to bf16, fp32 accumulate — against the oracle with the same operand rounding."""
    from crosslayer_transcoder_amd import precision
    from oracle import clt_oracle as O
    B, S, L, d, F = 4, 64, 3, 128, 512
    g = torch.Generator().manual_seed(4)
    enc, plt, clt = _modules(L, d, F, g)
    x = torch.randn(B, S, d, generator=g)
    f4 = torch.relu(torch.randn(B, S, L, F, generator=g))
    with precision("bf16-mixed"), O.operand_rounding(torch.bfloat16):
        for layer in range(L):
            want = O.encoder_fwd_layer(x, enc.W.detach().cpu(), enc.b.detach().cpu(), layer)
            # (the encoder product is rounded to bf16 before the bias joins it — autocast's order: summation-order noise of the
            # fp32 accumulator flips a few of those roundings)
            assert_close_rounded(enc(x.cuda(), layer=layer), want, 2.0 ** -7, 1e-4, 1e-5 * float(want.abs().max()), what="encoder bf16")
            fc = f4[:, :, :layer + 1].contiguous()
            want = O.crosslayer_decoder_fwd_layer(fc, [w.detach().cpu() for w in clt.weights()], clt.b.detach().cpu(), layer)
            assert_close(clt(fc.cuda(), layer=layer), want, 1e-4, 1e-5 * float(want.abs().max()), "crosslayer decoder bf16")
            want = O.plt_decoder_fwd_layer(f4, plt.W.detach().cpu(), plt.b.detach().cpu(), layer)
            assert_close(plt(f4.cuda(), layer=layer), want, 1e-4, 1e-5 * float(want.abs().max()), "decoder bf16")


def test_sparsifier_inference_forward():
    from crosslayer_transcoder_amd.model import BatchTopK, JumpReLU, PerLayerBatchTopK, PerLayerTopK, ReLU
    from oracle import clt_oracle as O
    g = torch.Generator().manual_seed(6)
    B, S, L, F, k = 3, 5, 4, 96, 7
    n = B * S * F
    x = ((torch.randperm(n, generator=g).float() - n / 3) / 17.0).reshape(B, S, F)      # tie-free
    # PerLayerTopK: inference = the per-row top-k of the training forward (topk.py:49-51), any leading shape
    got = PerLayerTopK(k=k)(x.cuda(), layer=2)
    want = O.per_layer_topk_fwd(x.reshape(B * S, 1, F), k)[0].reshape(B, S, F)
    assert torch.equal(got.cpu(), want)
    x3 = x.reshape(B * S // 3, 3, F)
    assert torch.equal(PerLayerTopK(k=k)(x3.cuda(), layer=0).cpu(), O.per_layer_topk_fwd(x3, k)[0])
    # BatchTopK / PerLayerBatchTopK: compare with max(0, threshold) (topk.py:71-74, 104-107), strict >
    for thr in (-0.5, 0.0, 1.25):
        m = BatchTopK(k=k, e=0.1).cuda()
        m.threshold.fill_(thr)
        xx = x.clone()
        xx[0, 0, :4] = torch.tensor([thr, max(thr, 0.0), -0.0, 0.0])               # boundary values stay out
        assert torch.equal(m(xx.cuda(), layer=1).cpu(), O.topk_inference(xx, thr))
        m = PerLayerBatchTopK(k=k, e=0.1, n_layers=L).cuda()
        m.threshold.copy_(torch.tensor([9.0, thr, -3.0, 0.5]))
        assert torch.equal(m(xx.cuda(), layer=1).cpu(), O.topk_inference(xx, thr))
        assert torch.equal(m(xx.cuda(), layer=3).cpu(), O.topk_inference(xx, 0.5))
    # ReLU / JumpReLU take no layer argument in the reference (jumprelu.py:21, 61): same forward
    assert torch.equal(ReLU()(x.cuda()).cpu(), O.relu_fwd(x))
    j = JumpReLU(theta=0.4, bandwidth=0.1, n_layers=S, d_features=F).cuda()
    assert torch.equal(j(x.cuda()).detach().cpu(), O.jumprelu_fwd(x, torch.full((1, S, F), 0.4)))
