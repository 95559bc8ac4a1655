"""Build crosslayer_transcoder_amd modules from a golden case's cfg dict (tests/golden/make_golden.py)."""
import torch

from crosslayer_transcoder_amd.model import (BatchTopK, CrossLayerTranscoder, CrosslayerDecoder, Decoder, Encoder, JumpReLU,
                                             PerLayerBatchTopK, PerLayerTopK, ReLU)
from crosslayer_transcoder_amd.model.clt_lightning import (CrossLayerTranscoderModule, JumpReLUCrossLayerTranscoderModule,
                                                           TopKCrossLayerTranscoderModule)
from crosslayer_transcoder_amd.model.standardize import DimensionwiseInputStandardizer, DimensionwiseOutputStandardizer


def build_nl(nl, L, F):
    k = nl["kind"]
    if k == "relu":
        return ReLU()
    if k == "jumprelu":
        return JumpReLU(theta=nl["theta"], bandwidth=nl["bandwidth"], n_layers=L, d_features=F)
    if k == "topk":
        return PerLayerTopK(k=nl["k"])
    if k == "batchtopk":
        return BatchTopK(k=nl["k"], e=nl["e"])
    if k == "perlayerbatchtopk":
        return PerLayerBatchTopK(k=nl["k"], e=nl["e"], n_layers=L)
    raise ValueError(k)


def build_module(case, device="cuda:0"):
    cfg, opt = case["cfg"], case["opt"]
    _, B, _, L, d = case["batches"].shape
    F = cfg["d_features"]
    dec = (CrosslayerDecoder if cfg["decoder"] == "clt" else Decoder)(d, F, L)
    model = CrossLayerTranscoder(build_nl(cfg["nonlinearity"], L, F), Encoder(d, F, L), dec,
                                 DimensionwiseInputStandardizer(L, d), DimensionwiseOutputStandardizer(L, d))
    lc = cfg["loss"]
    common = dict(model=model, learning_rate=opt["lr"], warmup_steps=opt.get("warmup_steps", 0),
                  lr_decay_step=opt.get("lr_decay_step", 0), lr_decay_factor=opt.get("lr_decay_factor", 1.0),
                  optimizer=opt.get("optimizer", "adam"), beta1=opt.get("beta1"), beta2=opt.get("beta2"))
    if lc["kind"] == "base":
        mod = CrossLayerTranscoderModule(tokens_till_dead=lc.get("tokens_till_dead", 1_000_000), **common)
    elif lc["kind"] == "jumprelu":
        mod = JumpReLUCrossLayerTranscoderModule(lambda_sparsity=lc["lambda_sparsity"], c_sparsity=lc["c_sparsity"],
                                                 use_tanh=lc["use_tanh"], pre_actv_loss=lc["pre_actv_loss"], **common)
    else:
        mod = TopKCrossLayerTranscoderModule(topk_aux=build_nl(lc["topk_aux"], L, F), tokens_till_dead=lc["tokens_till_dead"],
                                             aux_loss_scale=lc["aux_loss_scale"], **common)
    with torch.no_grad():
        for k, p in mod.model.named_parameters():
            p.copy_(case["init_params"][k])
        mod.last_active.copy_(case["last_active0"])
    mod.trainer.max_steps = case["max_steps"]
    return mod.to(device)
