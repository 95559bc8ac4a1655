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
