#!/usr/bin/env python3
"""What the decoder forward from the [F][d] weights (k-major B operand) would change in the dense 16-bit step: the grouped
cross-layer forward and the per-layer forward in both forms, and the optimizer's pass with / without the transposed copy.
python scripts/archive/km_adoption_probe.py   (one MI355X; HIP events, median of 5)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crosslayer_transcoder_amd as cta  # noqa: E402
from crosslayer_transcoder_amd import ops  # noqa: E402
from crosslayer_transcoder_amd._lib import lib  # noqa: E402

cta.set_precision("bf16-mixed")
dev = torch.device("cuda:0")
Bn, L, d, F = 4096, 12, 768, 32768


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


fo = (torch.rand(Bn, L, F, device=dev) * 2 - 1).bfloat16()
bias = torch.randn(L, d, device=dev)
Wp = ((torch.rand(L, F, d, device=dev) * 2 - 1) * 0.01).bfloat16()
WpT = Wp.transpose(1, 2).contiguous()
a = ops.plt_decoder_fwd_t(fo, WpT, bias)
b = ops.plt_decoder_fwd(fo, Wp, bias)
print("per-layer forward: NT + tail split %.3f ms, from [F][d] (one batched launch) %.3f ms, max |diff| %.3g" % (
    timed(lambda: ops.plt_decoder_fwd_t(fo, WpT, bias)), timed(lambda: ops.plt_decoder_fwd(fo, Wp, bias)), float((a - b).abs().max())))
del Wp, WpT, a, b
Ws = [((torch.rand(l + 1, F, d, device=dev) * 2 - 1) * 0.01).bfloat16() for l in range(L)]
WTs = [w.transpose(1, 2).contiguous() for w in Ws]
a = ops.clt_decoder_fwd_t(fo, WTs, bias)
b = ops.clt_decoder_fwd_t(fo, Ws, bias, k_major=True)
print("cross-layer forward (78 pairs, one grouped launch): NT %.3f ms, from [F][d] %.3f ms, equal bits: %s" % (
    timed(lambda: ops.clt_decoder_fwd_t(fo, WTs, bias)), timed(lambda: ops.clt_decoder_fwd_t(fo, Ws, bias, k_major=True)), torch.equal(a, b)))
del Ws, WTs, a, b, fo
torch.cuda.empty_cache()
p = torch.randn(L, F, d, device=dev) * 0.01
g, m, v = torch.randn_like(p) * 1e-3, torch.zeros_like(p), torch.zeros_like(p)
op, opT = torch.empty(L, F, d, dtype=torch.bfloat16, device=dev), torch.empty(L, d, F, dtype=torch.bfloat16, device=dev)
sq = torch.empty(L * F, device=dev)
st = torch.cuda.current_stream().cuda_stream


def prep(with_t):
    rc = lib.clt_adam_prep_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), L, F, d, 1e-4, 0.9, 0.999, 1e-8, 0.0, 0, 1, op.data_ptr(),
                               opT.data_ptr() if with_t else None, sq.data_ptr(), None, None, st)
    assert rc == 0, rc


t1, t0 = timed(lambda: prep(True)), timed(lambda: prep(False))
n = L * F * d
print("optimizer pass over a [12][32768][768] decoder: with the transposed copy %.3f ms (%.2f TB/s of 32 B/param), without %.3f ms (%.2f TB/s of 30 B/param)" % (
    t1, 32.0 * n / t1 * 1e-9, t0, 30.0 * n / t0 * 1e-9))
sh = torch.empty(L, F, d, dtype=torch.bfloat16, device=dev)
t2 = timed(lambda: ops.adam(p, g, m, v, 1, 1e-4, 0.9, 0.999, 1e-8, 0.0, False, bf16_shadow=sh))
print("plain adam_kernel + 16-bit copy (no row norms): %.3f ms (%.2f TB/s of 30 B/param)" % (t2, 30.0 * n / t2 * 1e-9))
