#!/usr/bin/env python3
"""Micro-benchmark of the encoder-forward GEMM (G32 shape) through the C ABI; HIP-event timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crosslayer_transcoder_amd import ops
B, L, d, F = 4096, 12, 768, 32768
if len(sys.argv) > 1:
    B, L, d, F = map(int, sys.argv[1:5])
dev = "cuda:0"
xs = torch.randn(B, L, d, device=dev).bfloat16()
Wt = (torch.rand(L, F, d, device=dev) * 2 - 1).mul_(F ** -0.5).bfloat16()
b = torch.zeros(L, F, device=dev)
for _ in range(2):
    pre = ops.encoder_fwd(xs, Wt, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    pre = ops.encoder_fwd(xs, Wt, b)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"diag={os.environ.get('CLT_GEMM_DIAG','0')} var={os.environ.get('CLT_GEMM_VAR','0')}  {ms:.3f} ms  {2*B*L*d*F/ms/1e9:.0f} TFLOP/s  out {B*L*F*4/ms/1e6:.0f} GB/s")
