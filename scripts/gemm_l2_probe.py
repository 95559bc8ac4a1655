#!/usr/bin/env python3
"""Does operand residency limit the NT GEMM main loop?  Same tile count and k length as the encoder forward, but every
batch re-reads the SAME 1024x768 A and B (3 MB: L2-resident) instead of streaming 300 MB of weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crosslayer_transcoder_amd import ops
from crosslayer_transcoder_amd._lib import lib
dev = "cuda:0"
M = N = 1024; K = 768; batch = 1536
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
C = torch.empty(batch, M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = lib.clt_gemm_bf16(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N, 0, 1, batch, 0, 0, M * N, None, 0, 0, st)
    assert rc == 0, rc
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"L2-resident operands diag={os.environ.get('CLT_GEMM_DIAG','0')}: {ms:.3f} ms  {2*M*N*K*batch/ms/1e9:.0f} TFLOP/s")
