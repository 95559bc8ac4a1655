"""List build from 16-bit pre-activations with and without a threshold vector (same active set): is the threshold's L2
traffic (re-read per row) what holds count_rows / fill_rows at 4.2 TB/s?  python scripts/build_probe.py"""
import torch
from crosslayer_transcoder_amd import ops
B, L, F = 4096, 12, 32768
g = torch.Generator(device="cuda").manual_seed(1)
pre = torch.empty(B, L, F, device="cuda", dtype=torch.bfloat16).normal_(0, 0.087, generator=g)   # ~0.15 % above 0.26
theta = torch.full((1, L, F), 0.26, device="cuda")
shifted = (pre.float() - 0.26).bfloat16()
cap = B * L * F // 256
for name, x, th in (("theta", pre, theta), ("none", shifted, None)):
    for _ in range(2):
        s = ops.sparse_build_threshold(x, th, cap)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        s = ops.sparse_build_threshold(x, th, cap)
    e1.record(); torch.cuda.synchronize()
    print(name, "build %.3f ms" % (e0.elapsed_time(e1) / 10), "entries", int(s.total), "overflow", int(s.overflow))
