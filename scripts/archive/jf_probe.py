"""jumprelu_fwd_bf16 tile pass at the PLT / CLT shape (bf16 pre-activations in): time per call, HIP events."""
import torch
from crosslayer_transcoder_amd import ops

B, L, F = 4096, 12, 32768
x = (torch.randn(B, L, F, device="cuda") * 0.5).bfloat16()
theta = torch.full((1, L, F), 0.03, device="cuda")
ts = []
for it in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = ops.jumprelu_fwd_bf16(x, theta)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    del out
t = sorted(ts[1:])[2]
print(f"jumprelu_fwd_bf16: {t * 1e3:.0f} us  {B * L * F * 6 / t / 1e6:.0f} GB/s")
