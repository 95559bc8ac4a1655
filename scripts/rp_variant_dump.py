"""One fused-Adam row pass on seeded inputs, results saved for a bitwise comparison between library variants
(CLT_HIP_LIBRARY=... python scripts/rp_variant_dump.py out.pt; python scripts/rp_variant_dump.py --compare a.pt b.pt)."""
import sys
import torch
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k], b[k]
        ne = int((x != y).sum())
        print(k, "equal" if ne == 0 else f"{ne} of {x.numel()} differ, max abs {float((x.float() - y.float()).abs().max()):.3g}")
    sys.exit(0)
from crosslayer_transcoder_amd import ops
B, L, F, d = 512, 3, 2048, 768
g = torch.Generator().manual_seed(3)
feat = torch.zeros(B, L, F)
idx = torch.randint(0, F, (B, L, 16), generator=g)
feat.scatter_(2, idx, torch.rand(B, L, 16, generator=g) + 0.1)
S = ops.sparse_build(feat.cuda(), 16 * B * L)
src = ops.cast_bf16((torch.randn(B, L, d, generator=g) * 0.1).cuda())
W = (torch.randn(L, F, d, generator=g) * 0.05).cuda()
m = (torch.randn(L, F, d, generator=g) * 0.01).cuda()
v = (torch.randn(L, F, d, generator=g) * 0.01).abs().cuda()
sh = torch.zeros(L, F, d, dtype=torch.bfloat16, device="cuda")
dots = torch.zeros(1, S.cap, device="cuda")
hp = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, adamw=False, step=3)
ops.sparse_row_pass(W, S, S.csc_val, src, 0, L - 1, adam=hp, m=m, v=v, shadow=sh, dots=dots, dot_slot=0)
torch.cuda.synchronize()
torch.save({"W": W.cpu(), "m": m.cpu(), "v": v.cpu(), "shadow": sh.cpu(), "dots": dots.cpu()}, sys.argv[1])
