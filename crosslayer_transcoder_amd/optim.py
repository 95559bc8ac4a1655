"""Fused Adam / AdamW on the HIP kernel clt_adam_f32 (replaces torch.optim.Adam(fused=True),
reference model/clt_lightning.py:339-352).  A torch.optim.Optimizer subclass so the reference's
LinearLR / MultiStepLR schedulers drive it unchanged."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, adamw=adamw))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2, group["eps"],
                         group["weight_decay"], group["adamw"], getattr(p, "_bf16_shadow", None))
        return loss
