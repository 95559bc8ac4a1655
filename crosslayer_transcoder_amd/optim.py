"""Fused Adam / AdamW on the HIP kernel clt_adam_f32 (replaces torch.optim.Adam(fused=True),
reference model/clt_lightning.py:339-352).  A torch.optim.Optimizer subclass so the reference's
LinearLR / MultiStepLR schedulers drive it unchanged.

Parameters may be stored in any dense (non-overlapping) layout — the encoder weight is kept feature-major —
so the moment buffers are allocated with the parameter's own strides and the kernel walks raw memory."""
import torch

from . import ops


def _same_layout(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    if t.stride() == like.stride() and t.shape == like.shape:
        return t
    out = torch.empty_like(like)                 # preserve_format: takes `like`'s strides
    out.copy_(t)
    return out


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, adamw=adamw))
        self._row_terms = {}
        #: dense 16-bit regime (engine._Decoder.register_prep / engine._encoder_operand): buffers the update kernel refreshes in
        #: the same pass — id(p) -> (op, opT, sq_part) of a decoder matrix, id(p) -> 16-bit shadow of the encoder weight
        self._preps = {}
        self._shadows = {}
        #: fp16 loss scaling only (set by the training step): remember the parameters whose step count moved, so that
        #: commit(skipped=True) can give the counts back after a step the kernels skipped
        self.track_steps = False
        self._stepped = []

    def add_row_term(self, p, dnorm, sq):
        """The next step() adds (dnorm[r] / sqrt(sq[r])) * p[r, :] to p's gradient inside the update kernel (the
        gradient that reaches the decoder weights through dec_norms, clt_lightning.py:438-446) — instead of a
        separate read-modify-write pass over the gradient."""
        self._row_terms[id(p)] = (dnorm, sq)

    def add_prep(self, p, op, opT, sq_part):
        """The next step() of the contiguous decoder matrix p also writes the updated matrix's 16-bit copy, transposed 16-bit
        copy and squared row norms into these buffers (clt_adam_prep_f32) and tags p with them (`p._clt_prep`)."""
        self._preps[id(p)] = (op, opT, sq_part)

    def add_shadow(self, p, shadow):
        """The next step() of p also writes the updated values' 16-bit copy (raw memory order of p) into `shadow`."""
        self._shadows[id(p)] = shadow

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p)          # same strides as p
            st["exp_avg_sq"] = torch.zeros_like(p)
        return st

    def _group_of(self, p):
        for group in self.param_groups:
            for q in group["params"]:
                if q is p:
                    return group
        raise KeyError("parameter is not managed by this optimizer")

    def begin_fused(self, p, storage=None):
        """For a kernel that applies the update itself (clt_sparse_row_pass_f32): counts the step and returns
        the keyword arguments ops.sparse_row_pass needs (`adam` hyper-parameters, moment buffers)."""
        group = self._group_of(p)
        st = self._state(p)
        st["step"] += 1
        if self.track_steps:
            self._stepped.append(p)
        b1, b2 = group["betas"]
        return dict(adam=dict(lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"],
                              adamw=group["adamw"], step=st["step"]), m=st["exp_avg"], v=st["exp_avg_sq"])

    def commit(self, skipped: bool = False):
        """End of an optimisation step under loss scaling: when the kernels skipped the update (overflow flag raised, see
        ops.set_loss_scaling) the step counts taken for it are given back, as GradScaler.step leaves them."""
        if skipped:
            for p in self._stepped:
                self.state[p]["step"] -= 1
                # (the skipped kernels wrote nothing: buffers registered for this step — a fresh squared-norm buffer among them —
                # do not describe the weights; the next step prepares its operands from the fp32 weights again)
                for tag in ("_clt_prep", "_clt_shadow"):
                    if hasattr(p, tag):
                        delattr(p, tag)
        self._stepped = []

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
                st = self._state(p)
                st["step"] += 1
                if self.track_steps:
                    self._stepped.append(p)
                g = _same_layout(p.grad, p)
                row_term = self._row_terms.pop(id(p), None)
                prep = self._preps.pop(id(p), None)
                shadow = self._shadows.pop(id(p), None)
                done = prep is not None and ops.adam_prep(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2,
                                                          group["eps"], group["weight_decay"], group["adamw"], *prep, row_term=row_term)
                if not done:
                    ops.adam(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2, group["eps"],
                             group["weight_decay"], group["adamw"], shadow, row_term)
                p._clt_version = getattr(p, "_clt_version", 0) + 1      # cached 16-bit operand copies: re-cast, unless tagged below
                key = (p._version, p._clt_version, ops.op16())
                if done:
                    p._clt_prep = (key,) + tuple(prep)
                elif shadow is not None:
                    p._clt_shadow = (key, shadow)
        self._preps.clear()                     # (registrations are for ONE step: nothing stale survives a parameter without gradient)
        self._shadows.clear()
        return loss
