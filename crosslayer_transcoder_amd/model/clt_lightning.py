"""Training modules with the reference's class names, constructor kwargs and `training_step` /
`configure_optimizers` contract (reference model/clt_lightning.py:29-567), running the fused HIP engine.

If `lightning` is importable the classes derive from `lightning.LightningModule`; otherwise from a small
stand-in base (nn.Module + log / global_step / trainer) so that the same code runs under
crosslayer_transcoder_amd.trainer.fit — Lightning is not installed in the build image.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import engine, ops
from .. import metrics as _metrics
from ..precision import get_precision
from ..optim import FusedAdam
from .clt import CrossLayerTranscoder

try:                                                   # pragma: no cover - not installed in the build image
    import lightning as _L
    _Base = _L.LightningModule
    HAVE_LIGHTNING = True
except Exception:                                      # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Trainer:
        max_steps = 1
        num_devices = 1

    class _Base(nn.Module):
        """What training_step touches of LightningModule: log(), global_step, trainer."""

        def __init__(self, *args, **kwargs):
            super().__init__()
            self.logged = {}
            self.global_step = 0
            self.trainer = _Trainer()

        def log(self, name, value, *args, **kwargs):
            self.logged[name] = value

        def save_hyperparameters(self, *args, **kwargs):
            pass

        # ---- what Lightning's manual-optimisation API offers a training_step -------------------
        def attach_optimizers(self, conf=None):
            """Stand-in for the trainer's optimizer set-up: keeps what configure_optimizers() returned, so that
            optimizers() / lr_schedulers() answer as under Lightning.  Returns (optimizer, scheduler or None)."""
            conf = self.configure_optimizers() if conf is None else conf
            if isinstance(conf, dict):
                self._optimizer, self._scheduler = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
            else:
                self._optimizer, self._scheduler = conf, None
            return self._optimizer, self._scheduler

        def optimizers(self):
            if getattr(self, "_optimizer", None) is None:
                self.attach_optimizers()
            return self._optimizer

        def lr_schedulers(self):
            if getattr(self, "_optimizer", None) is None:
                self.attach_optimizers()
            return self._scheduler


class _StepLoss(torch.autograd.Function):
    """Connects the engine's explicit gradients to `loss.backward()`: forward returns the loss value,
    backward hands each parameter its precomputed gradient times the incoming gradient (a loss scaler or
    `accumulate_grad_batches` multiplies the loss; `unit_root` skips that multiply for callers that guarantee
    a unit root gradient)."""

    @staticmethod
    def forward(ctx, loss, unit_root, *params_and_grads):
        n = len(params_and_grads) // 2
        ctx.grads = params_and_grads[n:]
        ctx.unit_root = unit_root
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        gs = ctx.grads if ctx.unit_root else tuple(g * gout for g in ctx.grads)
        return (None, None) + tuple(gs) + (None,) * len(gs)


class CrossLayerTranscoderModule(_Base):
    _loss_kind = "base"

    def __init__(self, model: CrossLayerTranscoder, replacement_model=None, dead_features=None,
                 learning_rate: float = 1e-3, compile: bool = False, warmup_steps: int = 0, lr_decay_step: int = 0,
                 lr_decay_factor: float = 1.0, compute_dead_features: bool = False,
                 compute_dead_features_every: int = 100, tokens_till_dead: int = 1_000_000, optimizer: str = "adam",
                 beta1: Optional[float] = None, beta2: Optional[float] = None, log_metrics_every: int = 50,
                 fused_step: bool = True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        #: training_step runs forward + backward + Adam itself (Lightning manual optimisation): the weight matrices
        #: are updated inside the backward row pass and no dense gradient tensor exists.  fused_step=False keeps
        #: Lightning's automatic optimisation: training_step returns a loss whose backward() hands out materialised
        #: gradients and the trainer calls optimizer.step() (reference clt_lightning.py:290-312 as is).
        self.fused_step = bool(fused_step)
        self.automatic_optimization = not self.fused_step
        self.model = model
        self.replacement_model = replacement_model      # eval-time metric objects are accepted and ignored
        self.dead_features = dead_features
        self.learning_rate = learning_rate
        #: reference clt_lightning.py:100-104 (`self = torch.compile(self)` in configure_model — it rebinds a local name, so the
        #: flag has no effect there).  Here forward() runs a dispatcher-level capture
        #: of the model (backend "aot_eager": a graph of the registered clt_hip ops, custom_ops.py; there is nothing for a
        #: code generator to fuse — the kernels are hand-written) and self.model stays unwrapped, so state-dict keys keep the
        #: reference's names.  The fused training step does not run the module forward and is unaffected.
        self.compile = compile
        object.__setattr__(self, "_compiled_model", None)
        self.tokens_till_dead = tokens_till_dead
        self.warmup_steps, self.lr_decay_step, self.lr_decay_factor = warmup_steps, lr_decay_step, lr_decay_factor
        self.compute_dead_features = compute_dead_features
        self.compute_dead_features_every = compute_dead_features_every
        self.optimizer, self.beta1, self.beta2 = optimizer, beta1, beta2
        self.log_metrics_every = log_metrics_every
        #: fused_step=False only: gradients handed out by loss.backward() are multiplied by the incoming gradient
        #: (GradScaler, accumulate_grad_batches); True skips the multiply for callers that guarantee a unit root.
        self.assume_unit_loss_grad = False
        assert model.encoder.n_layers == model.decoder.n_layers, "Encoder and decoder must have the same number of layers"
        self.register_buffer("last_active", torch.zeros((model.encoder.n_layers, model.encoder.d_features),
                                                        dtype=torch.long))
        self.last_step = None

    def configure_model(self):
        """reference clt_lightning.py:100-119 (tensor parallelism when trainer.num_devices > 1): with an initialised
        torch.distributed group of more than one rank, this rank keeps its F/world slice of every feature-axis tensor
        (parallel.shard_module) and the step's exchange points go through RCCL."""
        import torch.distributed as dist
        if getattr(self, "shard", None) is not None or not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() > 1:
            from ..parallel import FeatureShard, shard_module
            shard_module(self, FeatureShard(self.model.encoder.d_features))

    def forward(self, acts_norm):
        if self.compile:
            if self._compiled_model is None:
                object.__setattr__(self, "_compiled_model", torch.compile(self.model, backend="aot_eager"))   # (not a submodule)
            return self._compiled_model(acts_norm)
        return self.model(acts_norm)

    # ---- the hot path ---------------------------------------------------------------------------
    def _named_model_params(self):
        return dict(self.model.named_parameters())

    def _log_terms(self, terms):
        names = {"loss": "training/loss", "sparsity": "training/sparsity_loss", "pre_actv": "training/pre_actv_loss",
                 "aux": "training/aux_loss", "dead_fraction": "metrics/dead_features", "mse": "metrics/train_mse",
                 "sparsity_penalty": "training/sparsity_penalty",
                 "weighted_features_mean": "model/weighted_features_mean"}
        for k, v in terms.items():
            self.log(names.get(k, k), v)

    def compute_step(self, batch, batch_idx, optimizer=None, sparse=None):
        """Loss + gradients through the fused engine (no autograd).  Result kept in `self.last_step`.
        With `optimizer` (a FusedAdam) the sparse regime applies Adam to the weight matrices inside its
        backward row pass; those parameters are then absent from out["grads"]."""
        log_now = self.log_metrics_every > 0 and batch_idx % self.log_metrics_every == 0
        if batch.dtype != torch.float32:        # the shared-memory buffer hands over fp16 (datamodule contract): widen once
            batch = batch.float()
        with torch.no_grad():
            # (weight norms are logged from the pre-update weights, as the reference does inside training_step)
            norms = _metrics.weight_norm_metrics(self.model) if log_now and batch.is_cuda else None
            out = engine.training_step(self, batch, batch_idx, optimizer, sparse)
            self.last_step = out
            self._log_terms(out["terms"])
            if log_now:
                self.log_training_metrics(out, batch, batch_idx, norms)
        return out

    def log_training_metrics(self, out, batch, batch_idx, weight_norms=None):
        """clt_lightning.py:131-280 without the host copy of the feature tensor: every value comes from device-side
        reductions of what the step already produced (crosslayer_transcoder_amd/metrics.py)."""
        counts = _metrics.fire_counts(out)
        for k, v in _metrics.training_metrics(self, out, batch, counts, weight_norms).items():
            self.log(k, v)
        self.tokens_seen = getattr(self, "tokens_seen", 0) + batch.shape[0] * self.log_metrics_every      # :160-163
        self.log("training/tokens_seen", self.tokens_seen)
        if self.compute_dead_features and self.dead_features is not None:                               # :243-280
            self.dead_features.update(counts=counts, tokens=batch.shape[0])
            if self.global_step % self.compute_dead_features_every == 0:
                stats = self.dead_features.compute()
                self.log("metrics/dead_features_mean", stats["mean"])
                for i, v in enumerate(stats.get("per_layer", [])):
                    self.log(f"layers/dead_features/layer_{i}", v)
                if "log_freqs" in stats:
                    self.log("training/log_feature_density_mean", stats["log_freqs"].mean())
                self.dead_features.reset()

    def training_step(self, batch, batch_idx):
        """reference clt_lightning.py:290-312 / :420-475 / :519-567.  fused_step (default): the whole optimisation
        step — forward, backward, Adam (the trainer's own optimizer and LR scheduler objects), under Lightning's
        manual optimisation; returns the detached loss.  fused_step=False: returns a loss for loss.backward()."""
        if self.fused_step:
            opt = self.optimizers()
            raw = getattr(opt, "optimizer", opt)               # LightningOptimizer wraps the FusedAdam
            if not isinstance(raw, FusedAdam):
                raise TypeError("fused_step=True needs the FusedAdam that configure_optimizers() returns "
                                "(set fused_step=False to train with another optimizer)")
            sched = self.lr_schedulers()
            return self.fused_train_step(batch, batch_idx, raw, sched, stepper=opt)
        out = self.compute_step(batch, batch_idx)
        params = self._named_model_params()
        names = [n for n in params if n in out["grads"]]
        ps = [params[n] for n in names]
        gs = [out["grads"][n].reshape(params[n].shape) for n in names]
        if out["grad_scale"] != 1.0:                # fp16 mode: the engine's own loss scale leaves the handed-out gradients
            gs = [g * (1.0 / out["grad_scale"]) for g in gs]
        return _StepLoss.apply(out["loss"], self.assume_unit_loss_grad, *ps, *gs)

    def fused_train_step(self, batch, batch_idx, optimizer: "FusedAdam", scheduler=None, fuse_adam=True, sparse=None,
                         stepper=None):
        """fwd + bwd + Adam with no autograd bookkeeping.  `stepper`: the object whose step() is called for the
        parameters the backward pass did not update itself (Lightning's optimizer wrapper, which also counts
        global_step; default: `optimizer`)."""
        optimizer.track_steps = get_precision() == "fp16"
        out = self.compute_step(batch, batch_idx, optimizer if fuse_adam else None, sparse)
        params = self._named_model_params()
        for n, p in params.items():
            g = out["grads"].get(n)
            p.grad = None if g is None else g.reshape(p.shape)
        scaler = getattr(self, "loss_scaler", None) if get_precision() == "fp16" else None
        if scaler is not None:
            # fp16 loss scaling is the library's own (the kernels unscale and skip): the raw optimizer steps, not a wrapper that
            # would route through an outer GradScaler; one host read of the overflow flag closes the step.  Under Lightning's
            # manual optimisation trainer.global_step advances in the LightningOptimizer's _on_before_step / _on_after_step
            # hooks (optim_step_progress): they are called around the raw step, so max_steps, the lambda schedule
            # (global_step / max_steps) and step-based logging move in this mode as well.
            flag = ops.inf_flag(batch.device)
            engine.agree_on_overflow(getattr(self, "shard", None), flag)     # (feature shards: one decision for all ranks)
            before, after = getattr(stepper, "_on_before_step", None), getattr(stepper, "_on_after_step", None)
            if callable(before):
                before()
            optimizer.step()
            if callable(after):
                after()
            skipped = bool(flag.item())
            optimizer.commit(skipped)
            scaler.update(skipped)
            self.log("training/loss_scale", scaler.scale)
        else:
            (stepper if stepper is not None else optimizer).step()
            optimizer.commit()
        if scheduler is not None:
            scheduler.step()
        if not HAVE_LIGHTNING:
            self.global_step += 1
        return out["loss"]

    def validation_step(self, batch, batch_idx):
        pass

    def configure_optimizers(self):
        b1 = 0.9 if self.beta1 is None else self.beta1
        b2 = 0.999 if self.beta2 is None else self.beta2
        if self.optimizer == "adam":
            opt = FusedAdam(self.parameters(), lr=self.learning_rate, betas=(b1, b2))
        elif self.optimizer == "adamw":
            opt = FusedAdam(self.parameters(), lr=self.learning_rate, betas=(b1, b2), weight_decay=0.01, adamw=True)
        else:
            raise ValueError(f"Optimizer {self.optimizer} not supported")
        sched = None
        if self.warmup_steps > 0:
            sched = ("warmup", torch.optim.lr_scheduler.LinearLR(opt, start_factor=1e-7, end_factor=1.0,
                                                                 total_iters=self.warmup_steps))
        elif self.lr_decay_step > 0:
            sched = ("step_decay", torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[self.lr_decay_step],
                                                                       gamma=self.lr_decay_factor))
        if sched is None:
            return opt
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched[1], "interval": "step", "frequency": 1,
                                                   "name": sched[0]}}


class JumpReLUCrossLayerTranscoderModule(CrossLayerTranscoderModule):
    _loss_kind = "jumprelu"

    def __init__(self, lambda_sparsity: float = 0.0002, c_sparsity: float = 0.1, use_tanh: bool = True,
                 pre_actv_loss: float = 0.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._lambda = lambda_sparsity
        self.c = c_sparsity
        self.use_tanh = use_tanh
        self.pre_actv_loss = pre_actv_loss

    def current_sparsity_penalty(self):
        return self._lambda * (self.global_step / self.trainer.max_steps)


class TopKCrossLayerTranscoderModule(CrossLayerTranscoderModule):
    _loss_kind = "topk"

    def __init__(self, topk_aux: nn.Module, tokens_till_dead: int = 100_000, aux_loss_scale: float = 1 / 32,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tokens_till_dead = tokens_till_dead
        self.topk_aux = topk_aux
        self.aux_loss_scale = aux_loss_scale
