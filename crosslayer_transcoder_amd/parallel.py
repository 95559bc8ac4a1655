"""Feature-axis sharding of the training step over one process per GPU (replaces the DTensor tensor-parallel
plumbing of reference model/parallel.py:157-339, which shards the same axis: encoder W `Shard(-1)` :186-192,
theta `Shard(-1)` :235-255, decoder W_i `Shard(-2)` :303-310, decoder output Partial->Replicate :313-326).

Rank r of `world` owns features [f_off, f_off + F_local) of every layer: its rows of the encoder weight, of every
decoder matrix, of theta / last_active and of the Adam state.  The batch, decoder bias and standardizer buffers are
replicated.  Per step there is ONE exchange of activations — all-reduce(sum) of the partial reconstruction
[B, L, d] (a second one for the TopK-aux pass) — plus the few-KB exchanges of the global BatchTopK selection
(lower bound MIN, three radix-histogram SUMs, tie lists ALL-GATHER) and one packed scalar all-reduce.  Backward and
Adam are rank-local: g = dL/drecons is replicated, weight shards are disjoint.

Collectives go through torch.distributed (backend "nccl" = RCCL over xGMI).  With the gloo backend (CPU tests, or
several ranks sharing one GPU) device tensors are staged through host memory.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class FeatureShard:
    def __init__(self, F_global: int, group=None, rank: Optional[int] = None, world: Optional[int] = None,
                 keep_single: bool = False):
        self.group = group
        #: world == 1 normally drops to the unsharded step; keep_single runs the sharded schedule (all collectives)
        #: on the single rank anyway — the way to exercise the RCCL path on a one-GPU box
        self.keep_single = keep_single
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        if F_global % self.world:
            raise ValueError(f"d_features ({F_global}) must be divisible by the number of shards ({self.world})")
        self.F_global = F_global
        self.F_local = F_global // self.world
        self.f_off = self.rank * self.F_local
        self._host_staged = dist.get_backend(group) == "gloo"

    # ---- collectives (in place) -----------------------------------------------------------------
    def _run(self, fn, t: torch.Tensor):
        if self._host_staged and t.is_cuda:
            h = t.cpu()
            fn(h)
            t.copy_(h)
        else:
            fn(t)
        return t

    def all_reduce_sum(self, t):
        return self._run(lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group), t)

    def all_reduce_sum_async(self, t):
        """Starts all-reduce(sum) of `t` in place and returns wait(): RCCL runs it on its own stream behind the work
        already queued on the current stream; wait() makes the CURRENT STREAM wait for it (the host does not block), so
        kernels queued in between overlap the exchange.  Host-staged (gloo) groups reduce at once."""
        if self._host_staged and t.is_cuda:
            self.all_reduce_sum(t)
            return lambda: None
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work.wait

    def reduce_scatter_rows_async(self, outs, ins):
        """For every pair: outs[i] [rows / world, d] <- block `rank` of the sum over ranks of ins[i] [rows, d].  All pairs go
        out as ONE coalesced RCCL launch on RCCL's stream, behind the work already queued on the current stream; the returned
        wait() makes the stream that is current WHEN IT IS CALLED wait for it.  Host-staged (gloo) groups: all-reduce on the
        host, own block kept, at once."""
        if self._host_staged:
            for o, i in zip(outs, ins):
                h = i.cpu() if i.is_cuda else i.clone()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                n = o.shape[0]
                o.copy_(h[self.rank * n:(self.rank + 1) * n])
            return lambda: None
        with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
            for o, i in zip(outs, ins):
                dist.reduce_scatter_tensor(o, i, op=dist.ReduceOp.SUM, group=self.group)
        return cm.wait

    def all_gather_rows_async(self, outs, ins):
        """For every pair: outs[i] [rows, d] <- the ranks' ins[i] [rows / world, d] in rank order; one coalesced launch, wait()
        as in reduce_scatter_rows_async."""
        if self._host_staged:
            for o, i in zip(outs, ins):
                h = torch.empty(o.shape, dtype=o.dtype)
                dist.all_gather_into_tensor(h, i.cpu() if i.is_cuda else i, group=self.group)
                o.copy_(h)
            return lambda: None
        with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
            for o, i in zip(outs, ins):
                dist.all_gather_into_tensor(o, i, group=self.group)
        return cm.wait

    def all_reduce_min(self, t):
        return self._run(lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN, group=self.group), t)

    def all_gather_blocks(self, region: torch.Tensor):
        """region: [world, n]; block `rank` holds this rank's data; fills every block."""
        mine = region[self.rank].clone().reshape(-1)
        if self._host_staged and region.is_cuda:
            h = torch.empty(region.numel(), dtype=region.dtype)
            dist.all_gather_into_tensor(h, mine.cpu(), group=self.group)
            region.copy_(h.view(region.shape))
        else:
            assert region.is_contiguous()
            dist.all_gather_into_tensor(region.view(-1), mine, group=self.group)
        return region

    def all_gather_cat(self, t: torch.Tensor, dim: int) -> torch.Tensor:
        """Concatenation over ranks of every rank's `t` along `dim` (rank order = feature order)."""
        src = t.detach().contiguous()
        if self._host_staged and src.is_cuda:
            parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(self.world)]
            dist.all_gather(parts, src.cpu(), group=self.group)
        else:
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
        return torch.cat(parts, dim=dim)

    def slice_features(self, t: torch.Tensor, dim: int) -> torch.Tensor:
        """This rank's slice of a full-size tensor along its feature axis."""
        return t.narrow(dim, self.f_off, self.F_local)


def run_sharded_selection(stages, shard: Optional[FeatureShard], S: int, exchange, after_stage0=None, tie_radix=False):
    """Collective schedule of the sharded BatchTopK / PerLayerBatchTopK selection (include/clt_hip.h,
    clt_select_lists_sharded_f32).  `stages(lo, hi)` runs the local kernels of stages lo..hi; `exchange` maps
    "lb" -> int32 [S, 4] records (column 0 = lower-bound key), "hist" -> int32 [S*2048 + S], "ties" -> int32
    [world, tie_words] views of the selection workspace."""
    stages(0, 0)
    if after_stage0 is not None:             # fused encoder GEMM + filter: the bound is the same on every rank already
        after_stage0()
    elif shard is not None:
        lb = exchange["lb"]
        keys = lb[:, 0].to(torch.int64) & 0xFFFFFFFF            # uint32 keys: order them as unsigned
        shard.all_reduce_min(keys)
        lb[:, 0] = (keys - (keys >= 2 ** 31).to(torch.int64) * 2 ** 32).to(torch.int32)  # back to the uint32 bit pattern
    if shard is None:                        # one GPU: no exchange points
        stages(1, 5)
        return
    for st in (1, 2, 3):
        stages(st, st)
        shard.all_reduce_sum(exchange["hist"])
    if tie_radix:            # exact for any number of ties: radix select on the global flat index among the tied entries
        for st in (4, 5, 6, 7):
            stages(st, st)
            shard.all_reduce_sum(exchange["hist"])
        stages(8, 8)
        return
    stages(4, 4)
    shard.all_gather_blocks(exchange["ties"])
    stages(5, 5)


def run_sharded_rows_selection(stages, shard: FeatureShard, exchange):
    """Collective schedule of the sharded per-row TopK (clt_select_rows_sharded_f32): eight 4-bit radix rounds, each
    closed by an all-reduce(sum) of the [rows*16] bucket counts, then an all-gather of the per-row tie counts."""
    for st in range(8):
        stages(st, st)
        shard.all_reduce_sum(exchange["counts"])
    stages(8, 8)
    shard.all_gather_blocks(exchange["ties"])
    stages(9, 9)


def shard_module(mod, shard: FeatureShard):
    """Turn a full-size training module (identical on every rank, e.g. same seed or same checkpoint) into this
    rank's shard IN PLACE: parameters and buffers with a feature axis keep only [f_off, f_off + F_local).
    Afterwards `mod.shard` routes the step's exchange points through `shard` (engine.training_step)."""
    import torch.nn as nn
    model = mod.model
    enc, dec, nl = model.encoder, model.decoder, model.nonlinearity
    if enc.d_features != shard.F_global:
        raise ValueError(f"module has d_features={enc.d_features}, shard expects {shard.F_global}")
    f0, n = shard.f_off, shard.F_local
    with torch.no_grad():
        Wt = enc.Wt[:, f0:f0 + n].contiguous()                        # feature-major storage [L, F_local, d]
        enc.W = nn.Parameter(Wt.permute(0, 2, 1))
        enc.b = nn.Parameter(enc.b[:, f0:f0 + n].contiguous())
        enc.d_features = n
        if hasattr(dec, "W"):
            dec.W = nn.Parameter(dec.W[:, f0:f0 + n].contiguous())
        else:
            for l in range(dec.n_layers):
                setattr(dec, f"W_{l}", nn.Parameter(getattr(dec, f"W_{l}")[:, f0:f0 + n].contiguous()))
        dec.d_features = n
        if hasattr(nl, "theta"):
            nl.theta = nn.Parameter(nl.theta[..., f0:f0 + n].contiguous())
            if hasattr(nl, "d_features"):
                nl.d_features = n
        mod.last_active = mod.last_active[:, f0:f0 + n].contiguous()
    for m in (enc, dec):
        for attr in ("_bf16_shadow", "_wptr", "_wptr_sh", "_bf16_shadows"):
            if hasattr(m, attr):
                delattr(m, attr)
    mod.shard = shard
    return mod


#: feature axis of every state-dict entry that has one (reference model/parallel.py:186-192, 235-255, 303-310)
def _feature_axis(key: str):
    if key in ("encoder.W", "nonlinearity.theta"):
        return 2
    if key == "encoder.b" or key == "decoder.W" or key.startswith("decoder.W_"):
        return 1
    return None


def gather_full_state_dict(mod, shard: FeatureShard, to_cpu: bool = True):
    """Full-size state dict of a feature-sharded model (gather-on-save): entries with a feature axis are concatenated
    over ranks in feature order, tensor by tensor (moved to the host as they complete), the rest is rank 0's copy.
    Collective: every rank must call it; every rank gets the result."""
    full = {}
    for k, v in mod.model.state_dict().items():
        ax = _feature_axis(k)
        if ax is None:
            t = v.detach()
        elif k == "encoder.W":                  # stored feature-major [L, F, d]: gather that layout, return the logical view
            t = shard.all_gather_cat(mod.model.encoder.Wt, 1).permute(0, 2, 1)
        else:
            t = shard.all_gather_cat(v, ax)
        full[k] = t.cpu() if to_cpu else t
    return full


def save_pretrained_sharded(mod, shard: FeatureShard, directory, fold_standardizers: bool = True):
    """`CrossLayerTranscoder.save_pretrained` (reference clt.py:443-467) for a feature-sharded model: folding is local
    (it scales along d_acts only and the folded biases are per feature / replicated), the weights are gathered, and
    rank 0 writes `config.yaml` (global d_features) + `checkpoint.safetensors` with the reference's state-dict keys —
    loadable by `from_pretrained` on one GPU or by the reference's own tooling."""
    from pathlib import Path
    import yaml
    from safetensors.torch import save_file
    model = mod.model
    if fold_standardizers:
        model.fold()
    full = gather_full_state_dict(mod, shard)
    if shard.rank == 0:
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        config = model.to_config()

        def globalize(node):
            if isinstance(node, dict):
                for kk, vv in node.items():
                    if kk == "d_features" and vv == shard.F_local:
                        node[kk] = shard.F_global
                    else:
                        globalize(vv)
        globalize(config)
        config["is_folded"] = model._is_folded
        with open(directory / "config.yaml", "w") as fh:
            yaml.dump({"model": config}, fh)
        save_file({k: v.contiguous() for k, v in full.items()}, directory / "checkpoint.safetensors")
    if dist.is_initialized():
        dist.barrier(group=shard.group)
