"""One rank of the feature-sharded parity run (launched by tests/test_sharded.py through torch.distributed.run).

GPU mode: every rank builds the FULL golden-case module, keeps its feature shard (parallel.shard_module) and runs
the fused training steps; losses, local gradients, selected features, buffers and the parameters after the Adam
steps must equal the corresponding slices of the UNSHARDED reference golden vectors.  Ranks may share one GPU: the
collectives then run over gloo with host staging.

--sim mode (CPU only, no HIP): checks the collective schedule of the sharded selection
(parallel.run_sharded_selection) with a torch restatement of the kernel stages' contract.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KEEP_SINGLE = False          # --backend nccl with one rank: run the sharded schedule through RCCL anyway

FEATURE_AXIS = {"encoder.W": 2, "encoder.b": 1, "nonlinearity.theta": 2, "decoder.W": 1}


def feature_axis(name):
    if name.startswith("decoder.W_"):
        return 1
    return FEATURE_AXIS.get(name)


def gpu_case(name, fuse_adam):
    from crosslayer_transcoder_amd.parallel import FeatureShard, shard_module
    from tests.build_cases import build_module
    from tests.conftest import load_golden
    from tests.golden_cases import assert_close, check_first_adam_step
    case = load_golden(name)
    mod = build_module(case, "cuda:0")
    F = case["cfg"]["d_features"]
    if KEEP_SINGLE:
        shard = FeatureShard(F, keep_single=True)
        shard_module(mod, shard)
    else:
        mod.configure_model()                 # the module contract: shards when the process group has > 1 rank
        shard = mod.shard
        assert shard.world == dist.get_world_size() and mod.model.encoder.d_features == F // shard.world
    opt, sched = mod.attach_optimizers()
    sl = lambda t, ax: t if ax is None else t.narrow(ax, shard.f_off, shard.F_local)  # noqa: E731
    params = dict(mod.model.named_parameters())
    for t, rec in enumerate(case["steps"]):
        loss = mod.fused_train_step(case["batches"][t].cuda(), t, opt, sched, fuse_adam=fuse_adam)
        assert_close(loss, rec["loss"], 2e-5, 1e-7, f"{name} loss step {t}")
        out = mod.last_step
        if t == 0:
            feat = out["features"].cpu()
            want = sl(rec["features"], 2)
            assert torch.equal(feat > 0, want > 0), f"{name}: active set differs on rank {shard.rank}"
            assert_close(feat, want, 1e-5, 1e-6, "features")
            assert_close(out["recons_norm"], rec["recons_norm"], 1e-5, 2e-6, "recons_norm")
            for k, g in out["grads"].items():
                gref = sl(rec["grads"][k], feature_axis(k))
                assert_close(g.reshape(gref.shape), gref, 1e-4, 2e-6 * float(rec["grads"][k].abs().max()) + 1e-12, f"grad {k}")
            check_first_adam_step(name, opt, params, case, rec, sl=lambda k, t: sl(t, feature_axis(k)))
        assert torch.equal(mod.last_active.cpu(), sl(rec["last_active"], 1)), f"last_active {t}"
        assert_close(out["terms"]["dead_fraction"], rec["logged"]["metrics/dead_features"], 1e-6, 1e-7, "dead")
        if "training/aux_loss" in rec["logged"]:
            assert_close(out["terms"]["aux"], rec["logged"]["training/aux_loss"], 5e-5, 1e-7, "aux")
        if "training/sparsity_loss" in rec["logged"]:
            assert_close(out["terms"]["sparsity"], rec["logged"]["training/sparsity_loss"], 5e-5, 1e-8, "sparsity")
            assert_close(out["terms"]["pre_actv"], rec["logged"]["training/pre_actv_loss"], 5e-5, 1e-10, "pre_actv")
        if "threshold" in rec:
            assert_close(mod.model.nonlinearity.threshold, rec["threshold"], 1e-5, 1e-7, "threshold")
    lr = case["opt"]["lr"]
    for k, p in params.items():
        assert_close(p, sl(case["final_params"][k], feature_axis(k)), 1e-4, 0.3 * lr, f"{name} final {k}")
    # replicated parameters stay bit-identical across ranks
    b = mod.model.decoder.b.detach().clone()
    ref = b.clone()
    shard.all_reduce_min(ref)
    assert torch.equal(b, ref), "decoder.b diverged between ranks"


def gpu_selection_ties(rank, world):
    """Sharded selections on tie-heavy inputs vs the oracle's rule (larger value, then lower GLOBAL index)."""
    from crosslayer_transcoder_amd import ops
    from crosslayer_transcoder_amd.parallel import FeatureShard
    from oracle import clt_oracle as O
    g = torch.Generator().manual_seed(3)
    for B, L, F, ks in ((40, 3, 512, (1, 7, 100)), (9, 2, 64, (3, 40)), (33, 4, 256, (16,))):
        shard = FeatureShard(F, keep_single=KEEP_SINGLE)
        for tie_heavy in (True, False):
            # 24 distinct values: thousands of exact ties at every threshold, yet < 4096 per shard and segment (the
            # documented limit of the sharded batch modes; the per-row mode has no such limit)
            x = (torch.randint(0, 24, (B, L, F), generator=g).float() - 3.0) if tie_heavy else torch.randn(B, L, F, generator=g)
            xl = shard.slice_features(x, 2).contiguous().cuda()
            mask = (torch.rand(L, F, generator=g) < 0.4)
            for k in ks:
                for cm in (None, mask):
                    xm = x if cm is None else torch.where(cm.expand_as(x), x, torch.full_like(x, -float("inf")))
                    cml = None if cm is None else shard.slice_features(cm, 1).to(torch.uint8).contiguous().cuda()
                    S = ops.select_rows_sharded(xl, k, shard, cml)
                    want = shard.slice_features(O.per_layer_topk_fwd(xm, k)[0], 2)
                    assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), want), f"rows k={k} ties={tie_heavy}"
                    thr = torch.zeros(1, device="cuda")
                    S = ops.select_lists_sharded(xl, k, ops.SELECT_BATCH, shard, thr, 0.1, cml)
                    ref, _, mk = O.batch_topk_fwd(xm, k)
                    assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), shard.slice_features(ref, 2)), \
                        f"batch k={k} ties={tie_heavy} mask={cm is not None}"
                    S = ops.select_lists_sharded(xl, k, ops.SELECT_LAYER, shard, torch.zeros(L, device="cuda"), 0.1, cml)
                    ref = O.per_layer_batch_topk_fwd(xm, k)[0]
                    assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), shard.slice_features(ref, 2)), \
                        f"layer k={k} ties={tie_heavy}"
    # far more exact ties at the k-th value than a tie block holds (4096 per shard and segment): the exhaustive form with
    # the index radix select must take over — and still pick the lowest global indices
    B, L, F = 64, 3, 512
    shard = FeatureShard(F, keep_single=KEEP_SINGLE)
    x = torch.randint(0, 3, (B, L, F), generator=g).float() + 0.5        # three distinct values: ~32k ties per value
    xl = shard.slice_features(x, 2).contiguous().cuda()
    for k in (5, 100, 300):
        S = ops.select_lists_sharded(xl, k, ops.SELECT_BATCH, shard, torch.zeros(1, device="cuda"), 0.1, None)
        # (one rank: the kernels rank ties by the index radix select inside stage 4 already, no tie lists to overflow)
        assert S.tie_radix or world == 1, "expected the tie-radix form (tie lists overflow)"
        ref = O.batch_topk_fwd(x, k)[0]
        assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), shard.slice_features(ref, 2)), f"tie radix batch k={k}"
        S = ops.select_lists_sharded(xl, k, ops.SELECT_LAYER, shard, torch.zeros(L, device="cuda"), 0.1, None)
        ref = O.per_layer_batch_topk_fwd(x, k)[0]
        assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), shard.slice_features(ref, 2)), f"tie radix layer k={k}"


def gpu_list_capacity_overflow(rank, world):
    """A shard whose share of the selection exceeds its list capacity: all ranks redo the selection with full capacity."""
    from crosslayer_transcoder_amd import ops
    from crosslayer_transcoder_amd.parallel import FeatureShard
    from oracle import clt_oracle as O
    g = torch.Generator().manual_seed(5)
    B, L, F, k = 16, 2, 64, 4
    x = torch.randn(B, L, F, generator=g)
    x[:, :, : F // max(world, 1)] += 10.0                       # every selected entry lives on rank 0
    shard = FeatureShard(F, keep_single=KEEP_SINGLE)
    xl = shard.slice_features(x, 2).contiguous().cuda()
    S = ops.select_lists_sharded(xl, k, ops.SELECT_BATCH, shard, torch.zeros(1, device="cuda"), 0.1, None, cap=8)
    assert S.cap == k * B * L and int(S.overflow) == 0
    ref = O.batch_topk_fwd(x, k)[0]
    assert torch.equal(ops.sparse_scatter_dense(S, S.csc_val).cpu(), shard.slice_features(ref, 2))


def gpu_bf16_fused_sharded(rank, world):
    """bf16 mode under sharding, including the encoder GEMM fused with the candidate filter (steps >= 1): the sharded
    run must track the unsharded run of the same full model (same per-element GEMM sums; only the all-reduce changes
    the summation order of the reconstruction)."""
    import copy
    from crosslayer_transcoder_amd import precision
    from crosslayer_transcoder_amd.parallel import FeatureShard, shard_module
    from tests.build_cases import build_module
    from tests.golden_cases import assert_close
    B, L, d, F = 512, 3, 128, 1024
    g = torch.Generator().manual_seed(11)
    batches = [torch.randn(B, 2, L, d, generator=g) for _ in range(4)]
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d)}
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    for kind, prec in (("batchtopk", "bf16-mixed"), ("perlayerbatchtopk", "bf16-mixed"), ("batchtopk", "16-mixed")):
        case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": {"kind": kind, "k": 8, "e": 0.1},
                        "loss": {"kind": "base"}}, "opt": {"lr": 1e-3}, "batches": torch.stack(batches),
                "init_params": params, "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}
        full = build_module(case, "cuda:0")
        part = build_module(case, "cuda:0")
        shard = FeatureShard(F, keep_single=KEEP_SINGLE)
        shard_module(part, shard)
        of, op = full.configure_optimizers(), part.configure_optimizers()
        fused = 0
        with precision(prec):                     # ("16-mixed": fp16 operands, the library's loss scaling on every rank alike)
            for t, b in enumerate(batches):
                lf = full.fused_train_step(b.cuda(), t, of)
                lp = part.fused_train_step(b.cuda(), t, op)
                fused += "pre_actvs" not in dict.keys(part.last_step)
                assert_close(lp, lf, 1e-5, 1e-7, f"{kind} loss step {t}")
                want = shard.slice_features(full.last_step["features"], 2)
                assert torch.equal(part.last_step["features"] > 0, want > 0), f"{kind}: active set differs at step {t}"
        assert fused == 3, "the fused encoder+filter epilogue should run on steps 1..3"
        if prec == "16-mixed":
            assert part.loss_scaler.scale == full.loss_scaler.scale == 65536.0 and part.loss_scaler.skipped_steps == 0
        assert_close(part.model.nonlinearity.threshold, full.model.nonlinearity.threshold, 1e-6, 0, "threshold")
        fp = dict(full.model.named_parameters())
        for k, p in part.model.named_parameters():
            ax = feature_axis(k)
            ref = fp[k] if ax is None else fp[k].narrow(ax, shard.f_off, shard.F_local)
            # Adam's first steps are sign-like: a bf16 rounding flip of one gathered g element moves a weight by a
            # fraction of lr (same absolute allowance as the golden-trajectory tests: 0.3 * lr)
            assert_close(p, ref, 1e-4, 0.3 * 1e-3, f"{kind} param {k}")


def gpu_exchange_forms(rank, world):
    """The two forms of the pipelined exchange — one fp32 all-reduce per layer group with the whole loss on every rank, and
    reduce-scatter + loss on this rank's token block + all-gather of the 16-bit gradient rows — give the same step: loss to
    fp64-sum rounding, the reconstruction bit for bit (the reduce-scatter adds the same partial sums), the gathered
    gradient rows bit for bit (same kernel on the same values), parameters after three Adam steps equal."""
    from crosslayer_transcoder_amd import engine, precision
    from crosslayer_transcoder_amd.parallel import FeatureShard, shard_module
    from tests.build_cases import build_module
    from tests.golden_cases import assert_close
    B, L, d, F = 512, 5, 128, 1024
    g = torch.Generator().manual_seed(23)
    batches = [torch.randn(B, 2, L, d, generator=g) for _ in range(3)]
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d)}
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": {"kind": "batchtopk", "k": 8, "e": 0.1},
                    "loss": {"kind": "base"}}, "opt": {"lr": 1e-3}, "batches": torch.stack(batches),
            "init_params": params, "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}
    for prec in ("bf16-mixed", "32"):
        runs = {}
        for split in (False, True):
            keep, engine.EXCHANGE_SPLIT = engine.EXCHANGE_SPLIT, split
            try:
                mod = build_module(case, "cuda:0")
                shard = FeatureShard(F, keep_single=KEEP_SINGLE)
                shard_module(mod, shard)
                opt = mod.configure_optimizers()
                losses, recs = [], []
                with precision(prec):
                    for t, b in enumerate(batches):
                        losses.append(float(mod.fused_train_step(b.cuda(), t, opt)))
                        recs.append(mod.last_step["recons_norm"].clone())       # (split form: a collective, on every rank)
                runs[split] = (losses, recs, {k: p.detach().clone() for k, p in mod.model.named_parameters()})
            finally:
                engine.EXCHANGE_SPLIT = keep
        (l0, r0, p0), (l1, r1, p1) = runs[False], runs[True]
        for t in range(len(batches)):
            assert abs(l0[t] - l1[t]) <= 1e-6 * abs(l0[t]), f"{prec} loss step {t}: {l0[t]} vs {l1[t]}"
            if world <= 2 and t == 0:    # (two addends: any summation order gives the same bits)
                assert torch.equal(r0[t], r1[t]), f"{prec}: reconstruction differs at step {t}"
            else:                        # (later steps: decoder.b has moved by a differently rounded gradient, see below)
                assert_close(r1[t], r0[t], 1e-4, 1e-5, f"{prec} recons step {t}")
        for k in p0:
            # decoder.b's gradient is a column sum taken per token block and added over ranks in the split form; Adam's first
            # steps are sign-like, so a rounding flip moves a weight by a fraction of lr (allowance of the golden-trajectory tests)
            assert_close(p1[k], p0[k], 1e-4, 0.3 * 1e-3, f"{prec} param {k}")


def gpu_fp16_overflow_on_one_shard(rank, world, nonlinearity={"kind": "batchtopk", "k": 8, "e": 0.1}, loss={"kind": "base"}):
    """`precision: "16-mixed"` under feature sharding: an fp16 overflow that only ONE rank's shard produces (a decoder weight
    too large for fp16 in the last rank's slice: the 16-bit shadow cast raises that rank's flag) must skip the step on EVERY
    rank — same loss scale, same skip count, untouched parameters and optimizer state, replicated decoder.b still identical —
    and the next, clean step trains on every rank."""
    from crosslayer_transcoder_amd import precision
    from crosslayer_transcoder_amd.parallel import FeatureShard, shard_module
    from tests.build_cases import build_module
    B, L, d, F = 512, 3, 128, 1024
    g = torch.Generator().manual_seed(17)
    batches = [torch.randn(B, 2, L, d, generator=g) for _ in range(2)]
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * F ** -0.5, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d)}
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * (d * L) ** -0.5
    case = {"cfg": {"decoder": "clt", "d_features": F, "nonlinearity": nonlinearity, "loss": loss},
            "opt": {"lr": 1e-3}, "batches": torch.stack(batches),
            "init_params": params, "last_active0": torch.zeros(L, F, dtype=torch.long), "max_steps": 10}
    mod = build_module(case, "cuda:0")
    shard = FeatureShard(F, keep_single=KEEP_SINGLE)
    shard_module(mod, shard)
    opt = mod.configure_optimizers()
    W2 = mod.model.decoder.weights()[2]
    keep = W2[0, 0, 0].clone()
    if rank == world - 1:
        with torch.no_grad():
            W2[0, 0, 0] = 1e6
    before = {k: p.detach().clone() for k, p in mod.model.named_parameters()}
    with precision("16-mixed"):
        mod.fused_train_step(batches[0].cuda(), 0, opt)
        assert mod.loss_scaler.scale == 32768.0 and mod.loss_scaler.skipped_steps == 1, \
            f"rank {rank}: scale {mod.loss_scaler.scale}, skipped {mod.loss_scaler.skipped_steps}"
        for k, p in mod.model.named_parameters():
            assert torch.equal(p.detach(), before[k]), f"rank {rank}: the skipped step changed {k}"
            assert int(opt.state.get(p, {}).get("step", 0)) == 0, f"rank {rank}: {k} counted a step"
        with torch.no_grad():
            W2[0, 0, 0] = keep
        loss = mod.fused_train_step(batches[1].cuda(), 1, opt)
        assert torch.isfinite(loss) and mod.loss_scaler.scale == 32768.0 and mod.loss_scaler.skipped_steps == 1
        moved = sum(int(not torch.equal(p.detach(), before[k])) for k, p in mod.model.named_parameters())
        assert moved == len(before), f"rank {rank}: only {moved} of {len(before)} parameters trained after the skipped step"
    b = mod.model.decoder.b.detach().clone()
    ref = b.clone()
    shard.all_reduce_min(ref)
    assert torch.equal(b, ref), "decoder.b diverged between ranks after a one-shard overflow"


def gpu_save_sharded(rank, world):
    """Gather-on-save: the checkpoint written from two shards loads on one GPU and reproduces the unsharded model."""
    import tempfile
    from crosslayer_transcoder_amd.model import CrossLayerTranscoder
    from crosslayer_transcoder_amd.parallel import FeatureShard, save_pretrained_sharded, shard_module
    from tests.build_cases import build_module
    from tests.conftest import load_golden
    for name in ("topk_clt_aux", "jumprelu_plt_notanh"):
        case = load_golden(name)
        batch = case["batches"][0].cuda()
        full = build_module(case, "cuda:0")
        full.model.initialize_standardizers(batch)
        with torch.no_grad():
            want = full.model(batch[:, 0])[3]
        part = build_module(case, "cuda:0")
        part.model.initialize_standardizers(batch)
        shard = FeatureShard(case["cfg"]["d_features"], keep_single=KEEP_SINGLE)
        shard_module(part, shard)
        obj = [tempfile.mkdtemp() if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0)
        save_pretrained_sharded(part, shard, obj[0], fold_standardizers=True)
        loaded = CrossLayerTranscoder.from_pretrained(obj[0]).cuda()
        assert loaded._is_folded and loaded.encoder.d_features == case["cfg"]["d_features"]
        with torch.no_grad():
            got = loaded(batch[:, 0])[3]
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max())), f"{name}: reloaded sharded checkpoint"
        sd = loaded.state_dict()
        for k, v in full.model.state_dict().items():
            if k in sd and "standardizer" not in k:
                assert sd[k].shape == v.shape, k


def sim_selection(rank, world):
    """torch restatement of the stage contract of clt_select_lists_sharded_f32 on one shard."""
    from crosslayer_transcoder_amd.parallel import FeatureShard, run_sharded_selection
    from oracle import clt_oracle as O
    NBIN, TIECAP = 2048, 4096
    SHIFT, BITS = (21, 10, 0), (11, 11, 10)

    def key_of(x):
        u = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        u = torch.where(u == 0x80000000, torch.zeros_like(u), u)
        return torch.where(u >= 0x80000000, 0xFFFFFFFF - u, u | 0x80000000)

    for trial, (B, L, F, k, ties, radix) in enumerate([(6, 3, 16, 2, False, False), (5, 2, 24, 3, True, False), (4, 1, 8, 8, False, False),
                                                       (5, 2, 24, 3, True, True), (7, 3, 16, 5, True, True)]):
        g = torch.Generator().manual_seed(100 + trial)
        x = torch.randn(B, L, F, generator=g)
        if ties:
            x = torch.randint(0, 4, (B, L, F), generator=g).float() - 0.5
        shard = FeatureShard(F, rank=rank, world=world)
        xl = x[:, :, shard.f_off:shard.f_off + shard.F_local].contiguous()
        Fl = shard.F_local
        K = k * B * L
        keys = key_of(xl).reshape(-1)
        gidx = (torch.arange(B * L).repeat_interleave(Fl) * F + shard.f_off + torch.arange(Fl).repeat(B * L))
        tw = 1 + 1 + TIECAP
        ex = {"lb": torch.zeros(1, 4, dtype=torch.int32), "hist": torch.zeros(NBIN + 1, dtype=torch.int32),
              "ties": torch.zeros(world, tw, dtype=torch.int32)}
        st = {"prefix": 0, "mask": 0, "krem": K, "eq": 0}

        def hist(p):
            sel = keys[(keys & st["mask"]) == st["prefix"]]
            ex["hist"][:NBIN] += torch.bincount((sel >> SHIFT[p]) & ((1 << BITS[p]) - 1), minlength=NBIN).to(torch.int32)

        def find(p):
            h = ex["hist"][:NBIN].to(torch.int64)
            cum = 0
            for b in range((1 << BITS[p]) - 1, -1, -1):
                if cum + int(h[b]) >= st["krem"]:
                    st["prefix"] |= b << SHIFT[p]
                    st["mask"] |= ((1 << BITS[p]) - 1) << SHIFT[p]
                    st["krem"] -= cum
                    st["eq"] = int(h[b])
                    break
                cum += int(h[b])
            ex["hist"][:NBIN] = 0

        result = {}

        def stages(lo, hi):
            for s in range(lo, hi + 1):
                if s == 0:
                    lbk = int(keys.min()) if rank == 0 else int(keys.sort().values[len(keys) // 3])   # ranks disagree on purpose
                    ex["lb"][0, 0] = lbk - (1 << 32) if lbk >= (1 << 31) else lbk
                elif s == 1:
                    lb = int(ex["lb"][0, 0]) & 0xFFFFFFFF
                    assert lb == int(torch.tensor(lb)), "lb"
                    st["cand"] = keys >= lb
                    ex["hist"][NBIN] = int(st["cand"].sum())
                    st["all_keys"] = keys
                    keys_c = keys[st["cand"]]
                    st["ck"], st["ci"] = keys_c, gidx[st["cand"]]
                    sel = st["ck"]
                    ex["hist"][:NBIN] += torch.bincount((sel >> SHIFT[0]) & 2047, minlength=NBIN).to(torch.int32)
                elif s in (2, 3):
                    if s == 2:
                        assert int(ex["hist"][NBIN]) >= K, "global candidate count"
                    find(s - 2)
                    sel = st["ck"][(st["ck"] & st["mask"]) == st["prefix"]]
                    ex["hist"][:NBIN] += torch.bincount((sel >> SHIFT[s - 1]) & ((1 << BITS[s - 1]) - 1), minlength=NBIN).to(torch.int32)
                elif radix and s >= 4:
                    # exhaustive tie resolution: radix select (4 x 8 bits, ascending) on the global index among the tied entries
                    if s == 4:
                        find(2)
                        st["tx"] = {"prefix": 0, "mask": 0, "krem": st["krem"]}
                    else:
                        h = ex["hist"][:256].to(torch.int64)
                        shift, cum = 24 - 8 * (s - 5), 0
                        for b in range(256):
                            if cum + int(h[b]) >= st["tx"]["krem"] or b == 255:
                                break
                            cum += int(h[b])
                        st["tx"]["prefix"] |= b << shift
                        st["tx"]["mask"] |= 255 << shift
                        st["tx"]["krem"] -= cum
                        ex["hist"][:NBIN] = 0
                    tied = st["ci"][st["ck"] == st["prefix"]]
                    if s < 8:
                        if st["eq"] != st["krem"]:
                            sel = tied[(tied & st["tx"]["mask"]) == st["tx"]["prefix"]]
                            ex["hist"][:256] += torch.bincount((sel >> (24 - 8 * (s - 4))) & 255, minlength=256).to(torch.int32)
                    else:
                        T = st["prefix"]
                        take = st["ck"] > T
                        eq = st["ck"] == T
                        take |= eq & ((st["ci"] <= st["tx"]["prefix"]) if st["eq"] != st["krem"] else torch.ones_like(eq))
                        result["idx"] = st["ci"][take]
                elif s == 4:
                    find(2)
                    T = st["prefix"]
                    tied = st["ci"][st["ck"] == T]
                    blk = ex["ties"][rank]
                    blk[0] = len(tied)
                    blk[2:2 + len(tied)] = tied.to(torch.int32)
                elif s == 5:
                    T = st["prefix"]
                    allt = torch.cat([ex["ties"][r, 2:2 + int(ex["ties"][r, 0])] for r in range(world)]).to(torch.int64)
                    need_rank = st["eq"] != st["krem"]
                    take = st["ck"] > T
                    eq = st["ck"] == T
                    if need_rank:
                        ranks = (allt[None, :] < st["ci"][eq][:, None]).sum(1)
                        take[eq.nonzero().flatten()] = ranks < st["krem"]
                    else:
                        take |= eq
                    result["idx"] = st["ci"][take]

        run_sharded_selection(stages, shard, 1, ex, tie_radix=radix)
        ref, ridx, _ = O.batch_topk_fwd(x, k)
        mine = set(int(i) for i in result["idx"])
        want = set(int(i) for i in ridx if shard.f_off <= int(i) % F < shard.f_off + Fl)
        assert mine == want, f"trial {trial} rank {rank}: sharded selection differs from the oracle"


def sim_rows_selection(rank, world):
    """torch restatement of the stage contract of clt_select_rows_sharded_f32 on one shard: eight 4-bit radix rounds
    over the order-preserving keys (per-row bucket counts all-reduced), then the per-row tie counts all-gathered; ties
    are taken in rank order = global feature order."""
    from crosslayer_transcoder_amd.parallel import FeatureShard, run_sharded_rows_selection
    from oracle import clt_oracle as O

    def key_of(x):
        u = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        u = torch.where(u == 0x80000000, torch.zeros_like(u), u)
        return torch.where(u >= 0x80000000, 0xFFFFFFFF - u, u | 0x80000000)

    for trial, (B, L, F, k, ties) in enumerate([(5, 3, 16, 3, False), (4, 2, 24, 5, True), (3, 1, 8, 8, False), (6, 2, 12, 1, True)]):
        g = torch.Generator().manual_seed(200 + trial)
        x = torch.randn(B, L, F, generator=g)
        if ties:
            x = torch.randint(0, 3, (B, L, F), generator=g).float() + 0.5
        shard = FeatureShard(F, rank=rank, world=world)
        Fl = shard.F_local
        xl = x[:, :, shard.f_off:shard.f_off + Fl].contiguous()
        rows = B * L
        keys = key_of(xl).reshape(rows, Fl)
        ex = {"counts": torch.zeros(rows * 16, dtype=torch.int32), "ties": torch.zeros(world, rows, dtype=torch.int32)}
        st = {"prefix": torch.zeros(rows, dtype=torch.int64), "mask": torch.zeros(rows, dtype=torch.int64),
              "krem": torch.full((rows,), k, dtype=torch.int64)}
        result = {}

        def find(rnd):
            shift = 28 - 4 * rnd
            cnt = ex["counts"].view(rows, 16).to(torch.int64)
            for r in range(rows):
                cum = 0
                for b in range(15, -1, -1):
                    if cum + int(cnt[r, b]) >= int(st["krem"][r]):
                        st["prefix"][r] |= b << shift
                        st["mask"][r] |= 0xF << shift
                        st["krem"][r] -= cum
                        break
                    cum += int(cnt[r, b])
            ex["counts"].zero_()

        def count(rnd):
            shift = 28 - 4 * rnd
            match = (keys & st["mask"][:, None]) == st["prefix"][:, None]
            digit = (keys >> shift) & 0xF
            c = torch.zeros(rows, 16, dtype=torch.int32)
            for b in range(16):
                c[:, b] = (match & (digit == b)).sum(1).to(torch.int32)
            ex["counts"] += c.reshape(-1)

        def stages(lo, hi):
            for s in range(lo, hi + 1):
                if s <= 7:
                    if s > 0:
                        find(s - 1)
                    count(s)
                elif s == 8:
                    find(7)
                    ex["ties"][rank] = (keys == st["prefix"][:, None]).sum(1).to(torch.int32)
                else:
                    T = st["prefix"][:, None]
                    before = ex["ties"][:rank].sum(0).to(torch.int64)                       # ties on lower ranks
                    eq = keys == T
                    local_rank = torch.cumsum(eq.to(torch.int64), 1) - 1
                    sel = (keys > T) | (eq & ((before[:, None] + local_rank) < st["krem"][:, None]))
                    result["feat"] = torch.where(sel, xl.reshape(rows, Fl).clamp_min(0.0), torch.zeros(())).reshape(B, L, Fl)

        run_sharded_rows_selection(stages, shard, ex)
        want = O.per_layer_topk_fwd(x, k)[0][:, :, shard.f_off:shard.f_off + Fl]
        assert torch.equal(result["feat"], want), f"rows trial {trial} rank {rank}: sharded per-row selection differs from the oracle"


def sim_row_exchange(rank, world):
    """FeatureShard.reduce_scatter_rows_async / all_gather_rows_async (the split exchange's two collectives) on host tensors."""
    from crosslayer_transcoder_amd.parallel import FeatureShard
    shard = FeatureShard(8 * world, rank=rank, world=world)
    rows, d, nl = 6 * world, 5, 3
    full = [torch.arange(rows * d, dtype=torch.float32).view(rows, d) * (l + 1) + 100.0 * r for l in range(nl) for r in [rank]]
    outs = [torch.empty(rows // world, d) for _ in range(nl)]
    shard.reduce_scatter_rows_async(outs, [f.clone() for f in full])()
    for l in range(nl):
        want = sum(torch.arange(rows * d, dtype=torch.float32).view(rows, d) * (l + 1) + 100.0 * r for r in range(world))
        assert torch.equal(outs[l], want[rank * 6:(rank + 1) * 6]), f"reduce-scatter rows, layer {l}"
    for dt in (torch.float32, torch.bfloat16):
        mine = [(o * 0.5 + rank).to(dt) for o in outs]
        got = [torch.empty(rows, d, dtype=dt) for _ in range(nl)]
        shard.all_gather_rows_async(got, mine)()
        for l in range(nl):
            want = sum(torch.arange(rows * d, dtype=torch.float32).view(rows, d) * (l + 1) + 100.0 * r for r in range(world))
            want = torch.cat([(want[r * 6:(r + 1) * 6] * 0.5 + r).to(dt) for r in range(world)])
            assert torch.equal(got[l], want), f"all-gather rows {dt}, layer {l}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sim", action="store_true")
    ap.add_argument("--cases", default="batchtopk_clt_base,relu_clt_base,dbg_jumprelu_clt,jumprelu_plt_notanh,topk_clt_aux,"
                    "batchtopk_clt_aux,perlayerbatchtopk_clt_aux,topk_plt_aux_fewdead")
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"])
    args = ap.parse_args()
    global KEEP_SINGLE
    if args.backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        KEEP_SINGLE = True
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        if args.sim:
            sim_selection(rank, world)
            sim_rows_selection(rank, world)
            sim_row_exchange(rank, world)
        else:
            if args.backend == "gloo":
                torch.cuda.set_device(0)
            gpu_selection_ties(rank, world)
            gpu_list_capacity_overflow(rank, world)
            gpu_bf16_fused_sharded(rank, world)
            gpu_exchange_forms(rank, world)
            gpu_fp16_overflow_on_one_shard(rank, world)
            # ... and under the TopK-aux module, whose list regime holds a second agreement (compact or list form of the aux pass)
            gpu_fp16_overflow_on_one_shard(rank, world, {"kind": "topk", "k": 8},
                                           {"kind": "topk", "topk_aux": {"kind": "topk", "k": 32}, "tokens_till_dead": 512,
                                            "aux_loss_scale": 0.1})
            gpu_save_sharded(rank, world)
            from crosslayer_transcoder_amd import engine
            for name in args.cases.split(","):
                for fuse in (False, True):
                    gpu_case(name, fuse)
                    if "aux" in name:       # the aux pass as a compacted dense sub-problem, forced at these small sizes
                        share, engine.COMPACT_AUX_MAX_SHARE = engine.COMPACT_AUX_MAX_SHARE, 1e9
                        try:
                            gpu_case(name, fuse)
                        finally:
                            engine.COMPACT_AUX_MAX_SHARE = share
        print(f"rank {rank}: ok", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
