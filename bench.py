#!/usr/bin/env python3
"""Headline benchmark: activation tokens/s of one crosslayer-transcoder training step (fwd + bwd + Adam),
GPT-2-small CLT, 32k features per layer, BatchTopK(k=16), synthetic activations resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          # starts N ranks itself (torch.distributed.run, RCCL), relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for what `roofline` and `cpu_baseline` mean.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak
N_BATCHES = 8                  # distinct resident batches cycled through the timed steps


def synth_target_map(L, d, device, seed=42):
    return torch.randn(L, d, device=device, generator=torch.Generator(device=device).manual_seed(seed))


def synth_batch(B, L, d, device, seed=42, A=None):
    """batch[:,0] ~ N(0,1); batch[:,1] = 0.5*batch[:,0]*A + 0.1*N(0,1)  (A fixed over batches)."""
    if A is None:
        A = synth_target_map(L, d, device)
    g = torch.Generator(device=device).manual_seed(1000 + seed)
    x = torch.randn(B, L, d, device=device, generator=g)
    y = 0.5 * x * A + 0.1 * torch.randn(B, L, d, device=device, generator=g)
    return torch.stack([x, y], dim=1).contiguous()


def build(args, device, world=1):
    """The training module.  Every rank constructs the same full-size module (same seed); with world > 1
    `configure_model()` — the hook Lightning calls — keeps this rank's F/world slice of every feature-axis tensor
    (crosslayer_transcoder_amd.parallel.shard_module, reference clt_lightning.py:100-119)."""
    from crosslayer_transcoder_amd.model import (BatchTopK, CrossLayerTranscoder, CrosslayerDecoder, Decoder, Encoder, JumpReLU,
                                                 PerLayerTopK, ReLU)
    from crosslayer_transcoder_amd.model.clt_lightning import (CrossLayerTranscoderModule, JumpReLUCrossLayerTranscoderModule,
                                                               TopKCrossLayerTranscoderModule)
    from crosslayer_transcoder_amd.model.standardize import DimensionwiseInputStandardizer, DimensionwiseOutputStandardizer
    L, d, F = args.layers, args.d_acts, args.features
    torch.manual_seed(42)
    with torch.device(device):
        nl = {"batchtopk": lambda: BatchTopK(k=args.k, e=0.01), "topk": lambda: PerLayerTopK(k=args.k), "relu": ReLU,
              "jumprelu": lambda: JumpReLU(theta=args.theta, bandwidth=0.01, n_layers=L, d_features=F)}[args.nonlinearity]()
        dec = (CrosslayerDecoder if args.decoder == "clt" else Decoder)(d, F, L)
        model = CrossLayerTranscoder(nl, Encoder(d, F, L), dec, DimensionwiseInputStandardizer(L, d),
                                     DimensionwiseOutputStandardizer(L, d))
        if args.nonlinearity == "jumprelu":
            mod = JumpReLUCrossLayerTranscoderModule(model=model, learning_rate=1e-4, lambda_sparsity=7e-4, c_sparsity=1.0,
                                                     use_tanh=True, pre_actv_loss=1e-6)
            mod.global_step = 50_000               # global_step / max_steps = 0.5
        elif args.aux:
            mod = TopKCrossLayerTranscoderModule(model=model, topk_aux=PerLayerTopK(k=512), learning_rate=1e-4,
                                                 tokens_till_dead=args.batch * args.dead_steps)
        else:
            mod = CrossLayerTranscoderModule(model=model, learning_rate=1e-4)
    mod.trainer.max_steps = 100_000
    mod = mod.to(device)
    if world == 1 and args.shard_rehearsal:
        from crosslayer_transcoder_amd.parallel import FeatureShard
        mod.shard = FeatureShard(args.features, keep_single=True)
    mod.configure_model()                       # world > 1: feature-axis sharding
    assert world == 1 or mod.shard.world == world
    return mod


def apply_skew(mod, args, seed=7):
    """--skew zipf: per-feature encoder-bias offsets under which the firing frequency of a layer's features follows a Zipf law
    (p_r ~ 1 / r, clamped at 0.5, summing to k per token and layer as in the uniform workload — BatchTopK keeps the total at
    k*B*L by itself), i.e. a handful of features active in >= 10 % of the tokens and a long tail of rare ones: what real
    dictionaries look like, and what the one-wave-per-row weight pass is worst at.  With standardized inputs the pre-activation
    of feature f is ~ N(0, |W_f|^2); b_f moves its exceedance probability of the uniform workload's threshold to p_f."""
    L, F, k = args.layers, args.features, args.k
    r = torch.arange(1, F + 1, dtype=torch.float64)
    c = k / (1.0 / r).sum()
    for _ in range(60):                                   # rescale until the clamped frequencies sum to k again
        q = (c / r).clamp(max=0.5)
        c = c * k / q.sum()
    q = (c / r).clamp(max=0.5)
    z = lambda p: (2.0 ** 0.5) * torch.erfinv(1.0 - 2.0 * p)  # noqa: E731  (upper-tail normal quantile)
    enc = mod.model.encoder
    sig = enc.Wt.detach().double().norm(dim=2).cpu()      # [L, F]
    g = torch.Generator().manual_seed(seed)
    b = torch.empty(L, F, dtype=torch.float64)
    for l in range(L):
        perm = torch.randperm(F, generator=g)
        ql = torch.empty(F, dtype=torch.float64)
        ql[perm] = q
        b[l] = sig[l].mean() * z(torch.tensor(k / F, dtype=torch.float64)) - sig[l] * z(ql)
    with torch.no_grad():
        enc.b.copy_(b.float().to(enc.b.device))
    return {"law": "zipf", "features_ge_10pct_per_layer": int((q >= 0.1).sum()), "max_frequency": float(q.max())}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def physical_cores():
    """Physical cores of the host (the CPU leg's thread count is torch's; hyper-threads are not cores)."""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:                                  # noqa: BLE001
        pass
    return os.cpu_count() or 1


def kernel_source_digest():
    """sha256 of the source the dominant kernel is built from: profiles/pmc_traffic.json carries the digest (and the commit)
    it was measured on, and a stale measurement is not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sparse.hip", "lists_common.h", "common.h"):
        with open(os.path.join(ROOT, "crosslayer_transcoder_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(args):
    """The CPU oracle (a restatement of the reference's fp32 dense path) on a bounded sample of the same workload
    (SURVEY §8d): full model shape, batch of `--cpu-batch` tokens, loss + gradients + Adam; one warm-up step, then the
    median of up to 3 timed steps (fewer when `--cpu-budget` seconds are used up — said in `sample`)."""
    from oracle import clt_oracle as O
    L, d, F, Bc = args.layers, args.d_acts, args.features, args.cpu_batch
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(42)
    bound_e, bound_d = 1 / F ** 0.5, 1 / (d * L) ** 0.5
    params = {"encoder.W": (torch.rand(L, d, F, generator=g) * 2 - 1) * bound_e, "encoder.b": torch.zeros(L, F),
              "decoder.b": torch.zeros(L, d)}
    for l in range(L):
        params[f"decoder.W_{l}"] = (torch.rand(l + 1, F, d, generator=g) * 2 - 1) * bound_d
    batch = synth_batch(Bc, L, d, "cpu")
    cfg = {"decoder": "clt", "nonlinearity": {"kind": "batchtopk", "k": args.k, "e": 0.01}, "loss": {"kind": "base"}}
    buffers = {"last_active": torch.zeros(L, F, dtype=torch.long), "threshold": torch.zeros(1)}
    std = O.standardizer_init(batch)
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    times, t_start = [], time.perf_counter()
    for it in range(4):                       # 1 warm-up + up to 3 timed
        t0 = time.perf_counter()
        out = O.training_step(batch, params, std, cfg, buffers)
        for k in params:
            params[k], m[k], v[k] = O.adam_step(params[k], out["grads"][k], m[k], v[k], it + 1, 1e-4)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
        if time.perf_counter() - t_start + dt > args.cpu_budget and times:
            break
    med = statistics.median(times)
    return {"value": Bc / med, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle (restatement of the reference's dense fp32 step: loss+grads+Adam), full model shape L={L} d={d} "
                      f"F={F}, batch {Bc} tokens, {threads} torch threads on {cpu_model_name()} ({physical_cores()} physical cores, "
                      f"{os.cpu_count()} logical, on the host); 1 warm-up step, median of {len(times)} timed step(s) = {med:.1f} s"}


def roofline(timing, mod, args, world):
    """Dominant kernel = row_pass_kernel (sparse weight gradient + feature-gradient dots + Adam in one streaming
    pass over a weight matrix; 13 launches per step: 12 decoder matrices W_l + the encoder).  HBM-bound.
    Algorithmic bytes of a launch over `rows` weight rows of d floats whose CSC columns hold `hits` entries:
        24*rows*d  (W, m, v read + written, fp32)  [+ 2*rows*d where the pass also writes the weight's bf16 shadow]
      + hits*(e*d + 12)  (one gathered source row g[tok,l,:] or xs[tok,l,:] of e-byte elements — bf16 in bf16 mode —,
                          token id, coefficient, dot result)
      + 4*rows  (column offsets)
    `achieved` = sum of those bytes over the instrumented launches / sum of their HIP-event durations.  The events are
    recorded in a separate short pass AFTER the timed loop (same stream, same batches)."""
    # (feature shards launch the matrices of a layer group together, clt_sparse_row_pass_group_f32: one event pair, several
    # matrices' bytes)
    single, grouped = timing.get("row_pass", []), timing.get("row_pass_group", [])
    ev = single + grouped
    lists = mod.last_step.get("lists")
    if not ev or lists is None:            # dense regime: the MFMA GEMMs dominate
        gv = timing.get("gemm", [])
        if not gv:
            return None
        flops = sum(p for _, _, p in gv)
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gv)
        ach = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "gemm_nt_bf16_kernel", "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / BF16_PEAK_TFLOPS, "traffic": None, "launches": len(gv), "avg_launch_us": ms * 1e3 / len(gv),
                "gemm_ms_per_step": ms / timing["#steps"]}
    # the lists of the LAST instrumented step price its launches; earlier steps' launches use the same per-launch figure
    col_start = lists.col_start.cpu().long()
    n = timing["#steps"]
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ev)
    last = [[p] for _, _, p in single[len(single) - len(single) // n:]] + [m for _, _, m in grouped[len(grouped) - len(grouped) // n:]]
    nbytes = 0.0
    for mats in last:
        for rows, d, col0, src_layer, fused, src_bytes, shadow in mats:
            hits = int(col_start[col0 + rows] - col_start[col0])
            per_row = (24 if fused else 4) * d + (2 * d if shadow else 0) + 4
            nbytes += rows * per_row + hits * (src_bytes * d + 12)
    nbytes *= timing["#steps"]             # (hit counts differ by < 0.1 % between steps: k*B*L entries in total either way)
    ach = nbytes / (ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # written from the rocprofv3 --pmc passes
    default_workload = (args.batch, args.features, args.layers, args.d_acts, args.k, args.nonlinearity, args.decoder, args.aux,
                        world) == (4096, 32768, 12, 768, 16, "batchtopk", "clt", False, 1)
    traffic_of = None
    if os.path.exists(tf) and default_workload:       # (the PMC passes were taken on the headline configuration)
        with open(tf) as fh:
            rec = json.load(fh)
        if rec.get("kernel_source_digest") == kernel_source_digest():      # measured on THIS kernel source, else not reported
            traffic = rec.get("row_pass_bytes_per_launch")
            traffic_of = rec.get("commit")
        else:
            traffic_of = "stale: profiles/pmc_traffic.json was measured on another version of csrc/sparse.hip (scripts/gpu_pmc.sh)"
    return {"bound": "hbm", "kernel": "row_pass_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_measured_at": traffic_of, "launches": len(ev),
            "avg_launch_us": ms * 1e3 / len(ev), "bytes_per_launch": nbytes / len(ev)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--features", type=int, default=32768)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--d-acts", type=int, default=768)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--theta", type=float, default=0.03, help="JumpReLU threshold (raise it for a low-L0 workload)")
    ap.add_argument("--aux", action="store_true", help="TopK module with the k_aux=512 dead-feature loss")
    ap.add_argument("--dead-steps", type=int, default=1,
                    help="--aux: a feature is dead after this many steps without firing (1 -> about 14 %% dead on the synthetic "
                         "data, the ~10 %% of BASELINE.md config 3; 3 -> 0.3 %%)")
    ap.add_argument("--nonlinearity", default="batchtopk", choices=["batchtopk", "topk", "relu", "jumprelu"])
    ap.add_argument("--decoder", default="clt", choices=["clt", "plt"])
    ap.add_argument("--precision", default="bf16-mixed", help="bf16-mixed | 16-mixed (fp16 operands + dynamic loss scaling) | 32")
    ap.add_argument("--api", default="training_step", choices=["training_step", "fused_train_step"],
                    help="training_step: the Lightning-facing contract (manual optimisation, optimizers()/lr_schedulers()); "
                         "fused_train_step: the same step called directly")
    ap.add_argument("--sparse", default="auto", choices=["auto", "on", "off"],
                    help="activation-list regime: auto = the module's own choice")
    ap.add_argument("--ingest", default="resident", choices=["resident", "host"],
                    help="host: every step's batch arrives as a pinned fp16 host tensor through the double-buffered "
                         "H2D prefetcher (PCIe-inclusive rate; not the headline value)")
    ap.add_argument("--shard-rehearsal", action="store_true",
                    help="one rank, but run the feature-sharded schedule (every RCCL exchange point) anyway: measures the "
                         "fixed cost the sharded step adds; not the headline value")
    ap.add_argument("--rehearsal", action="store_true",
                    help="--gpus N on a machine with fewer than N GPUs: ranks share the GPUs and exchange over gloo with host "
                         "staging.  A schedule rehearsal, NOT a scaling measurement (the line says so)")
    ap.add_argument("--select-margin", type=float, default=None,
                    help="fused encoder+select: candidate bound = margin x last step's k-th value (default: engine.SELECT_MARGIN)")
    ap.add_argument("--distinct-batches", type=int, default=N_BATCHES,
                    help="resident synthetic batches cycled through the steps (fewer only where HBM is short: config 5 on one GPU)")
    ap.add_argument("--skew", default="none", choices=["none", "zipf"],
                    help="zipf: heavy-tailed feature firing frequencies through encoder-bias offsets (apply_skew); not the headline")
    ap.add_argument("--cpu-batch", type=int, default=256)
    ap.add_argument("--cpu-budget", type=float, default=150.0,
                    help="seconds of CPU-oracle time after which no further step starts (1 warm-up + 3 timed steps of ~21 s "
                         "each at the default sample: SURVEY 8(d))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no rank environment: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay rank 0's JSON line.  Nothing here touches the GPU (counting devices does not
    initialise it), and no process that has is ever replaced by another."""
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    if ndev < args.gpus:
        if not args.rehearsal:
            print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible; pass --rehearsal to let ranks share GPUs "
                  f"over gloo (a schedule rehearsal, not a scaling measurement)", file=sys.stderr)
            return 2
        env["CLT_DIST_BACKEND"] = "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode or lines else 1


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("CLT_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    if world > ndev and backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank (see --rehearsal)")
    local = local % max(ndev, 1)           # (rehearsal: several ranks share one GPU under CLT_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    elif args.shard_rehearsal:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1, device_id=device)

    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import engine, ops
    cta.set_precision(args.precision)
    if args.select_margin is not None:
        import crosslayer_transcoder_amd.engine as _engine
        _engine.SELECT_MARGIN = args.select_margin
    mod = build(args, device, world)
    skew = apply_skew(mod, args) if args.skew == "zipf" and world == 1 else None
    opt, sched = mod.attach_optimizers()
    A = synth_target_map(args.layers, args.d_acts, device)
    batches = [synth_batch(args.batch, args.layers, args.d_acts, device, seed=i, A=A) for i in range(args.distinct_batches)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    feed = None
    if args.ingest == "host":
        import itertools
        from crosslayer_transcoder_amd.data.prefetch import PinnedPrefetcher
        host = [b.to(torch.float16).cpu().pin_memory() for b in batches]
        feed = PinnedPrefetcher(itertools.cycle(host), device)
    step = 0

    def nxt():
        return next(feed) if feed is not None else batches[step % args.distinct_batches]

    sparse = {"auto": None, "on": True, "off": False}[args.sparse]
    if sparse is not None:
        mod.sparse_regime = sparse

    def one_step():
        nonlocal step
        if args.api == "training_step":
            mod.training_step(nxt(), step)
        else:
            mod.fused_train_step(nxt(), step, opt, sched)
        step += 1

    def density():
        """Active share of the last step's lists (a thresholding sparsifier's active set shrinks while it trains: the timed
        window is described by the value before its first and after its last step)."""
        S = (getattr(mod, "last_step", None) or {}).get("lists")
        return None if S is None else round(float(S.total) / (args.batch * args.layers * (args.features // max(world, 1))), 6)

    for _ in range(args.warmup):
        one_step()
    barrier()
    density_first = density()
    engine.STATS["select_fallbacks"] = engine.STATS["fused_select_steps"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    density_last = density()
    stats = dict(engine.STATS)
    if dist is not None:
        tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms = dt / args.steps * 1e3
    tokens_per_s = args.batch * args.steps / dt      # feature-sharded: every rank sees the same tokens

    # roofline pass: HIP events around the dominant kernel's launches, in a short pass of its own after the timed loop
    n_inst = 2
    mod.log_metrics_every = 0
    ops.TIMING = {"#steps": n_inst}
    for _ in range(n_inst):
        one_step()
    barrier()
    timing, ops.TIMING = ops.TIMING, None

    L, d, F = args.layers, args.d_acts, args.features
    par = f"feature-shard x{world}" if world > 1 else ("single GPU, sharded schedule rehearsal" if args.shard_rehearsal else "single GPU")
    if world > 1 and backend != "nccl":
        par += f" REHEARSAL ({world} ranks on {ndev} GPU(s), gloo + host staging: not a scaling measurement)"
    line = {
        "metric": "activation tokens/sec (fwd+bwd+Adam), GPT-2 CLT 32k feats, 1/2/4/8 MI355X",
        "value": tokens_per_s, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp16": "f16", "32": "f32"}[cta.get_precision()], "data": "synthetic",
        "rccl_ranks": world if (world > 1 and backend == "nccl") else (1 if args.shard_rehearsal else 0),
        "config": {"workload": f"gpt2-small {args.decoder.upper()} {args.nonlinearity}"
                               + (f"(k={args.k})" if args.nonlinearity in ("batchtopk", "topk") else "")
                               + f" L={L} d={d} F={F}/layer B={args.batch}"
                               + (" +aux(k=512)" if args.aux else ""),
                   "parallelism": par, "ingest": args.ingest, "api": args.api, "distinct_batches": args.distinct_batches,
                   "regime": "activation lists" if mod.last_step.get("lists") is not None else "dense GEMM",
                   "fused_select_steps": stats.get("fused_select_steps", 0), "select_fallbacks": stats.get("select_fallbacks", 0),
                   "select_drift": (lambda t: None if t is None else [round(float(v), 5) for v in t[: t.numel() // 2].tolist()])(
                       getattr(mod.model.nonlinearity, "_kth_drift", None)),
                   "list_density": density_last, "list_density_first": density_first,
                   "loss": float(mod.last_step["loss"])},
    }
    ms_ = torch.cuda.memory_stats(device)
    line["config"]["hbm"] = {"peak_allocated_gb": round(ms_.get("allocated_bytes.all.peak", 0) / 2 ** 30, 2),
                             "peak_reserved_gb": round(ms_.get("reserved_bytes.all.peak", 0) / 2 ** 30, 2),
                             "alloc_retries": int(ms_.get("num_alloc_retries", 0))}
    if skew is not None:
        S_ = mod.last_step.get("lists")
        if S_ is not None:
            lens = (S_.col_start[1:] - S_.col_start[:-1]).long()
            skew.update(longest_column=int(lens.max()), hot_columns=int((lens > ops.HOT_MIN_LEN).sum()) if ops.HOT_MIN_LEN > 0 else None,
                        entries_in_hot_columns=(float(lens[lens > ops.HOT_MIN_LEN].sum()) / max(int(lens.sum()), 1)) if ops.HOT_MIN_LEN > 0 else None)
        line["config"]["skew"] = skew
    if rank == 0:
        line["roofline"] = roofline(timing, mod, args, world)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
