"""Can the next step's encoder GEMM (MFMA-bound, persistent grid limited to G workgroups) hide under the weight passes
(HBM-bound) of the current step when both run on the same GPU from two streams?  Replays the recorded launches of one headline
step: the 13 row passes on stream A, the select-epilogue GEMM on stream B, alone and together.

    python scripts/overlap_probe.py            (on the GPU box; prints one line per grid limit)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


class Recorder:
    def __init__(self, lib, names):
        self._lib, self._names, self.calls, self.pre = lib, names, [], None

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in self._names:
            return fn

        def wrapped(*args):
            if self.pre is not None:
                self.pre(name)
            self.calls.append((name, args))
            return fn(*args)
        return wrapped


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse_args()
    import crosslayer_transcoder_amd as cta
    from crosslayer_transcoder_amd import _lib, ops
    cta.set_precision(args.precision)
    dev = torch.device("cuda", 0)
    mod = bench.build(args, dev)
    mod.attach_optimizers()
    A = bench.synth_target_map(args.layers, args.d_acts, dev)
    batches = [bench.synth_batch(args.batch, args.layers, args.d_acts, dev, seed=i, A=A) for i in range(4)]
    for i in range(6):
        mod.training_step(batches[i % 4], i)
    torch.cuda.synchronize()

    rec = Recorder(_lib.lib, {"clt_gemm_bf16_select"})
    snap = {}

    def pre(name):
        key = (dev.index, torch.cuda.current_stream().cuda_stream)
        snap["ws"] = ops._ws[key]
        snap["copy"] = ops._ws[key].clone()
    rec.pre = pre
    rows = []
    orig_call = ops.call

    def call_rec(name, *a):
        if name == "clt_sparse_row_pass_f32":
            rows.append(a)
        return orig_call(name, *a)
    ops.lib, ops.call = rec, call_rec
    mod.training_step(batches[2], 6)
    torch.cuda.synchronize()
    ops.lib, ops.call = _lib.lib, orig_call
    gemm = [c for c in rec.calls if c[0] == "clt_gemm_bf16_select"]
    assert len(gemm) == 1 and len(rows) == 13, (len(gemm), len(rows))
    gargs = gemm[0][1]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def run_rows(stream):
        for a in rows:
            orig_call("clt_sparse_row_pass_f32", *a[:-1], stream.cuda_stream)      # (the stream is the last argument)

    def run_gemm(stream, limit):
        _lib.lib.clt_gemm_nt_grid_limit(limit)
        rc = _lib.lib.clt_gemm_bf16_select(*gargs[:-1], stream.cuda_stream)
        _lib.lib.clt_gemm_nt_grid_limit(0)
        assert rc == 0

    def timed(fn_a, fn_b, reps=3):
        best = []
        for _ in range(reps):
            snap["ws"].copy_(snap["copy"])
            torch.cuda.synchronize()
            e0, e1, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            go = torch.cuda.Event()
            go.record(torch.cuda.current_stream())
            sa.wait_event(go); sb.wait_event(go)
            e0.record(sa)
            if fn_a:
                fn_a(sa)
            if fn_b:
                fn_b(sb)
            eb.record(sb)
            sa.wait_event(eb)
            e1.record(sa)
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1))
        return min(best), sorted(best)[len(best) // 2]

    t_rows = timed(run_rows, None)
    print(f"row passes alone (13 launches): min {t_rows[0]:.3f} ms  median {t_rows[1]:.3f}", flush=True)
    for G in (0, 128, 96, 64, 48, 32):
        t_g = timed(None, lambda s: run_gemm(s, G))
        t_both = timed(run_rows, lambda s: run_gemm(s, G))
        print(f"G={G or 256:3d}: GEMM alone {t_g[0]:.3f} ms | rows + GEMM together {t_both[0]:.3f} (median {t_both[1]:.3f}) | "
              f"serial sum at full grid = rows + GEMM(256) ; hidden = {t_rows[0] + t_g[0] - t_both[0]:.3f} of this GEMM", flush=True)


if __name__ == "__main__":
    main()
