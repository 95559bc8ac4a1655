#!/usr/bin/env python3
"""Probe: does the row stride of the A operand set the rate of the long-K contractions?  Per-layer decoder forward shape
(M 4096 tokens, N 768, K 32768 features, 12 layers): A = the 16-bit feature operand as [B][L][F] (rows 786 KB apart: what the
step uses, the reference's tensor layout) against [L][B][F] (rows 64 KB apart) and padded forms.

    python scripts/gemm_lda_probe.py
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from crosslayer_transcoder_amd._lib import lib
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    Bn, Lr, d, F = 4096, 12, 768, 32768
    WT = ((torch.rand(Lr, d, F, device=dev) * 2 - 1) * F ** -0.5).bfloat16()
    C = torch.empty(Bn, Lr, d, dtype=torch.float32, device=dev)
    flops = 2.0 * Bn * F * d * Lr

    def timed(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            assert fn() == 0
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    layouts = [("[B][L][F]", Lr * F, F)] + [(f"[B][L][F] row pad {p} el", Lr * F + p, F) for p in (64, 1024)] + \
              [("[L][B][F]", F, Bn * F)] + [(f"[L][B][F] row pad {p} el", F + p, Bn * (F + p)) for p in (64, 1024)]
    for name, lda, sA in layouts:
        elems = (Lr - 1) * sA + (Bn - 1) * lda + F if sA > lda else (Bn - 1) * lda + Lr * F
        A = (torch.rand(elems, device=dev) - 0.5).bfloat16()
        r = [flops / timed(lambda: lib.clt_gemm_bf16(A.data_ptr(), WT.data_ptr(), C.data_ptr(), Bn, d, F, lda, F, Lr * d, 0, 1, Lr, sA, d * F, d,
                                                     None, 0, 0, st)) / 1e12 for _ in range(6)][1:]
        print(f"decoder forward, A as {name:28s} {statistics.median(r):7.1f} TFLOP/s", flush=True)
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
