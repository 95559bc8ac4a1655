#!/usr/bin/env python3
"""A/B of the forms of the NT-GEMM store-epilogue kernel in ONE library build (clt_gemm_nt_variant): 1 = every wave loads
and stores, 2 = loader / storer waves, 0 = chosen by shape (the default).
Interleaved rounds, random operands, HIP events; results must be bit-identical (same k order per element).

    python scripts/gemm_nt2_ab.py 1 2 0
"""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.gemm_ab import f32_key, shapes  # noqa: E402

P, I, L = C.c_void_p, C.c_int, C.c_long


def main():
    variants = sys.argv[1:] or ["1", "2:2"]
    from crosslayer_transcoder_amd._lib import lib
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    rounds = int(os.environ.get("ROUNDS", 7))

    def use(v):
        a, _, b = v.partition(":")
        lib.clt_gemm_nt_variant(int(a), int(b) if b else 0)

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rc = fn()
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def report(name, res):
        print(f"{name:40s} " + "  ".join(f"v{v}: med {statistics.median(res[v]):7.1f} max {max(res[v]):7.1f}" for v in variants), flush=True)

    # ---- encoder forward fused with the candidate filter ----------------------------------------------------------
    Bn, Lr, d, F = 4096, 12, 768, 32768
    xs = torch.randn(Bn, Lr, d, device=dev).bfloat16()
    Wt = ((torch.rand(Lr, F, d, device=dev) * 2 - 1) * F ** -0.5).bfloat16()
    bias = torch.randn(Lr, F, device=dev) * 0.01
    sigma = (d / (3.0 * F)) ** 0.5
    capS = 8 << 20
    ctl = torch.zeros(4, dtype=torch.int32, device=dev)
    ck = torch.empty(capS, dtype=torch.int32, device=dev)
    ci = torch.empty(capS, dtype=torch.int32, device=dev)
    lb = f32_key(3.2 * sigma)
    init = torch.tensor([lb - (1 << 32) if lb >= (1 << 31) else lb, 0, 0, 0], dtype=torch.int32, device=dev)
    res, ref = {v: [] for v in variants}, None
    flops = 2.0 * Bn * F * d * Lr
    for r in range(rounds + 1):
        for v in variants:
            use(v)
            ctl.copy_(init)
            dt = timed(lambda: lib.clt_gemm_bf16_select(xs.data_ptr(), Wt.data_ptr(), Bn, F, d, Lr * d, d, Lr * F, Lr, d, d * F, F, bias.data_ptr(), F,
                                                        ctl.data_ptr(), ck.data_ptr(), ci.data_ptr(), capS, 0, 2, st), 1)
            n = int(ctl[1])
            sig = (n, int(ck[:n].long().sum()), int(ci[:n].long().sum()))
            ref = ref or sig
            assert sig == ref, f"variant {v}: candidate set differs {sig} vs {ref}"
            if r:
                res[v].append(flops / dt / 1e12)
    report("enc fwd + filter epilogue (select)", res)
    # ---- encoder forward, 16-bit C -----------------------------------------------------------------------------------
    C16 = torch.empty(Bn, Lr, F, dtype=torch.bfloat16, device=dev)
    res, ref = {v: [] for v in variants}, None
    for r in range(rounds + 1):
        for v in variants:
            use(v)
            dt = timed(lambda: lib.clt_gemm_bf16_nt_out16(xs.data_ptr(), Wt.data_ptr(), C16.data_ptr(), Bn, F, d, Lr * d, d, Lr * F, Lr, d, d * F, F,
                                                          bias.data_ptr(), F, st), 3)
            if r == 0:
                chk = C16[:64].clone()
                ref = chk if ref is None else ref
                assert torch.equal(chk, ref), f"variant {v} differs (16-bit C)"
            else:
                res[v].append(flops / dt / 1e12)
    report("enc fwd, bf16 C", res)
    del C16, ck, ci
    torch.cuda.empty_cache()
    # ---- plain fp32-C shapes ---------------------------------------------------------------------------------------------
    for (name, M, N, K, batch, lda, ldb, ldc, sA, sB, sC) in shapes():
        a_elems = max((batch - 1) * sA + (M - 1) * lda + K, 1)
        b_elems = max((batch - 1) * sB + (N - 1) * ldb + K, 1)
        c_elems = max((batch - 1) * sC + (M - 1) * ldc + N, 1)
        A = (torch.rand(a_elems, device=dev) * 2 - 1).bfloat16()
        Bm = (torch.rand(b_elems, device=dev) * 2 - 1).bfloat16()
        Cm = torch.empty(c_elems, dtype=torch.float32, device=dev)
        res, ref = {v: [] for v in variants}, None
        for r in range(rounds + 1):
            for v in variants:
                use(v)
                dt = timed(lambda: lib.clt_gemm_bf16(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), M, N, K, lda, ldb, ldc, 0, 1, batch, sA, sB, sC,
                                                     None, 0, 0, st), 3)
                if r == 0:
                    chk = Cm[: min(c_elems, 1 << 22)].clone()
                    ref = chk if ref is None else ref
                    assert torch.equal(chk, ref), f"variant {v} differs from variant {variants[0]} on {name}"
                else:
                    res[v].append(2.0 * M * N * K * batch / dt / 1e12)
        report(name, res)
        del A, Bm, Cm
        torch.cuda.empty_cache()
    lib.clt_gemm_nt_variant(0, 0)


if __name__ == "__main__":
    main()
