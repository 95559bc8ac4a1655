#!/usr/bin/env python3
"""The dense step's contractions in the k-contiguous NT form (operands transposed beforehand, not timed) against the k-major
forms of the same kernel (operands as they lie: transposed LDS reads), one process, interleaved rounds, HIP events; TFLOP/s.

    python scripts/gemm_km_ab.py
"""
import ctypes as C
import os
import statistics

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, I, L = C.c_void_p, C.c_int, C.c_long
lib = C.CDLL(os.environ.get("CLT_HIP_LIBRARY") or os.path.join(ROOT, "crosslayer_transcoder_amd", "libclt_hip.so"))
lib.clt_gemm_bf16.restype = I
lib.clt_gemm_bf16.argtypes = [P, P, P, I, I, I, L, L, L, I, I, I, L, L, L, P, L, I, P]


def main():
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    rounds = int(os.environ.get("ROUNDS", 7))
    Bn, Lr, d, F = 4096, 12, 768, 32768
    feat = (torch.rand(Bn, Lr, F, device=dev) * 2 - 1).bfloat16()           # [B][L][F]
    g = (torch.rand(Bn, Lr, d, device=dev) * 2 - 1).bfloat16()              # [B][L][d]
    W = ((torch.rand(Lr, F, d, device=dev) * 2 - 1) * 0.01).bfloat16()      # [L][F][d]
    featT = feat.permute(1, 2, 0).contiguous()                               # [L][F][B]
    gT = g.permute(1, 2, 0).contiguous()                                     # [L][d][B]
    WT = W.transpose(1, 2).contiguous()                                      # [L][d][F]
    sq = (torch.rand(4096, 4096, device=dev) * 2 - 1).bfloat16()
    sqT = sq.t().contiguous()
    rec = torch.empty(Bn, Lr, d, device=dev)
    dW = torch.empty(Lr, F, d, device=dev)
    c4 = torch.empty(4096, 4096, device=dev)
    # name, flops, {form: (A, B, C, M, N, K, lda, ldb, ldc, tA, tB, batch, sA, sB, sC)}
    cases = [
        ("dec fwd  M4096 N768 K32768 x12", 2.0 * Bn * d * F * Lr, {
            "NT": (feat, WT, rec, Bn, d, F, Lr * F, F, Lr * d, 0, 1, Lr, F, d * F, d),
            "B k-major": (feat, W, rec, Bn, d, F, Lr * F, d, Lr * d, 0, 0, Lr, F, F * d, d)}),
        ("wgrad    M32768 N768 K4096 x12", 2.0 * Bn * d * F * Lr, {
            "NT": (featT, gT, dW, F, d, Bn, Bn, Bn, d, 0, 1, Lr, F * Bn, d * Bn, F * d),
            "A, B k-major": (feat, g, dW, F, d, Bn, Lr * F, Lr * d, d, 1, 0, Lr, F, d, F * d),
            "A k-major": (feat, gT, dW, F, d, Bn, Lr * F, Bn, d, 1, 1, Lr, F, d * Bn, F * d)}),
        ("4096^3", 2.0 * 4096 ** 3, {
            "NT": (sq, sq, c4, 4096, 4096, 4096, 4096, 4096, 4096, 0, 1, 1, 0, 0, 0),
            "B k-major": (sq, sqT, c4, 4096, 4096, 4096, 4096, 4096, 4096, 0, 0, 1, 0, 0, 0),
            "A k-major": (sqT, sq, c4, 4096, 4096, 4096, 4096, 4096, 4096, 1, 1, 1, 0, 0, 0),
            "A, B k-major": (sqT, sqT, c4, 4096, 4096, 4096, 4096, 4096, 4096, 1, 0, 1, 0, 0, 0)}),
    ]
    for name, flops, forms in cases:
        res = {k: [] for k in forms}
        ref = None
        for r in range(rounds + 1):
            for k, (A, B, Cm, M, N, K, lda, ldb, ldc, tA, tB, batch, sA, sB, sC) in forms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    rc = lib.clt_gemm_bf16(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N, K, lda, ldb, ldc, tA, tB, batch, sA, sB, sC, None, 0, 0, st)
                    assert rc == 0, rc
                e1.record()
                torch.cuda.synchronize()
                if r == 0:
                    chk = Cm.reshape(-1)[: 1 << 22].clone()
                    if ref is None:
                        ref = chk
                    else:
                        assert os.environ.get("NOCHECK") or torch.equal(chk, ref), f"{name}: form {k} differs from NT"
                    continue
                res[k].append(flops * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        print(f"{name:32s} " + "   ".join(f"{k}: med {statistics.median(v):6.1f} max {max(v):6.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
