#!/usr/bin/env python3
"""Probe: does the row stride of C (HBM / L2 channel aliasing of a tile's 256 row segments) set the store epilogue's rate?
Encoder forward shape (M 4096, K 768, N 32768, 12 layers), fp32 and 16-bit C, for several C layouts.

    python scripts/gemm_ldc_probe.py [variant]
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from crosslayer_transcoder_amd._lib import lib
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    lib.clt_gemm_nt_variant(variant, 0)
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    Bn, Lr, d, F = 4096, 12, 768, 32768
    xs = torch.randn(Bn, Lr, d, device=dev).bfloat16()
    Wt = ((torch.rand(Lr, F, d, device=dev) * 2 - 1) * F ** -0.5).bfloat16()
    flops = 2.0 * Bn * F * d * Lr

    def timed(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            assert fn() == 0
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    layouts = [("[B][L][F]", Lr * F, F)] + [(f"[B][L][F] row pad {p} el", Lr * F + p, F) for p in (32, 64, 256, 1024, 4096, 4096 + 64)] + \
              [("[L][B][F]", F, Bn * F)] + [(f"[L][B][F] row pad {p} el", F + p, Bn * (F + p)) for p in (32, 64, 256, 1024)]
    for name, ldc, sC in layouts:
        elems = (Lr - 1) * sC + (Bn - 1) * ldc + F if sC > ldc else (Bn - 1) * ldc + Lr * F
        C32 = torch.empty(elems, dtype=torch.float32, device=dev)
        r32 = [flops / timed(lambda: lib.clt_gemm_bf16(xs.data_ptr(), Wt.data_ptr(), C32.data_ptr(), Bn, F, d, Lr * d, d, ldc, 0, 1, Lr, d, d * F, sC,
                                                        None, 0, 0, st)) / 1e12 for _ in range(6)][1:]
        C16 = C32.view(torch.bfloat16)[:elems]
        r16 = [flops / timed(lambda: lib.clt_gemm_bf16_nt_out16(xs.data_ptr(), Wt.data_ptr(), C16.data_ptr(), Bn, F, d, Lr * d, d, ldc, Lr, d, d * F, sC,
                                                                 None, 0, st)) / 1e12 for _ in range(6)][1:]
        print(f"v{variant} {name:32s} fp32 C {statistics.median(r32):7.1f}   16-bit C {statistics.median(r16):7.1f} TFLOP/s", flush=True)
        del C32, C16
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
