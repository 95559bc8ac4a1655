"""A/B of clt_adam_f32 variants (ab/libclt_hip_ADAM_VARIANT_<v>.so, scripts/build_variants.sh): GB/s of the 28 B/parameter
update on a PLT-sized weight tensor (302 M parameters), interleaved rounds in one process."""
import ctypes
import sys

import torch

N = 12 * 32768 * 768
variants = sys.argv[1:] or ["0", "1", "2", "3", "4", "5", "8", "9"]
libs = {}
for v in variants:
    lib = ctypes.CDLL(f"ab/libclt_hip_ADAM_VARIANT_{v}.so")
    lib.clt_adam_f32.restype = ctypes.c_int
    lib.clt_adam_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long] + [ctypes.c_double] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    libs[v] = lib
p, g, m, v_ = (torch.randn(N, device="cuda") * 0.01 for _ in range(4))
v_.abs_()
st = torch.cuda.current_stream().cuda_stream
res = {v: [] for v in variants}
for rnd in range(6):
    for v, lib in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(3):
            rc = lib.clt_adam_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v_.data_ptr(), N, 1e-4, 0.9, 0.999, 1e-8, 0.0, 0, rnd * 3 + it + 1, None, None, None, 0, st)
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[v].append(e0.elapsed_time(e1) / 3)
for v in variants:
    t = sorted(res[v])[len(res[v]) // 2]
    print(f"variant {v}: {t * 1e3:.0f} us  {28 * N / t / 1e6:.0f} GB/s")
