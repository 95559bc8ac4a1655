"""Where weight_prep_kernel's time goes: the entry point with each output switched off (PLT decoder: 12 x 32768 x 768)."""
import torch
from crosslayer_transcoder_amd import ops
from crosslayer_transcoder_amd._lib import lib

nb, rows, n = 12, 32768, 768
W = torch.randn(nb, rows, n, device="cuda") * 0.02
op = torch.empty(nb, rows, n, dtype=torch.bfloat16, device="cuda")
opT = torch.empty(nb, n, rows, dtype=torch.bfloat16, device="cuda")
sq = torch.empty(nb, rows, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(a, b, c, label, bytes_per):
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.clt_weight_prep_bf16(W.data_ptr(), nb, rows, n, a.data_ptr() if a is not None else None, b.data_ptr() if b is not None else None,
                                      c.data_ptr() if c is not None else None, 0, st)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts[1:])[2]
    print(f"{label}: {t * 1e3:.0f} us  {bytes_per * W.numel() / t / 1e6:.0f} GB/s")


run(op, opT, sq, "all (op, opT, sq)", 8)
run(op, None, sq, "op + sq", 6)
run(None, opT, sq, "opT + sq", 6)
run(None, None, sq, "sq only", 4)
run(op, opT, None, "op + opT", 8)
