"""Time clt_topk_rows_f32 (list output only) on a [B*L, F] tensor:  topk_probe.py [F] [k]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from crosslayer_transcoder_amd._lib import lib
rows, F, k = 4096 * 12, int(sys.argv[1]) if len(sys.argv) > 1 else 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 16
x = torch.randn(rows, F, device="cuda")
idx = torch.empty(rows, k, dtype=torch.int32, device="cuda"); val = torch.empty(rows, k, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    lib.clt_topk_rows_f32(x.data_ptr(), None, rows, F, k, idx.data_ptr(), val.data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(10):
    lib.clt_topk_rows_f32(x.data_ptr(), None, rows, F, k, idx.data_ptr(), val.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"F={F} k={k}: {ms:.3f} ms  ({rows * F * 4 / ms / 1e6:.0f} GB/s)")
