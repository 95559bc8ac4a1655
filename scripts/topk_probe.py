"""Time clt_topk_rows_f32 on a [B*L, F] tensor:  topk_probe.py [F] [k] [dense]   (default: list output only)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from crosslayer_transcoder_amd._lib import lib
rows, F, k = 4096 * 12, int(sys.argv[1]) if len(sys.argv) > 1 else 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 16
x = torch.randn(rows, F, device="cuda")
idx = torch.empty(rows, k, dtype=torch.int32, device="cuda"); val = torch.empty(rows, k, device="cuda")
dense = len(sys.argv) > 3 and sys.argv[3] == 'dense'
feat = torch.empty_like(x) if dense else None
fp, ip, vp = (feat.data_ptr(), None, None) if dense else (None, idx.data_ptr(), val.data_ptr())
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    lib.clt_topk_rows_f32(x.data_ptr(), fp, rows, F, k, ip, vp, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(10):
    lib.clt_topk_rows_f32(x.data_ptr(), fp, rows, F, k, ip, vp, st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"F={F} k={k} {'dense' if dense else 'lists'} wave={0 if os.environ.get('CLT_TOPK_NO_WAVE') else 1}: {ms:.3f} ms  ({rows * F * 4 / ms / 1e6:.0f} GB/s)")
