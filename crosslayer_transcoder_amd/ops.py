"""Tensor-level wrappers over the C ABI (include/clt_hip.h).

PyTorch is used here only as plumbing: device memory (caching allocator), the current HIP stream and
tensor metadata.  Every arithmetic operation below is one or more calls into libclt_hip.so; there is no
torch / CPU fallback — a non-CUDA tensor raises.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional, Tuple

import torch

from ._lib import CltError, call, lib
from .precision import ab_switch, operand_dtype as op16

_ws = {}
_DT16 = (torch.bfloat16, torch.float16)

# ---- 16-bit operand format + loss scaling: process-wide library state (clt_set_loss_scaling) -------------------------------
#: gradients are formed multiplied by this factor (the loss kernels' coefficients carry it) and unscaled inside the Adam
#: kernels; 1.0 outside fp16 mode.  Set through set_loss_scaling only.
GRAD_SCALE = 1.0
_lib_state = {}            # per device: the library keeps this state in each device's constant memory
_STATE0 = (False, 1.0, 0)
_flags = {}


def inf_flag(device) -> torch.Tensor:
    """The device int32 that fp32 -> fp16 conversions raise on overflow (one per device, owned here)."""
    key = torch.device(device).index or 0
    if key not in _flags:
        _flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _flags[key]


def set_loss_scaling(fp16: bool, scale: float = 1.0, flag: Optional[torch.Tensor] = None):
    """Bring the library state in line with (operand format, loss scale, overflow flag); a C call (which synchronises the
    device) only when something changed."""
    global GRAD_SCALE
    dev = torch.cuda.current_device()
    if fp16 and flag is None:
        flag = inf_flag(dev)
    if flag is not None and flag.device.index != dev:
        raise RuntimeError("loss-scaling state applies to the current device: run the fp16 step with torch.cuda.set_device(module's device)")
    want = (bool(fp16), float(scale), flag.data_ptr() if flag is not None else 0)
    if want != _lib_state.get(dev, _STATE0):
        call("clt_set_loss_scaling", int(want[0]), want[1], want[2] or None)
        _lib_state[dev] = want
    GRAD_SCALE = want[1]


def sync_operand_format():
    """Called by the host paths that launch 16-bit kernels outside the fused step (module forward, inference): operand format
    from the active precision, no loss scaling of the library's own (an outer GradScaler scales and unscales)."""
    from .precision import get_precision
    fp16 = get_precision() == "fp16"
    state = _lib_state.get(torch.cuda.current_device(), _STATE0)
    # (the scale goes back to 1 as well: an optimizer stepping outside the fused step — fused_step=False under an outer
    # GradScaler — must not unscale a second time)
    if fp16 != state[0] or state[1] != 1.0:
        set_loss_scaling(fp16, 1.0)

#: kernel timing hook (bench.py's roofline pass, run AFTER its timed loop): when a dict, selected launches are bracketed
#: by HIP events recorded on the launch stream (= torch's current stream) and appended as name -> [(start, end,
#: payload), ...].  No sync.
TIMING = None


def _timed_call(tag, payload, name, *args):
    if TIMING is None:
        return call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(name, *args)
    e1.record()
    TIMING.setdefault(tag, []).append((e0, e1, payload))


def _timed_raw(tag, payload, name, *args) -> int:
    """Like _timed_call for entry points whose return code 1 means "shape does not qualify": returns the code."""
    fn = getattr(lib, name)
    if TIMING is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    if rc == 0:
        TIMING.setdefault(tag, []).append((e0, e1, payload))
    return rc


def _stream():
    return torch.cuda.current_stream().cuda_stream


#: K-segmented feature-gradient GEMMs of a cross-layer decoder over two streams when their tile count leaves a short last
#: round (see clt_decoder_dgrad_kseg); 1 = one stream
KSEG_DGRAD_STREAMS = ab_switch("kseg_dgrad_streams", 2)
_side_streams = {}


def _side_stream(dev):
    key = torch.device(dev).index or 0
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def _chk(t: torch.Tensor, dtype=torch.float32, contiguous=True):
    if not t.is_cuda:
        raise RuntimeError("crosslayer_transcoder_amd ops need CUDA (HIP) tensors; there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return t


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-device scratch buffer, grown on demand (stream-ordered reuse on the current stream)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


#: (CLT_AB=table_cache=0: every table uploaded from pageable memory when it is needed, for A/B runs)
TABLE_CACHE = bool(ab_switch("table_cache", 1))
_TABLES = {}          # (device index, values) -> (device int64 tensor, pinned source); insertion-ordered, bounded
_TABLES_MAX = 64
_TABLES_OLD = []


def _device_table(values, device) -> torch.Tensor:
    """Device int64 array of `values` (pointer / offset tables of the grouped and K-segmented GEMMs, of the gather forward).
    The tables of a training step repeat from step to step (weights, operand copies and — through the caching allocator —
    even the per-step buffers keep their addresses), and an upload from pageable memory makes the host wait for everything
    queued before it: 0.4 ms of idle GPU per dense CLT step in front of the feature-gradient GEMMs.  A table is a pure function
    of its values, so a cached one is valid whenever the same values are asked for again; a new one goes up from pinned
    memory without blocking."""
    device = torch.device(device)
    if not TABLE_CACHE:
        return torch.tensor(values, dtype=torch.int64).to(device)
    key = (device.index or 0, tuple(values))
    hit = _TABLES.get(key)
    if hit is None:
        src = torch.tensor(values, dtype=torch.int64).pin_memory()
        hit = (src.to(device, non_blocking=True), src)
        if len(_TABLES) >= _TABLES_MAX:
            # (an evicted table may still be read by a queued kernel of any stream: it stays alive for another _TABLES_MAX
            # evictions — whole steps later — instead of being freed, or waited for, now)
            _TABLES_OLD.append(_TABLES.pop(next(iter(_TABLES))))
            if len(_TABLES_OLD) > _TABLES_MAX:
                _TABLES_OLD.pop(0)
        _TABLES[key] = hit
    return hit[0]


def _rows_view(x: torch.Tensor) -> Tuple[int, int, int]:
    """x: [B, ...] with contiguous inner dims (e.g. batch[:, 0]) -> (rows, n, row_stride)."""
    B = x.shape[0]
    n = x[0].numel()
    if not x[0].is_contiguous():
        raise ValueError("inner dimensions must be contiguous")
    return B, n, (x.stride(0) if B > 1 else n)


# ----------------------------------------------------------------------------------------------- K1
def column_stats(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(x, contiguous=False)
    rows, n, rs = _rows_view(x)
    mean = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
    std = torch.empty_like(mean)
    call("clt_column_stats_f32", x.data_ptr(), rs, rows, n, mean.data_ptr(), std.data_ptr(), _stream())
    return mean, std


def standardize(x, mean, std, want_f32=True, want_bf16=False):
    _chk(x, contiguous=False); _chk(mean); _chk(std)
    rows, n, rs = _rows_view(x)
    o32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    o16 = torch.empty(x.shape, dtype=op16(), device=x.device) if want_bf16 else None
    call("clt_standardize_f32", x.data_ptr(), rs, mean.data_ptr(), std.data_ptr(), rows, n,
         o32.data_ptr() if want_f32 else None, o16.data_ptr() if want_bf16 else None, _stream())
    return o32, o16


def destandardize(r, mean, std):
    _chk(r); _chk(mean); _chk(std)
    out = torch.empty_like(r)
    call("clt_destandardize_f32", r.data_ptr(), mean.data_ptr(), std.data_ptr(), r.shape[0], r[0].numel(),
         out.data_ptr(), _stream())
    return out


# ------------------------------------------------------------------------------------- dense GEMMs
#: flag bits of the bf16 GEMM entry points' `accumulate` argument (include/clt_hip.h)
GEMM_ACCUMULATE, GEMM_ROUND_ACC = 1, 2


def _gemm(fn, A, aoff, B, boff, Cm, coff, M, N, K, lda, ldb, ldc, tA, tB, batch=1, sA=0, sB=0, sC=0,
          bias=None, bias_off=0, sBias=0, accumulate=False, round_acc=False):
    es = A.element_size()
    flags = int(bool(accumulate)) | (GEMM_ROUND_ACC if (round_acc and es == 2) else 0)
    _timed_call("gemm", 2.0 * M * N * K * batch, fn, A.data_ptr() + aoff * es, B.data_ptr() + boff * es, Cm.data_ptr() + coff * 4, M, N, K, lda, ldb, ldc,
         int(tA), int(tB), batch, sA, sB, sC, (bias.data_ptr() + bias_off * 4) if bias is not None else None, sBias,
         flags, _stream())


#: 16-bit modes: the encoder's product is rounded to the operand format before the fp32 bias is added — the values the
#: reference's sparsifier sees under torch.autocast (clt.py:157-165: the einsum returns a 16-bit tensor, `+ self.b.to(...)`
#: promotes it back to fp32); pinned by tests/golden/ac_*_{bf16,fp16}.pt.  False: the fp32 accumulator values (A/B runs).
AUTOCAST_PRE = True


def _fn_for(t):
    return "clt_gemm_bf16" if t.dtype in _DT16 else "clt_gemm_f32"


def encoder_fwd(xs: torch.Tensor, Wt: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """pre[b,l,:] = xs[b,l,:] @ W[l] + b[l]   (clt.py:157-165).  `Wt` is the feature-major storage of the
    encoder weight, [L,F,d] (= W.permute(0,2,1)), so both operands are k-contiguous; fp32 or 16-bit operands (then, with
    AUTOCAST_PRE, the product is rounded to the operand format before the bias is added, as under the reference's autocast)."""
    Bn, L, d = xs.shape
    F = Wt.shape[1]
    _chk(xs, xs.dtype); _chk(Wt, xs.dtype)
    pre = torch.empty(Bn, L, F, dtype=torch.float32, device=xs.device)
    _gemm(_fn_for(xs), xs, 0, Wt, 0, pre, 0, Bn, F, d, L * d, d, L * F, False, True, batch=L, sA=d, sB=d * F, sC=F,
          bias=b, sBias=F, round_acc=AUTOCAST_PRE)
    return pre


def encoder_fwd_out16(xs: torch.Tensor, Wt: torch.Tensor, b: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """encoder_fwd with the result leaving in the 16-bit operand format [B, L, F] (dense 16-bit regime).  With b = None this is
    exactly the reference's einsum result under autocast; the consumers (jumprelu_fwd_bf16, jumprelu_loss_bwd_bf16,
    sparse_build_threshold) then add the fp32 bias in fp32, in autocast's order.  None when the shape does not qualify for
    the fast kernel."""
    Bn, L, d = xs.shape
    F = Wt.shape[1]
    _chk(xs, op16()); _chk(Wt, op16())
    if F % 8 or d % 8 or Bn * F < 256 * 256:
        return None
    pre = torch.empty(Bn, L, F, dtype=op16(), device=xs.device)
    rc = _timed_raw("gemm", 2.0 * Bn * F * d * L, "clt_gemm_bf16_nt_out16", xs.data_ptr(), Wt.data_ptr(), pre.data_ptr(), Bn, F, d, L * d, d,
                    L * F, L, d, d * F, F, b.data_ptr() if b is not None else None, F, _stream())
    if rc == 1:
        return None
    if rc != 0:
        raise CltError("clt_gemm_bf16_nt_out16", rc)
    return pre


def encoder_wgrad(xs: torch.Tensor, dpre: torch.Tensor) -> torch.Tensor:
    """dWt[l] = dpre[:,l]^T @ xs[:,l]  -> [L,F,d] (feature-major, like the weight storage)."""
    Bn, L, d = xs.shape
    F = dpre.shape[2]
    _chk(xs, xs.dtype); _chk(dpre, xs.dtype)
    dWt = torch.empty(L, F, d, dtype=torch.float32, device=xs.device)
    _gemm(_fn_for(xs), dpre, 0, xs, 0, dWt, 0, F, d, Bn, L * F, L * d, d, True, False, batch=L, sA=F, sB=d, sC=d * F)
    return dWt


def encoder_dgrad(dpre: torch.Tensor, Wt: torch.Tensor) -> torch.Tensor:
    """dx[:,l] = dpre[:,l] @ W[l]^T (only when the caller differentiates w.r.t. the activations)."""
    Bn, L, F = dpre.shape
    d = Wt.shape[2]
    dx = torch.empty(Bn, L, d, dtype=torch.float32, device=dpre.device)
    _gemm(_fn_for(dpre), _chk(dpre, dpre.dtype), 0, _chk(Wt, dpre.dtype), 0, dx, 0, Bn, d, F, L * F, d, L * d, False,
          False, batch=L, sA=F, sB=d * F, sC=d)
    return dx


def clt_decoder_fwd(feat: torch.Tensor, Ws: List[torch.Tensor], b: Optional[torch.Tensor]) -> torch.Tensor:
    """recons[:,l] = feat[:, :l+1].flatten(1) @ W_l.flatten(0,1) + b[l]   (clt.py:334-343)."""
    Bn, L, F = feat.shape
    d = Ws[0].shape[2]
    _chk(feat, feat.dtype)
    r = torch.empty(Bn, L, d, dtype=torch.float32, device=feat.device)
    for l in range(L):
        _gemm(_fn_for(feat), feat, 0, _chk(Ws[l], feat.dtype), 0, r, l * d, Bn, d, (l + 1) * F, L * F, d, L * d,
              False, False, bias=b, bias_off=l * d)
    return r


def clt_decoder_dgrad(g: torch.Tensor, Ws: List[torch.Tensor]) -> torch.Tensor:
    """dfeat[:, :l+1] += g[:,l] @ W_l.flatten(0,1)^T, l = L-1 first (covers every column, no zero fill)."""
    Bn, L, d = g.shape
    F = Ws[0].shape[1]
    _chk(g, g.dtype)
    df = torch.empty(Bn, L, F, dtype=torch.float32, device=g.device)
    for l in range(L - 1, -1, -1):
        _gemm(_fn_for(g), g, l * d, _chk(Ws[l], g.dtype), 0, df, 0, Bn, (l + 1) * F, d, L * d, d, L * F, False, True,
              accumulate=(l != L - 1))
    return df


def plt_decoder_dgrad_out16(g: torch.Tensor, W: torch.Tensor) -> Optional[torch.Tensor]:
    """plt_decoder_dgrad with the feature gradient leaving as bf16 [B, L, F] (bf16 operands).  None: shape does not qualify."""
    Bn, L, d = g.shape
    F = W.shape[1]
    _chk(g, op16()); _chk(W, op16())
    if F % 8 or d % 8 or Bn * F < 256 * 256:
        return None
    df = torch.empty(Bn, L, F, dtype=op16(), device=g.device)
    rc = _timed_raw("gemm", 2.0 * Bn * F * d * L, "clt_gemm_bf16_nt_out16", g.data_ptr(), W.data_ptr(), df.data_ptr(), Bn, F, d, L * d, d,
                    L * F, L, d, F * d, F, None, 0, _stream())
    if rc == 1:
        return None
    if rc != 0:
        raise CltError("clt_gemm_bf16_nt_out16", rc)
    return df


def clt_decoder_dgrad_kseg(g: torch.Tensor, Ws: List[torch.Tensor], out_bf16: bool = False) -> Optional[torch.Tensor]:
    """bf16 fast form of clt_decoder_dgrad: per source layer l' ONE GEMM contracting over (target layer l >= l', d_acts)
    with the B operand K-segmented over the per-target-layer weight tensors — the result is written once instead of
    accumulated over L - l' passes.  g: bf16 [B, L, d]; Ws[l]: bf16 [(l+1), F, d].  out_bf16: the result as bf16.
    None when the shape does not qualify."""
    Bn, L, d = g.shape
    F = Ws[0].shape[1]
    _chk(g, op16())
    if d % 64 or F % 4 or Bn * F < 256 * 256 or (out_bf16 and F % 8):
        return None
    ptrs = []
    for lp in range(L):
        ptrs += [_chk(Ws[l], op16()).data_ptr() + 2 * lp * F * d for l in range(lp, L)]
    table = _device_table(ptrs, g.device)
    df = torch.empty(Bn, L, F, dtype=op16() if out_bf16 else torch.float32, device=g.device)
    # One persistent block per CU walks 256 x 256 tiles: a launch with a few tiles more than a multiple of 256 (the compacted
    # TopK-aux pass: 16 x 18 = 288) spends a whole extra round on them.  The L launches write disjoint columns, so they
    # alternate over two streams and one launch's short last round runs beside the next launch's first.
    tiles = -(-Bn // 256) * -(-F // 256)
    two = (KSEG_DGRAD_STREAMS > 1 and TIMING is None and L > 1 and 0 < tiles % 256 <= 128)      # (also: at most half the CUs busy)
    if two:
        main, side = torch.cuda.current_stream(g.device), _side_stream(g.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
    off = 0
    rc_bad = None
    for lp in range(L):
        nseg = L - lp
        with torch.cuda.stream(side) if (two and lp % 2 == 1) else contextlib.nullcontext():
            if out_bf16:
                rc = _timed_raw("gemm", 2.0 * Bn * F * d * nseg, "clt_gemm_bf16_nt_kseg_out16", g.data_ptr() + 2 * lp * d,
                                table.data_ptr() + 8 * off, df.data_ptr() + 2 * lp * F, Bn, F, nseg, d, L * d, d, L * F, _stream())
            else:
                rc = _timed_raw("gemm", 2.0 * Bn * F * d * nseg, "clt_gemm_bf16_nt_kseg", g.data_ptr() + 2 * lp * d, table.data_ptr() + 8 * off,
                                df.data_ptr() + 4 * lp * F, Bn, F, nseg, d, L * d, d, L * F, None, 0, _stream())
        if rc != 0:
            rc_bad = rc
            break
        off += nseg
    if two:                                    # (also on the error paths: nothing of this call may outlive it on the side stream)
        ev = torch.cuda.Event()
        ev.record(side)
        main.wait_event(ev)
    if rc_bad == 1:
        return None
    if rc_bad is not None:
        raise CltError("clt_gemm_bf16_nt_kseg", rc_bad)
    return df


def clt_decoder_wgrad(feat: torch.Tensor, g: torch.Tensor, l: int, out: Optional[torch.Tensor] = None,
                      accumulate=False) -> torch.Tensor:
    """dW_l = feat[:, :l+1].flatten(1)^T @ g[:,l]   -> [(l+1), F, d]."""
    Bn, L, F = feat.shape
    d = g.shape[2]
    if out is None:
        out = torch.empty(l + 1, F, d, dtype=torch.float32, device=feat.device)
    _gemm(_fn_for(feat), feat, 0, g, l * d, out, 0, (l + 1) * F, d, Bn, L * F, L * d, d, True, False,
          accumulate=accumulate)
    return out


def plt_decoder_fwd(feat, W, b):
    Bn, L, F = feat.shape
    d = W.shape[2]
    r = torch.empty(Bn, L, d, dtype=torch.float32, device=feat.device)
    _gemm(_fn_for(feat), _chk(feat, feat.dtype), 0, _chk(W, feat.dtype), 0, r, 0, Bn, d, F, L * F, d, L * d, False,
          False, batch=L, sA=F, sB=F * d, sC=d, bias=b, sBias=d)
    return r


def plt_decoder_dgrad(g, W):
    Bn, L, d = g.shape
    F = W.shape[1]
    df = torch.empty(Bn, L, F, dtype=torch.float32, device=g.device)
    _gemm(_fn_for(g), _chk(g, g.dtype), 0, _chk(W, g.dtype), 0, df, 0, Bn, F, d, L * d, d, L * F, False, True, batch=L,
          sA=d, sB=F * d, sC=F)
    return df


def plt_decoder_wgrad(feat, g, out=None, accumulate=False):
    Bn, L, F = feat.shape
    d = g.shape[2]
    if out is None:
        out = torch.empty(L, F, d, dtype=torch.float32, device=feat.device)
    _gemm(_fn_for(feat), feat, 0, g, 0, out, 0, F, d, Bn, L * F, L * d, d, True, False, batch=L, sA=F, sB=d, sC=F * d,
          accumulate=accumulate)
    return out


# ---- dense decoder / encoder-wgrad contractions with every operand k-contiguous (bf16 fast path) -------------
# The LDS-DMA MFMA kernel wants both operands k-contiguous ("NT").  Contractions whose natural layout is not
# (decoder forward: K = features, weights stored [F][d]; weight gradients: K = tokens) get a transposed bf16
# copy of one operand (cast_transpose_bf16) — the cast to bf16 is needed anyway.
def transpose_cast_2d(x: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, ...] -> bf16 [prod(...), rows] (e.g. features [B, L, F] -> [L, F, B] flattened)."""
    rows = x.shape[0]
    return cast_transpose_bf16(x.reshape(1, rows, -1))[0]


#: the last round of a persistent batched GEMM with a long contraction, when it is short: its problems go through a second,
#: grouped launch with the contraction cut in pieces (see _tail_split); CLT_AB=gemm_tail_split=0: one launch
GEMM_TAIL_SPLIT = ab_switch("gemm_tail_split", 1)             # (2: adopt a split the cost model prices at 5 % instead of 10 %, for A/B runs)
_cus = {}


def _persistent_blocks(device) -> int:
    """Blocks of the persistent NT GEMM's grid: one per CU, a multiple of 8 (gemm_nt_bf16.hip: persistent_grid)."""
    key = torch.device(device).index or 0
    if key not in _cus:
        n = torch.cuda.get_device_properties(device).multi_processor_count
        _cus[key] = n // 8 * 8 if n >= 8 else 8
    return _cus[key]


def _tail_split(items: int, tiles: int, K: int, blocks: int, gain: float = 0.10) -> Tuple[int, int]:
    """One persistent block per CU walks 256 x 256 tiles, so `items` batched problems of `tiles` tiles each take
    ceil(items * tiles / blocks) rounds of K / 64 k-tiles — 576 tiles on 256 CUs pay 3 rounds for 2.25.  (t, s): the last t
    problems leave the batched launch and run as t * s problems of a grouped launch, each over K / s of the contraction
    (ceil(t * tiles * s / blocks) rounds of 1 / s the length) into fp32 slabs that clt_sum_slabs_f32 adds in piece order.
    (0, 1): no split pays.  Priced in rounds of the whole contraction, as measured at K = 32768 (853 us per round): a piece
    keeps >= 2048 of the contraction, a piece's tile costs 3 % of a round beyond its k-tiles (prologue + a 256 KB fp32 tile
    stored while every other CU stores one: 23 us), a problem's slab sum 2 %; the estimate has to beat the one launch by 10 %
    (B = 4096, d = 768, K = 32768 x 12 layers: 2.56 -> 1.73 + 0.39 + 0.07 ms, B = 2048: step 11.8 -> 11.1-11.4 ms; B = 1024 with its
    144 tiles and B = 8192 with 4.5 rounds, estimated at 7 %, measured equal or slower: left alone)."""
    plain = -(-items * tiles // blocks)
    best, pick = (1.0 - gain) * plain, (0, 1)
    for t in range(1, items + 1):
        main = -(-(items - t) * tiles // blocks)
        for s in (2, 3, 4, 6, 8, 12, 16):
            if K % (128 * s) or K // s < 2048:
                continue
            est = main + -(-t * tiles * s // blocks) * (1.0 / s + 0.03) + 0.02 * t
            if est < best - 1e-9:
                best, pick = est, (t, s)
    return pick


def plt_decoder_fwd_t(fo: torch.Tensor, WT: torch.Tensor, b) -> torch.Tensor:
    """fo: bf16 [B, L, F]; WT: bf16 [L, d, F] (transposed weights)."""
    Bn, L, F = fo.shape
    d = WT.shape[1]
    r = torch.empty(Bn, L, d, dtype=torch.float32, device=fo.device)
    _chk(fo, op16()); _chk(WT, op16())
    t, s = 0, 1
    if GEMM_TAIL_SPLIT and F % 8 == 0 and d % 4 == 0 and Bn * d >= 256 * 256:
        t, s = _tail_split(L, -(-Bn // 256) * -(-d // 256), F, _persistent_blocks(fo.device), 0.05 if GEMM_TAIL_SPLIT == 2 else 0.10)
    if t:
        # the last t layers: s pieces of the feature contraction each, one fp32 slab per piece
        Kp, slab_b = F // s, 4 * Bn * d
        slabs = torch.empty(t * s, Bn, d, dtype=torch.float32, device=fo.device)
        rec = []
        for i in range(t):
            l = L - t + i
            for j in range(s):
                rec += [l * F + j * Kp, WT.data_ptr() + 2 * (l * d * F + j * Kp), slabs.data_ptr() + (i * s + j) * slab_b]
        table = _device_table(rec, fo.device)
    if L - t:
        _gemm("clt_gemm_bf16", fo, 0, WT, 0, r, 0, Bn, d, F, L * F, F, L * d, False, True,
              batch=L - t, sA=F, sB=d * F, sC=d, bias=b, sBias=d)
    if t:
        rc = _timed_raw("gemm", 2.0 * Bn * d * F * t, "clt_gemm_bf16_nt_grouped", fo.data_ptr(), table.data_ptr(), t * s, Bn, d, Kp,
                        L * F, F, d, 0, _stream())
        if rc != 0:
            raise CltError("clt_gemm_bf16_nt_grouped", rc)
        for i in range(t):
            l = L - t + i
            call("clt_sum_slabs_f32", slabs.data_ptr() + i * s * slab_b, s, Bn * d, Bn, d, L * d,
                 (b.data_ptr() + 4 * l * d) if b is not None else None, r.data_ptr() + 4 * l * d, _stream())
    return r


#: upper bound on the fp32 slab memory of one grouped decoder-forward launch
GROUPED_SLAB_BYTES = 4 << 30


def clt_decoder_fwd_t(fo: torch.Tensor, WTs: List[torch.Tensor], b, k_major: bool = False) -> torch.Tensor:
    """fo: bf16 [B, L, F]; WTs[l]: bf16 [(l+1), d, F].  Per target layer one batched GEMM over its source layers
    (split-K by source layer) into fp32 slabs, summed in source-layer order (+ bias).  k_major: WTs[l] is the weight as it
    lies, bf16 [(l+1), F, d] — the grouped launch takes it as a k-major B operand (no transposed copy; same bits)."""
    Bn, L, F = fo.shape
    d = WTs[0].shape[2 if k_major else 1]
    r = torch.empty(Bn, L, d, dtype=torch.float32, device=fo.device)
    # (target, source) pairs in persistent grouped launches, one fp32 slab per pair; consecutive target layers share a
    # launch as long as their slabs fit GROUPED_SLAB_BYTES (G32: all 78 pairs, 1 GB, one launch)
    if F % 8 == 0 and d % (8 if k_major else 4) == 0 and Bn * d >= 256 * 256:
        slab_b = 4 * Bn * d
        l0, ok = 0, True
        while l0 < L and ok:
            l1, P = l0, 0
            while l1 < L and (P == 0 or (P + l1 + 1) * slab_b <= GROUPED_SLAB_BYTES):
                P += l1 + 1
                l1 += 1
            slabs = torch.empty(P, Bn, d, dtype=torch.float32, device=fo.device)
            rec, p = [], 0
            for l in range(l0, l1):
                wt = _chk(WTs[l], op16()).data_ptr()
                for lp in range(l + 1):
                    rec += [lp * F, wt + 2 * lp * d * F, slabs.data_ptr() + p * slab_b]
                    p += 1
            table = _device_table(rec, fo.device)
            rc = _timed_raw("gemm", 2.0 * Bn * d * F * P, "clt_gemm_bf16_nt_grouped", _chk(fo, op16()).data_ptr(),
                            table.data_ptr(), P, Bn, d, F, L * F, d if k_major else F, d, 2 if k_major else 0, _stream())
            if rc == 1 and l0 == 0:
                ok = False
                break
            if rc != 0:
                raise CltError("clt_gemm_bf16_nt_grouped", rc)
            p = 0
            for l in range(l0, l1):
                call("clt_sum_slabs_f32", slabs.data_ptr() + p * slab_b, l + 1, Bn * d, Bn, d, L * d,
                     (b.data_ptr() + 4 * l * d) if b is not None else None, r.data_ptr() + 4 * l * d, _stream())
                p += l + 1
            l0 = l1
        if ok:
            return r
    slabs = torch.empty(L, Bn, d, dtype=torch.float32, device=fo.device)
    for l in range(L):
        _gemm("clt_gemm_bf16", fo, 0, _chk(WTs[l], op16()), 0, slabs, 0, Bn, d, F, L * F, d if k_major else F, d, False, not k_major,
              batch=l + 1, sA=F, sB=d * F, sC=Bn * d)
        call("clt_sum_slabs_f32", slabs.data_ptr(), l + 1, Bn * d, Bn, d, L * d, (b.data_ptr() + 4 * l * d) if b is not None else None,
             r.data_ptr() + 4 * l * d, _stream())
    return r


def wgrad_ta_ok(Bn: int, F: int, d: int) -> bool:
    """Whether wgrad_ta's shape goes through the LDS-DMA kernel (else the generic tile kernel would take it: keep the copy)."""
    return F % 8 == 0 and Bn % 8 == 0 and d % 4 == 0 and F * d >= 256 * 256


def wgrad_ta(feat_op: torch.Tensor, gT: torch.Tensor, out: torch.Tensor, accumulate=False) -> torch.Tensor:
    """Per-layer decoder weight gradient out[l] [F, d] (+)= feat[:, l]^T @ g[:, l] with the features as they lie: feat_op bf16
    [B, L, F] is the k-major A operand (stored [K = B][M = F], row stride L * F), gT bf16 [L, d, B] the k-contiguous B."""
    Bn, L, F = feat_op.shape
    d = gT.shape[1]
    _gemm("clt_gemm_bf16", _chk(feat_op, op16()), 0, _chk(gT, op16()), 0, out, 0, F, d, Bn, L * F, Bn, d, True, True,
          batch=L, sA=F, sB=d * Bn, sC=F * d, accumulate=accumulate)
    return out


def wgrad_t(aT: torch.Tensor, bT: torch.Tensor, out: torch.Tensor, batch: int, sA: int, sB: int, accumulate=False):
    """out[i] [M, N] (+)= aT[i] [M, K] @ bT[i] [N, K]^T with K = tokens; aT/bT are the token-minor (transposed) bf16
    operands, e.g. features^T [L, F, B] and g^T [L, d, B]."""
    M, N, K = out.shape[-2], out.shape[-1], aT.shape[-1]
    _gemm("clt_gemm_bf16", _chk(aT, op16()), 0, _chk(bT, op16()), 0, out, 0, M, N, K, K, K, N, False, True,
          batch=batch, sA=sA, sB=sB, sC=M * N, accumulate=accumulate)
    return out


def clt_wgrad_t_grouped(fT: torch.Tensor, gT: torch.Tensor, outs: List[torch.Tensor]) -> bool:
    """Cross-layer decoder weight gradients in ONE grouped launch: outs[l][l'] [F, d] = fT[l'] [F, B] @ gT[l] [d, B]^T for all
    l' <= l.  fT: bf16 [L, F, B], gT: bf16 [L, d, B] (token-minor operands), outs[l]: fp32 [(l+1), F, d] (overwritten).
    False when the shape does not qualify (the caller then runs the per-target-layer launches)."""
    L, F, Bn = fT.shape
    d = gT.shape[1]
    _chk(fT, op16()); _chk(gT, op16())
    if Bn % 8 or d % 4 or F * d < 256 * 256:
        return False
    rec = []
    for l in range(L):
        _chk(outs[l])
        for lp in range(l + 1):
            rec += [lp * F * Bn, gT.data_ptr() + 2 * l * d * Bn, outs[l].data_ptr() + 4 * lp * F * d]
    table = _device_table(rec, fT.device)
    P = L * (L + 1) // 2
    rc = _timed_raw("gemm", 2.0 * F * d * Bn * P, "clt_gemm_bf16_nt_grouped", fT.data_ptr(), table.data_ptr(), P, F, d, Bn, Bn, Bn, d,
                    0, _stream())
    if rc == 1:
        return False
    if rc != 0:
        raise CltError("clt_gemm_bf16_nt_grouped", rc)
    return True


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    _chk(x)
    out = torch.empty(x.shape, dtype=op16(), device=x.device)
    call("clt_cast_bf16", x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def cast_transpose_bf16(x: torch.Tensor) -> torch.Tensor:
    """[batch, rows, cols] fp32 -> [batch, cols, rows] bf16."""
    _chk(x)
    nb, rows, cols = x.shape
    out = torch.empty(nb, cols, rows, dtype=op16(), device=x.device)
    call("clt_cast_transpose_bf16", x.data_ptr(), cols, out.data_ptr(), rows, rows, cols, nb, rows * cols, rows * cols,
         _stream())
    return out


def weight_prep_bf16(w: torch.Tensor, sq: Optional[torch.Tensor] = None, accumulate: bool = True):
    """w: fp32 [nb, rows, n] -> (bf16 copy, bf16 [nb, n, rows] transposed copy) in one read; sq (fp32 [>= nb*rows]) gets
    the squared row norms added (or stored).  None when the shape does not qualify (rows % 8, n % 4)."""
    _chk(w)
    nb, rows, n = w.shape
    if rows % 8 or n % 4:
        return None
    op = torch.empty(nb, rows, n, dtype=op16(), device=w.device)
    opT = torch.empty(nb, n, rows, dtype=op16(), device=w.device)
    call("clt_weight_prep_bf16", w.data_ptr(), nb, rows, n, op.data_ptr(), opT.data_ptr(),
         _chk(sq).data_ptr() if sq is not None else None, int(accumulate), _stream())
    return op, opT


def weight_prep_gather_bf16(w: torch.Tensor, idx: torch.Tensor):
    """weight_prep_bf16(gather_rows(w, idx)) without the gathered fp32 copy: w fp32 [groups, F, n]; idx int32 [>= groups, n_pad]
    (group g uses row g; -1 = zero row) -> (16-bit [groups, n_pad, n], 16-bit [groups, n, n_pad])."""
    _chk(w); _chk(idx, torch.int32)
    groups, F, n = w.shape
    n_pad = idx.shape[1]
    assert idx.shape[0] >= groups and n_pad % 8 == 0 and n % 4 == 0
    op = torch.empty(groups, n_pad, n, dtype=op16(), device=w.device)
    opT = torch.empty(groups, n, n_pad, dtype=op16(), device=w.device)
    call("clt_weight_prep_gather_bf16", w.data_ptr(), idx.data_ptr(), groups, F, n_pad, n, op.data_ptr(), opT.data_ptr(), _stream())
    return op, opT


# ----------------------------------------------------------------------------------------------- K3
def relu_fwd(x):
    out = torch.empty_like(_chk(x))
    call("clt_relu_fwd_f32", x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def relu_bwd(x, g):
    dx = torch.empty_like(_chk(x))
    call("clt_relu_bwd_f32", x.data_ptr(), _chk(g).data_ptr(), dx.data_ptr(), x.numel(), _stream())
    return dx


def jumprelu_fwd(x, theta):
    _chk(x); _chk(theta)
    rows, n = x.shape[0], x[0].numel()
    assert theta.numel() == n
    out = torch.empty_like(x)
    call("clt_jumprelu_fwd_f32", x.data_ptr(), theta.data_ptr(), out.data_ptr(), rows, n, _stream())
    return out


def jumprelu_bwd(x, theta, bandwidth: float, g):
    _chk(x); _chk(theta); _chk(g)
    rows, n = x.shape[0], x[0].numel()
    dx = torch.empty_like(x)
    dth = torch.empty_like(theta)
    nb = lib.clt_jumprelu_bwd_workspace_bytes(rows, n)
    ws = workspace(nb, x.device)
    call("clt_jumprelu_bwd_f32", x.data_ptr(), theta.data_ptr(), float(bandwidth), g.data_ptr(), dx.data_ptr(),
         dth.data_ptr(), rows, n, ws.data_ptr(), ws.numel(), _stream())
    return dx, dth


# ----------------------------------------------------------------------------------------------- K4
def topk_rows(x, k: int, want_compact=False):
    _chk(x)
    Bn, L, F = x.shape
    feat = torch.empty_like(x)
    idx = val = None
    if want_compact:
        idx = torch.empty(Bn * L, max(k, 1), dtype=torch.int32, device=x.device)
        val = torch.empty(Bn * L, max(k, 1), dtype=torch.float32, device=x.device)
    call("clt_topk_rows_f32", x.data_ptr(), feat.data_ptr(), Bn * L, F, k, idx.data_ptr() if want_compact else None,
         val.data_ptr() if want_compact else None, _stream())
    return (feat, idx, val) if want_compact else feat


def topk_segmented(x, k: int, per_layer: bool, threshold: Optional[torch.Tensor], e: float):
    """BatchTopK (per_layer=False) / PerLayerBatchTopK (True); `threshold` is updated in place."""
    _chk(x)
    Bn, L, F = x.shape
    feat = torch.empty_like(x)
    nb = lib.clt_topk_segmented_workspace_bytes(Bn, L, F, int(per_layer))
    ws = workspace(nb, x.device)
    if threshold is not None:
        _chk(threshold)
    call("clt_topk_segmented_f32", x.data_ptr(), feat.data_ptr(), Bn, L, F, k, int(per_layer),
         threshold.data_ptr() if threshold is not None else None, float(e or 0.0), ws.data_ptr(), ws.numel(), _stream())
    return feat


def mask_positive(feat, g, out=None, accumulate=False):
    dx = torch.empty_like(_chk(feat)) if out is None else _chk(out)
    call("clt_mask_positive_f32", feat.data_ptr(), _chk(g).data_ptr(), dx.data_ptr(), feat.numel(), int(accumulate),
         _stream())
    return dx


def mask_positive_t(feat: torch.Tensor, dfeat: torch.Tensor):
    """dpre = feat > 0 ? dfeat : 0 as the transposed 16-bit GEMM operand + its column sums, one pass (clt_mask_positive_t_f32).
    feat / dfeat: fp32 [B, L, n] -> (dpreT 16-bit [L, n, B], colsum fp32 [L, n])."""
    _chk(feat); _chk(dfeat)
    Bn, L, n = feat.shape
    dpreT = torch.empty(L, n, Bn, dtype=op16(), device=feat.device)
    cs = torch.empty(L, n, dtype=torch.float32, device=feat.device)
    ws = workspace(lib.clt_mask_positive_t_workspace_bytes(Bn, L * n), feat.device)
    call("clt_mask_positive_t_f32", feat.data_ptr(), dfeat.data_ptr(), Bn, L * n, dpreT.data_ptr(), cs.data_ptr(), ws.data_ptr(),
         ws.numel(), _stream())
    return dpreT, cs


def threshold_fwd(x, threshold, per_layer: bool, L: int):
    """Inference path; x: [..., F] of one layer or [B, L, F]."""
    _chk(x); _chk(threshold)
    out = torch.empty_like(x)
    F = x.shape[-1]
    rows = x.numel() // F
    if per_layer:
        call("clt_threshold_fwd_f32", x.data_ptr(), threshold.data_ptr(), 1, out.data_ptr(), rows // L, L, F, _stream())
    else:
        call("clt_threshold_fwd_f32", x.data_ptr(), threshold.data_ptr(), 0, out.data_ptr(), rows, 1, F, _stream())
    return out


def mask_dead(x, dead_u8):
    _chk(x); _chk(dead_u8, torch.uint8)
    out = torch.empty_like(x)
    call("clt_mask_dead_f32", x.data_ptr(), dead_u8.data_ptr(), out.data_ptr(), x.shape[0], x[0].numel(), _stream())
    return out


# ----------------------------------------------------------------------------------------- K8 - K12
def mse(r, y, mean, std, want_grad=True, want_ys=False, count: Optional[int] = None):
    """Returns (loss_sum fp64 [1], g or None, ys or None); g = 2/count * (r - ys), count = number of elements of the
    whole reconstruction (default: of r; a caller that passes one layer's slab gives B*L*d)."""
    _chk(r); _chk(y, contiguous=False); _chk(mean); _chk(std)
    rows, n, rs = _rows_view(y)
    g = torch.empty_like(r) if want_grad else None
    ys = torch.empty_like(r) if want_ys else None
    loss_sum = torch.empty(1, dtype=torch.float64, device=r.device)
    nb = lib.clt_reduce_workspace_bytes(rows * n)
    ws = workspace(nb, r.device)
    call("clt_mse_f32", r.data_ptr(), y.data_ptr(), rs, mean.data_ptr(), std.data_ptr(), rows, n, GRAD_SCALE * 2.0 / (count or rows * n),
         g.data_ptr() if want_grad else None, ys.data_ptr() if want_ys else None, loss_sum.data_ptr(), ws.data_ptr(),
         ws.numel(), _stream())
    return loss_sum, g, ys


def mse_bf16_colsum(r, y, mean, std, count: Optional[int] = None, g16=None, colsum=None):
    """Base loss of the list regime in bf16 mode, one pass: (loss_sum fp64 [1], g16 = bf16 2/count * (r - ys),
    column sums of that gradient in fp32 = the decoder bias gradient).  count: see mse().  g16 / colsum: preallocated
    outputs (slab views).  None when the layout does not qualify."""
    _chk(r); _chk(y, contiguous=False); _chk(mean); _chk(std)
    rows, n, rs = _rows_view(y)
    if n % 4 or rs % 4 or y.data_ptr() % 16 or r.data_ptr() % 16 or mean.data_ptr() % 16 or std.data_ptr() % 16:
        return None
    if g16 is None:
        g16 = torch.empty(r.shape, dtype=op16(), device=r.device)
    if colsum is None:
        colsum = torch.empty(mean.shape, dtype=torch.float32, device=r.device)
    _chk(g16, op16()); _chk(colsum)
    loss_sum = torch.empty(1, dtype=torch.float64, device=r.device)
    ws = workspace(lib.clt_mse_bf16_colsum_workspace_bytes(rows, n), r.device)
    call("clt_mse_bf16_colsum_f32", r.data_ptr(), y.data_ptr(), rs, mean.data_ptr(), std.data_ptr(), rows, n, GRAD_SCALE * 2.0 / (count or rows * n),
         g16.data_ptr(), colsum.data_ptr(), 0, loss_sum.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return loss_sum, g16, colsum


def mse_bf16_colsum_layers(r_lm, y, mean, std, count: int, g16_lm, colsum, loss_sum):
    """mse_bf16_colsum over a group of layers of a layer-major reconstruction: r_lm / g16_lm [nl, B, d] (contiguous slices of
    the [L, B, d] tensors), y = batch[:, 1, lo:lo + nl] ([B, nl, d] view), mean / std / colsum [nl, d], loss_sum fp64 [1]
    (all preallocated views).  False when the layout does not qualify (B % 64, d % 4, alignment)."""
    _chk(r_lm); _chk(g16_lm, op16()); _chk(mean); _chk(std); _chk(colsum); _chk(y, contiguous=False)
    nl, Bn, d = r_lm.shape
    rs = y.stride(0)
    if (Bn % 64 or d % 4 or rs % 4 or y.stride(1) != d or y.stride(2) != 1 or any(t.data_ptr() % 16 for t in (r_lm, y, mean, std))
            or g16_lm.data_ptr() % 8):
        return False
    ws = workspace(lib.clt_mse_bf16_colsum_workspace_bytes(nl * Bn, d), r_lm.device)
    call("clt_mse_bf16_colsum_layers_f32", r_lm.data_ptr(), y.data_ptr(), rs, mean.data_ptr(), std.data_ptr(), nl * Bn, d, Bn,
         GRAD_SCALE * 2.0 / count, g16_lm.data_ptr(), colsum.data_ptr(), 0, loss_sum.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return True


def aux_mse(aux_r, ys, r, g_main, scale: float):
    """g_aux = scale*2/N * e (finite only); g_main += g_aux in place.  Returns (aux_sum fp64 [1], g_aux)."""
    for t in (aux_r, ys, r, g_main):
        _chk(t)
    n = r.numel()
    g_aux = torch.empty_like(r)
    s = torch.empty(1, dtype=torch.float64, device=r.device)
    nb = lib.clt_reduce_workspace_bytes(n)
    ws = workspace(nb, r.device)
    call("clt_aux_mse_f32", aux_r.data_ptr(), ys.data_ptr(), r.data_ptr(), n, GRAD_SCALE * scale * 2.0 / n, g_aux.data_ptr(),
         g_main.data_ptr(), s.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return s, g_aux


def column_sum(x, out=None, accumulate=False):
    _chk(x)
    rows, n = x.shape[0], x[0].numel()
    if out is None:
        out = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
    nb = lib.clt_column_sum_workspace_bytes(rows, n)
    ws = workspace(nb, x.device)
    call("clt_column_sum_f32", x.data_ptr(), rows, n, out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel(),
         _stream())
    return out


def row_sqnorm(W, sq, sq_off=0, accumulate=True):
    """sq[sq_off + row] (+)= sum_d W[row]^2 for the rows of W flattened to [rows, d]."""
    _chk(W); _chk(sq)
    d = W.shape[-1]
    rows = W.numel() // d
    call("clt_row_sqnorm_f32", W.data_ptr(), rows, d, sq.data_ptr() + 4 * sq_off, int(accumulate), _stream())


def jumprelu_loss_bwd(pre, theta, sq, dfeat_main, bandwidth, lam_over_B, c, use_tanh, coef_over_BL):
    """Fused JumpReLU-module loss terms + backward through the sparsifier (clt_lightning.py:438-468).
    Returns (sums fp64 [3] = [sum t, sum relu(-pre), sum w], dpre, dtheta, dnorm)."""
    _chk(pre); _chk(theta); _chk(sq)
    rows, n = pre.shape[0], pre[0].numel()
    sums = torch.empty(3, dtype=torch.float64, device=pre.device)
    have = dfeat_main is not None
    dpre = torch.empty_like(pre) if have else None
    dth = torch.empty_like(theta) if have else None
    dn = torch.empty(n, dtype=torch.float32, device=pre.device) if have else None
    nb = lib.clt_jumprelu_loss_workspace_bytes(rows, n)
    ws = workspace(nb, pre.device)
    call("clt_jumprelu_loss_bwd_f32", pre.data_ptr(), theta.data_ptr(), sq.data_ptr(),
         _chk(dfeat_main).data_ptr() if have else None, float(bandwidth), float(GRAD_SCALE * lam_over_B), float(c), int(use_tanh),
         float(GRAD_SCALE * coef_over_BL), dpre.data_ptr() if have else None, dth.data_ptr() if have else None,
         dn.data_ptr() if have else None, sums.data_ptr(), rows, n, ws.data_ptr(), ws.numel(), _stream())
    return sums, dpre, dth, dn


def jumprelu_fwd_bf16(x, theta=None, bias=None, want_T=True):
    """Dense bf16 regime: JumpReLU (theta=None: ReLU) straight to the GEMM operands.  x: fp32 or bf16 [B, L, F]; bias (fp32
    [L, F], optional) is added to x first (a 16-bit x then holds the encoder product alone: encoder_fwd_out16(..., None)).
    Returns (feat bf16 [B, L, F], featT bf16 [L, F, B], fired uint8 [L, F]); want_T=False: no transposed copy (featT = None)."""
    _chk(x, x.dtype)
    if bias is not None:
        _chk(bias)
        assert bias.numel() == x.shape[1] * x.shape[2]
    assert x.dtype in (torch.float32, op16())
    Bn, L, F = x.shape
    if theta is not None:
        _chk(theta)
        assert theta.numel() == L * F
    feat = torch.empty(Bn, L, F, dtype=op16(), device=x.device)
    featT = torch.empty(L, F, Bn, dtype=op16(), device=x.device) if want_T else None
    fired = torch.empty(L, F, dtype=torch.uint8, device=x.device)
    call("clt_jumprelu_fwd_bf16", x.data_ptr(), int(x.dtype in _DT16), theta.data_ptr() if theta is not None else None,
         bias.data_ptr() if bias is not None else None, Bn, L * F, feat.data_ptr(), featT.data_ptr() if want_T else None, fired.data_ptr(), _stream())
    return feat, featT, fired


def jumprelu_loss_bwd_bf16(pre, theta, sq, dfeat_main, bandwidth, lam_over_B, c, use_tanh, coef_over_BL, bias=None):
    """jumprelu_loss_bwd for the dense bf16 regime: returns (sums fp64 [3], dpreT bf16 [L, F, B], dbias fp32 [L, F],
    dtheta, dnorm) — no fp32 dpre tensor.  pre / dfeat_main: fp32 or bf16; bias: as in jumprelu_fwd_bf16."""
    _chk(pre, pre.dtype); _chk(theta); _chk(sq); _chk(dfeat_main, dfeat_main.dtype)
    if bias is not None:
        _chk(bias)
        assert bias.numel() == pre.shape[1] * pre.shape[2]
    assert pre.dtype in (torch.float32, op16()) and dfeat_main.dtype in (torch.float32, op16())
    Bn, L, F = pre.shape
    n = L * F
    sums = torch.empty(3, dtype=torch.float64, device=pre.device)
    dpreT = torch.empty(L, F, Bn, dtype=op16(), device=pre.device)
    db = torch.empty(L, F, dtype=torch.float32, device=pre.device)
    dth = torch.empty_like(theta)
    dn = torch.empty(n, dtype=torch.float32, device=pre.device)
    ws = workspace(lib.clt_jumprelu_loss_bf16_workspace_bytes(Bn, n), pre.device)
    call("clt_jumprelu_loss_bwd_bf16", pre.data_ptr(), int(pre.dtype in _DT16), theta.data_ptr(),
         bias.data_ptr() if bias is not None else None, sq.data_ptr(), dfeat_main.data_ptr(), int(dfeat_main.dtype in _DT16), float(bandwidth),
         float(GRAD_SCALE * lam_over_B), float(c), int(use_tanh), float(GRAD_SCALE * coef_over_BL), dpreT.data_ptr(), db.data_ptr(), dth.data_ptr(),
         dn.data_ptr(), sums.data_ptr(), Bn, n, ws.data_ptr(), ws.numel(), _stream())
    return sums, dpreT, db, dth, dn


def rowscale_add(W, dnorm, sq, dW, off=0):
    """dW[row] += dnorm[off+row]/sqrt(sq[off+row]) * W[row]."""
    d = W.shape[-1]
    rows = W.numel() // d
    call("clt_rowscale_add_f32", _chk(W).data_ptr(), _chk(dnorm).data_ptr() + 4 * off, _chk(sq).data_ptr() + 4 * off,
         _chk(dW).data_ptr(), rows, d, _stream())


def recon_metrics(r, y, mean, std):
    """Per-column fp64 sums [L, d, 6] for the logged reconstruction metrics (see clt_recon_metrics_f32)."""
    _chk(r); _chk(y, contiguous=False); _chk(mean); _chk(std)
    rows, n, rs = _rows_view(y)
    out = torch.empty(*r.shape[1:], 6, dtype=torch.float64, device=r.device)
    call("clt_recon_metrics_f32", r.data_ptr(), y.data_ptr(), rs, mean.data_ptr(), std.data_ptr(), rows, n, out.data_ptr(),
         _stream())
    return out


def count_positive(feat):
    """int32 [L, F]: number of rows (tokens) in which each feature is positive."""
    _chk(feat)
    rows, n = feat.shape[0], feat[0].numel()
    counts = torch.empty(feat.shape[1:], dtype=torch.int32, device=feat.device)
    call("clt_count_positive_f32", feat.data_ptr(), rows, n, counts.data_ptr(), _stream())
    return counts


def column_any_positive(feat):
    _chk(feat)
    rows, n = feat.shape[0], feat[0].numel()
    fired = torch.empty(feat.shape[1:], dtype=torch.uint8, device=feat.device)
    call("clt_column_any_positive_f32", feat.data_ptr(), rows, n, fired.data_ptr(), _stream())
    return fired


def dead_update(fired, tokens: int, last_active, tokens_till_dead: int):
    """In-place last_active update (clt_lightning.py:282-288); returns (dead uint8 [L,F], dead_count int32 [1])."""
    _chk(fired, torch.uint8); _chk(last_active, torch.int64)
    dead = torch.empty_like(fired)
    cnt = torch.empty(1, dtype=torch.int32, device=fired.device)
    call("clt_dead_update", fired.data_ptr(), fired.numel(), int(tokens), last_active.data_ptr(), int(tokens_till_dead),
         dead.data_ptr(), cnt.data_ptr(), _stream())
    return dead, cnt


# ---------------------------------------------------------------------------------------------- K13
def adam(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, adamw=False,
         bf16_shadow=None, row_term=None):
    """row_term = (dnorm, sq): adds (dnorm[r] / sqrt(sq[r])) * p to the gradient of every row r of the last dimension
    (contiguous parameters only) inside the update."""
    for t in (p, g, m, v):                       # any dense layout, as long as all four share it
        _chk(t, contiguous=False)
        if t.stride() != p.stride() or t.shape != p.shape:
            raise ValueError("adam: p, g, m, v must share one memory layout")
    if not (p.is_contiguous() or p.permute(*sorted(range(p.dim()), key=lambda i: -p.stride(i))).is_contiguous()):
        raise ValueError("adam: parameter storage must be dense")
    if row_term is not None:
        if not p.is_contiguous():
            raise ValueError("adam: a row term needs a contiguous parameter")
        _chk(row_term[0]); _chk(row_term[1])
        assert row_term[0].numel() >= p.numel() // p.shape[-1]
    call("clt_adam_f32", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1),
         float(beta2), float(eps), float(weight_decay), int(adamw), int(step),
         bf16_shadow.data_ptr() if bf16_shadow is not None else None,
         row_term[0].data_ptr() if row_term is not None else None, row_term[1].data_ptr() if row_term is not None else None,
         p.shape[-1] if row_term is not None else 0, _stream())


def adam_prep(p, g, m, v, step: int, lr: float, beta1, beta2, eps, weight_decay, adamw, op, opT, sq_part, row_term=None) -> bool:
    """clt_adam_prep_f32: the Adam step of a contiguous decoder matrix p [nb, rows, n] that also leaves the updated matrix's
    16-bit copy (op, same shape), transposed 16-bit copy (opT [nb, n, rows]) and squared row norms (sq_part [nb * rows]) — the
    next step's weight preparation.  False when the layout does not qualify (the caller then runs ops.adam)."""
    if p.dim() != 3 or not all(t.is_contiguous() and t.shape == p.shape for t in (p, g, m, v)):
        return False
    nb, rows, n = p.shape
    if rows % 8 or n % 4 or op.dtype != op16() or opT.dtype != op16() or op.shape != p.shape or opT.shape != (nb, n, rows):
        return False
    for t in (p, g, m, v, sq_part):
        _chk(t)
    assert sq_part.numel() >= nb * rows
    if row_term is not None:
        _chk(row_term[0]); _chk(row_term[1])
        assert row_term[0].numel() >= nb * rows
    rc = lib.clt_adam_prep_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), nb, rows, n, float(lr), float(beta1), float(beta2),
                               float(eps), float(weight_decay), int(adamw), int(step), _chk(op, op16()).data_ptr(), _chk(opT, op16()).data_ptr(),
                               sq_part.data_ptr(), row_term[0].data_ptr() if row_term is not None else None,
                               row_term[1].data_ptr() if row_term is not None else None, _stream())
    if rc == -4:                                  # CLT_ERR_ALIGN
        return False
    if rc != 0:
        raise CltError("clt_adam_prep_f32", rc)
    return True


# ------------------------------------------------------------------------------ sparse decoder path
class SparseLists:
    """Token-major CSR + feature-major CSC lists of the active features (include/clt_hip.h, sparse section)."""
    __slots__ = ("B", "L", "F", "cap", "row_start", "ent_col", "ent_val", "col_start", "csc_tok", "csc_val",
                 "csc_ent", "fired", "total", "overflow", "used_fallback", "kth", "tie_radix", "speculated", "confirm", "hot", "overflowed", "join")


def sparse_build(feat: torch.Tensor, cap: int) -> SparseLists:
    _chk(feat)
    Bn, L, F = feat.shape
    dev = feat.device
    s = SparseLists()
    s.B, s.L, s.F, s.cap = Bn, L, F, int(cap)
    i32 = dict(dtype=torch.int32, device=dev)
    s.row_start = torch.empty(Bn * L + 1, **i32)
    s.ent_col = torch.empty(cap, **i32)
    s.ent_val = torch.empty(cap, dtype=torch.float32, device=dev)
    s.col_start = torch.empty(L * F + 1, **i32)
    s.csc_tok = torch.empty(cap, **i32)
    s.csc_val = torch.empty(cap, dtype=torch.float32, device=dev)
    s.csc_ent = torch.empty(cap, **i32)
    s.fired = torch.empty(L, F, dtype=torch.uint8, device=dev)
    s.total = torch.empty(1, **i32)
    s.overflow = torch.empty(1, **i32)
    nb = lib.clt_sparse_build_workspace_bytes(Bn, L, F)
    ws = workspace(nb, dev)
    call("clt_sparse_build_f32", feat.data_ptr(), Bn, L, F, s.cap, s.row_start.data_ptr(), s.ent_col.data_ptr(),
         s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(), s.csc_val.data_ptr(), s.csc_ent.data_ptr(),
         s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return s


def sparse_build_threshold(x: torch.Tensor, theta: Optional[torch.Tensor], cap: int, bias: Optional[torch.Tensor] = None,
                           split: bool = False) -> SparseLists:
    """Lists of JumpReLU(x) (theta [1, L, F]) or ReLU(x) (theta None) straight from the pre-activations x [B, L, F]:
    entries = positions with x > theta and x > 0 (x fp32, or 16-bit in the active operand format: values widened; a 16-bit
    x may hold the encoder product alone, the fp32 `bias` [L, F] is then added to the widened values first).  `overflow`
    (device int32 [1]) = 1 when more than `cap` entries exist
    (the lists are then incomplete and the caller takes the dense regime); `total` holds the true count.
    split: phases 1 (counts, offsets, `overflow`) and 2 (token-major lists) on the launch stream, phase 3 (feature-major lists) on
    the second stream of the list builds; the result then carries `overflowed()` -> bool — one host read, of a flag that is final
    BEFORE the lists are filled, so the launch queue does not drain behind it — and `join()`, which makes the launch stream
    wait for phase 3 and has to be called before anything reads col_start / csc_* / fired."""
    x16 = x.dtype in _DT16
    _chk(x, x.dtype)
    assert x.dtype == torch.float32 or (x.dtype == op16() and x.shape[-1] % 4 == 0)
    Bn, L, F = x.shape
    if theta is not None:
        _chk(theta)
        assert theta.numel() == L * F
    if bias is not None:
        assert x16, "a separate bias only with 16-bit pre-activations"
        _chk(bias)
        assert bias.numel() == L * F
    s = _empty_lists(Bn, L, F, int(cap), x.device)
    split = bool(split and LIST_CSC_STREAM and TIMING is None)
    need = lib.clt_sparse_build_workspace_bytes(Bn, L, F)
    if split:                                   # (phase 2 runs beside kernels that take their scratch from ops.workspace)
        ws = _select_workspace(need, x.device)
        _wait_last_csc(x.device)
    else:
        ws = workspace(need, x.device)
    extra = ((bias.data_ptr() if bias is not None else None),) if x16 else ()

    def phase(ph):
        call("clt_sparse_build_threshold_x16" if x16 else "clt_sparse_build_threshold_f32", x.data_ptr(), theta.data_ptr() if theta is not None else None, *extra, Bn, L, F, s.cap,
             s.row_start.data_ptr(), s.ent_col.data_ptr(), s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(),
             s.csc_val.data_ptr(), s.csc_ent.data_ptr(), s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(),
             ws.data_ptr(), ws.numel(), ph, _stream())

    s.used_fallback = None
    s.kth = None
    flag, dev = s.overflow, x.device            # (the closures below must hold neither the lists — a cycle — nor x)
    if not split:
        phase(0)
        s.overflowed = lambda: read_flag(flag) != 0
        s.join = lambda: None
        return s
    phase(1)                                    # counts, offsets, gate: `overflow` is final
    ev = torch.cuda.Event()
    ev.record()
    side, host = _flag_reader(dev)
    with torch.cuda.stream(side):               # (the flag's copy waits for phase 1 alone, not for the fill or the forward)
        side.wait_event(ev)
        host.copy_(flag, non_blocking=True)
    phase(2)                                    # token-major lists
    done = _fork_csc(dev, lambda: phase(3), True)

    def overflowed() -> bool:
        side.synchronize()
        return int(host) != 0

    def join():
        if done is not None:
            torch.cuda.current_stream(dev).wait_event(done)

    s.overflowed, s.join = overflowed, join
    return s


def read_flag(flag: torch.Tensor) -> int:
    """Host value of a device int32 flag through a side stream behind an event: work queued on the launch stream after
    this call's event keeps running while the host waits (the launch queue never drains)."""
    ev = torch.cuda.Event()
    ev.record()
    side, host = _flag_reader(flag.device)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        host.copy_(flag, non_blocking=True)
    side.synchronize()
    return int(host)


def _empty_lists(Bn, L, F, cap, dev) -> SparseLists:
    s = SparseLists()
    s.B, s.L, s.F, s.cap = Bn, L, F, int(cap)
    i32 = dict(dtype=torch.int32, device=dev)
    s.row_start = torch.empty(Bn * L + 1, **i32)
    s.ent_col = torch.empty(cap, **i32)
    s.ent_val = torch.empty(cap, dtype=torch.float32, device=dev)
    s.col_start = torch.empty(L * F + 1, **i32)
    s.csc_tok = torch.empty(cap, **i32)
    s.csc_val = torch.empty(cap, dtype=torch.float32, device=dev)
    s.csc_ent = torch.empty(cap, **i32)
    s.fired = torch.empty(L, F, dtype=torch.uint8, device=dev)
    s.total = torch.empty(1, **i32)
    s.overflow = torch.empty(1, **i32)
    return s


SELECT_ROWS, SELECT_BATCH, SELECT_LAYER = 0, 1, 2


def select_lists(x: torch.Tensor, k: int, mode: int, threshold: Optional[torch.Tensor] = None, e: float = 0.0,
                 col_mask: Optional[torch.Tensor] = None, cap: Optional[int] = None) -> SparseLists:
    """TopK-family selection straight to the active-entry lists (no dense feature tensor).  mode: SELECT_ROWS /
    SELECT_BATCH / SELECT_LAYER.  `threshold` (modes 1/2) is EMA-updated in place.  The returned lists carry
    `used_fallback` (device int32 [1]): 1 when the exhaustive path had to run."""
    _chk(x)
    Bn, L, F = x.shape
    if cap is None:
        cap = max(1, min(k, F) * Bn * L)
    s = _empty_lists(Bn, L, F, cap, x.device)
    nb = lib.clt_select_lists_workspace_bytes(Bn, L, F, k, mode)
    ws = workspace(nb, x.device)
    dense = torch.empty_like(x) if mode != SELECT_ROWS else None      # fallback scratch (untouched on the fast path)
    fb = torch.empty(1, dtype=torch.int32, device=x.device)
    if col_mask is not None:
        _chk(col_mask, torch.uint8)
    if threshold is not None:
        _chk(threshold)
    call("clt_select_lists_f32", x.data_ptr(), col_mask.data_ptr() if col_mask is not None else None, Bn, L, F, k, mode,
         threshold.data_ptr() if threshold is not None else None, float(e or 0.0), s.cap, s.row_start.data_ptr(),
         s.ent_col.data_ptr(), s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(), s.csc_val.data_ptr(),
         s.csc_ent.data_ptr(), s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(),
         dense.data_ptr() if dense is not None else None, fb.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    s.used_fallback = fb
    s.kth = selection_kth_keys(Bn, L, F, k, mode, x.device) if (mode != SELECT_ROWS and k > 0) else None
    return s


def select_lists_sharded(x: torch.Tensor, k: int, mode: int, shard, threshold: Optional[torch.Tensor] = None, e: float = 0.0,
                         col_mask: Optional[torch.Tensor] = None, cap: Optional[int] = None) -> SparseLists:
    """Feature-sharded BatchTopK / PerLayerBatchTopK selection (x = this rank's [B, L, F_local] pre-activations,
    k per (b, l) row over all shard.F_global features).  Lists are rank-local.  One host read of the outcome flag
    per call; if the sampled candidate set was insufficient on any rank, all ranks re-run exhaustively."""
    from .parallel import run_sharded_selection
    _chk(x)
    if mode not in (SELECT_BATCH, SELECT_LAYER):
        raise NotImplementedError("feature-sharded selection covers BatchTopK / PerLayerBatchTopK (per-row TopK across "
                                  "shards needs a per-row candidate exchange: not implemented)")
    Bn, L, F = x.shape
    assert F == shard.F_local
    if k > shard.F_global:
        raise AssertionError("k must be <= d_features")
    kglob = max(1, k * Bn * L)
    if cap is None:
        cap = min(kglob, 2 * kglob // shard.world + 65536)
    S = L if mode == SELECT_LAYER else 1
    fb = torch.zeros(1, dtype=torch.int32, device=x.device)
    if col_mask is not None:
        _chk(col_mask, torch.uint8)
    if threshold is not None:
        _chk(threshold)
    attempts = [(0, cap), (1, cap)]
    while attempts:
        exhaustive, cap = attempts.pop(0)
        s = _empty_lists(Bn, L, F, cap, x.device)
        dims = (Bn, L, F, k, mode, shard.world, exhaustive)
        nb = lib.clt_select_lists_sharded_workspace_bytes(*dims)
        ws = workspace(nb, x.device)
        off = [lib.clt_select_lists_exchange_info(*dims, w) for w in range(5)]
        i32 = lambda o, n: ws[o:o + 4 * n].view(torch.int32)  # noqa: E731
        exchange = {"lb": i32(off[0], 4 * S).view(S, 4), "hist": i32(off[1], off[2]),
                    "ties": i32(off[3], off[4] * shard.world).view(shard.world, off[4])}

        def stages(lo, hi):
            call("clt_select_lists_sharded_f32", lo, hi, shard.world, shard.rank, shard.f_off, shard.F_global, exhaustive,
                 x.data_ptr(), col_mask.data_ptr() if col_mask is not None else None, Bn, L, F, k, mode,
                 threshold.data_ptr() if threshold is not None else None, float(e or 0.0), s.cap, s.row_start.data_ptr(),
                 s.ent_col.data_ptr(), s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(),
                 s.csc_val.data_ptr(), s.csc_ent.data_ptr(), s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(),
                 fb.data_ptr(), ws.data_ptr(), ws.numel(), _stream())

        if k == 0:
            stages(0, 5)
            break
        run_sharded_selection(stages, shard, S, exchange, tie_radix=bool(exhaustive & 8))
        outcome = int(fb)                          # host read of the gate every rank agrees on (it rides in the tie gather)
        if outcome & 2 and cap < kglob:
            # a shard holding far more than its share of the selection would overflow its list capacity: redo the
            # selection with room for the whole selection on every shard
            attempts.insert(0, (exhaustive, kglob))
            continue
        if outcome != 0:                           # not every rank had enough candidates (or too many exact ties)
            if exhaustive & 8:
                raise CltError("clt_select_lists_sharded_f32", outcome)      # (cannot happen: this form has no capacity limits)
            if exhaustive:
                # more entries tied at the k-th value than a tie block holds (4096 per shard and segment): the form that
                # resolves ties by a radix select on the global index — four more small all-reduces, any number of ties
                attempts.append((9, kglob))
            continue
        break
    s.used_fallback = torch.full((1,), exhaustive, dtype=torch.int32, device=x.device)
    s.tie_radix = bool(exhaustive & 8)
    s.kth = None
    if k > 0:
        o6 = lib.clt_select_lists_exchange_info(*dims, 6)
        s.kth = ws[o6:o6 + 16 * S].view(torch.int32).view(S, 4)[:, 0].clone()
    return s


def encoder_select_lists(xs: torch.Tensor, Wt: torch.Tensor, b: Optional[torch.Tensor], k: int, mode: int, prev_T: torch.Tensor,
                         margin: float, threshold: Optional[torch.Tensor], e: float, shard=None, cap: Optional[int] = None,
                         drift: Optional[torch.Tensor] = None, speculate=None, defer: bool = False):
    """Encoder forward fused with the BatchTopK / PerLayerBatchTopK selection: the bf16 NT GEMM's epilogue appends the
    candidates (>= margin * last step's k-th value, `prev_T` int32 keys [S]) and the pre-activation tensor is never
    written.  Returns (lists, T) — T = int32 keys [S] of this step's exact k-th values — or (None, None) when the
    attempt was void (one host read) and the caller must run the ordinary GEMM + selection.
    drift: fp32 [2 S] kept by the caller across steps (new_select_drift): the bound then tightens from `margin` towards
    0.98 as far as the observed step-to-step change of the k-th value allows (clt_select_lists_ext_f32).
    speculate(lists): called after the list construction has been queued and BEFORE the host learns whether the attempt held —
    for work that only reads the lists and writes fresh buffers (the decoder's gather forward); its result is kept in
    lists.speculated.  After a void attempt the lists are empty (row offsets all zero), so such work is harmless and discarded.
    defer: return (lists, T) at once, the host read postponed to lists.confirm() -> bool (False: void; everything computed
    from the lists must then be thrown away) — the caller keeps queueing work that mutates no state and asks before the
    first kernel that does."""
    from .parallel import run_sharded_selection
    Bn, L, d = xs.shape
    F = Wt.shape[1]
    _chk(xs, op16()); _chk(Wt, op16())
    world, rank, f_off, Fg = (shard.world, shard.rank, shard.f_off, shard.F_global) if shard is not None else (1, 0, 0, F)
    kglob = max(1, k * Bn * L)
    if cap is None:
        cap = kglob if world == 1 else min(kglob, 2 * kglob // world + 65536)
    S = L if mode == SELECT_LAYER else 1
    s = _empty_lists(Bn, L, F, cap, xs.device)
    dims = (Bn, L, F, k, mode, world, 2)
    # (a workspace of this entry point's own, not the scratch buffer every op of the launch stream shares: stage 7 runs on a
    # second stream while the launch stream goes on to kernels that take their scratch from ops.workspace)
    ws = _select_workspace(lib.clt_select_lists_sharded_workspace_bytes(*dims), xs.device)
    off = [lib.clt_select_lists_exchange_info(*dims, w) for w in range(10)]
    i32 = lambda o, n: ws[o:o + 4 * n].view(torch.int32)  # noqa: E731
    exchange = {"lb": i32(off[0], 4 * S).view(S, 4), "hist": i32(off[1], off[2]),
                "ties": i32(off[3], off[4] * world).view(world, off[4])}
    fb = torch.zeros(1, dtype=torch.int32, device=xs.device)
    base = ws.data_ptr()
    _wait_last_csc(xs.device)

    def stages(lo, hi):
        call("clt_select_lists_ext_f32", lo, hi, world, rank, f_off, Fg, prev_T.data_ptr(), float(margin),
             _chk(drift).data_ptr() if drift is not None else None, Bn, L, F, k, mode, threshold.data_ptr() if threshold is not None else None, float(e or 0.0), s.cap, s.row_start.data_ptr(),
             s.ent_col.data_ptr(), s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(), s.csc_val.data_ptr(),
             s.csc_ent.data_ptr(), s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(), fb.data_ptr(), base,
             ws.numel(), _stream())

    def gemm():
        rc = lib.clt_gemm_bf16_select(xs.data_ptr(), Wt.data_ptr(), Bn, F, d, L * d, d, L * F, L, d, d * F, F,
                                      b.data_ptr() if b is not None else None, F, base + off[0], base + off[7], base + off[8],
                                      off[9], int(mode == SELECT_LAYER), GEMM_ROUND_ACC if AUTOCAST_PRE else 0, _stream())
        if rc != 0:
            raise CltError("clt_gemm_bf16_select", rc)

    run_sharded_selection(stages, shard, S, exchange, after_stage0=gemm)      # ... stage 5: outcome flag published
    # the step's host read — did the bound from the last step hold? — goes through a side stream behind an event, so it
    # completes while stage 6 (emit + list construction) is still running and the launch queue never drains
    ev = torch.cuda.Event()
    ev.record()
    if speculate is not None or defer:
        s.row_start.zero_()                  # (a void attempt emits nothing: speculative readers then see no entries)
        s.col_start.zero_()
    stages(6, 6)                              # ... token-major lists: what the gather forward reads
    # stage 7, the feature-major lists (scan, fill, column sorts: short latency-bound launches, half a millisecond on heavy-tailed
    # lists), on a second stream beside the speculated forward; confirm() — called before anything reads col_start / csc_* /
    # fired — makes the launch stream wait for it
    csc_done = _fork_csc(xs.device, lambda: stages(7, 7), speculate is not None)
    spec = speculate(s) if speculate is not None else None
    side, host = _flag_reader(xs.device)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        host.copy_(fb, non_blocking=True)
    T = i32(off[6], 4 * S).view(S, 4)[:, 0].clone()
    s.used_fallback = fb
    s.kth = T
    s.speculated = spec

    def confirm() -> bool:
        if csc_done is not None:
            torch.cuda.current_stream(xs.device).wait_event(csc_done)
        side.synchronize()
        return int(host) == 0

    s.confirm = confirm
    if defer:
        return s, T
    if not confirm():
        return None, None
    return s, T


#: fused encoder + BatchTopK step: the feature-major half of the list build on a stream of its own beside the gather forward
#: (CLT_AB=list_csc_stream=0: on the launch stream, in front of it)
LIST_CSC_STREAM = bool(ab_switch("list_csc_stream", 1))
_csc_streams = {}
_csc_last = {}             # device -> event of the last stage 7 queued on the second stream


_select_ws = {}


def _select_workspace(nbytes: int, device) -> torch.Tensor:
    key = device.index or 0
    buf = _select_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        last = _csc_last.get(key)
        if last is not None:                  # (the buffer about to be dropped may still be read by a queued stage 7)
            torch.cuda.current_stream(device).wait_event(last)
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _select_ws[key] = buf
    return buf


def _fork_csc(device, run_stage, enabled: bool):
    """Queue `run_stage()` — the feature-major half of a list build — on the second stream behind what the launch stream holds
    so far; returns the event that marks its end (None: it ran on the launch stream)."""
    if not (enabled and LIST_CSC_STREAM and TIMING is None):
        run_stage()
        return None
    key = device.index or 0
    main, cs = torch.cuda.current_stream(device), _csc_stream(device)
    ev = torch.cuda.Event()
    ev.record(main)
    cs.wait_event(ev)
    with torch.cuda.stream(cs):
        run_stage()
        done = torch.cuda.Event()
        done.record(cs)
    _csc_last[key] = done
    return done


def _wait_last_csc(device):
    """(a step that never asked confirm(): its feature-major stage still owns the selection workspace)"""
    last = _csc_last.get(device.index or 0)
    if last is not None:
        torch.cuda.current_stream(device).wait_event(last)


def _csc_stream(dev):
    key = torch.device(dev).index or 0
    if key not in _csc_streams:
        _csc_streams[key] = torch.cuda.Stream(device=dev)
    return _csc_streams[key]


#: fused encoder + per-row TopK: every ROWS_SAMPLE_STRIDE-th feature is sampled; a row's bound is the j-th largest sampled
#: value, j chosen so that the bound exceeds the row's k-th value with probability <= ROWS_FAIL_ODDS per row
ROWS_SAMPLE_STRIDE, ROWS_FAIL_ODDS = 16, 3e-9


ROWS_SLOTS = 1024


def _rows_plan(k: int):
    """(j, slots) of the fused per-row selection for k, or None when it does not apply.  The bound fails when j of the sampled
    features are among the row's k - 1 largest: P[Binomial(k - 1, 1 / stride) >= j].  The number of a row's elements above the
    j-th largest of its sample is about stride * Gamma(j): the slot block must hold that with the same odds
    (P[Gamma(j) > slots / stride] = P[Poisson(slots / stride) < j])."""
    import math
    if not 1 <= k <= 64:
        return None
    p = 1.0 / ROWS_SAMPLE_STRIDE
    n = k - 1
    tail = lambda j: sum(math.comb(n, i) * p ** i * (1 - p) ** (n - i) for i in range(j, n + 1))  # noqa: E731
    j = next((j for j in range(1, 65) if tail(j) <= ROWS_FAIL_ODDS), None)
    if j is None:
        return None
    x = ROWS_SLOTS / ROWS_SAMPLE_STRIDE
    overflow = sum(math.exp(-x) * x ** i / math.factorial(i) for i in range(j))
    return (j, ROWS_SLOTS) if overflow <= ROWS_FAIL_ODDS else None


def encoder_select_rows_applicable(Bn: int, L: int, d: int, F: int, k: int) -> bool:
    plan = _rows_plan(k)
    Fs = F // ROWS_SAMPLE_STRIDE
    return (plan is not None and F % ROWS_SAMPLE_STRIDE == 0 and Fs % 4 == 0 and Fs >= 4 * plan[0] and d % 8 == 0 and F % 4 == 0
            and Bn * L * F < 2 ** 32)


def encoder_select_rows_fits(Bn: int, L: int, d: int, F: int, k: int, device, margin: float = 0.04) -> bool:
    """Is there HBM for the fused per-row selection's scratch (slot blocks of 8 KB per row + the sampled pre-activations) with
    `margin` of the device's memory to spare?  A model that fills the card (Llama-3-8B shape on one GPU: 283 of 288 GB) is
    better off unfused: measured there, the step with the extra 1.3 GB ran 404 instead of 228 ms — the allocator at its limit."""
    plan = _rows_plan(k)
    if plan is None:
        return False
    j, slots = plan
    need = lib.clt_select_rows_ext_workspace_bytes(Bn, L, F, k, slots, F // ROWS_SAMPLE_STRIDE, j) + 4 * Bn * L * (F // ROWS_SAMPLE_STRIDE)
    free, total = torch.cuda.mem_get_info(device)
    st = torch.cuda.memory_stats(device)
    free += st.get("reserved_bytes.all.current", 0) - st.get("allocated_bytes.all.current", 0)      # (torch's cached, unused blocks)
    return free - need >= margin * total


def encoder_select_rows_lists(xs: torch.Tensor, Wt: torch.Tensor, b: Optional[torch.Tensor], k: int, offset: int = 0,
                              speculate=None) -> SparseLists:
    """Encoder forward fused with the PerLayerTopK selection (topk.py:53-63): the pre-activation tensor is never written.
    A first, narrow GEMM on the features offset, offset + stride, ... gives every (token, layer) row a lower bound of its k-th
    largest value; the full GEMM's epilogue files the elements above their row's bound per row; the exact selection runs on
    those.  Returns the lists at once with lists.confirm() -> bool pending (False: the attempt was void — some row's bound did
    not hold or its slot block overflowed — the lists are then EMPTY and the caller reruns the unfused path).
    speculate(lists): work that only reads the lists, queued before the host read (as in encoder_select_lists)."""
    Bn, L, d = xs.shape
    F = Wt.shape[1]
    _chk(xs, op16()); _chk(Wt, op16())
    stride = ROWS_SAMPLE_STRIDE
    assert encoder_select_rows_applicable(Bn, L, d, F, k)
    Fs = F // stride
    j, slots = _rows_plan(k)
    offset = int(offset) % stride
    dev = xs.device
    dims = (Bn, L, F, k, slots, Fs, j)
    ws = _select_workspace(lib.clt_select_rows_ext_workspace_bytes(*dims), dev)     # (its own: stage 3 runs on a second stream)
    off = [lib.clt_select_rows_ext_info(*dims, w) for w in range(4)]
    base = ws.data_ptr()
    _wait_last_csc(dev)
    s = _empty_lists(Bn, L, F, max(1, k * Bn * L), dev)
    fb = torch.zeros(1, dtype=torch.int32, device=dev)
    pre_s = torch.empty(Bn, L, Fs, dtype=torch.float32, device=dev)
    bias_s = b[:, offset::stride].contiguous() if b is not None else None
    _gemm("clt_gemm_bf16", xs, 0, Wt, offset * d, pre_s, 0, Bn, Fs, d, L * d, stride * d, L * Fs, False, True, batch=L, sA=d, sB=d * F,
          sC=Fs, bias=bias_s, sBias=Fs, round_acc=AUTOCAST_PRE)

    def stages(lo, hi):
        call("clt_select_rows_ext_f32", lo, hi, pre_s.data_ptr(), *dims, s.cap, s.row_start.data_ptr(), s.ent_col.data_ptr(),
             s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(), s.csc_val.data_ptr(), s.csc_ent.data_ptr(),
             s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(), fb.data_ptr(), base, ws.numel(), _stream())

    stages(0, 0)
    rc = _timed_raw("gemm", 2.0 * Bn * F * d * L, "clt_gemm_bf16_select_rows", xs.data_ptr(), Wt.data_ptr(), Bn, F, d, L * d, d, L, d, d * F,
                    b.data_ptr() if b is not None else None, F, base + off[0], base + off[1], base + off[2], base + off[3], slots,
                    GEMM_ROUND_ACC if AUTOCAST_PRE else 0, _stream())
    if rc != 0:
        raise CltError("clt_gemm_bf16_select_rows", rc)
    stages(1, 1)                         # outcome flag published: the host read below completes while stage 2 runs
    ev = torch.cuda.Event()
    ev.record()
    stages(2, 2)                         # ... selection + token-major lists; stage 3 (feature-major lists) beside the forward
    csc_done = _fork_csc(dev, lambda: stages(3, 3), speculate is not None)
    spec = speculate(s) if speculate is not None else None
    side, host = _flag_reader(dev)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        host.copy_(fb, non_blocking=True)
    s.used_fallback = fb
    s.kth = None
    s.speculated = spec

    def confirm() -> bool:
        if csc_done is not None:
            torch.cuda.current_stream(dev).wait_event(csc_done)
        side.synchronize()
        return int(host) == 0

    s.confirm = confirm
    return s


def new_select_drift(S: int, device) -> torch.Tensor:
    """Initial state of the adaptive candidate bound: no history (the bound starts at its floor and tightens over ~30 steps)."""
    return torch.zeros(2 * S, device=device)


class FlagProbe:
    """Host view of a device int32 flag that never blocks: the copy to pinned memory is queued on the current stream now;
    value() returns the flag once that copy has completed, None before."""

    def __init__(self, flag: torch.Tensor):
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.host.copy_(flag, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record()

    def value(self) -> Optional[int]:
        return int(self.host) if self.ev.query() else None


_FLAG_READERS = {}


def _flag_reader(device):
    """(side stream, pinned int32 [1]) per device for early reads of a device flag."""
    key = torch.device(device).index
    if key not in _FLAG_READERS:
        _FLAG_READERS[key] = (torch.cuda.Stream(device=device), torch.zeros(1, dtype=torch.int32).pin_memory())
    return _FLAG_READERS[key]


def selection_kth_keys(Bn, L, F, k, mode, device) -> torch.Tensor:
    """int32 keys [S] of the exact k-th values left in the workspace by the last single-GPU ops.select_lists call
    with these dimensions (valid when its fast path ran)."""
    dims = (Bn, L, F, k, mode, 1, 0)
    ws = workspace(lib.clt_select_lists_sharded_workspace_bytes(*dims), device)
    S = L if mode == SELECT_LAYER else 1
    o = lib.clt_select_lists_exchange_info(*dims, 6)
    return ws[o:o + 16 * S].view(torch.int32).view(S, 4)[:, 0].clone()


def select_rows_sharded(x: torch.Tensor, k: int, shard, col_mask: Optional[torch.Tensor] = None,
                        cap: Optional[int] = None, local_columns: bool = False) -> SparseLists:
    """Feature-sharded PerLayerTopK: k of shard.F_global per (b, l) row; x = this rank's [B, L, F_local].
    local_columns: x holds an arbitrary rank-local subset of columns in feature order (the compacted dead columns of
    the aux pass; widths may differ between ranks) — the exchange buffers depend on rows and k only."""
    from .parallel import run_sharded_rows_selection
    _chk(x)
    Bn, L, F = x.shape
    assert local_columns or F == shard.F_local
    if k > shard.F_global:
        raise AssertionError("k must be <= d_features")
    if min(k, F) > 2048:
        raise NotImplementedError("sharded per-row TopK supports k <= 2048")
    if cap is None:
        cap = max(1, min(k, F) * Bn * L)
    s = _empty_lists(Bn, L, F, cap, x.device)
    dims = (Bn, L, F, k, shard.world)
    ws = workspace(lib.clt_select_rows_sharded_workspace_bytes(*dims), x.device)
    off = [lib.clt_select_rows_sharded_exchange_info(*dims, w) for w in range(4)]
    i32 = lambda o, n: ws[o:o + 4 * n].view(torch.int32)  # noqa: E731
    exchange = {"counts": i32(off[0], off[1]), "ties": i32(off[2], off[3] * shard.world).view(shard.world, off[3])}
    if col_mask is not None:
        _chk(col_mask, torch.uint8)

    def stages(lo, hi):
        call("clt_select_rows_sharded_f32", lo, hi, shard.world, shard.rank, shard.F_global, x.data_ptr(),
             col_mask.data_ptr() if col_mask is not None else None, Bn, L, F, k, s.cap, s.row_start.data_ptr(),
             s.ent_col.data_ptr(), s.ent_val.data_ptr(), s.col_start.data_ptr(), s.csc_tok.data_ptr(), s.csc_val.data_ptr(),
             s.csc_ent.data_ptr(), s.fired.data_ptr(), s.total.data_ptr(), s.overflow.data_ptr(), ws.data_ptr(), ws.numel(),
             _stream())

    run_sharded_rows_selection(stages, shard, exchange)
    s.used_fallback = torch.zeros(1, dtype=torch.int32, device=x.device)
    return s


def pointer_table(tensors: List[torch.Tensor]) -> torch.Tensor:
    """Device array of device pointers (one small H2D copy; cache it while the storages are stable)."""
    return _device_table([t.data_ptr() for t in tensors], tensors[0].device)


def sparse_decoder_fwd(s: SparseLists, wptr: torch.Tensor, bias: Optional[torch.Tensor], d: int, cross_layer: bool,
                       w_bf16: bool = False, layer_major: bool = False, out: Optional[torch.Tensor] = None, layers=None):
    """wptr: device table of per-target-layer weight pointers (fp32 rows, or bf16 shadow rows when w_bf16).
    layer_major: the result as [L, B, d] (every layer's slab contiguous) instead of [B, L, d].
    out / layers = (lo, hi): write only the target layers [lo, hi) of a result tensor allocated earlier."""
    if out is None:
        out = torch.empty((s.L, s.B, d) if layer_major else (s.B, s.L, d), dtype=torch.float32, device=wptr.device)
    ts, ls = (d, s.B * d) if layer_major else (s.L * d, d)
    lo, hi = layers if layers is not None else (0, s.L)
    call("clt_sparse_decoder_fwd_f32", s.row_start.data_ptr(), s.ent_col.data_ptr(), s.ent_val.data_ptr(),
         wptr.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), s.B, s.L, s.F, d,
         int(cross_layer), s.cap, int(w_bf16), ts, ls, lo, hi, _stream())
    return out


#: columns (features) with more list entries than this are "hot": their weight rows go through the multi-wave launch of the row
#: pass (clt_sparse_hot_columns / the hot_* arguments of clt_sparse_row_pass_f32).  0: off (A/B: CLT_AB=hot_min_len=0)
HOT_MIN_LEN = ab_switch("hot_min_len", 64)
#: segment slots of the hot-row scratch (64 entries each; 3.3 KB of partial sums a slot at d = 768: 105 MB per stream)
HOT_SEG_SLOTS = 32768


_HOT_SEEN = {}            # (device, B, L, F) -> [pinned int32 [1] with the last lists' hot-column count, event or None]


def hot_columns(s: SparseLists):
    """(hot block int32 [2 cap + 2 seg_cap], counts int32 [2], cap, scratch by stream, seg_cap) of the lists' columns longer than HOT_MIN_LEN (the columns, their
    segment slots and the segment table: include/clt_hip.h), computed once per lists object (one small kernel) and kept on it;
    None when switched off — or when the LAST lists of this shape on this device had no such column: the count is copied to
    pinned memory behind the kernel and read a step later (it has long arrived: no stall), and a workload without hot columns
    pays for neither the extra launches of every row pass nor its skip test.  The row passes are correct with or without the
    list (a hot row then takes one wave); only the summation order of hot rows depends on it, and it is a deterministic
    function of the previous step.  (Within the hot-row form the order does not depend on the segment table's size: a column
    that found no slots is summed by one wave in the segmented order.)"""
    hot = getattr(s, "hot", None)
    if hot is None and HOT_MIN_LEN > 0:
        dev = s.col_start.device
        cap = s.cap // (HOT_MIN_LEN + 1) + 1                 # more hot columns than this cannot exist
        seg_cap = min(2 * cap, HOT_SEG_SLOTS)                # (segments beyond it: those columns are walked whole, by one wave)
        lst = torch.empty(2 * cap + 2 * seg_cap, dtype=torch.int32, device=dev)
        cnt = torch.empty(2, dtype=torch.int32, device=dev)
        call("clt_sparse_hot_columns", s.col_start.data_ptr(), s.L * s.F, HOT_MIN_LEN, lst.data_ptr(), cap, seg_cap, cnt.data_ptr(), _stream())
        seen = _HOT_SEEN.setdefault((dev.index, s.B, s.L, s.F), [torch.zeros(2, dtype=torch.int32).pin_memory(), None])
        enabled = True
        if seen[1] is not None:
            seen[1].synchronize()
            enabled = int(seen[0][0]) > 0
        seen[0].copy_(cnt, non_blocking=True)
        seen[1] = torch.cuda.Event()
        seen[1].record()
        hot = s.hot = (lst, cnt, cap, {}, seg_cap) if enabled else False
    return hot or None


def _hot_args(s, d: int = 0, nmat: int = 1):
    """The hot-row arguments of the row pass entry points.  The segment scratch (`hot_partials`) is per stream — passes of
    different matrices may run on two streams at once — and grows to the largest (d, nmat) asked for."""
    hot = hot_columns(s)
    if hot is None:
        return (None, None, 0, 0, 0, None)
    part = None
    if 0 < d <= 1024 and d % 4 == 0:
        need = nmat * hot[4] * (-(-d // 256) * 256 + 64)
        key = torch.cuda.current_stream().cuda_stream
        part = hot[3].get(key)
        if part is None or part.numel() < need:
            part = hot[3][key] = torch.empty(need, dtype=torch.float32, device=s.col_start.device)
    return (hot[0].data_ptr(), hot[1].data_ptr(), hot[2], HOT_MIN_LEN, hot[4], part.data_ptr() if part is not None else None)


def sparse_row_pass(W, s: SparseLists, coef, src, col0: int, src_layer: int, *, grad_out=None, adam=None, m=None, v=None,
                    shadow=None, dots=None, dot_slot=0, colsum=None, second=None, dots2=None, extra=None, row_term=None,
                    src_slab=False, sq_out=None, hot_phase=0):
    """See clt_sparse_row_pass_f32.  W: contiguous [..., d] weight rows.  `adam` = dict(lr, beta1, beta2, eps,
    weight_decay, adamw, step) for the in-place update, or grad_out for the gradient only.
    `second` = (lists2, coef2, src2): an additional (CSC list, coefficient, source) contribution whose
    per-entry dots go to `dots2` ([slots, lists2.cap]).  `extra` = (compact gradient [n, d] fp32, map int32 [L*F],
    compact colsum or None): dense contribution of the compacted TopK-aux pass for the mapped rows.
    `row_term` = (dnorm [L*F], sq [L*F]): adds (dnorm[col] / sqrt(sq[col])) * W[row] to the row's gradient.
    `src_slab`: `src` is ONE layer's slab [B, d] (layer `src_layer` of a layer-major source) instead of [B, L, d].
    `sq_out` ([rows] fp32; Adam form, d <= 1024): squared length of every UPDATED row (= row_sqnorm of the new matrix).
    `hot_phase`: 1 = the common launch alone, 2 = the hot rows' launches alone (for a caller that queues the two on different
    streams, engine.HOT_ROWS_STREAM); without a hot-column list phase 1 is the whole pass and phase 2 nothing."""
    if hot_phase and hot_columns(s) is None:
        if hot_phase == 2:
            return
        hot_phase = 0
    _chk(W); _chk(coef)
    if sq_out is not None:
        _chk(sq_out)
        assert grad_out is None and W.shape[-1] <= 1024 and sq_out.numel() >= W.numel() // W.shape[-1]
    rdn, rsq = row_term if row_term is not None else (None, None)
    if rdn is not None:
        _chk(rdn); _chk(rsq)
        assert rdn.numel() >= col0 + W.numel() // W.shape[-1] and rsq.numel() >= rdn.numel()
    ex, exmap, excs = extra if extra is not None else (None, None, None)
    if ex is not None:
        _chk(ex); _chk(exmap, torch.int32)
        assert ex.shape[-1] == W.shape[-1]
    src_bf16 = src.dtype in _DT16
    _chk(src, src.dtype)
    d = W.shape[-1]
    rows = W.numel() // d
    a = adam or dict(lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, adamw=False, step=1)
    s2, c2, x2 = second if second is not None else (None, None, None)
    src_ptr, src_L = src.data_ptr(), s.L
    if src_slab:
        # the kernel reads src + tok * (L * d) + layer * d: with L = 1 and the base moved back by layer * d elements that
        # is slab + tok * d (the moved base itself is never dereferenced)
        assert src_layer >= 0 and second is None and src.shape == (s.B, d)
        src_ptr, src_L = src.data_ptr() - src_layer * d * src.element_size(), 1
    if x2 is not None and (x2.dtype in _DT16) != src_bf16:
        raise TypeError("both gather sources must have the same dtype")
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    _timed_call("row_pass", (rows, d, col0, src_layer, grad_out is None, 2 if src_bf16 else 4, shadow is not None), "clt_sparse_row_pass_f32", W.data_ptr(), ptr(m), ptr(v), ptr(shadow), ptr(grad_out), rows, d, s.F, src_L, col0,
         src_layer, s.col_start.data_ptr(), s.csc_tok.data_ptr(), coef.data_ptr(), src_ptr,
         ptr(s2.col_start) if s2 is not None else None, ptr(s2.csc_tok) if s2 is not None else None, ptr(c2), ptr(x2),
         ptr(dots), ptr(dots2), s.cap, s2.cap if s2 is not None else 0, dot_slot, ptr(colsum), float(a["lr"]),
         float(a["beta1"]), float(a["beta2"]), float(a["eps"]), float(a["weight_decay"]), int(a["adamw"]),
         int(a["step"]), int(src_bf16), ptr(ex), ptr(exmap), ptr(excs), ptr(rdn), ptr(rsq), ptr(sq_out), *_hot_args(s, d), int(hot_phase),
         _stream())


def sparse_row_pass_group(Ws, s: SparseLists, coef, srcs, layers, *, adam, ms, vs, shadows=None, dots=None, src_slab=False) -> bool:
    """clt_sparse_row_pass_group_f32: the fused-Adam row pass of several matrices of one cross-layer decoder in ONE launch
    (matrix i: rows Ws[i].numel() / d, target layer layers[i], gather source srcs[i] — a [B, L, d] tensor, or with
    `src_slab` that layer's slab [B, d]).  False when the shape does not qualify (the caller launches per matrix)."""
    import ctypes as C
    n = len(Ws)
    d = Ws[0].shape[-1]
    src_bf16 = srcs[0].dtype in _DT16
    ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    for w, m, v, x in zip(Ws, ms, vs, srcs):
        _chk(w); _chk(m); _chk(v); _chk(x, srcs[0].dtype)
    rows = (C.c_int * n)(*[w.numel() // d for w in Ws])
    if src_slab:             # (see sparse_row_pass: L = 1 and the base moved back by layer * d elements)
        sp = (C.c_void_p * n)(*[x.data_ptr() - l * d * x.element_size() for x, l in zip(srcs, layers)])
        src_L = 1
    else:
        sp, src_L = ptrs(srcs), s.L
    payload = [(w.numel() // d, d, 0, l, True, 2 if src_bf16 else 4, shadows is not None) for w, l in zip(Ws, layers)]
    rc = _timed_raw("row_pass_group", payload, "clt_sparse_row_pass_group_f32", n, ptrs(Ws), ptrs(ms), ptrs(vs),
                    ptrs(shadows) if shadows is not None else None, rows, (C.c_int * n)(*layers), sp, d, s.F, src_L, s.col_start.data_ptr(),
                    s.csc_tok.data_ptr(), _chk(coef).data_ptr(), dots.data_ptr() if dots is not None else None, s.cap, float(adam["lr"]),
                    float(adam["beta1"]), float(adam["beta2"]), float(adam["eps"]), float(adam["weight_decay"]), int(adam["adamw"]),
                    int(adam["step"]), int(src_bf16), *_hot_args(s, d, n), _stream())
    if rc == 1:
        return False
    if rc != 0:
        raise CltError("clt_sparse_row_pass_group_f32", rc)
    return True


def compact_columns(dead_u8: torch.Tensor):
    """Dead-column compaction maps (clt_compact_columns).  Returns (n_pad, idx int32 [L, n_pad], map int32 [L*F]) or
    None when no feature is dead.  One host read (the per-layer counts)."""
    _chk(dead_u8, torch.uint8)
    L, F = dead_u8.shape
    counts = torch.empty(L, dtype=torch.int32, device=dead_u8.device)
    call("clt_compact_columns", dead_u8.data_ptr(), L, F, 0, counts.data_ptr(), None, None, _stream())
    nmax = int(counts.max())
    if nmax == 0:
        return None
    n_pad = (nmax + 63) // 64 * 64
    idx = torch.empty(L, n_pad, dtype=torch.int32, device=dead_u8.device)
    cmap = torch.empty(L * F, dtype=torch.int32, device=dead_u8.device)
    call("clt_compact_columns", dead_u8.data_ptr(), L, F, n_pad, counts.data_ptr(), idx.data_ptr(), cmap.data_ptr(), _stream())
    return n_pad, idx, cmap


def gather_cols(x: torch.Tensor, idx: torch.Tensor, fill: float) -> torch.Tensor:
    """x: fp32 [B, L, F]; idx: int32 [L, n_pad] -> [B, L, n_pad]."""
    _chk(x); _chk(idx, torch.int32)
    Bn, L, F = x.shape
    n_pad = idx.shape[1]
    out = torch.empty(Bn, L, n_pad, dtype=torch.float32, device=x.device)
    call("clt_gather_cols_f32", x.data_ptr(), idx.data_ptr(), Bn * L, L, F, n_pad, float(fill), out.data_ptr(), _stream())
    return out


def encoder_fwd_columns(xs: torch.Tensor, Wt32: torch.Tensor, b: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """gather_cols(encoder_fwd(xs, Wt, b), idx, -inf) without reading the [B, L, F] pre-activations again: the encoder GEMM
    on the gathered weight rows (xs 16-bit [B, L, d]; Wt32 fp32 [L, F, d]; idx int32 [L, n_pad], -1 = pad).  Every output
    element of the NT kernel accumulates over k in an order that does not depend on its place in the tile grid, so the values
    are bit-identical to the full GEMM's as long as both launches take the same kernel (the caller checks the shape
    conditions that decide that); a pad column is 0 * x + (-inf).  The compacted TopK-aux pass: 1.5 -> 0.55 ms at 14 % dead."""
    L, F, d = Wt32.shape
    n_pad = idx.shape[1]
    Wc = gather_rows(Wt32, idx)                                  # zero rows at the pads
    if xs.dtype in _DT16:
        Wc = cast_bf16(Wc)
    bc = gather_cols(b.reshape(1, L, F), idx, float("-inf")).view(L, n_pad)
    return encoder_fwd(xs, Wc, bc)


def gather_rows(W: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """W: fp32 [groups, F, d]; idx: int32 [>= groups, n_pad] (group g uses row g) -> [groups, n_pad, d]."""
    _chk(W); _chk(idx, torch.int32)
    groups, F, d = W.shape
    n_pad = idx.shape[1]
    assert idx.shape[0] >= groups
    out = torch.empty(groups, n_pad, d, dtype=torch.float32, device=W.device)
    call("clt_gather_rows_f32", W.data_ptr(), idx.data_ptr(), groups, F, n_pad, d, out.data_ptr(), _stream())
    return out


def sparse_reduce_dots(dots, s: SparseLists, cross_layer: bool):
    dval = torch.empty(s.cap, dtype=torch.float32, device=dots.device)
    call("clt_sparse_reduce_dots_f32", dots.data_ptr(), s.cap, s.col_start.data_ptr(), s.L, s.F, int(cross_layer),
         dval.data_ptr(), _stream())
    return dval


def sparse_scatter_dense(s: SparseLists, val):
    out = torch.empty(s.B, s.L, s.F, dtype=torch.float32, device=val.device)
    call("clt_sparse_scatter_dense_f32", s.col_start.data_ptr(), s.csc_tok.data_ptr(), val.data_ptr(), s.B, s.L * s.F,
         out.data_ptr(), _stream())
    return out
