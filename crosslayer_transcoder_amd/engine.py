"""Fused training-step engine: loss and ALL parameter gradients of one step, computed by explicit calls
into the HIP kernels (no autograd graph), in the order the reference's autograd would produce them.

Mirrors, op for op, reference model/clt_lightning.py:290-312 (base), :420-475 (JumpReLU module) and
:519-567 (TopK module with auxiliary loss).  Gradient formulas are those validated in the CPU oracle
against the reference's autograd (oracle/clt_oracle.py:training_step — not imported here).

Two regimes:
  * dense  (ReLU / JumpReLU): encoder, decoder fwd/dgrad/wgrad are MFMA GEMMs; Adam is a separate pass.
  * sparse (TopK family): after the select the active entries are compacted into token-major (CSR) and
    feature-major (CSC) lists.  The decoder forward is a row gather; the decoder and encoder backward are
    ONE streaming pass over each weight matrix that forms the row's gradient from its feature's token list,
    emits the d(loss)/d(feature) dots and — when an optimizer is handed in — applies Adam in place, so the
    dense gradient tensors (7.9 GB for the GPT-2 32k CLT) never exist.
"""
from __future__ import annotations

from typing import Dict, Optional

import weakref

import torch

from . import ops
from .model.clt import CrosslayerDecoder, Decoder
from .model.jumprelu import JumpReLU, ReLU
from .model.topk import BatchTopK, PerLayerBatchTopK, PerLayerTopK
from .precision import LossScaler, ab_check, ab_switch, get_precision, mixed16


def _op(t):
    return ops.cast_bf16(t.contiguous()) if mixed16() else t.contiguous()


def _select(nl, pre):
    """Training forward of a TopK-family module on dense pre-activations (updates its EMA buffer)."""
    assert nl.k <= pre.shape[-1]
    if isinstance(nl, PerLayerTopK):
        return ops.topk_rows(pre, nl.k)
    if nl.k > 0 and nl.e is None:
        raise TypeError("BatchTopK needs `e` (threshold EMA rate) for training")
    return ops.topk_segmented(pre, nl.k, isinstance(nl, PerLayerBatchTopK), nl.threshold if nl.k > 0 else None,
                              nl.e or 0.0)


def _select_lists(nl, pre, col_mask=None, shard=None):
    """TopK-family training forward straight to the active-entry lists (no dense feature tensor).  `col_mask`
    (uint8 [L,F]) restricts the candidates to the masked-in features: where(mask, pre, -inf) (clt_lightning.py:544-545)."""
    Bn, L, F = pre.shape
    assert nl.k <= (F if shard is None else shard.F_global)
    if shard is not None:
        if isinstance(nl, PerLayerTopK):
            return ops.select_rows_sharded(pre, nl.k, shard, col_mask)
        if nl.k > 0 and nl.e is None:
            raise TypeError("BatchTopK needs `e` (threshold EMA rate) for training")
        mode = ops.SELECT_LAYER if isinstance(nl, PerLayerBatchTopK) else ops.SELECT_BATCH
        return ops.select_lists_sharded(pre, nl.k, mode, shard, nl.threshold if nl.k > 0 else None, nl.e or 0.0, col_mask)
    if isinstance(nl, PerLayerTopK):
        if nl.k > 2048:                         # compact-row buffer limit of the rows kernel: go through the dense result
            x = pre if col_mask is None else ops.mask_dead(pre, col_mask)
            return ops.sparse_build(ops.topk_rows(x, nl.k), _list_capacity(nl, Bn, L, F))
        return ops.select_lists(pre, nl.k, ops.SELECT_ROWS, col_mask=col_mask)
    if nl.k > 0 and nl.e is None:
        raise TypeError("BatchTopK needs `e` (threshold EMA rate) for training")
    mode = ops.SELECT_LAYER if isinstance(nl, PerLayerBatchTopK) else ops.SELECT_BATCH
    return ops.select_lists(pre, nl.k, mode, nl.threshold if nl.k > 0 else None, nl.e or 0.0, col_mask)


class StepOutput(dict):
    """Result of training_step.  In the sparse regime the dense `features` tensor is not part of the computation —
    nor, when the encoder GEMM is fused with the selection, the pre-activations: both are materialised on first
    access (tests, metric logging)."""

    lazy = None

    def __missing__(self, key):
        if key == "features" and "lists" in self:
            S = self["lists"]
            self[key] = ops.sparse_scatter_dense(S, S.csc_val)
            return self[key]
        if self.lazy and key in self.lazy:
            self[key] = self.lazy[key]()
            return self[key]
        raise KeyError(key)


#: counters the benchmark reports: steps whose encoder GEMM carried the candidate filter, and void attempts among them
#: (the bound from the last step did not hold: the step reran the unfused GEMM + selection)
STATS = {"fused_select_steps": 0, "select_fallbacks": 0}

#: dense 16-bit JumpReLU step of a per-layer decoder: the fused forward pass writes no transposed feature copy (a third of its
#: bytes) and the decoder weight gradient reads the features as a k-major GEMM operand (ds_read_b64_tr_b16; 10 % slower than from
#: the transposed copy, same bits): PLT JumpReLU step -0.5 ms.  Not for the cross-layer decoder, whose 78 weight-gradient pairs
#: lose more than the pass gains (profiles/r03_gemm_ab.md section 11).  CLT_AB=km_wgrad_plt=0: transposed copy.
KM_WGRAD_PLT = bool(ab_switch("km_wgrad_plt", 1))
#: fuse the encoder GEMM with the BatchTopK candidate filter when last step's k-th value is known (bf16, base loss)
FUSE_ENCODER_SELECT = True
#: lower bound of the fused filter = SELECT_MARGIN * (last step's exact k-th value)
SELECT_MARGIN = 0.9
#: tighten that bound from the floor SELECT_MARGIN towards 0.98 as far as the observed step-to-step change of the k-th value
#: allows (device-side peak-hold of |T_t / T_{t-1} - 1|, clt_select_lists_ext_f32's drift state): fewer candidates in the epilogue
ADAPTIVE_SELECT_MARGIN = True
#: PerLayerTopK: fuse the encoder GEMM with the per-row selection (row bounds from a 1/16 feature sample; bf16, base loss, one GPU)
FUSE_ENCODER_ROWS = bool(ab_switch("fuse_encoder_rows", 1))
#: queue the decoder's gather forward before the host reads the fused selection's outcome flag
SPECULATIVE_FORWARD = True
#: streams the decoder's weight passes alternate over when the matrices are small (CLT_AB=rowpass_streams=1: off, for A/B runs)
ROW_PASS_STREAMS = ab_switch("rowpass_streams", 2)
ROW_PASS_SMALL = 50_000_000          # mean parameters per decoder matrix below which a pass is "short"
#: short passes of one layer group in one launch (CLT_AB=group_row_passes=0: off, for A/B runs)
GROUP_ROW_PASSES = bool(ab_switch("group_row_passes", 1))
_side_stream = ops._side_stream          # (one side stream per device, shared with ops.clt_decoder_dgrad_kseg)
#: heavy-tailed lists: the hot rows' launches of a weight pass (segments + rows, ~0.1 ms of short latency-bound work per matrix)
#: run on a stream of their own under the common launch instead of behind it (CLT_AB=hot_rows_stream=0: behind it, for A/B runs)
HOT_ROWS_STREAM = bool(ab_switch("hot_rows_stream", 1))
#: TopK-aux pass (k_aux of the DEAD features per row, second decoder pass) as a dense sub-problem on compacted dead
#: columns (csrc/compact.hip) instead of activation lists, as long as the dead columns are at most this share of F
COMPACT_AUX = True
COMPACT_AUX_MAX_SHARE = 0.5
COMPACT_AUX_FUSED_OPERANDS = True     # (False: separate cast / transpose / mask passes; A/B verdict r02: 35.0 -> 32.7 ms)
COMPACT_AUX_GEMM_COLUMNS = True       # (False: gather the dead columns from `pre`)
#: dense bf16 regime of the JumpReLU module: the sparsifier forward and the loss backward write the bf16 GEMM operands
#: (features, features^T, dpre^T) directly — no fp32 feature / dpre tensors, no separate cast / transpose / column-sum passes
#: dense bf16 JumpReLU regime: pre-activations / feature gradient stored as bf16 (CLT_AB=dense_bf16_act=0: fp32, for A/B runs)
DENSE_BF16_ACTIVATIONS = bool(ab_switch("dense_bf16_act", 1))
#: ... and in the steps that build activation lists from them (CLT_AB=list_bf16_act=0: fp32 pre-activations there)
LIST_BF16_ACTIVATIONS = bool(ab_switch("list_bf16_act", 1))
FUSE_DENSE_JUMPRELU = True
#: dense bf16 regime: bf16 copy + transposed bf16 copy + squared row norms of each decoder matrix in one pass over it
FUSE_WEIGHT_PREP = True
#: ... and that pass folded into the optimizer's update of the matrix (clt_adam_prep_f32: the updated values leave in all four
#: forms, the next step reads no fp32 weight for its operands); likewise the encoder's 16-bit shadow (CLT_AB=fuse_adam_prep=0: off)
FUSE_ADAM_PREP = bool(ab_switch("fuse_adam_prep", 1))
#: list regime, bf16 mode, base loss: loss sum + bf16 gradient rows + decoder bias gradient in one pass over the error
FUSE_MSE_BF16 = True
#: feature-sharded list regime, cross-layer decoder, base loss: the partial reconstruction is written layer-major and
#: exchanged as L per-layer all-reduces issued at once on RCCL's stream; the loss of layer l and the weight pass of W_l
#: (which gathers only g[:, l, :]) start as soon as slab l has arrived, so the exchange of the other layers runs under
#: them instead of in front of the whole backward pass
PIPELINE_EXCHANGE = True
#: ... in its reduce-scatter (fp32 partial sums) + all-gather (16-bit gradient rows) form, the loss pass on 1/world of the
#: tokens per rank (CLT_AB=exchange_split=0: one fp32 all-reduce per layer group, every rank computes the whole loss)
EXCHANGE_SPLIT = bool(ab_switch("exchange_split", 1))
#: thresholding sparsifiers (JumpReLU / ReLU): break-even between the activation-list regime and the dense GEMM regime as
#: a share of the B*L*F entries that are active (L0 / F).  Below it the decoder forward is a row gather and the decoder
#: weight gradient + Adam one fused row pass; above it (e.g. an untrained JumpReLU at theta = 0.03: ~45 % active) the
#: MFMA GEMMs are cheaper.  The count is taken on the device (list construction); one host read of its overflow flag.
#: Measured (scripts/archive/gpu_breakeven.sh, B = 4096, F = 32768, bf16): list step = 14.9 ms + 2.4 ms per 0.1 % active (PLT) /
#: 37.4 + 11.4 (CLT) against 19.8 / 61.5 ms dense: both cross at ~0.2 % active, i.e. L0 ~ 65 features per token and layer.
LIST_MAX_DENSITY = 1.0 / 512
#: list regime with the fused optimizer: the Adam row passes write each updated decoder row's squared length, which is
#: the next step's dec_norms (CLT at G32: 78 row_sqnorm launches = 1.36 ms of 34.8 per step otherwise)
ROW_PASS_KEEPS_NORMS = True
#: after a step whose density was above the break-even, the next attempt is made this many steps later (doubling, capped)
LIST_RETRY_FIRST, LIST_RETRY_MAX = 8, 512
ab_check()                  # (every switch of the package is registered by now: ops.py's and the ones above)


def _encoder_operand(enc):
    """Feature-major [L,F,d] encoder weight in the active precision.  In bf16 mode a persistent bf16 shadow is
    kept on the module; the fused Adam row pass refreshes it, any other in-place change of W (optimizer,
    load_state_dict) bumps the tensor version and triggers a re-cast."""
    Wt = enc.Wt
    if not mixed16():
        return Wt
    sh = getattr(enc, "_bf16_shadow", None)
    ver = (enc.W._version, getattr(enc.W, "_clt_version", 0), Wt.data_ptr(), ops.op16())
    if sh is None or enc._shadow_version != ver:
        tagged = getattr(enc.W, "_clt_shadow", None)          # written by the optimizer's own update pass (optim.add_shadow)
        if (FUSE_ADAM_PREP and sh is not None and tagged is not None and tagged[1] is sh
                and tagged[0] == (enc.W._version, getattr(enc.W, "_clt_version", 0), ops.op16())):
            enc._shadow_version = ver
        else:
            sh = ops.cast_bf16(Wt)
            enc._bf16_shadow, enc._shadow_version = sh, ver
    return sh


_hot_streams = {}


class _HotFork:
    """Weight passes whose hot rows run on a second stream (ops.sparse_row_pass hot_phase 1 / 2: disjoint rows and dot slots).
    Created AFTER everything the passes write has been allocated and the optimizer state exists (a buffer allocated or zero-
    filled on the launch stream after the fork is not ordered against the second stream); join() before anything reads the
    passes' results.  Off without hot columns, under bench.py's per-launch timing, and when not asked for."""

    def __init__(self, S, dev, enabled=True):
        self.on = bool(enabled and HOT_ROWS_STREAM and ops.TIMING is None and ops.hot_columns(S) is not None)
        if self.on:
            key = torch.device(dev).index or 0
            if key not in _hot_streams:
                _hot_streams[key] = torch.cuda.Stream(device=dev)
            self.main, self.side = torch.cuda.current_stream(dev), _hot_streams[key]
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.side.wait_event(ev)

    def row_pass(self, *args, **kw):
        if not self.on:
            return ops.sparse_row_pass(*args, **kw)
        ops.sparse_row_pass(*args, hot_phase=1, **kw)
        with torch.cuda.stream(self.side):
            ops.sparse_row_pass(*args, hot_phase=2, **kw)

    def join(self):
        if self.on:
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)
            self.on = False


class _CompactDecoder:
    """Stand-in for a decoder module restricted to n_pad gathered feature rows per source layer (the compacted
    TopK-aux pass): what _Decoder needs of a decoder, on the gathered weights."""

    def __init__(self, dec, Wc, n_pad):
        self.n_layers, self.d_acts, self.d_features, self.b, self._W = dec.n_layers, dec.d_acts, n_pad, dec.b, Wc

    def weights(self):
        return self._W


class _Decoder:
    """Decoder passes for either decoder class.  Dense passes take fp32 tensors and form their GEMM operands in
    the active precision: fp32 -> the exact-fp32 MFMA kernel on the natural layouts; bf16 -> every contraction in
    the k-contiguous "NT" form of the LDS-DMA MFMA kernel, with transposed bf16 copies where the natural layout has
    the contraction index outermost (decoder forward: weights^T; weight gradients: features^T and g^T)."""

    def __init__(self, dec, shard=None, shadow=False, clt=None):
        self.dec = dec
        self.shard = shard
        #: sparse regime, bf16 mode, Adam fused into the row pass: the gather forward reads bf16 shadow copies of the
        #: decoder matrices that the same row pass keeps current (half the gathered bytes)
        self.shadow = shadow
        self.clt = isinstance(dec, CrosslayerDecoder) if clt is None else clt
        self.L = dec.n_layers
        self.W32 = dec.weights()
        self.bf16 = mixed16()            # (16-bit operands, either format)
        self._Wop = self._WT = self._sq = None

    def _prep(self):
        """bf16 dense regime: bf16 copy, transposed bf16 copy and squared row norms of every decoder matrix from ONE
        read of it (`clt_weight_prep_bf16`) instead of three passes over the decoder weights."""
        if self._Wop is not None or self._WT is not None or not (self.bf16 and FUSE_WEIGHT_PREP):
            return
        F = self.dec.d_features
        sq = torch.zeros(self.L * F, dtype=torch.float32, device=self.W32[0].device)
        if FUSE_ADAM_PREP:
            # the optimizer's update pass of the last step left all three forms of every UPDATED matrix behind (optim.add_prep,
            # clt_adam_prep_f32): same bits as the preparation pass below would produce, no read of the fp32 weights
            tags = [getattr(w, "_clt_prep", None) for w in self.W32]
            if all(t is not None and t[0] == (w._version, getattr(w, "_clt_version", 0), ops.op16()) for t, w in zip(tags, self.W32)):
                for t in tags:                  # ascending l: the reference's accumulation order (:440-444)
                    sq[:t[3].numel()] += t[3]
                self._Wop, self._WT, self._sq = [t[1] for t in tags], [t[2] for t in tags], sq
                return
        Wop, WT = [], []
        for w in self.W32:                      # ascending l: the reference's accumulation order (:440-444)
            r = ops.weight_prep_bf16(w, sq, True) if w.is_contiguous() else None
            if r is None:
                return
            Wop.append(r[0]); WT.append(r[1])
        self._Wop, self._WT, self._sq = Wop, WT, sq

    def register_prep(self, optimizer, enc=None):
        """Dense 16-bit regime with a FusedAdam: the coming optimizer.step() refreshes this step's operand buffers in its own
        pass over each decoder matrix (and the encoder's 16-bit shadow), so the next step's _prep / _encoder_operand read no
        fp32 weights."""
        if not (FUSE_ADAM_PREP and self.bf16 and hasattr(optimizer, "add_prep")):
            return
        if self._Wop is not None and self._WT is not None and not self.shadow:
            for w, op, opT in zip(self.W32, self._Wop, self._WT):
                if w.is_contiguous() and w.dim() == 3:
                    optimizer.add_prep(w, op, opT, torch.empty(w.shape[0] * w.shape[1], dtype=torch.float32, device=w.device))
        if enc is not None and getattr(enc, "_bf16_shadow", None) is not None:
            optimizer.add_shadow(enc.W, enc._bf16_shadow)

    @property
    def Wop(self):
        if self.shadow:                     # list regime: the bf16 copies the fused Adam row pass keeps current
            return self.shadows()
        self._prep()
        if self._Wop is None:
            self._Wop = [_op(w) for w in self.W32]
        return self._Wop

    @property
    def WT(self):
        """bf16 [n, d, F] transposed copies of the decoder matrices (forward B operand, k = feature contiguous)."""
        self._prep()
        if self._WT is None:
            self._WT = [ops.cast_transpose_bf16(w.contiguous()) for w in self.W32]
        return self._WT

    def names(self):
        return [f"decoder.W_{l}" for l in range(self.L)] if self.clt else ["decoder.W"]

    def _bias(self):
        """The (replicated) bias enters the all-reduced sum exactly once."""
        return self.dec.b if self.shard is None or self.shard.rank == 0 else None

    def _reduce(self, r):
        return r if self.shard is None else self.shard.all_reduce_sum(r)

    def fwd(self, feat, feat_op=None):
        """feat: fp32 [B, L, F]; feat_op: its bf16 copy when the caller already has one (then feat may be None)."""
        if self.bf16:
            fo = feat_op if feat_op is not None else ops.cast_bf16(feat)
            r = ops.clt_decoder_fwd_t(fo, self.WT, self._bias()) if self.clt else ops.plt_decoder_fwd_t(fo, self.WT[0], self._bias())
        else:
            r = ops.clt_decoder_fwd(feat, self.Wop, self._bias()) if self.clt else ops.plt_decoder_fwd(feat, self.Wop[0], self._bias())
        return self._reduce(r)

    def dgrad(self, g, go=None, out_bf16=False):
        """Feature gradient [B, L, F]; out_bf16: as bf16 when the bf16 kernels take the shape (else fp32 as always)."""
        go = _op(g) if go is None else go
        if self.clt and self.bf16:          # one K-segmented GEMM per source layer: the result is written once
            df = ops.clt_decoder_dgrad_kseg(go, self.Wop, out_bf16)
            if df is None and out_bf16:
                df = ops.clt_decoder_dgrad_kseg(go, self.Wop)
            if df is not None:
                return df
        if out_bf16 and self.bf16 and not self.clt:
            df = ops.plt_decoder_dgrad_out16(go, self.Wop[0])
            if df is not None:
                return df
        return ops.clt_decoder_dgrad(go, self.Wop) if self.clt else ops.plt_decoder_dgrad(go, self.Wop[0])

    def wgrad(self, feat, g, grads: Dict[str, torch.Tensor], featT=None, feat_op=None):
        """grads[name] (+)= feat^T @ g; accumulates when the entry already exists.  featT: bf16 [L, F, B] when the
        caller already has the transposed operand (then feat may be None); feat_op (per-layer decoder): the 16-bit features
        [B, L, F] as they lie — the GEMM takes them as its k-major A operand (same bits as through featT)."""
        Bn, d = g.shape[0], g.shape[2]
        L, F = self.L, self.dec.d_features
        if self.bf16:
            gT = ops.transpose_cast_2d(g).view(L, d, Bn)             # [L, d, B]
            if featT is None and feat_op is not None and not self.clt:
                acc = "decoder.W" in grads
                out = grads["decoder.W"] if acc else torch.empty(L, F, d, dtype=torch.float32, device=g.device)
                grads["decoder.W"] = ops.wgrad_ta(feat_op, gT, out, acc)
                return
            fT = featT if featT is not None else ops.transpose_cast_2d(feat).view(L, F, Bn)          # [L, F, B]
            if self.clt and not any(f"decoder.W_{l}" in grads for l in range(L)):
                outs = [torch.empty(l + 1, F, d, dtype=torch.float32, device=g.device) for l in range(L)]
                if ops.clt_wgrad_t_grouped(fT, gT, outs):           # all (target, source) pairs in one persistent launch
                    for l in range(L):
                        grads[f"decoder.W_{l}"] = outs[l]
                    return
            if self.clt:
                for l in range(L):
                    k = f"decoder.W_{l}"
                    acc = k in grads
                    out = grads[k] if acc else torch.empty(l + 1, F, d, dtype=torch.float32, device=g.device)
                    grads[k] = ops.wgrad_t(fT, gT[l], out, l + 1, F * Bn, 0, acc)
            else:
                acc = "decoder.W" in grads
                out = grads["decoder.W"] if acc else torch.empty(L, F, d, dtype=torch.float32, device=g.device)
                grads["decoder.W"] = ops.wgrad_t(fT, gT, out, L, F * Bn, d * Bn, acc)
            return
        if self.clt:
            for l in range(self.L):
                k = f"decoder.W_{l}"
                grads[k] = ops.clt_decoder_wgrad(feat, g, l, grads.get(k), accumulate=k in grads)
        else:
            grads["decoder.W"] = ops.plt_decoder_wgrad(feat, g, grads.get("decoder.W"), accumulate="decoder.W" in grads)

    def _version_key(self):
        return [(w._version, getattr(w, "_clt_version", 0), w.data_ptr(), ops.op16()) for w in self.W32]

    def sqnorms(self):
        if self._sq is not None:
            return self._sq
        F = self.dec.d_features
        sq = torch.zeros(self.L * F, dtype=torch.float32, device=self.W32[0].device)
        kept = getattr(self.dec, "_row_sq_parts", None) if ROW_PASS_KEEPS_NORMS else None
        if kept is not None and kept[0] == self._version_key():
            # list regime: last step's fused Adam row passes left every updated row's squared length behind (same bits
            # as row_sqnorm of the matrix); summed in the same ascending order, no read of the 7.85 GB of weights
            for part in kept[1]:
                sq[:part.numel()] += part
            return sq
        for w in self.W32:                      # ascending l: the reference's accumulation order (:440-444)
            ops.row_sqnorm(w, sq, 0, True)
        return sq

    def add_norm_grads(self, dn, sq, grads, optimizer=None):
        """Gradient through dec_norms.  With a FusedAdam the term is handed to the optimizer, whose update kernel adds
        it on the fly (grads then hold the reconstruction part only)."""
        for l, w in enumerate(self.W32):
            if optimizer is not None and hasattr(optimizer, "add_row_term") and w.is_contiguous():
                optimizer.add_row_term(w, dn, sq)
            else:
                ops.rowscale_add(w, dn, sq, grads[f"decoder.W_{l}" if self.clt else "decoder.W"])

    # ---- sparse (list) passes -------------------------------------------------------------------
    def shadows(self):
        """bf16 copies of the decoder matrices, cast once and then refreshed by the fused Adam row pass; any other
        in-place change of a weight (optimizer.step, load_state_dict) changes its version key and forces a re-cast."""
        ver = [(w._version, getattr(w, "_clt_version", 0), w.data_ptr(), ops.op16()) for w in self.W32]
        cached = getattr(self.dec, "_bf16_shadows", None)
        if cached is None or cached[0] != ver:
            cached = (ver, [ops.cast_bf16(w.contiguous()) for w in self.W32])
            self.dec._bf16_shadows = cached
        return cached[1]

    def pointer_table(self):
        """Device array of the L per-target-layer weight base pointers (cached while storages are stable)."""
        d, F = self.dec.d_acts, self.dec.d_features
        Ws, es, slot = (self.shadows(), 2, "_wptr_sh") if self.shadow else (self.W32, 4, "_wptr")
        ptrs = [w.data_ptr() for w in Ws] if self.clt else [Ws[0].data_ptr() + es * l * F * d for l in range(self.L)]
        cached = getattr(self.dec, slot, None)
        if cached is None or cached[0] != ptrs:
            cached = (ptrs, torch.tensor(ptrs, dtype=torch.int64).to(self.W32[0].device))
            setattr(self.dec, slot, cached)
        return cached[1]

    def sparse_fwd(self, S):
        return self._reduce(ops.sparse_decoder_fwd(S, self.pointer_table(), self._bias(), self.dec.d_acts, self.clt,
                                                   w_bf16=self.shadow))

    def sparse_wgrad(self, S, g, grads, fused, row_term=None, g16=None):
        """Decoder weight gradient alone from the activation lists (thresholding sparsifiers: d(loss)/d(feature) comes
        from the dense feature-gradient GEMM there, not from per-entry dots): one row pass per weight matrix, with the
        gradient through dec_norms (`row_term` = (dnorm, sq)) added per row, Adam in place when `fused`."""
        if self.bf16:
            g = g16 if g16 is not None else ops.cast_bf16(g)
        keep = ROW_PASS_KEEPS_NORMS and fused is not None and self.dec.d_acts <= 1024
        self.dec._row_sq_parts = None
        todo = list(zip(self.names(), self.W32))
        # (optimizer state and the kept-norm buffers exist before the hot rows' stream forks off)
        begun = [fused.begin_fused(w) for _, w in todo] if fused is not None else None
        parts = [torch.empty(w.numel() // w.shape[-1], dtype=torch.float32, device=w.device) for _, w in todo] if keep else []
        ops.hot_columns(S)
        fork = _HotFork(S, g.device, enabled=fused is not None)
        for l, (name, w) in enumerate(todo):
            kw = dict(row_term=row_term)
            if fused is not None:
                kw.update(begun[l])
                if self.shadow:
                    kw["shadow"] = self.shadows()[l]
                if keep:
                    kw["sq_out"] = parts[l]
            else:
                grads[name] = kw["grad_out"] = torch.empty_like(w)
            fork.row_pass(w, S, S.csc_val, g, 0, l if self.clt else -1, **kw)
        fork.join()
        if keep:                                # valid while nothing else changes the weights (same key as the shadows)
            self.dec._row_sq_parts = (self._version_key(), parts)

    def sparse_bwd(self, S, g, S2, g2, grads, fused, extra=None, g16=None, slabs=None, groups=None):
        """Row pass over every decoder weight row: dW (or in-place Adam) + d(loss)/d(feature) per list entry.
        Returns (dval [S.cap], dval2 [S2.cap] or None).  extra = ({name: compact dW}, map): dense gradient of the
        compacted aux pass, added to the mapped rows.  slabs (cross-layer decoder, pipelined exchange): slabs(l) returns
        layer l's gradient slab [B, d] once it can be computed; the matrices are then walked from the last layer down."""
        dev = S.csc_val.device
        slots = self.L if self.clt else 1
        ops.hot_columns(S)                      # (listed once, on this stream, before passes are dealt to other streams)
        dots = torch.empty(slots, S.cap, dtype=torch.float32, device=dev)
        dots2 = torch.empty(slots, S2.cap, dtype=torch.float32, device=dev) if S2 is not None else None
        if self.bf16 and slabs is None:    # bf16 mode: gathered g rows as bf16 (fp32 accumulate), half the gather bytes
            g = g16 if g16 is not None else ops.cast_bf16(g)      # (g16: the loss kernel already wrote the bf16 rows)
            g2 = ops.cast_bf16(g2) if g2 is not None else None
        second = (S2, S2.csc_val, g2) if S2 is not None else None
        todo = list(enumerate(zip(self.names(), self.W32)))
        # the passes of different matrices are independent (own rows, own dot slots): alternating them over two streams lets
        # one pass's tail overlap the next one's ramp (not while bench.py's roofline pass times the launches one by one)
        # — measured: 8-way shard of the headline (passes of 35..180 us) 2.67 -> 2.61 ms/step; whole model (850 us passes)
        # 15.60 -> 15.87, two bandwidth-bound kernels only get in each other's way: short passes only
        small = sum(w.numel() for w in self.W32) < ROW_PASS_SMALL * len(self.W32)
        if (GROUP_ROW_PASSES and small and self.clt and fused is not None and extra is None and S2 is None and len(todo) > 1
                and len(todo) <= 16):
            # short passes (feature shards): the matrices of a layer group — of the whole decoder without the pipelined
            # exchange — go through ONE launch (clt_sparse_row_pass_group_f32)
            kws = [fused.begin_fused(w) for _, (_, w) in todo]
            same = all(k["adam"] == kws[0]["adam"] for k in kws)
            sh = self.shadows() if self.shadow else None
            done_all = same
            # (one stream: alternating the group launches over two streams measured slower, 2.555 -> 2.588 ms)
            for gi, (lo, hi) in enumerate(groups if slabs is not None else [(0, len(todo))]):
                if not done_all:
                    break
                ls = list(range(hi - 1, lo - 1, -1)) if slabs is not None else list(range(lo, hi))
                srcs = [slabs(l) for l in ls] if slabs is not None else [g] * len(ls)
                ok = ops.sparse_row_pass_group([self.W32[l] for l in ls], S, S.csc_val, srcs, ls, adam=kws[0]["adam"],
                                               ms=[kws[l]["m"] for l in ls], vs=[kws[l]["v"] for l in ls],
                                               shadows=[sh[l] for l in ls] if sh is not None else None, dots=dots,
                                               src_slab=slabs is not None)
                if not ok:
                    assert gi == 0                       # (a shape test: it fails on the first launch or never)
                    done_all = False
            if done_all:
                return ops.sparse_reduce_dots(dots, S, self.clt), None
            begun = dict(zip((l for l, _ in todo), kws))
        else:
            begun = None
        two = ROW_PASS_STREAMS > 1 and small and ops.TIMING is None and len(todo) > 1
        hot_fork = fused is not None and slabs is None and not two
        if (two or hot_fork) and fused is not None and begun is None:
            # the optimizer creates (zero-fills) a matrix's moment buffers in begin_fused on its first step: all of them on this
            # stream BEFORE the side stream's wait point below — created inside the loop, a side-stream pass raced its own
            # matrix's zero fill (first step only; found by the heavy-tail test of round 3)
            begun = {l: fused.begin_fused(w) for l, (_, w) in todo}
        if hot_fork and self.shadow:
            self.shadows()                       # (cast on the launch stream before the fork, when they do not exist yet)
        fork = _HotFork(S, dev, enabled=hot_fork)
        if two:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        for idx, (l, (name, w)) in enumerate(reversed(todo) if slabs is not None else todo):
            kw = dict(dots=dots, dots2=dots2, second=second, dot_slot=-1 if self.clt else 0)
            if slabs is not None:
                g = slabs(l)
                kw["src_slab"] = True
                if two and idx % 2 == 1:         # (the slab was produced on the main stream)
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
            if extra is not None:
                kw["extra"] = (extra[0][name].view(-1, w.shape[-1]), extra[1], None)
            if fused is not None:
                kw.update(begun[l] if begun is not None else fused.begin_fused(w))
                if self.shadow:
                    kw["shadow"] = self.shadows()[l]
            else:
                grads[name] = kw["grad_out"] = torch.empty_like(w)
            if two and idx % 2 == 1:
                with torch.cuda.stream(side):
                    ops.sparse_row_pass(w, S, S.csc_val, g, 0, l if self.clt else -1, **kw)
            else:
                fork.row_pass(w, S, S.csc_val, g, 0, l if self.clt else -1, **kw)
        fork.join()
        if two:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)
        dval = ops.sparse_reduce_dots(dots, S, self.clt)
        dval2 = ops.sparse_reduce_dots(dots2, S2, self.clt) if S2 is not None else None
        return dval, dval2


def _threshold_lists(mod, pre, theta, force: bool, bias=None):
    """Activation lists of a thresholding sparsifier (JumpReLU: theta; ReLU: None) from its pre-activations, or None
    when more than LIST_MAX_DENSITY of the entries are active (the dense regime is then cheaper).  force: no gate."""
    Bn, L, F = pre.shape
    full = Bn * L * F
    # (split build: the host reads the gate's outcome while the lists are still being filled, and the feature-major half runs on
    # the list builds' second stream beside the gather forward; the step joins it before it reads col_start / csc_* / fired)
    if force:
        return ops.sparse_build_threshold(pre, theta, full, bias, split=True)
    if getattr(mod, "_list_skip", 0) > 0:            # recently above the break-even: no attempt this step
        mod._list_skip -= 1
        return None
    S = ops.sparse_build_threshold(pre, theta, max(1, int(LIST_MAX_DENSITY * full)), bias, split=True)
    if S.overflowed():
        S.join()                                     # (its gated-off kernels still hold the lists' memory)
        mod._list_retry = min(LIST_RETRY_MAX, 2 * getattr(mod, "_list_retry", LIST_RETRY_FIRST // 2))
        mod._list_skip = mod._list_retry
        STATS["list_gate_dense"] = STATS.get("list_gate_dense", 0) + 1
        return None
    mod._list_retry = LIST_RETRY_FIRST // 2
    return S


class _SplitExchange:
    """State of the split exchange of one step (see _forward_and_exchange): this rank's token block of the reduced
    reconstruction, the gathered gradient slabs, and the two small sums that are complete only after finish()."""

    def __init__(self, shard, L, Bn, d, g_dtype, dev):
        self.shard, self.Bc = shard, Bn // shard.world
        self.r_part = torch.empty(L, self.Bc, d, dtype=torch.float32, device=dev)
        self.g_part = torch.empty(L, self.Bc, d, dtype=g_dtype, device=dev)
        self.g_lm = torch.empty(L, Bn, d, dtype=g_dtype, device=dev)
        self.b_grad = torch.empty(L, d, dtype=torch.float32, device=dev)
        self.loss_parts = torch.zeros(L, dtype=torch.float64, device=dev)
        self._late = []

    def tensors(self):
        return (self.r_part, self.g_part, self.g_lm, self.b_grad, self.loss_parts)

    def finish(self):
        """The current stream waits for the sums over ranks of the bias-gradient rows and the loss partials."""
        for w in self._late:
            w()
        self._late = []

    def recons_norm(self):
        """The reduced reconstruction [B, L, d], gathered from the ranks' token blocks.  A collective: every rank has to ask
        for it in the same step (the training metrics do; the step itself never needs it)."""
        L, Bc, d = self.r_part.shape
        full = torch.empty(L, Bc * self.shard.world, d, dtype=torch.float32, device=self.r_part.device)
        self.shard.all_gather_rows_async([full[l] for l in range(L)], [self.r_part[l] for l in range(L)])()
        return full.permute(1, 0, 2).contiguous()


_loss_streams = {}


def _loss_stream(dev):
    key = torch.device(dev).index or 0
    if key not in _loss_streams:
        _loss_streams[key] = torch.cuda.Stream(device=dev)
    return _loss_streams[key]


def _forward_and_exchange(D, S, shard, d, split=None):
    """Pipelined exchange, forward half: the partial reconstruction layer-major [L, B, d], gathered layer group by layer
    group (last layers first).  Returns (r_lm, [wait per group], split state or None).

    split is None: each group's all-reduce is queued on RCCL's stream as soon as its slabs are written, so it runs under the
    gather of the next group (and later under the weight passes); the caller computes the group's loss after its wait.

    split = (mlp_out, output standardizer, N, 16-bit?): reduce-scatter + all-gather form.  The group's fp32 slabs are
    reduce-scattered over token blocks (rank r receives tokens [r*B/world, (r+1)*B/world) of every layer of the group), the
    loss kernel runs on that block alone — on a stream of its own, so that it waits for nobody but its reduce-scatter — and
    the gradient rows it writes (16-bit in the 16-bit modes: the form the weight passes gather) are all-gathered.  Per rank
    (world-1)/world * (4 + 2) bytes per element cross the links instead of the all-reduce's 8, and the loss pass reads
    1/world of the rows.  The whole chain of a group is queued here, in front of the weight passes: wait i is the all-gather's."""
    r_lm, waits = None, []
    if split is None:
        for lo, hi in _exchange_groups(D.L):
            r_lm = ops.sparse_decoder_fwd(S, D.pointer_table(), D._bias(), d, True, w_bf16=D.shadow, layer_major=True, out=r_lm,
                                          layers=(lo, hi))
            waits.append(shard.all_reduce_sum_async(r_lm[lo:hi]))
        return r_lm, waits, None
    mlp_out, ost, N, bf16 = split
    Bn, dev = mlp_out.shape[0], mlp_out.device
    ex = _SplitExchange(shard, D.L, Bn, d, ops.op16() if bf16 else torch.float32, dev)
    side = _loss_stream(dev)
    for t in ex.tensors():
        t.record_stream(side)                    # (written on the loss stream; a void attempt drops them without waiting)
    y = mlp_out[shard.rank * ex.Bc:(shard.rank + 1) * ex.Bc]
    for gi, (lo, hi) in enumerate(_exchange_groups(D.L)):
        r_lm = ops.sparse_decoder_fwd(S, D.pointer_table(), D._bias(), d, True, w_bf16=D.shadow, layer_major=True, out=r_lm,
                                      layers=(lo, hi))
        scattered = shard.reduce_scatter_rows_async([ex.r_part[l] for l in range(lo, hi)], [r_lm[l] for l in range(lo, hi)])
        with torch.cuda.stream(side):
            scattered()
            if not (bf16 and ops.mse_bf16_colsum_layers(ex.r_part[lo:hi], y[:, lo:hi], ost.mean[lo:hi], ost.std[lo:hi], N,
                                                        ex.g_part[lo:hi], ex.b_grad[lo:hi], ex.loss_parts[gi:gi + 1])):
                for ll in range(lo, hi):         # generic form, layer by layer (fp32 mode, odd shapes)
                    ls, gl, _ = ops.mse(ex.r_part[ll], y[:, ll], ost.mean[ll], ost.std[ll], True, False, count=N)
                    ex.loss_parts[gi:gi + 1].add_(ls)
                    ops.column_sum(gl, ex.b_grad[ll])
                    ex.g_part[ll].copy_(ops.cast_bf16(gl) if bf16 else gl)
            waits.append(shard.all_gather_rows_async([ex.g_lm[l] for l in range(lo, hi)], [ex.g_part[l] for l in range(lo, hi)]))
    with torch.cuda.stream(side):                # the two small sums, behind the last group's gather on RCCL's stream
        ex._late = [shard.all_reduce_sum_async(ex.b_grad), shard.all_reduce_sum_async(ex.loss_parts)]
    return r_lm, waits, ex


def _exchange_groups(L):
    """Layer groups of the pipelined exchange, last layers first, as (lo, hi) ranges: sizes 1, 2, 3, rest — the first group
    is small (its all-reduce is the only one nothing can hide) and every later group's all-reduce runs under the weight
    passes of the groups before it (W_l costs (l + 1) / sum of the work)."""
    out, hi = [], L
    for size in (1, 2, 3, L):
        if hi <= 0:
            break
        lo = max(0, hi - size)
        out.append((lo, hi))
        hi = lo
    return out


def agree_on_overflow(shard, flag):
    """fp16 mode under feature sharding: the conversions of a step are per shard, so an overflow may raise the flag on one
    rank only; the Adam kernels and the loss scaler of EVERY rank must see it (replicated parameters and the scale would
    diverge for good otherwise).  SUM over ranks of the device flag, in place; non-zero = skip."""
    if shard is not None and flag is not None and (shard.world > 1 or shard.keep_single):
        flag.copy_(shard.all_reduce_sum(flag.clone()))


def _rows_fit(mod, Bn, L, d, F, k, dev) -> bool:
    """HBM headroom for the fused per-row selection: asked once per module and shape, at its second step (the first step
    creates the optimizer state, bf16 shadows and workspaces that the answer depends on)."""
    key = (Bn, L, d, F, k)
    st = getattr(mod, "_rows_fit_state", None)
    if st is None or st[0] != key:
        mod._rows_fit_state = (key, None)
        return False                                 # (first step of this shape: unfused)
    if st[1] is None:
        mod._rows_fit_state = (key, ops.encoder_select_rows_fits(Bn, L, d, F, k, dev))
    return mod._rows_fit_state[1]


def _list_capacity(nl, Bn, L, F):
    """Upper bound on the number of positive entries a TopK-family module leaves."""
    return max(1, min(nl.k, F) * Bn * L)


class _VoidAttempt(Exception):
    """The fused encoder+select attempt of this step did not hold (learnt late, see _settle): run the step again unfused."""


class _StepCtx:
    """What the phases of one training step share and none of them changes: the module's parts, the shapes, the precision
    mode, the standardized inputs."""
    __slots__ = ("mod", "enc", "dec", "nl", "ost", "optimizer", "shard", "dev", "Bn", "L", "d", "F", "N", "bf16", "fp16", "flag",
                 "kind", "xo", "xs32", "xs16", "mlp_out", "batch_idx")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _aux_forward_lists(c, D, dead, pre_acts, ys, r, g, terms, grads):
    """TopK-aux pass of the list regime, forward half (clt_lightning.py:539-553): top k_aux of the DEAD features per row, second
    decoder pass, aux loss; g += g_aux in place.  Either as a dense sub-problem on the compacted dead columns (returns
    `compact` = (n_pad, cmap, Dc, feat_c, fc16) for _aux_compact_backward) or as a second pair of activation lists (returns
    S2).  Sets terms["aux"] and the aux part of grads["decoder.b"].  Returns (S2, g_aux, compact)."""
    mod, enc, dec, shard, dev, bf16, xo = c.mod, c.enc, c.dec, c.shard, c.dev, c.bf16, c.xo
    Bn, L, d, F, N = c.Bn, c.L, c.d, c.F, c.N
    S2 = cx = None
    aux = mod.topk_aux
    if COMPACT_AUX and isinstance(aux, PerLayerTopK) and 0 < aux.k <= 2048:
        cx = ops.compact_columns(dead)                      # one host read: the per-layer dead counts
        if cx is not None and cx[0] > COMPACT_AUX_MAX_SHARE * F:
            cx = None
        if cx is not None:
            # gathered weights (fp32 + bf16 + bf16 transposed) + their fp32 gradients + compact activations
            rows_c = sum(w.shape[0] for w in D.W32) * cx[0]
            need = rows_c * d * 12 + Bn * L * cx[0] * 16
            if need > 0.8 * torch.cuda.mem_get_info(dev)[0]:
                cx = None                                   # not enough HBM left: the activation-list form
        if shard is not None:                               # both forms hold collectives: every rank takes the same one
            agreed = torch.tensor([0 if cx is None else 1], dtype=torch.int32, device=dev)
            if int(shard.all_reduce_min(agreed)) == 0:
                cx = None
    if cx is not None:
        # dense sub-problem on the dead columns: where(dead, pre, -inf) -> top k_aux -> relu restricted to them
        n_pad, cidx, cmap = cx
        # (16-bit modes: the dead columns' pre-activations come from the encoder GEMM on the gathered weight rows — same
        # kernel, same bits — instead of a second read of `pre`: every 64-byte line of it holds a dead column or two)
        if bf16 and F % 4 == 0 and d % 8 == 0 and Bn * n_pad >= 256 * 256 and COMPACT_AUX_GEMM_COLUMNS:
            pre_c = ops.encoder_fwd_columns(xo, enc.Wt, enc.b, cidx)
        else:
            pre_c = ops.gather_cols(pre_acts(), cidx, float("-inf"))
        if shard is None:
            feat_c = ops.topk_rows(pre_c, min(aux.k, n_pad))
        else:               # k_aux of the dead features of ALL shards per row: exact global k-th value, then dense again
            Sc = ops.select_rows_sharded(pre_c, aux.k, shard, local_columns=True)
            feat_c = ops.sparse_scatter_dense(Sc, Sc.csc_val)
        if bf16 and COMPACT_AUX_FUSED_OPERANDS and d % 4 == 0 and all(w.is_contiguous() for w in D.W32):
            # 16-bit operand forms of the dead rows straight from the decoder matrices (no gathered fp32 copies)
            Dc = _Decoder(_CompactDecoder(dec, [None] * len(D.W32), n_pad), shard, clt=D.clt)
            prepped = [ops.weight_prep_gather_bf16(w, cidx) for w in D.W32]
            Dc._Wop, Dc._WT = [p[0] for p in prepped], [p[1] for p in prepped]
        else:
            Dc = _Decoder(_CompactDecoder(dec, [ops.gather_rows(w, cidx) for w in D.W32], n_pad), shard, clt=D.clt)
        # 16-bit modes: both GEMM operand forms of the compact features (row-major for the forward, transposed for the
        # weight gradient) from ONE read of them (the ReLU tile pass of the dense regime: feat_c >= 0 already)
        fc16 = ops.jumprelu_fwd_bf16(feat_c, None)[:2] if (bf16 and COMPACT_AUX_FUSED_OPERANDS and Bn % 8 == 0) else None
        aux_r = Dc.fwd(feat_c, feat_op=fc16[0] if fc16 is not None else None)
    else:
        S2 = _select_lists(aux, pre_acts(), dead, shard)
        aux_r = D.sparse_fwd(S2)
    aux_sum, g_aux = ops.aux_mse(aux_r, ys, r, g, mod.aux_loss_scale)      # g += g_aux in place
    terms["aux"] = (aux_sum[0] / N).float()
    grads["decoder.b"] = ops.column_sum(g_aux)
    compact = (n_pad, cmap, Dc, feat_c, fc16) if cx is not None else None
    return S2, g_aux, compact


def _aux_compact_backward(c, compact, g_aux):
    """Backward of the compacted aux pass: the dead rows' decoder gradients and the dead columns' encoder gradients, compact; they
    re-enter the Adam row passes through the row -> compact-row map.  Returns (extra_dec, extra_enc)."""
    n_pad, cmap, Dc, feat_c, fc16 = compact
    Bn, L, d, bf16, dev, xs32 = c.Bn, c.L, c.d, c.bf16, c.dev, c.xs32
    gc: Dict[str, torch.Tensor] = {}
    Dc.wgrad(feat_c, g_aux, gc, featT=fc16[1] if fc16 is not None else None)
    if fc16 is not None:
        # masked feature gradient straight to the transposed 16-bit operand + its column sums (no fp32 dpre tensor)
        dT_c, db_c = ops.mask_positive_t(feat_c, Dc.dgrad(g_aux))
    else:
        dpre_c = ops.mask_positive(feat_c, Dc.dgrad(g_aux))
        dT_c = ops.transpose_cast_2d(dpre_c).view(L, n_pad, Bn) if bf16 else None
        db_c = ops.column_sum(dpre_c)
    if bf16:
        dWt_c = torch.empty(L, n_pad, d, dtype=torch.float32, device=dev)
        ops.wgrad_t(dT_c, ops.transpose_cast_2d(xs32).view(L, d, Bn), dWt_c, L, n_pad * Bn, d * Bn)
    else:
        dWt_c = ops.encoder_wgrad(xs32, dpre_c)
    extra_dec = (gc, cmap)
    extra_enc = (dWt_c.view(-1, d), cmap, db_c.view(-1))
    return extra_dec, extra_enc


def _encoder_pass_lists(c, S, dval, S2, dval2, xsrc, extra_enc, grads):
    """Encoder weight / bias gradient (+ Adam) of the list regime from d(loss)/d(feature) per list entry."""
    enc, optimizer, bf16, dev, L, F = c.enc, c.optimizer, c.bf16, c.dev, c.L, c.F
    # encoder: dpre is non-zero only at the list entries (scatter(relu(gather)) backward), so the weight
    # gradient of feature (l,i) is sum_p dval[p] * xs[tok[p], l, :] and its bias gradient sum_p dval[p]
    db = torch.empty(L, F, dtype=torch.float32, device=dev)
    Wt = enc.Wt
    kw = dict(colsum=db, second=(S2, dval2, xsrc) if S2 is not None else None, extra=extra_enc)
    if optimizer is not None:
        kw.update(optimizer.begin_fused(enc.W, storage=Wt))
        if bf16:
            kw["shadow"] = _encoder_operand(enc)
    else:
        dWt = torch.empty_like(Wt)
        kw["grad_out"] = dWt
        grads["encoder.W"] = dWt.permute(0, 2, 1)
    fork = _HotFork(S, dev, enabled=optimizer is not None and c.shard is None)
    fork.row_pass(Wt, S, dval, xsrc, 0, -1, **kw)
    fork.join()
    grads["encoder.b"] = db


def _backward_jumprelu(c, D, S, pre, pre_bias, theta, feat, fused_ops, fuse_jump, g, terms, grads, out):
    """JumpReLU module (clt_lightning.py:420-475): sparsity and pre-activation loss terms, their gradients and the STE backward in
    one pass over the pre-activations; decoder weight gradients from the lists (S) or the dense GEMMs.  Returns (loss, dpre,
    dpreT, db): d(loss)/d(pre) as the fp32 tensor or as the transposed 16-bit operand + its column sums."""
    mod, nl, shard, optimizer, bf16, fp16, flag = c.mod, c.nl, c.shard, c.optimizer, c.bf16, c.fp16, c.flag
    Bn, L, F = c.Bn, c.L, c.F
    dpre = dpreT = db = None
    if not isinstance(nl, (JumpReLU, ReLU)):
        raise NotImplementedError("JumpReLU loss module needs a JumpReLU/ReLU nonlinearity in the fused step")
    lam = mod._lambda * (mod.global_step / mod.trainer.max_steps)
    th = theta if theta is not None else torch.zeros(1, L, F, device=pre.device)
    bw = nl._bandwidth if theta is not None else 1.0
    sq = D.sqnorms()
    g16 = ops.cast_bf16(g) if (bf16 and S is not None) else None
    loss16 = fused_ops is not None or (S is not None and fuse_jump)
    dfeat = D.dgrad(g, go=g16, out_bf16=bool(loss16 and DENSE_BF16_ACTIVATIONS))
    if loss16:
        sums, dpreT, db, dth, dn = ops.jumprelu_loss_bwd_bf16(pre, th, sq, dfeat, bw, lam / Bn, mod.c, mod.use_tanh,
                                                              mod.pre_actv_loss / (Bn * L), bias=pre_bias)
    else:
        sums, dpre, dth, dn = ops.jumprelu_loss_bwd(pre, th, sq, dfeat, bw, lam / Bn, mod.c, mod.use_tanh,
                                                    mod.pre_actv_loss / (Bn * L))
    if shard is not None:                 # loss terms are sums over the feature axis (gradients stay local)
        sums = shard.all_reduce_sum(sums.clone())
        if fp16:                          # d(pre) is per shard: every rank must take the same skip decision
            agree_on_overflow(shard, flag)
    if S is not None:
        # list regime: dW_l[l', i, :] = sum over the tokens where feature (l', i) is active of val * g[tok, l, :], plus the
        # row's dec_norms term — one streaming pass per decoder matrix, Adam applied in it when an optimizer is handed in
        D.sparse_wgrad(S, g, grads, optimizer, (dn, sq), g16)
        out["lists"] = S
    else:
        D.wgrad(feat, g, grads, featT=fused_ops[1] if fused_ops is not None else None, feat_op=fused_ops[0] if fused_ops is not None else None)
        D.add_norm_grads(dn, sq, grads, optimizer)
    grads["decoder.b"] = ops.column_sum(g)
    if theta is not None:
        grads["nonlinearity.theta"] = dth
    terms["sparsity"] = (lam * sums[0] / Bn).float()
    terms["pre_actv"] = (mod.pre_actv_loss * sums[1] / (Bn * L)).float()
    terms["weighted_features_mean"] = (sums[2] / (Bn * L * F * (shard.world if shard is not None else 1))).float()
    terms["sparsity_penalty"] = torch.tensor(lam)
    loss = terms["mse"] + terms["sparsity"] + terms["pre_actv"]
    return loss, dpre, dpreT, db


def _backward_dense(c, D, pre, feat, theta, dead, ys, r, g, terms, grads):
    """Dense regime under the base / TopK-aux losses: aux pass (dense selection on the dead columns, second decoder pass),
    decoder gradients by GEMM, d(loss)/d(pre) through the sparsifier's backward.  Returns (loss, dpre)."""
    mod, nl, kind, N = c.mod, c.nl, c.kind, c.N
    if kind == "topk":
        aux = mod.topk_aux
        dead_pre = ops.mask_dead(pre, dead)
        dead_feat = _select(aux, dead_pre)
        aux_r = D.fwd(dead_feat)
        aux_sum, g_aux = ops.aux_mse(aux_r, ys, r, g, mod.aux_loss_scale)      # g += g_aux in place
        terms["aux"] = (aux_sum[0] / N).float()
        loss = terms["mse"] + mod.aux_loss_scale * terms["aux"]
        D.wgrad(dead_feat, g_aux, grads)
        grads["decoder.b"] = ops.column_sum(g_aux)
    else:
        loss = terms["mse"]
    D.wgrad(feat, g, grads)
    grads["decoder.b"] = ops.column_sum(g, grads.get("decoder.b"), accumulate="decoder.b" in grads)
    dfeat = D.dgrad(g)
    if theta is not None:
        dpre, grads["nonlinearity.theta"] = ops.jumprelu_bwd(pre, theta, nl._bandwidth, dfeat)
    elif isinstance(nl, ReLU):
        dpre = ops.relu_bwd(pre, dfeat)
    else:
        dpre = ops.mask_positive(feat, dfeat)
    if kind == "topk":
        ops.mask_positive(dead_feat, D.dgrad(g_aux), out=dpre, accumulate=True)   # aux path reaches dead entries only
    return loss, dpre


def _encoder_wgrad_dense(c, pre, dpre, dpreT, db, grads):
    """Encoder weight / bias gradient of the dense regime: contraction over the tokens."""
    bf16, Bn, L, d, F, xs32, xo = c.bf16, c.Bn, c.L, c.d, c.F, c.xs32, c.xo
    if bf16:       # dWt[l] [F, d] = dpre[:, l]^T @ xs[:, l], contraction over tokens: both operands token-minor
        dWt = torch.empty(L, F, d, dtype=torch.float32, device=pre.device)
        dT = dpreT if dpreT is not None else ops.transpose_cast_2d(dpre).view(L, F, Bn)
        ops.wgrad_t(dT, ops.transpose_cast_2d(xs32).view(L, d, Bn), dWt, L, F * Bn, d * Bn)
        grads["encoder.W"] = dWt.permute(0, 2, 1)
    else:
        grads["encoder.W"] = ops.encoder_wgrad(xo, dpre).permute(0, 2, 1)
    grads["encoder.b"] = db if dpreT is not None else ops.column_sum(dpre)


def _fused_batch_select(c, split_args):
    """BatchTopK / PerLayerBatchTopK with last step's k-th value known: encoder GEMM whose epilogue files the candidates, exact
    selection on those (clt_gemm_bf16_select + clt_select_lists_ext_f32).  Returns (lists or None, deferred host read or None);
    the deferred read raises _VoidAttempt when the bound did not hold."""
    nl, enc, dec, shard, optimizer, fp16, dev, xo, L, d = c.nl, c.enc, c.dec, c.shard, c.optimizer, c.fp16, c.dev, c.xo, c.L, c.d
    mode = ops.SELECT_LAYER if isinstance(nl, PerLayerBatchTopK) else ops.SELECT_BATCH
    nseg = L if isinstance(nl, PerLayerBatchTopK) else 1
    drift = getattr(nl, "_kth_drift", None)
    if ADAPTIVE_SELECT_MARGIN and (drift is None or drift.numel() != 2 * nseg or drift.device != dev):
        drift = nl._kth_drift = ops.new_select_drift(nseg, dev)
    # the decoder's gather forward is queued BEFORE the host reads whether the bound held (it only reads the lists, which
    # are empty after a void attempt): the launch queue does not run dry behind that read — on an 8-way shard, where
    # every kernel is 8 times shorter, the GPU otherwise idled ~0.2 ms a step there
    def _spec_fwd(S_):
        D_ = _Decoder(dec, shard, shadow=optimizer is not None)
        lm = bool(PIPELINE_EXCHANGE and shard is not None and D_.clt and not fp16)
        if lm:
            return (D_, True) + _forward_and_exchange(D_, S_, shard, d, split_args)
        return D_, False, ops.sparse_decoder_fwd(S_, D_.pointer_table(), D_._bias(), d, D_.clt, w_bf16=D_.shadow), None, None

    S, kth = ops.encoder_select_lists(xo, _encoder_operand(enc), enc.b, nl.k, mode, nl._kth, SELECT_MARGIN, nl.threshold,
                                      nl.e, shard, drift=drift if ADAPTIVE_SELECT_MARGIN else None,
                                      speculate=_spec_fwd if SPECULATIVE_FORWARD else None, defer=SPECULATIVE_FORWARD)
    STATS["fused_select_steps"] += 1

    def _settle_select():
        """The step's one host read: did the bound hold?  Asked as late as possible — before the first kernel that
        changes module state (dead-feature counters, weights) — so that the forward, the exchange and the first loss
        kernel are already queued behind the selection and the GPU does not idle while the host waits."""
        if S.confirm():
            nl._kth = kth
            return
        nl._kth = None
        STATS["select_fallbacks"] += 1
        if drift is not None:
            drift[:nseg] = 0.1               # the bound did not hold: back to the floor for ~30 steps, then tighten again
        raise _VoidAttempt()

    if SPECULATIVE_FORWARD:
        return S, _settle_select
    if S is not None:
        nl._kth = kth
    else:
        STATS["select_fallbacks"] += 1
        if drift is not None:
            drift[:nseg] = 0.1
    return S, None


def _fused_rows_select(c):
    """PerLayerTopK fused with the encoder GEMM (clt_select_rows_ext_f32 + clt_gemm_bf16_select_rows).  Returns (lists, deferred
    host read); the read raises _VoidAttempt when some row's bound did not hold, and the rerun of the step is unfused."""
    mod, nl, enc, dec, optimizer, xo, d, batch_idx = c.mod, c.nl, c.enc, c.dec, c.optimizer, c.xo, c.d, c.batch_idx
    # per-row TopK without the pre-activation tensor: lower bound of every row from a narrow GEMM on a rotating 1/16 sample
    # of the features, candidates filed per row by the full GEMM's epilogue, exact selection on those
    def _spec_rows(S_):
        D_ = _Decoder(dec, None, shadow=optimizer is not None)
        return D_, False, ops.sparse_decoder_fwd(S_, D_.pointer_table(), D_._bias(), d, D_.clt, w_bf16=D_.shadow), None, None

    S = ops.encoder_select_rows_lists(xo, _encoder_operand(enc), enc.b, nl.k, offset=batch_idx,
                                      speculate=_spec_rows if SPECULATIVE_FORWARD else None)
    STATS["fused_select_steps"] += 1

    def _settle_rows():
        if S.confirm():
            return
        STATS["select_fallbacks"] += 1
        mod._rows_unfused_once = True                # some row's bound did not hold: this step again, unfused
        raise _VoidAttempt()

    return S, _settle_rows


def _group_slabs(c, groups, waits, ex, r_lm):
    """Pipelined exchange, backward half.  Returns (slabs, loss_parts, b_grad): slabs(l) makes the current stream wait for layer
    l's group — in the all-reduce form it then computes the group's loss partial, gradient slabs and bias-gradient rows; in the
    split form (ex) those came with the exchange — and returns layer l's gradient slab [B, d]; the weight passes call it right
    before the pass of W_l.  loss_parts fp64 [L] (one entry per group used), b_grad [L, d]."""
    ost, mlp_out, bf16, dev, Bn, L, d, N = c.ost, c.mlp_out, c.bf16, c.dev, c.Bn, c.L, c.d, c.N
    if ex is not None:                                   # split form: the loss ran on this rank's token block, per group
        loss_parts, b_grad, g_lm = ex.loss_parts, ex.b_grad, ex.g_lm
    else:
        loss_parts = torch.zeros(L, dtype=torch.float64, device=dev)
        b_grad = torch.empty(L, d, dtype=torch.float32, device=dev)
        g_lm = torch.empty(L, Bn, d, dtype=ops.op16() if bf16 else torch.float32, device=dev)
    group_of = {l: gi for gi, (lo, hi) in enumerate(groups) for l in range(lo, hi)}
    done = set()

    def slabs(l):
        gi = group_of[l]
        if gi not in done:                               # first layer of its group: wait for the group's exchange, then its loss
            done.add(gi)
            waits[gi]()                                  # (the current stream waits; the host does not)
            if ex is not None:                           # (... for the group's gathered gradient slabs: nothing left to compute)
                return g_lm[l]
            lo, hi = groups[gi]
            if not (bf16 and ops.mse_bf16_colsum_layers(r_lm[lo:hi], mlp_out[:, lo:hi], ost.mean[lo:hi], ost.std[lo:hi], N,
                                                        g_lm[lo:hi], b_grad[lo:hi], loss_parts[gi:gi + 1])):
                for ll in range(lo, hi):                 # generic form, layer by layer (fp32 mode, odd shapes)
                    ls, gl, _ = ops.mse(r_lm[ll], mlp_out[:, ll], ost.mean[ll], ost.std[ll], True, False, count=N)
                    loss_parts[gi:gi + 1].add_(ls)
                    ops.column_sum(gl, b_grad[ll])
                    g_lm[ll].copy_(ops.cast_bf16(gl) if bf16 else gl)
        return g_lm[l]

    return slabs, loss_parts, b_grad


def training_step(mod, batch: torch.Tensor, batch_idx: int, optimizer=None, sparse: Optional[bool] = None) -> Dict[str, object]:
    try:
        return _training_step(mod, batch, batch_idx, optimizer, sparse)
    except _VoidAttempt:
        # nothing of the module's state was touched: the forward, the exchange and the loss computed from the void lists only
        # wrote scratch buffers.  The sparsifier's k-th value is reset, so this pass takes the unfused GEMM + selection.
        return _training_step(mod, batch, batch_idx, optimizer, sparse)


def _training_step(mod, batch: torch.Tensor, batch_idx: int, optimizer=None, sparse: Optional[bool] = None) -> Dict[str, object]:
    """Returns {"loss": 0-dim fp32 tensor, "terms": {name: tensor}, "grads": {param_name: tensor},
    "dead": uint8 [L,F], ...}.  `mod` is a CrossLayerTranscoderModule (or subclass).

    optimizer: a FusedAdam — in the sparse regime the encoder/decoder weight matrices are then updated in
    place by the backward row pass and do not appear in "grads" (the optimizer's step() afterwards covers the
    remaining parameters).  sparse: force (True) / forbid (False) the list regime; default = TopK family."""
    model = mod.model
    enc, dec, nl = model.encoder, model.decoder, model.nonlinearity
    if not batch.is_cuda:
        raise RuntimeError("training_step needs a CUDA (HIP) batch; there is no CPU path")
    if batch.dtype != torch.float32:
        batch = batch.float()                   # the shm buffer hands over fp16 (datamodule contract)
    if batch_idx == 0:
        model.initialize_standardizers(batch)
    ist, ost = model.input_standardizer, model.output_standardizer
    if ist is None or ost is None:
        raise ValueError("training needs both standardizers")
    ist._require_init()
    resid, mlp_out = batch[:, 0], batch[:, 1]
    Bn, L, d = resid.shape
    F = enc.d_features
    bf16 = mixed16()                            # 16-bit GEMM operands (bf16, or fp16 with loss scaling), fp32 accumulation
    fp16 = get_precision() == "fp16"
    if fp16:
        # `precision: "16-mixed"`: gradients are formed multiplied by the loss scale (the loss kernels' coefficients carry it)
        # so that they survive the fp16 operand casts, and are unscaled inside the Adam kernels; an overflowing conversion
        # raises the device flag and every Adam kernel of the step then leaves its parameters untouched.  The caller
        # (fused_train_step) reads the flag once at the end of the step and moves the scale (precision.LossScaler)
        scaler = getattr(mod, "loss_scaler", None)
        if scaler is None:
            scaler = mod.loss_scaler = LossScaler()
        flag = ops.inf_flag(batch.device)
        flag.zero_()
        ops.set_loss_scaling(True, scaler.scale, flag)
    else:
        ops.set_loss_scaling(False)
    kind = mod._loss_kind
    topk_family = isinstance(nl, (PerLayerTopK, BatchTopK, PerLayerBatchTopK))
    # thresholding sparsifiers take the list regime when few entries are active: JumpReLU / ReLU under the JumpReLU module's
    # losses (decoder forward + decoder weight gradient from the lists; the STE passes gradient wherever pre >= 0, so the
    # feature gradient and the encoder stay dense GEMMs), ReLU under the base loss (everything from the lists)
    thr_family = (isinstance(nl, (JumpReLU, ReLU)) and kind == "jumprelu") or (isinstance(nl, ReLU) and kind == "base")
    if sparse is None:
        sparse = getattr(mod, "sparse_regime", None)
    force_lists = sparse is True
    if sparse is None:
        sparse = (topk_family and kind in ("base", "topk")) or thr_family
    if sparse and not ((topk_family and kind in ("base", "topk")) or thr_family):
        raise NotImplementedError("the activation-list regime covers TopK-family sparsifiers with the base / TopK-aux losses, "
                                  "JumpReLU / ReLU with the JumpReLU module's losses and ReLU with the base loss")

    # (the fp32 standardized activations are read only by the fp32 mode, the dense regime and the compacted aux pass)
    need_xs32 = not (bf16 and sparse and kind == "base" and topk_family)
    xs32, xs16 = ops.standardize(resid, ist.mean, ist.std, need_xs32, bf16)
    xo = xs16 if bf16 else xs32
    dev = batch.device
    shard = getattr(mod, "shard", None)
    if shard is not None and shard.world == 1 and not shard.keep_single:
        shard = None
    theta = None
    feat = S = pre = fused_ops = dpreT = None
    batch_mode = isinstance(nl, (BatchTopK, PerLayerBatchTopK))
    # pipelined exchange in its reduce-scatter + all-gather form (every rank decides from the shapes alone)
    split_args = ((mlp_out, ost, Bn * L * d, bf16) if (EXCHANGE_SPLIT and shard is not None and Bn % shard.world == 0
                                                      and (shard.world > 1 or shard.keep_single)) else None)
    pending = None                               # deferred host read of a fused selection's outcome
    c = _StepCtx(mod=mod, enc=enc, dec=dec, nl=nl, ost=ost, optimizer=optimizer, shard=shard, dev=dev, Bn=Bn, L=L, d=d, F=F,
                 N=Bn * L * d, bf16=bf16, fp16=fp16, flag=flag if fp16 else None, kind=kind, xo=xo, xs32=xs32, xs16=xs16,
                 mlp_out=mlp_out, batch_idx=batch_idx)
    probe = getattr(nl, "_kth_probe", None)
    if probe is not None and probe.value():
        # the last unfused selection ended in its exhaustive fallback, which leaves no k-th key behind: a fused attempt
        # from it would be void (one wasted encoder GEMM) — skip it this step
        nl._kth = None
    nl._kth_probe = None
    if (sparse and bf16 and kind == "base" and batch_mode and nl.k > 0 and nl.e is not None and FUSE_ENCODER_SELECT
            and getattr(nl, "_kth", None) is not None and d % 8 == 0 and F % 4 == 0):
        S, pending = _fused_batch_select(c, split_args)
    rows_unfused = getattr(mod, "_rows_unfused_once", False)       # (the rerun after a void per-row attempt)
    mod._rows_unfused_once = False
    if (S is None and sparse and bf16 and kind in ("base", "topk") and isinstance(nl, PerLayerTopK) and shard is None and FUSE_ENCODER_ROWS
            and not rows_unfused and ops.encoder_select_rows_applicable(Bn, L, d, F, nl.k) and _rows_fit(mod, Bn, L, d, F, nl.k, dev)):
        S, pending = _fused_rows_select(c)
    fuse_jump = (FUSE_DENSE_JUMPRELU and bf16 and kind == "jumprelu" and isinstance(nl, (JumpReLU, ReLU)) and Bn % 8 == 0
                 and (L * F) % 4 == 0)
    # 16-bit regime of the JumpReLU module: the pre-activations (and later the feature gradient) leave their GEMMs as bf16 —
    # the K = d_acts GEMMs are bound by their epilogue stores, and the consumers (tile passes; the list build when a list
    # attempt is made this step, clt_sparse_build_threshold_x16) read half the bytes
    attempt = bool(sparse and thr_family and (force_lists or getattr(mod, "_list_skip", 0) == 0))
    act16 = bool(DENSE_BF16_ACTIVATIONS and fuse_jump and S is None and (not attempt or (LIST_BF16_ACTIVATIONS and F % 4 == 0)))
    # 16-bit pre-activation tensor: with ops.AUTOCAST_PRE it holds the encoder PRODUCT alone (the reference's einsum result
    # under autocast is this 16-bit tensor) and every consumer adds the fp32 bias in fp32 (`pre_bias`), autocast's order
    pre_bias = None
    if S is None:
        split_bias = bool(act16 and ops.AUTOCAST_PRE and enc.b is not None)
        pre = ops.encoder_fwd_out16(xo, _encoder_operand(enc), None if split_bias else enc.b) if act16 else None
        if pre is None:
            act16 = False
            pre = ops.encoder_fwd(xo, _encoder_operand(enc), enc.b)
        elif split_bias:
            pre_bias = enc.b
    if S is None and sparse and thr_family:
        theta = nl.theta if isinstance(nl, JumpReLU) else None
        S = _threshold_lists(mod, pre, theta, force_lists, pre_bias)
        if S is None:
            sparse = False                       # above the break-even density: dense GEMM regime for this step
    if S is not None:
        pass
    elif sparse:
        S = _select_lists(nl, pre, shard=shard)
        if batch_mode:
            nl._kth = getattr(S, "kth", None)
            if shard is None and nl._kth is not None and S.used_fallback is not None:
                nl._kth_probe = ops.FlagProbe(S.used_fallback)        # (read at the next step, without blocking)
    elif fuse_jump:
        theta = nl.theta if isinstance(nl, JumpReLU) else None
        # (feat bf16, feat^T bf16, fired); per-layer decoder: no transposed copy, its weight gradient takes the features k-major
        fused_ops = ops.jumprelu_fwd_bf16(pre, theta, pre_bias, want_T=not (KM_WGRAD_PLT and not isinstance(dec, CrosslayerDecoder) and ops.wgrad_ta_ok(Bn, F, d)))
    elif isinstance(nl, JumpReLU):
        theta = nl.theta
        feat = ops.jumprelu_fwd(pre, theta)
    elif isinstance(nl, ReLU):
        feat = ops.relu_fwd(pre)
    elif topk_family:
        feat = _select(nl, pre)
    else:
        raise NotImplementedError(f"fused step: unsupported nonlinearity {type(nl).__name__}")

    spec = getattr(S, "speculated", None) if S is not None else None
    D = spec[0] if spec is not None else _Decoder(dec, shard, shadow=bool(sparse and bf16 and optimizer is not None))
    N = c.N
    grads: Dict[str, torch.Tensor] = {}
    want_ys = kind == "topk"

    def pre_acts():
        """The pre-activations, computed now when the fused encoder + selection never wrote them (aux pass on a path that reads them)."""
        nonlocal pre
        if pre is None:
            pre = ops.encoder_fwd(xo, _encoder_operand(enc), enc.b)
        return pre

    lists_only = sparse and kind != "jumprelu"           # every gradient comes from the lists (TopK family; ReLU + base loss)
    # (not under fp16 loss scaling: a later group's gradient slab could overflow after an earlier group's weights were updated)
    pipelined = bool(PIPELINE_EXCHANGE and shard is not None and lists_only and kind == "base" and D.clt and not fp16)
    waits = None
    if sparse:
        fired = S.fired
    elif fused_ops is not None:
        fired = fused_ops[2]
    else:
        fired = ops.column_any_positive(feat)

    def dead_features():
        join = getattr(S, "join", None) if S is not None else None
        if join is not None:                  # (threshold lists: `fired` comes with the feature-major half of the build)
            join()
        dd, cnt = ops.dead_update(fired, Bn, mod.last_active, mod.tokens_till_dead)
        if shard is not None:                 # features are split over ranks: count over all shards
            if pipelined:
                # (queued behind the layer groups' all-reduces on RCCL's stream: nobody waits for it before the end of the
                # step — a blocking all-reduce here would hold the weight passes until the whole exchange had finished)
                cnt = cnt.clone()
                return dd, (cnt, shard.all_reduce_sum_async(cnt))
            cnt = shard.all_reduce_sum(cnt.clone())
        return dd, cnt

    if pipelined:
        # partial reconstruction layer-major [L, B, d].  Launched first (the host has just read the selection's outcome flag
        # and the launch queue is empty); the dead-feature update and its tiny all-reduce follow, and only then are the
        # all-reduces of the layer groups queued on RCCL's stream — no small collective waits behind them
        groups = _exchange_groups(L)                       # [(lo, hi)], last layers first
        r_lm, waits, ex = (spec[2], spec[3], spec[4]) if spec is not None and spec[1] else _forward_and_exchange(D, S, shard, d,
                                                                                                                  split_args)
        if pending is None:
            dead, dead_cnt = dead_features()
        r = None
    else:
        if sparse:
            r = D._reduce(spec[2]) if spec is not None and not spec[1] else D.sparse_fwd(S)
        elif fused_ops is not None:
            r = D.fwd(None, feat_op=fused_ops[0])
        else:
            r = D.fwd(feat)
        if pending is None:
            dead, dead_cnt = dead_features()
    g16 = b_grad = slabs = None
    fm = ops.mse_bf16_colsum(r, mlp_out, ost.mean, ost.std) if (FUSE_MSE_BF16 and lists_only and bf16 and kind == "base"
                                                                 and not pipelined) else None
    if pipelined:
        # per layer, once its slab has been reduced: loss partial + gradient slab (+ bias-gradient row); called by the
        # backward pass right before the weight pass of W_l
        g = ys = None
        slabs, loss_parts, b_grad = _group_slabs(c, groups, waits, ex, r_lm)
        if pending is not None:
            slabs(L - 1)                                 # (first group's loss kernel queued before the host read below)
        terms = {}
    elif fm is not None:                     # loss + bf16 gradient rows + decoder bias gradient in one pass, no fp32 gradient
        loss_sum, g16, b_grad = fm
        g = ys = None
        terms = {"mse": (loss_sum[0] / N).float()}
    else:
        loss_sum, g, ys = ops.mse(r, mlp_out, ost.mean, ost.std, True, want_ys)
        terms = {"mse": (loss_sum[0] / N).float()}
    if pending is not None:
        pending()                                        # raises _VoidAttempt when the fused selection did not hold
        dead, dead_cnt = dead_features()
    if isinstance(dead_cnt, tuple):                      # pipelined exchange: the count's all-reduce is waited for at the end
        late_dead, dead_cnt = dead_cnt, None
    else:
        late_dead = None
        terms["dead_fraction"] = dead_cnt[0].float() / (L * F * (shard.world if shard is not None else 1))
    out = StepOutput(terms=terms, grads=grads, dead=dead)
    out["grad_scale"] = ops.GRAD_SCALE           # out["grads"] carry this factor (1.0 outside fp16 mode)
    out.lazy = {}
    if pipelined:
        out.lazy["recons_norm"] = ex.recons_norm if ex is not None else (lambda: r_lm.permute(1, 0, 2).contiguous())
    else:
        out["recons_norm"] = r
    if pre is not None and not act16:
        out["pre_actvs"] = pre
    elif pre is not None:
        out.lazy["pre_actvs"] = lambda: pre.float() if pre_bias is None else pre.float() + pre_bias
    else:
        out.lazy["pre_actvs"] = lambda: ops.encoder_fwd(xo, _encoder_operand(enc), enc.b)
    if feat is not None:
        out["features"] = feat
    elif fused_ops is not None:
        # (a weak reference: `out` holds this closure — a strong one would make every step's output, with its gradients and
        # the 16-bit pre-activations, a reference cycle that lives until Python's cyclic collector gets to it: +6 GB per few
        # steps at B = 8192)
        wout = weakref.ref(out)
        pre32 = (lambda: wout()["pre_actvs"]) if act16 else (lambda: pre)
        out.lazy["features"] = (lambda: ops.jumprelu_fwd(pre32(), theta)) if theta is not None else (lambda: ops.relu_fwd(pre32()))

    if lists_only:
        S2 = g_aux = compact = None
        if kind == "topk":
            S2, g_aux, compact = _aux_forward_lists(c, D, dead, pre_acts, ys, r, g, terms, grads)
            loss = terms["mse"] + mod.aux_loss_scale * terms["aux"]
        else:
            loss = terms.get("mse")              # (pipelined exchange: known after the backward pass)
        if b_grad is not None:
            grads["decoder.b"] = b_grad
        elif not pipelined:
            grads["decoder.b"] = ops.column_sum(g, grads.get("decoder.b"), accumulate="decoder.b" in grads)
        xsrc = xs16 if bf16 else xs32
        extra_dec = extra_enc = None
        if compact is not None:
            extra_dec, extra_enc = _aux_compact_backward(c, compact, g_aux)
        if fp16:                                  # every rank takes the same skip decision in the Adam-carrying row passes
            agree_on_overflow(shard, flag)
        dval, dval2 = D.sparse_bwd(S, g, S2, g_aux if S2 is not None else None, grads, optimizer, extra=extra_dec, g16=g16,
                                   slabs=slabs, groups=groups if pipelined else None)
        if pipelined:
            if ex is not None:
                ex.finish()
            terms["mse"] = (loss_parts.sum() / N).float()
            loss = terms["mse"]
            late_dead[1]()
            terms["dead_fraction"] = late_dead[0][0].float() / (L * F * shard.world)
        _encoder_pass_lists(c, S, dval, S2, dval2, xsrc, extra_enc, grads)
        out["lists"] = S
        terms["loss"] = loss
        out["loss"] = loss
        return out

    if kind == "jumprelu":
        loss, dpre, dpreT, db = _backward_jumprelu(c, D, S, pre, pre_bias, theta, feat, fused_ops, fuse_jump, g, terms, grads, out)
    else:
        loss, dpre = _backward_dense(c, D, pre, feat, theta, dead, ys, r, g, terms, grads)
        dpreT = db = None

    _encoder_wgrad_dense(c, pre, dpre, dpreT, db, grads)
    if optimizer is not None:
        D.register_prep(optimizer, enc)
    terms["loss"] = loss
    out["loss"] = loss
    return out
