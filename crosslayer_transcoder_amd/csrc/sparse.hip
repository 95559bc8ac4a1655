// Sparse (activation-list) decoder path of the crosslayer-transcoder step for gfx950.
//
// When the sparsifier leaves k << F active features per (token, layer) the reference still runs dense
// bmm's over [B, (l+1)F] x [(l+1)F, d] (model/clt.py:334-343) and their autograd.  Here the active
// entries are compacted once into a token-major CSR and a feature-major CSC list, and
//   * forward   : recons[b,l,:] = b[l] + sum_{entries (l',i,val) of token b, l' <= l} val * W_l[l',i,:]
//   * backward  : one streaming pass over every decoder row W_l[l',i,:] that (a) forms the row's weight
//                 gradient  sum_b val * g[b,l,:]  from the feature's token list, (b) emits the partial
//                 d(loss)/d(feature) dots  g[b,l,:] . W_l[l',i,:]  for those tokens, and (c) applies the
//                 Adam update in the same pass — so the dense dW tensor never exists in HBM.
// All kernels are HBM/L2-bandwidth bound (row gathers of d floats); lists are ordered, so every sum has a
// fixed order and results are bitwise reproducible.
#include <math.h>
#include <type_traits>
#include "common.h"
#include "lists_common.h"

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------------
// forward gather: one wave per (token, target layer)
// ------------------------------------------------------------------------------------------------
// FWD_U: gathered rows in flight per wave for d <= 768.  4, 8 and 16 measured the same (15.39 / 15.39 / 15.39 ms headline,
// scripts/archive/gpu_fwdu_ab.sh): the gather is not latency-bound at 4
#ifndef FWD_U
#define FWD_U 4
#endif
// F16: the 16-bit rows are fp16 instead of bf16 (clt_set_loss_scaling)
template <int NT, bool VEC, bool BF, bool F16 = false>
__global__ __launch_bounds__(TPB) void sparse_decoder_fwd_kernel(const uint32_t* __restrict__ row_start, const int32_t* __restrict__ ent_col,
                                                                 const float* __restrict__ ent_val, const float* const* __restrict__ Wptr,
                                                                 const float* __restrict__ bias, float* __restrict__ out, int L, int F,
                                                                 int d, int cross_layer, uint32_t cap, long out_tok_stride,
                                                                 long out_layer_stride, int l_lo, int l_hi) {
    constexpr int W = VEC ? 4 : 1;
    constexpr int U = NT <= 3 ? FWD_U : (NT <= 4 ? 4 : (NT <= 8 ? 2 : 1));   // entries in flight per wave (a divisor of 64; bounded by the register budget)
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // target layers are dealt to the 4 waves heaviest first in serpentine order (a cross-layer target l gathers the
    // entries of l + 1 source layers): 11,4,3 | 10,5,2 | 9,6,1 | 8,7,0 for L = 12
    // (target layers [l_lo, l_hi): the feature-sharded step launches the forward layer group by layer group, so that a
    // group's all-reduce runs under the forward of the next)
    for (int rnd = 0; rnd * 4 < l_hi - l_lo; ++rnd) {
        const int l = l_hi - 1 - rnd * 4 - ((rnd & 1) ? 3 - w : w);
        if (l < l_lo) continue;
        const float* Wl = Wptr[l];
        const uint16_t* Wh = reinterpret_cast<const uint16_t*>(Wl);   // BF: the table points at bf16 shadow rows
        const int lo = cross_layer ? 0 : l;
        const uint32_t e0 = row_start[(long)b * L + lo], e1 = min(row_start[(long)b * L + l + 1], cap);
        const int colsub = cross_layer ? 0 : l * F;            // per-layer decoder: W[l] rows are indexed by i only
        float acc[NT][W];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < W; ++c) { const int j = (t * 64 + lane) * W + c; acc[t][c] = (bias && j < d) ? bias[l * d + j] : 0.0f; }
        for (uint32_t base = e0; base < e1; base += 64) {
            const uint32_t mine = base + lane;
            // lanes past the end carry (row 0, value 0): the entry loop runs in whole groups of U without a tail
            const int myrow = mine < e1 ? ent_col[mine] - colsub : 0;
            const float myval = mine < e1 ? ent_val[mine] : 0.0f;
            const int cnt = (int)min(64u, e1 - base);
            for (int j = 0; j < cnt; j += U) {
                float v[U]; long ro[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { ro[u] = (long)__shfl(myrow, j + u, 64) * d; v[u] = __shfl(myval, j + u, 64); }
                if (VEC) {
                    typedef typename std::conditional<BF, ushort4, float4>::type raw_t;    // converted at use: half the registers
                    raw_t r[U][NT];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int o = (t * 64 + lane) * W;
                            if (o < d) {
                                if constexpr (BF) r[u][t] = *reinterpret_cast<const ushort4*>(Wh + ro[u] + o);
                                else r[u][t] = *reinterpret_cast<const float4*>(Wl + ro[u] + o);
                            } else r[u][t] = raw_t{};
                        }
#pragma unroll
                    for (int u = 0; u < U; ++u)            // entry order = list order: fixed summation order
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            float4 x;
                            if constexpr (BF) x = make_float4(cvt16<F16>(r[u][t].x), cvt16<F16>(r[u][t].y), cvt16<F16>(r[u][t].z), cvt16<F16>(r[u][t].w));
                            else x = r[u][t];
                            acc[t][0] += v[u] * x.x; acc[t][1 % W] += v[u] * x.y; acc[t][2 % W] += v[u] * x.z; acc[t][3 % W] += v[u] * x.w;
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int o = t * 64 + lane;
                            if (o < d) acc[t][0] += v[u] * (BF ? cvt16<F16>(Wh[ro[u] + o]) : Wl[ro[u] + o]);
                        }
                }
            }
        }
        float* dst = out + (long)b * out_tok_stride + (long)l * out_layer_stride;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = (t * 64 + lane) * W;
            if (o < d) {
                if (VEC) *reinterpret_cast<float4*>(dst + o) = make_float4(acc[t][0], acc[t][1 % W], acc[t][2 % W], acc[t][3 % W]);
                else dst[o] = acc[t][0];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward row pass: sparse weight gradient (+ partial feature-gradient dots) (+ Adam) per weight row
// ------------------------------------------------------------------------------------------------
struct RowPassArgs {
    float* W; float* m; float* v; uint16_t* shadow; float* grad_out;   // grad_out != null: write dW, no Adam
    int rows, d, F;
    int col0;                    // CSC column of row 0 (row r <-> column col0 + r)
    int src_layer;               // >= 0: fixed source layer; < 0: layer = (col0 + r) / F
    int L;                       // layers in the source tensors ([B][L][d])
    const uint32_t* col_start; const int32_t* tok; const float* coef; const float* src;
    const uint32_t* col_start2; const int32_t* tok2; const float* coef2; const float* src2;   // optional 2nd list
    float* dot_out; float* dot_out2; long dot_stride, dot_stride2; int dot_slot_base;   // dot_out[slot*dot_stride + csc_pos]; slot_base < 0: slot = src layer - own layer
    float* colsum_out;           // optional: colsum_out[col] = sum coef (encoder bias gradient)
    float lr, omb1, b2, omb2, eps, wd, step_size, isb2; int adamw;
    int src_bf16;                // src / src2 hold bf16 (uint16_t) elements instead of fp32: halves the gather traffic
    // optional dense contribution for a subset of the rows (the compacted TopK-aux pass, compact.hip):
    //   grad[r,:] += extra[extra_map[col0 + r], :] ; colsum_out[col] += extra_colsum[extra_map[col]]   (map < 0: none)
    const float* extra; const int32_t* extra_map; const float* extra_colsum;
    // optional per-row term: grad[r,:] += (row_dnorm[col] / sqrt(row_sq[col])) * W[r,:] — the gradient that reaches a decoder
    // row through dec_norms (clt_lightning.py:438-448; same term as clt_adam_f32's row_dnorm / row_sq)
    const float* row_dnorm; const float* row_sq;
    // optional (one wave per row, i.e. d <= 1024): squared length of the UPDATED row r to sq_out[r] — next step's dec_norms
    // without another read of the matrix; same per-lane order and wave reduction as row_sqnorm_kernel
    float* sq_out;
    // > 0: rows whose (first) list holds more than this many entries are left to row_pass_hot_kernel (heavy-tailed feature
    // frequencies: a feature active in 10 % of the tokens is a 400-entry list, which one wave would walk alone)
    int hot_min_len;
};

// CW = waves per weight row.  CW == 1: four rows per 256-thread block, one wave each (d <= 1024 floats per wave with
// 16-byte lanes).  CW > 1 (long rows, e.g. d = 4096): one row per block, wave c owns the c-th 1024-float chunk —
// keeps the per-lane state at NT = 4 (high occupancy) and combines the per-entry dots through LDS in wave order.
// RP_HU: list entries of a column whose source rows are requested together (measured on the headline, 30 steps x 3 runs:
// 1 / 2 / 3 / 4 / 6 -> 15.60 / 15.74 / 15.31 / 15.40 / 15.73 ms per step, the pass itself 850 -> 828 us = 0.79 of the HBM peak;
// JumpReLU lists at 0.7 % active, PLT: 32.0 -> 28.0 ms).  With the non-temporal weight streams (RP_NT): 2 / 3 / 4 -> 14.92 / 15.00 /
// 15.20 ms — 3 is kept for the longer lists of the thresholding sparsifiers
#ifndef RP_HU
#define RP_HU 3
#endif
// RP_TAIL2: a remainder of two entries after the batches of RP_HU is requested together as well (the TopK lists' columns hold
// ~2 entries: headline launch 806 -> 791 us, 15.04 -> 14.83 ms, scripts/archive/gpu_tail2_ab.sh)
#ifndef RP_TAIL2
#define RP_TAIL2 1
#endif
// one list entry's contribution to a lane's four gradient elements and to its share of the entry's dot — one fixed sequence
// of fused multiply-adds wherever an entry is accumulated, so that the result does not depend on the code path that loaded it
__device__ __forceinline__ void rp_acc4(float* __restrict__ g4, float& dot, float cv, const float4 s, const float* __restrict__ w4) {
    g4[0] = fmaf(cv, s.x, g4[0]); g4[1] = fmaf(cv, s.y, g4[1]); g4[2] = fmaf(cv, s.z, g4[2]); g4[3] = fmaf(cv, s.w, g4[3]);
    dot = fmaf(s.x, w4[0], dot); dot = fmaf(s.y, w4[1], dot); dot = fmaf(s.z, w4[2], dot); dot = fmaf(s.w, w4[3], dot);
}
// RP_NT (bit 0 loads, bit 1 stores): the weight / moment streams of the row pass (read once, written once per step) as
// non-temporal accesses, so that they do not displace the gathered source rows ([B, L, d], 75 MB at G32 in bf16) from the
// caches.  Headline launch 829 -> 815 us (loads) -> 812 us (loads + stores; stores alone 835), scripts/archive/gpu_rpnt_ab.sh
#ifndef RP_NT
#define RP_NT 3
#endif
typedef float rp_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int rp_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 rp_ld4(const float* q) {
#if RP_NT & 1
    const rp_f4 t = __builtin_nontemporal_load(reinterpret_cast<const rp_f4*>(q)); return make_float4(t[0], t[1], t[2], t[3]);
#else
    return *reinterpret_cast<const float4*>(q);
#endif
}
__device__ __forceinline__ void rp_st4(float* q, float a, float b, float c, float d) {
#if RP_NT & 2
    const rp_f4 t = {a, b, c, d}; __builtin_nontemporal_store(t, reinterpret_cast<rp_f4*>(q));
#else
    *reinterpret_cast<float4*>(q) = make_float4(a, b, c, d);
#endif
}
__device__ __forceinline__ void rp_st2(uint16_t* q, uint32_t a, uint32_t b) {
#if RP_NT & 2
    const rp_u2 t = {a, b}; __builtin_nontemporal_store(t, reinterpret_cast<rp_u2*>(q));
#else
    *reinterpret_cast<uint2*>(q) = make_uint2(a, b);
#endif
}
// SQ: also write the updated row's squared length (a.sq_out) — its own instantiation: compiled into the common kernel the
// few extra instructions cost the headline launch 20 us (830 -> 851, register allocation of the whole body moved)
// The entries [p, pend) of one list of column `col` (a whole column, or one wave's segment of a hot column): gradient sums into
// g, dots written, coefficient sum into csum.  p0 = first entry of the column (parity of the CW > 1 dot exchange).
// HU: source rows requested together (6 in the hot-row kernel spills under its 128-register budget and gains nothing: 3 everywhere; the sums do not depend on it)
template <int NT, bool VEC, int CW, bool F16, int HU = RP_HU>
__device__ __forceinline__ void row_pass_walk(const RowPassArgs& a, const int pass, const int col, const int layer, const int lane, const int wave,
                                              const int cb, uint32_t p, const uint32_t pend, const uint32_t p0, const float (&wv)[NT][VEC ? 4 : 1],
                                              float (&g)[NT][VEC ? 4 : 1], float& csum, float (*shdot)[CW]) {
    constexpr int W = VEC ? 4 : 1;
    const long srs = (long)a.L * a.d;
    const int32_t* tk = pass ? a.tok2 : a.tok;
    const float* cf = pass ? a.coef2 : a.coef;
    const float* src = (pass ? a.src2 : a.src) + (a.src_bf16 ? 0 : (long)layer * a.d);
    const uint16_t* srch = reinterpret_cast<const uint16_t*>(pass ? a.src2 : a.src) + (long)layer * a.d;
    float* dout = pass ? a.dot_out2 : a.dot_out;
    const long dstride = pass ? a.dot_stride2 : a.dot_stride;
#if RP_HU > 1
    // Columns with several entries (the JumpReLU lists' tens of tokens per feature): RP_HU entries' source rows are
    // requested together and then accumulated in list order — the same sums, a fraction of the serialised round trips.
    if constexpr (VEC && CW == 1) if (a.src_bf16) {
        auto batch = [&](auto nu) {                     // the next NU entries: all source-row loads first, then the sums in list order
            constexpr int NU = decltype(nu)::value;
            float cvs[NU];
            ushort4 hh[NU][NT];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                cvs[u] = cf[p + u];
                const uint16_t* rowh = srch + (long)tk[p + u] * srs;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int o = (t * 64 + lane) * 4;
                    hh[u][t] = *reinterpret_cast<const ushort4*>(rowh + (o < a.d ? o : 0));
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float cv = cvs[u];
                float dot = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int o = (t * 64 + lane) * 4;
                    if (o < a.d) {
                        const ushort4 h = hh[u][t];
                        rp_acc4(g[t], dot, cv, make_float4(cvt16<F16>(h.x), cvt16<F16>(h.y), cvt16<F16>(h.z), cvt16<F16>(h.w)), wv[t]);
                    }
                }
                csum += cv;
                if (dout) {
                    dot = wave_sum(dot);
                    if (lane == 0) dout[(long)(a.dot_slot_base < 0 ? layer - col / a.F : a.dot_slot_base) * dstride + p + u] = dot;
                }
            }
            p += NU;
        };
        while (p + HU <= pend) batch(std::integral_constant<int, HU>{});
        if constexpr (HU > RP_HU) { if (p + RP_HU <= pend) batch(std::integral_constant<int, RP_HU>{}); }
#if RP_HU > 2 && RP_TAIL2
        if (p + 2 <= pend) batch(std::integral_constant<int, 2>{});      // (a remainder of two: together as well)
#endif
    }
#endif
    for (; p < pend; ++p) {
        const float cv = cf[p];
        const float* srow = src + (long)tk[p] * srs;
        const uint16_t* srowh = srch + (long)tk[p] * srs;
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = cb + (t * 64 + lane) * W;
            if (o < a.d) {
                if constexpr (VEC) {
                    float4 s;
                    if (a.src_bf16) {
                        const ushort4 h = *reinterpret_cast<const ushort4*>(srowh + o);
                        s = make_float4(cvt16<F16>(h.x), cvt16<F16>(h.y), cvt16<F16>(h.z), cvt16<F16>(h.w));
                    } else s = *reinterpret_cast<const float4*>(srow + o);
                    rp_acc4(g[t], dot, cv, s, wv[t]);
                } else { const float s = a.src_bf16 ? cvt16<F16>(srowh[o]) : srow[o]; g[t][0] += cv * s; dot += s * wv[t][0]; }
            }
        }
        csum += cv;
        if (dout) {
            dot = wave_sum(dot);
            if (CW > 1) {                                   // every wave of the block walks the same entry list
                const int par = (int)((p - p0) & 1u);
                if (lane == 0) shdot[par][wave] = dot;
                __syncthreads();
                dot = 0.0f;
#pragma unroll
                for (int c = 0; c < CW; ++c) dot += shdot[par][c];
            }
            if (lane == 0 && wave == (CW == 1 ? wave : 0))
                dout[(long)(a.dot_slot_base < 0 ? layer - col / a.F : a.dot_slot_base) * dstride + p] = dot;
        }
    }
}

// What follows the entry lists for a row whose gradient sums (g) and coefficient sum are complete: the dense contribution
// of the compacted aux pass, the dec_norms term, then the materialised gradient or the Adam update (+ shadow, + squared length).
template <int NT, bool VEC, int CW, bool F16, bool SQ>
__device__ __forceinline__ void row_pass_finish(const RowPassArgs& a, const int r, const int col, const int lane, const int wave, const int cb,
                                                float (&wv)[NT][VEC ? 4 : 1], float (&g)[NT][VEC ? 4 : 1], float (&mm)[NT][VEC ? 4 : 1],
                                                float (&vv)[NT][VEC ? 4 : 1], float csum) {
    constexpr int W = VEC ? 4 : 1;
    const long ro = (long)r * a.d;
    if (a.extra) {
        const int mr = a.extra_map[col];
        if (mr >= 0) {
            const float* er = a.extra + (long)mr * a.d;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int o = cb + (t * 64 + lane) * W;
                if (o < a.d) {
                    if (VEC) { const float4 x = *reinterpret_cast<const float4*>(er + o); g[t][0] += x.x; g[t][1 % W] += x.y; g[t][2 % W] += x.z; g[t][3 % W] += x.w; }
                    else g[t][0] += er[o];
                }
            }
            if (a.extra_colsum) csum += a.extra_colsum[mr];
        }
    }
    if (a.colsum_out && lane == 0 && (CW == 1 || wave == 0)) a.colsum_out[col] = csum;
    if (a.row_dnorm) {
        const float cf = a.row_dnorm[col] / sqrtf(a.row_sq[col]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < W; ++c) g[t][c] += cf * wv[t][c];
    }
    if (a.grad_out) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = cb + (t * 64 + lane) * W;
            if (o < a.d) {
                if (VEC) *reinterpret_cast<float4*>(a.grad_out + ro + o) = make_float4(g[t][0], g[t][1 % W], g[t][2 % W], g[t][3 % W]);
                else a.grad_out[ro + o] = g[t][0];
            }
        }
        return;
    }
    // loss scaling (clt_set_loss_scaling): the gathered sources carry the loss scale, the gradient is unscaled here; a step
    // whose gradients overflowed leaves weights, moments and shadow untouched (the dots and column sums above are still
    // written: the later passes of this step run, and skip likewise)
    bool skip = false;
    if constexpr (SQ) skip = clt_skip_update();     // (the row's length is still written: of the unchanged row)
    else if (clt_skip_update()) return;
    const float unscale = clt_dev_state.grad_unscale;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = cb + (t * 64 + lane) * W;
        if (o >= a.d || skip) continue;
#pragma unroll
        for (int c = 0; c < W; ++c) {
            float gg = g[t][c] * unscale, pp = wv[t][c];
            if (a.wd != 0.0f) { if (a.adamw) pp *= (1.0f - a.lr * a.wd); else gg += a.wd * pp; }
            // (explicit fused multiply-adds: left to the compiler, the contraction of these lines differed between the
            // instantiations of this body — 1 ulp in v and W between kernels that must agree bit for bit)
            mm[t][c] = __fmaf_rn(gg - mm[t][c], a.omb1, mm[t][c]);
            vv[t][c] = __fmaf_rn(vv[t][c], a.b2, (a.omb2 * gg) * gg);
            pp = __fmaf_rn(-a.step_size, mm[t][c] / __fmaf_rn(sqrtf(vv[t][c]), a.isb2, a.eps), pp);
            wv[t][c] = pp;
        }
        if (VEC) {
            rp_st4(a.W + ro + o, wv[t][0], wv[t][1 % W], wv[t][2 % W], wv[t][3 % W]);
            rp_st4(a.m + ro + o, mm[t][0], mm[t][1 % W], mm[t][2 % W], mm[t][3 % W]);
            rp_st4(a.v + ro + o, vv[t][0], vv[t][1 % W], vv[t][2 % W], vv[t][3 % W]);
            if (a.shadow) rp_st2(a.shadow + ro + o, pack16<F16>(wv[t][0], wv[t][1 % W]), pack16<F16>(wv[t][2 % W], wv[t][3 % W]));
        } else {
            a.W[ro + o] = wv[t][0]; a.m[ro + o] = mm[t][0]; a.v[ro + o] = vv[t][0];
            if (a.shadow) a.shadow[ro + o] = (uint16_t)pack16<F16>(wv[t][0], 0.0f);
        }
    }
    if constexpr (SQ && CW == 1) if (a.sq_out) {
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = (t * 64 + lane) * W;
            if (o < a.d) s = VEC ? s + sq4(wv[t][0], wv[t][1 % W], wv[t][2 % W], wv[t][3 % W]) : __fmaf_rn(wv[t][0], wv[t][0], s);
        }
        s = wave_sum(s);
        if (lane == 0) a.sq_out[r] = s;
    }
}

// the row's weights (when anything needs them) and, for the Adam form, its moments: issued together, the moments are in
// flight during the entry loop (used only after it)
template <int NT, bool VEC>
__device__ __forceinline__ void row_pass_load(const RowPassArgs& a, const long ro, const int lane, const int cb, float (&wv)[NT][VEC ? 4 : 1],
                                              float (&g)[NT][VEC ? 4 : 1], float (&mm)[NT][VEC ? 4 : 1], float (&vv)[NT][VEC ? 4 : 1],
                                              const bool moments) {
    constexpr int W = VEC ? 4 : 1;
    const bool need_w = a.dot_out != nullptr || a.dot_out2 != nullptr || a.grad_out == nullptr || a.row_dnorm != nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = cb + (t * 64 + lane) * W;
#pragma unroll
        for (int c = 0; c < W; ++c) { g[t][c] = 0.0f; wv[t][c] = 0.0f; mm[t][c] = 0.0f; vv[t][c] = 0.0f; }
        if (need_w && o < a.d) {
            if (VEC) { const float4 x = rp_ld4(a.W + ro + o); wv[t][0] = x.x; wv[t][1 % W] = x.y; wv[t][2 % W] = x.z; wv[t][3 % W] = x.w; }
            else wv[t][0] = a.W[ro + o];
        }
    }
    if (moments && !a.grad_out) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = cb + (t * 64 + lane) * W;
            if (o < a.d) {
                if (VEC) {
                    const float4 x = rp_ld4(a.m + ro + o); mm[t][0] = x.x; mm[t][1 % W] = x.y; mm[t][2 % W] = x.z; mm[t][3 % W] = x.w;
                    const float4 y = rp_ld4(a.v + ro + o); vv[t][0] = y.x; vv[t][1 % W] = y.y; vv[t][2 % W] = y.z; vv[t][3 % W] = y.w;
                } else { mm[t][0] = a.m[ro + o]; vv[t][0] = a.v[ro + o]; }
            }
        }
    }
}

// The streaming pass itself, one wave per row (or CW waves per long row).  It is written as ONE body on purpose: the same steps
// factored into the row_pass_load / _walk / _finish helpers above (which row_pass_hot_kernel is made of) compile to 144 instead of
// 110 registers per lane here, i.e. three instead of four waves per SIMD for the kernel that is 70 % of the headline step.
// SQ: also write the updated row's squared length (a.sq_out) — its own instantiation: compiled into the common kernel the
// few extra instructions cost the headline launch 20 us (830 -> 851, register allocation of the whole body moved)
template <int NT, bool VEC, int CW, bool F16, bool SQ = false>
__device__ __forceinline__ void row_pass_body(const RowPassArgs& a, const int r) {
    constexpr int W = VEC ? 4 : 1;
    __shared__ float shdot[2][CW];
    const int wave = threadIdx.x >> 6;
    if (r >= a.rows) return;
    const int lane = threadIdx.x & 63;
    const int cb = CW == 1 ? 0 : wave * NT * 64 * W;          // first element of this wave's chunk
    const int col = a.col0 + r;
    if (CW == 1 && a.hot_min_len > 0 && a.col_start[col + 1] - a.col_start[col] > (uint32_t)a.hot_min_len) return;   // row_pass_hot_kernel's
    const int layer = a.src_layer >= 0 ? a.src_layer : col / a.F;
    const long srs = (long)a.L * a.d;
    const long ro = (long)r * a.d;
    float wv[NT][W], g[NT][W];
    const bool need_w = a.dot_out != nullptr || a.dot_out2 != nullptr || a.grad_out == nullptr || a.row_dnorm != nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = cb + (t * 64 + lane) * W;
#pragma unroll
        for (int c = 0; c < W; ++c) { g[t][c] = 0.0f; wv[t][c] = 0.0f; }
        if (need_w && o < a.d) {
            if (VEC) { const float4 x = rp_ld4(a.W + ro + o); wv[t][0] = x.x; wv[t][1 % W] = x.y; wv[t][2 % W] = x.z; wv[t][3 % W] = x.w; }
            else wv[t][0] = a.W[ro + o];
        }
    }
    // Adam moments: issued now so that they are in flight during the entry loop (used only after it)
    float mm[NT][W], vv[NT][W];
    if (!a.grad_out) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = cb + (t * 64 + lane) * W;
#pragma unroll
            for (int c = 0; c < W; ++c) { mm[t][c] = 0.0f; vv[t][c] = 0.0f; }
            if (o < a.d) {
                if (VEC) {
                    const float4 x = rp_ld4(a.m + ro + o); mm[t][0] = x.x; mm[t][1 % W] = x.y; mm[t][2 % W] = x.z; mm[t][3 % W] = x.w;
                    const float4 y = rp_ld4(a.v + ro + o); vv[t][0] = y.x; vv[t][1 % W] = y.y; vv[t][2 % W] = y.z; vv[t][3 % W] = y.w;
                } else { mm[t][0] = a.m[ro + o]; vv[t][0] = a.v[ro + o]; }
            }
        }
    }
    float csum = 0.0f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t* cs = pass ? a.col_start2 : a.col_start;
        if (!cs) continue;
        const int32_t* tk = pass ? a.tok2 : a.tok;
        const float* cf = pass ? a.coef2 : a.coef;
        const float* src = (pass ? a.src2 : a.src) + (a.src_bf16 ? 0 : (long)layer * a.d);
        const uint16_t* srch = reinterpret_cast<const uint16_t*>(pass ? a.src2 : a.src) + (long)layer * a.d;
        float* dout = pass ? a.dot_out2 : a.dot_out;
        const long dstride = pass ? a.dot_stride2 : a.dot_stride;
        const uint32_t p0 = cs[col], p1 = cs[col + 1];
        uint32_t p = p0;
#if RP_HU > 1
        // Columns with several entries (the JumpReLU lists' tens of tokens per feature): RP_HU entries' source rows are
        // requested together and then accumulated in list order — the same sums, a fraction of the serialised round trips.
        if constexpr (VEC && CW == 1) if (a.src_bf16) {
            auto batch = [&](auto nu) {                     // the next NU entries: all source-row loads first, then the sums in list order
                constexpr int NU = decltype(nu)::value;
                float cvs[NU];
                ushort4 hh[NU][NT];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    cvs[u] = cf[p + u];
                    const uint16_t* rowh = srch + (long)tk[p + u] * srs;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int o = (t * 64 + lane) * 4;
                        hh[u][t] = *reinterpret_cast<const ushort4*>(rowh + (o < a.d ? o : 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const float cv = cvs[u];
                    float dot = 0.0f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int o = (t * 64 + lane) * 4;
                        if (o < a.d) {
                            const ushort4 h = hh[u][t];
                            rp_acc4(g[t], dot, cv, make_float4(cvt16<F16>(h.x), cvt16<F16>(h.y), cvt16<F16>(h.z), cvt16<F16>(h.w)), wv[t]);
                        }
                    }
                    csum += cv;
                    if (dout) {
                        dot = wave_sum(dot);
                        if (lane == 0) dout[(long)(a.dot_slot_base < 0 ? layer - col / a.F : a.dot_slot_base) * dstride + p + u] = dot;
                    }
                }
                p += NU;
            };
            while (p + RP_HU <= p1) batch(std::integral_constant<int, RP_HU>{});
#if RP_HU > 2 && RP_TAIL2
            if (p + 2 <= p1) batch(std::integral_constant<int, 2>{});      // (a remainder of two: together as well)
#endif
        }
#endif
        for (; p < p1; ++p) {
            const float cv = cf[p];
            const float* srow = src + (long)tk[p] * srs;
            const uint16_t* srowh = srch + (long)tk[p] * srs;
            float dot = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int o = cb + (t * 64 + lane) * W;
                if (o < a.d) {
                    if constexpr (VEC) {
                        float4 s;
                        if (a.src_bf16) {
                            const ushort4 h = *reinterpret_cast<const ushort4*>(srowh + o);
                            s = make_float4(cvt16<F16>(h.x), cvt16<F16>(h.y), cvt16<F16>(h.z), cvt16<F16>(h.w));
                        } else s = *reinterpret_cast<const float4*>(srow + o);
                        rp_acc4(g[t], dot, cv, s, wv[t]);
                    } else { const float s = a.src_bf16 ? cvt16<F16>(srowh[o]) : srow[o]; g[t][0] += cv * s; dot += s * wv[t][0]; }
                }
            }
            csum += cv;
            if (dout) {
                dot = wave_sum(dot);
                if (CW > 1) {                                   // every wave of the block walks the same entry list
                    const int par = (int)((p - p0) & 1u);
                    if (lane == 0) shdot[par][wave] = dot;
                    __syncthreads();
                    dot = 0.0f;
#pragma unroll
                    for (int c = 0; c < CW; ++c) dot += shdot[par][c];
                }
                if (lane == 0 && wave == (CW == 1 ? wave : 0))
                    dout[(long)(a.dot_slot_base < 0 ? layer - col / a.F : a.dot_slot_base) * dstride + p] = dot;
            }
        }
    }
    if (a.extra) {
        const int mr = a.extra_map[col];
        if (mr >= 0) {
            const float* er = a.extra + (long)mr * a.d;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int o = cb + (t * 64 + lane) * W;
                if (o < a.d) {
                    if (VEC) { const float4 x = *reinterpret_cast<const float4*>(er + o); g[t][0] += x.x; g[t][1 % W] += x.y; g[t][2 % W] += x.z; g[t][3 % W] += x.w; }
                    else g[t][0] += er[o];
                }
            }
            if (a.extra_colsum) csum += a.extra_colsum[mr];
        }
    }
    if (a.colsum_out && lane == 0 && (CW == 1 || wave == 0)) a.colsum_out[col] = csum;
    if (a.row_dnorm) {
        const float cf = a.row_dnorm[col] / sqrtf(a.row_sq[col]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < W; ++c) g[t][c] += cf * wv[t][c];
    }
    if (a.grad_out) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = cb + (t * 64 + lane) * W;
            if (o < a.d) {
                if (VEC) *reinterpret_cast<float4*>(a.grad_out + ro + o) = make_float4(g[t][0], g[t][1 % W], g[t][2 % W], g[t][3 % W]);
                else a.grad_out[ro + o] = g[t][0];
            }
        }
        return;
    }
    // loss scaling (clt_set_loss_scaling): the gathered sources carry the loss scale, the gradient is unscaled here; a step
    // whose gradients overflowed leaves weights, moments and shadow untouched (the dots and column sums above are still
    // written: the later passes of this step run, and skip likewise)
    bool skip = false;
    if constexpr (SQ) skip = clt_skip_update();     // (the row's length is still written: of the unchanged row)
    else if (clt_skip_update()) return;
    const float unscale = clt_dev_state.grad_unscale;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int o = cb + (t * 64 + lane) * W;
        if (o >= a.d || skip) continue;
#pragma unroll
        for (int c = 0; c < W; ++c) {
            float gg = g[t][c] * unscale, pp = wv[t][c];
            if (a.wd != 0.0f) { if (a.adamw) pp *= (1.0f - a.lr * a.wd); else gg += a.wd * pp; }
            // (explicit fused multiply-adds: left to the compiler, the contraction of these lines differed between the
            // instantiations of this body — 1 ulp in v and W between kernels that must agree bit for bit)
            mm[t][c] = __fmaf_rn(gg - mm[t][c], a.omb1, mm[t][c]);
            vv[t][c] = __fmaf_rn(vv[t][c], a.b2, (a.omb2 * gg) * gg);
            pp = __fmaf_rn(-a.step_size, mm[t][c] / __fmaf_rn(sqrtf(vv[t][c]), a.isb2, a.eps), pp);
            wv[t][c] = pp;
        }
        if (VEC) {
            rp_st4(a.W + ro + o, wv[t][0], wv[t][1 % W], wv[t][2 % W], wv[t][3 % W]);
            rp_st4(a.m + ro + o, mm[t][0], mm[t][1 % W], mm[t][2 % W], mm[t][3 % W]);
            rp_st4(a.v + ro + o, vv[t][0], vv[t][1 % W], vv[t][2 % W], vv[t][3 % W]);
            if (a.shadow) rp_st2(a.shadow + ro + o, pack16<F16>(wv[t][0], wv[t][1 % W]), pack16<F16>(wv[t][2 % W], wv[t][3 % W]));
        } else {
            a.W[ro + o] = wv[t][0]; a.m[ro + o] = mm[t][0]; a.v[ro + o] = vv[t][0];
            if (a.shadow) a.shadow[ro + o] = (uint16_t)pack16<F16>(wv[t][0], 0.0f);
        }
    }
    if constexpr (SQ && CW == 1) if (a.sq_out) {
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = (t * 64 + lane) * W;
            if (o < a.d) s = VEC ? s + sq4(wv[t][0], wv[t][1 % W], wv[t][2 % W], wv[t][3 % W]) : __fmaf_rn(wv[t][0], wv[t][0], s);
        }
        s = wave_sum(s);
        if (lane == 0) a.sq_out[r] = s;
    }
}

template <int NT, bool VEC, int CW, bool F16 = false, bool SQ = false>
// (about 100 registers: 4 waves per SIMD.  Forcing 5 or 6 with amdgpu_waves_per_eu spills 60 / 132 bytes per lane and the headline
// launch goes from 787 to 971 / 1607 us, scripts/archive/gpu_rpwaves_ab.sh)
__global__ __launch_bounds__(CW == 1 ? TPB : 64 * CW) void row_pass_kernel(RowPassArgs a) {
    row_pass_body<NT, VEC, CW, F16, SQ>(a, CW == 1 ? blockIdx.x * 4 + (int)(threadIdx.x >> 6) : (int)blockIdx.x);
}

// Several weight matrices in ONE launch (clt_sparse_row_pass_group_f32): the rows of the matrices are laid end to end over
// the grid; a wave finds its matrix in the (at most 16 entries long) table and runs the ordinary pass with that matrix's
// pointers, source and layer.  On a feature shard the per-matrix launches are 35..180 us long and spend a fifth of that
// ramping up and draining.
constexpr int RP_GROUP_MAX = 16;
constexpr int HOT_GRID = 1024;                            // blocks of the hot-row kernels (each walks its table with this stride)
struct RowPassGroup {
    int n; int row_end[RP_GROUP_MAX];                     // cumulative row counts
    float* W[RP_GROUP_MAX]; float* m[RP_GROUP_MAX]; float* v[RP_GROUP_MAX]; uint16_t* shadow[RP_GROUP_MAX];
    const float* src[RP_GROUP_MAX]; int src_layer[RP_GROUP_MAX];
};
template <int NT, bool F16>
__global__ __launch_bounds__(TPB) void row_pass_group_kernel(RowPassArgs a, RowPassGroup g) {
    int r = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (r >= g.row_end[g.n - 1]) return;
    int i = 0;
    while (r >= g.row_end[i]) ++i;
    i = __builtin_amdgcn_readfirstlane(i);                // (one row per wave: the matrix index is wave-uniform)
    r -= i ? g.row_end[i - 1] : 0;
    a.W = g.W[i]; a.m = g.m[i]; a.v = g.v[i]; a.shadow = g.shadow[i]; a.src = g.src[i]; a.src_layer = g.src_layer[i];
    a.rows = g.row_end[i] - (i ? g.row_end[i - 1] : 0);
    row_pass_body<NT, true, 1, F16>(a, r);
}


// ---- heavy-tailed feature frequencies ------------------------------------------------------------------------------
// A feature that fires in 10 % of the tokens is a 400-entry list, which one wave of row_pass_kernel would walk alone
// while the rows around it finish in a few entries.  Columns longer than `hot_min_len` are therefore listed once per step
// (hot_columns_kernel) and left out by the common pass.  Their lists are walked in SEGMENTS by row_pass_seg_kernel (a block of
// HW waves per segment, a wave per piece of it; dots are per entry and written directly; the pieces' gradient sums are
// combined through LDS in wave order and the segment's sum goes to scratch), and row_pass_hot_kernel (a wave per row) adds a
// row's segment sums in segment order and finishes the row (Adam): a fixed summation order, as everywhere.
// The hot block (int32 [2 cap + 2 seg_cap]): [0, cap) the hot columns | [cap, 2 cap) per hot column its first SEGMENT slot (-1: no
// room in the table, the column is walked whole) | then the segment table, seg_cap pairs (hot index, segment); hot_count[0] =
// hot columns, hot_count[1] = segments asked for (may exceed seg_cap).  Every hot column is cut into segments of HOT_SEG entries, each walked by a block of its
// own (row_pass_seg_kernel): equal work items whatever the column lengths — a feature that fires in half of the tokens is 2000
// entries, which the waves of ONE block would walk for ~100 us while the chip waits for that block.
// cap >= entries / (min_len + 1) + 1 bounds the hot columns; segments <= entries / HOT_SEG + hot columns (the caller sizes seg_cap).
constexpr int HOT_SEG = 64;
__global__ void hot_columns_kernel(const uint32_t* __restrict__ col_start, int ncols, uint32_t min_len, int32_t* __restrict__ hot_list,
                                   uint32_t hot_cap, uint32_t seg_cap, uint32_t* __restrict__ hot_count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const uint32_t len = col_start[c + 1] - col_start[c];
    if (len > min_len) {
        const uint32_t pos = atomicAdd(&hot_count[0], 1u);
        if (pos >= hot_cap) return;                 // (hot_cap >= entries / (min_len + 1) + 1 by the caller: never exceeded)
        hot_list[pos] = c;
        int32_t base = -1;
        const uint32_t nseg = (len + HOT_SEG - 1) / HOT_SEG;
        const uint32_t b = atomicAdd(&hot_count[1], nseg);
        if (b + nseg <= seg_cap) {
            base = (int32_t)b;
            for (uint32_t q = 0; q < nseg; ++q) { hot_list[2 * hot_cap + 2 * (b + q)] = (int32_t)pos; hot_list[2 * hot_cap + 2 * (b + q) + 1] = (int32_t)q; }
        }
        hot_list[hot_cap + pos] = base;
    }
}

#ifndef HOT_HU
#define HOT_HU 3
#endif
template <int NT> struct HotWaves { static constexpr int value = 8; };   // waves of a segment block (LDS: HW * NT KB for the piece sums)
// floats of one segment's partial result: the gradient row (NT x 256) + the coefficient sum
template <int NT> struct HotPart { static constexpr int value = NT * 256 + 64; };

// one block per (hot column, segment) of the segment table: the HW waves walk the segment's HW pieces (dots written per entry,
// as everywhere), the piece sums are combined through LDS in wave order and the segment's sum goes to partials[slot]
template <int NT, bool F16>
__global__ __launch_bounds__(64 * HotWaves<NT>::value) void row_pass_seg_kernel(RowPassArgs a, RowPassGroup grp, int use_group,
                                                                               const int32_t* __restrict__ hot_list,
                                                                               const uint32_t* __restrict__ hot_count, uint32_t hot_cap,
                                                                               uint32_t seg_cap, float* __restrict__ partials) {
    constexpr int HW = HotWaves<NT>::value;
    __shared__ float part[HW][NT * 256];
    __shared__ float pcs[HW];
    __shared__ float shdot[2][1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (use_group) {
        const int i = blockIdx.y;
        a.W = grp.W[i]; a.m = grp.m[i]; a.v = grp.v[i]; a.shadow = grp.shadow[i]; a.src = grp.src[i]; a.src_layer = grp.src_layer[i];
        a.rows = grp.row_end[i] - (i ? grp.row_end[i - 1] : 0);
        partials += (long)i * seg_cap * HotPart<NT>::value;
    }
    const uint32_t ns = min(hot_count[1], seg_cap);
    for (uint32_t it = blockIdx.x; it < ns; it += gridDim.x) {
        const int h = hot_list[2 * hot_cap + 2 * it], sg = hot_list[2 * hot_cap + 2 * it + 1];
        if (h < 0) continue;                                       // (a hole: its column was not segmented)
        const int col = hot_list[h];
        const int r = col - a.col0;
        if (r < 0 || r >= a.rows) continue;                        // (block-uniform)
        const int layer = a.src_layer >= 0 ? a.src_layer : col / a.F;
        float wv[NT][4], g[NT][4], mm[NT][4], vv[NT][4];
        row_pass_load<NT, true>(a, (long)r * a.d, lane, 0, wv, g, mm, vv, false);
        float csum = 0.0f;
        const uint32_t p0 = a.col_start[col], p1 = a.col_start[col + 1];
        const uint32_t s0 = p0 + (uint32_t)sg * HOT_SEG, s1 = min(p1, s0 + HOT_SEG);
        const uint32_t per = (s1 - s0 + HW - 1) / HW;
        const uint32_t q0 = min(s1, s0 + (uint32_t)wave * per), q1 = min(s1, q0 + per);
        row_pass_walk<NT, true, 1, F16, HOT_HU>(a, 0, col, layer, lane, wave, 0, q0, q1, p0, wv, g, csum, shdot);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            *reinterpret_cast<float4*>(&part[wave][(t * 64 + lane) * 4]) = make_float4(g[t][0], g[t][1], g[t][2], g[t][3]);
        if (lane == 0) pcs[wave] = csum;
        __syncthreads();
        if (wave == 0) {
#pragma unroll 1
            for (int w = 1; w < HW; ++w) {                         // wave order: a fixed summation order
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 x = *reinterpret_cast<const float4*>(&part[w][(t * 64 + lane) * 4]);
                    g[t][0] += x.x; g[t][1] += x.y; g[t][2] += x.z; g[t][3] += x.w;
                }
                csum += pcs[w];
            }
            float* dst = partials + (long)it * HotPart<NT>::value;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<float4*>(dst + (t * 64 + lane) * 4) = make_float4(g[t][0], g[t][1], g[t][2], g[t][3]);
            if (lane == 0) dst[NT * 256] = csum;
        }
        __syncthreads();
    }
}

// the hot rows themselves, one wave each: the segments' sums are added in segment order (the row's fixed summation order), a
// second list is walked as in the common pass, then the row is finished (Adam, shadow, squared length) as there
template <int NT, bool F16>
__global__ __launch_bounds__(TPB) void row_pass_hot_kernel(RowPassArgs a, RowPassGroup grp, int use_group,
                                                          const int32_t* __restrict__ hot_list, const uint32_t* __restrict__ hot_count,
                                                          uint32_t hot_cap, uint32_t seg_cap, const float* __restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (use_group) {
        const int i = blockIdx.y;
        a.W = grp.W[i]; a.m = grp.m[i]; a.v = grp.v[i]; a.shadow = grp.shadow[i]; a.src = grp.src[i]; a.src_layer = grp.src_layer[i];
        a.rows = grp.row_end[i] - (i ? grp.row_end[i - 1] : 0);
        if (partials) partials += (long)i * seg_cap * HotPart<NT>::value;
    }
    const uint32_t nh = min(hot_count[0], hot_cap);
    for (uint32_t h = blockIdx.x * 4 + wave; h < nh; h += gridDim.x * 4) {
        const int col = hot_list[h];
        const int r = col - a.col0;
        if (r < 0 || r >= a.rows) continue;                        // (wave-uniform; nothing below synchronises the block)
        const int layer = a.src_layer >= 0 ? a.src_layer : col / a.F;
        const int sb = partials ? hot_list[hot_cap + h] : -1;      // >= 0: the first list was walked by row_pass_seg_kernel
        float wv[NT][4], g[NT][4], mm[NT][4], vv[NT][4];
        row_pass_load<NT, true>(a, (long)r * a.d, lane, 0, wv, g, mm, vv, true);
        float csum = 0.0f;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t* cs = pass ? a.col_start2 : a.col_start;
            if (!cs) continue;
            const uint32_t p0 = cs[col], p1 = cs[col + 1];
            if (pass == 0 && sb >= 0) {
                const uint32_t nseg = (p1 - p0 + HOT_SEG - 1) / HOT_SEG;
#pragma unroll 1
                for (uint32_t q = 0; q < nseg; ++q) {
                    const float* src = partials + (long)(sb + q) * HotPart<NT>::value;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 x = *reinterpret_cast<const float4*>(src + (t * 64 + lane) * 4);
                        g[t][0] += x.x; g[t][1] += x.y; g[t][2] += x.z; g[t][3] += x.w;
                    }
                    csum += src[NT * 256];
                }
                continue;
            }
            if (pass == 0) {
                // Not segmented (the segment table was full when hot_columns_kernel reached this column — which columns that hits
                // depends on the order its atomics arrive in — or there is no scratch): the wave forms the SAME sums in the same
                // order as the segment blocks would — a segment's HW pieces from zero each, added in piece order, the segments
                // added in segment order — so a row's update does not depend on whether its column got slots.
                constexpr int HW = HotWaves<NT>::value;
#pragma unroll 1
                for (uint32_t s0 = p0; s0 < p1; s0 += HOT_SEG) {
                    const uint32_t s1 = min(p1, s0 + HOT_SEG), per = (s1 - s0 + HW - 1) / HW;
                    float sg[NT][4], sc = 0.0f;
#pragma unroll 1
                    for (int w = 0; w < HW; ++w) {
                        const uint32_t q0 = min(s1, s0 + (uint32_t)w * per), q1 = min(s1, q0 + per);
                        float gp[NT][4], cp = 0.0f;
#pragma unroll
                        for (int t = 0; t < NT; ++t) { gp[t][0] = 0.0f; gp[t][1] = 0.0f; gp[t][2] = 0.0f; gp[t][3] = 0.0f; }
                        row_pass_walk<NT, true, 1, F16, HOT_HU>(a, 0, col, layer, lane, 0, 0, q0, q1, p0, wv, gp, cp, nullptr);
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int c = 0; c < 4; ++c) sg[t][c] = w ? sg[t][c] + gp[t][c] : gp[t][c];
                        sc = w ? sc + cp : cp;
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) { g[t][0] += sg[t][0]; g[t][1] += sg[t][1]; g[t][2] += sg[t][2]; g[t][3] += sg[t][3]; }
                    csum += sc;
                }
                continue;
            }
            row_pass_walk<NT, true, 1, F16>(a, pass, col, layer, lane, 0, 0, p0, p1, p0, wv, g, csum, nullptr);
        }
        row_pass_finish<NT, true, 1, F16, true>(a, r, col, lane, 0, 0, wv, g, mm, vv, csum);
    }
}

// dval[p] = sum_{s < nslots(p)} dots[s*stride + p], nslots = L - layer(column of p) (cross-layer) or 1.  One thread per ENTRY
// (a thread per column serialises a hot column's thousands of entries); the entry's layer from the layers' first CSC positions.
__global__ void reduce_dots_kernel(const float* __restrict__ dots, long stride, const uint32_t* __restrict__ col_start, int ncols,
                                   int F, int L, int cross_layer, float* __restrict__ dval) {
    const uint32_t n = col_start[ncols];
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        int ns = 1;
        if (cross_layer) {
            int layer = 0;
            while (layer + 1 < L && col_start[(long)(layer + 1) * F] <= p) ++layer;
            ns = L - layer;
        }
        float s = 0.0f;
        for (int k = 0; k < ns; ++k) s += dots[(long)k * stride + p];
        dval[p] = s;
    }
}

// dense [rows][n] gradient of the sparsifier input from CSC-ordered values (for the un-fused autograd path)
__global__ void scatter_dense_kernel(const uint32_t* __restrict__ col_start, const int32_t* __restrict__ tok, const float* __restrict__ val,
                                     int ncols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    for (uint32_t p = col_start[c]; p < col_start[c + 1]; ++p) out[(long)tok[p] * ncols + c] = val[p];
}

// threshold lists: once the entry count is known, everything after the row scan runs only when the lists fit `cap`
// (above the caller's break-even the dense regime takes over and the truncated lists would be thrown away)
__global__ void list_gate_kernel(const uint32_t* __restrict__ total, uint32_t cap, int32_t* __restrict__ gate, int32_t* __restrict__ overflow) {
    const bool fits = *total <= cap;
    *gate = fits ? 1 : 0;
    if (!fits) *overflow = 1;
}

}  // namespace

CLT_DEFINE_TU_STATE_SETTER(clt_internal_set_state_sparse)

// dispatch on the number of 64-lane chunks a row needs
#define DISPATCH_NT(KERNEL, GRID, STREAM, D, VECOK, ...)                                                         \
    do {                                                                                                          \
        const int chunks_ = ((D) + ((VECOK) ? 256 : 64) - 1) / ((VECOK) ? 256 : 64);                              \
        if (VECOK) {                                                                                              \
            if (chunks_ <= 1) hipLaunchKernelGGL((KERNEL<1, true>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__);       \
            else if (chunks_ <= 2) hipLaunchKernelGGL((KERNEL<2, true>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__);  \
            else if (chunks_ <= 3) hipLaunchKernelGGL((KERNEL<3, true>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__);  \
            else if (chunks_ <= 4) hipLaunchKernelGGL((KERNEL<4, true>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__);  \
            else if (chunks_ <= 8) hipLaunchKernelGGL((KERNEL<8, true>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__);  \
            else if (chunks_ <= 16) hipLaunchKernelGGL((KERNEL<16, true>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__); \
            else return CLT_ERR_RANGE;                                                                            \
        } else {                                                                                                  \
            if (chunks_ <= 1) hipLaunchKernelGGL((KERNEL<1, false>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__);      \
            else if (chunks_ <= 2) hipLaunchKernelGGL((KERNEL<2, false>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__); \
            else if (chunks_ <= 4) hipLaunchKernelGGL((KERNEL<4, false>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__); \
            else if (chunks_ <= 16) hipLaunchKernelGGL((KERNEL<16, false>), GRID, dim3(TPB), 0, STREAM, __VA_ARGS__); \
            else return CLT_ERR_RANGE;                                                                            \
        }                                                                                                         \
    } while (0)

// workspace layout: row_cnt[rows] | col_cnt[ncols] | cursor[ncols] | long_list[ncols] | long_count
// gate: NULL = always; otherwise every kernel runs only when *gate != 0 (the fallback of clt_select_lists_f32).
int clt_internal_sparse_build(const void* feat, int B, int L, int F, uint32_t cap, uint32_t* row_start, int32_t* ent_col,
                              float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val, int32_t* csc_ent,
                              uint8_t* fired, uint32_t* total, int32_t* overflow, void* ws, size_t wsb, hipStream_t st,
                              const int32_t* gate, const float* theta, bool theta_gate, bool x16, const float* bias, int phase) {
    const int xt = x16 ? (clt_host_state.fp16 ? 2 : 1) : 0;     // element type of feat: fp32 / bf16 / fp16
    if (B <= 0 || L <= 0 || F <= 0 || cap == 0) return CLT_ERR_SHAPE;
    if ((unsigned long long)L * F >= 0x7fffffffull || (unsigned long long)B * L >= 0x7fffffffull) return CLT_ERR_RANGE;
    if (!ws || wsb < clt_sparse_build_workspace_bytes(B, L, F)) return CLT_ERR_WORKSPACE;
    const bool own_gate = !gate && theta_gate;
    const int rows = B * L, ncols = L * F;
    uint32_t* row_cnt = (uint32_t*)ws;
    uint32_t* col_cnt = row_cnt + rows;
    uint32_t* cursor = col_cnt + ncols;
    int32_t* long_list = (int32_t*)(cursor + ncols);
    uint32_t* long_count = (uint32_t*)(long_list + ncols);
    uint32_t* wave_cnt = long_count + 64;                  // [rows][4]: per-wave entry counts, count_rows -> fill_rows
    // phase (threshold lists): 0 = the whole build; 1 = counts, row offsets and the gate (overflow is final); 2 = the token-major
    // lists; 3 = the feature-major lists from them (the caller may queue phase 3 on a second stream: it reads what the earlier
    // phases wrote and the workspace)
    const dim3 g4((unsigned)(((B + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * L));
    const bool th4 = F % 4 == 0 && (reinterpret_cast<uintptr_t>(feat) & (uintptr_t)(xt == 0 ? 15 : 7)) == 0 &&
                     (!theta || (reinterpret_cast<uintptr_t>(theta) & 15) == 0);
    if (bias && (!th4 || (reinterpret_cast<uintptr_t>(bias) & 15))) return CLT_ERR_ALIGN;      // (a bias only in the vector form)
    if (phase >= 2) {
        if (own_gate) gate = (int32_t*)(long_count + 8);
    } else {
    if (!gate) {            // the gated caller clears these itself (a memset cannot be gated)
        if (hipMemsetAsync(col_cnt, 0, (size_t)2 * ncols * sizeof(uint32_t), st) != hipSuccess) return CLT_ERR_LAUNCH;
        if (hipMemsetAsync(long_count, 0, 64, st) != hipSuccess) return CLT_ERR_LAUNCH;
        if (hipMemsetAsync(overflow, 0, sizeof(int32_t), st) != hipSuccess) return CLT_ERR_LAUNCH;
    }
    // thresholds + vector form (every row has it when the bases are aligned and F % 4 == 0): four rows of a layer per block
    if (th4) {
#define COUNT_ROWS_TH(XT_)                                                                                                 \
    do {                                                                                                                    \
        if (theta) hipLaunchKernelGGL((count_rows_th_kernel<XT_, true>), g4, dim3(TPB), 0, st, feat, B, F, L, theta, bias, row_cnt, gate, 1, wave_cnt); \
        else hipLaunchKernelGGL((count_rows_th_kernel<XT_, false>), g4, dim3(TPB), 0, st, feat, B, F, L, theta, bias, row_cnt, gate, 1, wave_cnt);    \
    } while (0)
        if (xt == 0) COUNT_ROWS_TH(0); else if (xt == 1) COUNT_ROWS_TH(1); else COUNT_ROWS_TH(2);
#undef COUNT_ROWS_TH
    } else
    if (xt == 0) hipLaunchKernelGGL((count_rows_kernel<0>), dim3(rows), dim3(TPB), 0, st, feat, F, L, theta, row_cnt, gate, 1, wave_cnt);
    else if (xt == 1) hipLaunchKernelGGL((count_rows_kernel<1>), dim3(rows), dim3(TPB), 0, st, feat, F, L, theta, row_cnt, gate, 1, wave_cnt);
    else hipLaunchKernelGGL((count_rows_kernel<2>), dim3(rows), dim3(TPB), 0, st, feat, F, L, theta, row_cnt, gate, 1, wave_cnt);
    launch_scan(row_cnt, rows, row_start, total, (uint8_t*)nullptr, gate, 1, (uint32_t*)long_list, st);   // (long_list is idle until the row sort)
    if (own_gate) {         // (threshold lists: the remaining kernels exit at once when the count exceeds cap)
        int32_t* gw = (int32_t*)(long_count + 8);
        hipLaunchKernelGGL(list_gate_kernel, dim3(1), dim3(1), 0, st, total, cap, gw, overflow);
        gate = gw;
    }
    if (phase == 1) { CLT_LAUNCH_CHECK(); return CLT_OK; }
    }
    if (phase <= 2) {
#define FILL_ROWS(XT_) hipLaunchKernelGGL((fill_rows_kernel<XT_>), dim3(rows), dim3(TPB), 0, st, feat, F, L, theta, row_start, ent_col, ent_val, cap, \
                                          overflow, gate, 1, (const uint32_t*)wave_cnt)
#define FILL_ROWS_TH(XT_)                                                                                                  \
    do {                                                                                                                    \
        if (theta) hipLaunchKernelGGL((fill_rows_th_kernel<XT_, true>), g4, dim3(TPB), 0, st, feat, B, F, L, theta, bias, row_start, ent_col, ent_val, \
                                      cap, overflow, gate, 1, (const uint32_t*)wave_cnt);                                    \
        else hipLaunchKernelGGL((fill_rows_th_kernel<XT_, false>), g4, dim3(TPB), 0, st, feat, B, F, L, theta, bias, row_start, ent_col, ent_val,     \
                                cap, overflow, gate, 1, (const uint32_t*)wave_cnt);                                          \
    } while (0)
    if (th4) { if (xt == 0) FILL_ROWS_TH(0); else if (xt == 1) FILL_ROWS_TH(1); else FILL_ROWS_TH(2); }
    else if (xt == 0) FILL_ROWS(0); else if (xt == 1) FILL_ROWS(1); else FILL_ROWS(2);
#undef FILL_ROWS
#undef FILL_ROWS_TH
    if (phase == 2) { CLT_LAUNCH_CHECK(); return CLT_OK; }
    }
    hipLaunchKernelGGL(count_cols_kernel, dim3(1024), dim3(TPB), 0, st, ent_col, total, cap, col_cnt, gate, 1);
    launch_scan(col_cnt, ncols, col_start, nullptr, fired, gate, 1, (uint32_t*)long_list, st);   // (long_list is idle until sort_cols)
    hipLaunchKernelGGL(fill_cols_kernel, dim3((rows + 3) / 4), dim3(TPB), 0, st, row_start, ent_col, ent_val, rows, L, cap,
                       col_start, cursor, csc_tok, csc_val, csc_ent, gate, 1);
    hipLaunchKernelGGL(sort_cols_kernel, dim3((ncols + TPB - 1) / TPB), dim3(TPB), 0, st, col_start, ncols, csc_tok, csc_val,
                       csc_ent, long_list, long_count, gate, 1);
    launch_sort_queued_cols(st, col_start, csc_tok, csc_val, csc_ent, long_list, long_count, gate, 1);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

extern "C" {

size_t clt_sparse_build_workspace_bytes(int B, int L, int F) {
    const size_t rows = (size_t)B * L, ncols = (size_t)L * F;
    return (5 * rows + 3 * ncols + 64) * sizeof(uint32_t);
}

int clt_sparse_build_f32(const float* feat, int B, int L, int F, uint32_t cap, uint32_t* row_start, int32_t* ent_col,
                         float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val, int32_t* csc_ent,
                         uint8_t* fired, uint32_t* total, int32_t* overflow, void* ws, size_t wsb, void* stp) {
    return clt_internal_sparse_build(feat, B, L, F, cap, row_start, ent_col, ent_val, col_start, csc_tok, csc_val, csc_ent, fired,
                                     total, overflow, ws, wsb, as_stream(stp), nullptr, nullptr);
}

int clt_sparse_build_threshold_f32(const float* x, const float* theta, int B, int L, int F, uint32_t cap, uint32_t* row_start,
                                   int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val,
                                   int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow, void* ws, size_t wsb,
                                   int phase, void* stp) {
    if (phase < 0 || phase > 3) return CLT_ERR_SHAPE;
    return clt_internal_sparse_build(x, B, L, F, cap, row_start, ent_col, ent_val, col_start, csc_tok, csc_val, csc_ent, fired,
                                     total, overflow, ws, wsb, as_stream(stp), nullptr, theta, true, false, nullptr, phase);
}

int clt_sparse_build_threshold_x16(const uint16_t* x, const float* theta, const float* bias, int B, int L, int F, uint32_t cap, uint32_t* row_start,
                                   int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val,
                                   int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow, void* ws, size_t wsb,
                                   int phase, void* stp) {
    if (phase < 0 || phase > 3) return CLT_ERR_SHAPE;
    if (F % 4 || (reinterpret_cast<uintptr_t>(x) & 7)) return CLT_ERR_ALIGN;
    return clt_internal_sparse_build(x, B, L, F, cap, row_start, ent_col, ent_val, col_start, csc_tok, csc_val, csc_ent, fired,
                                     total, overflow, ws, wsb, as_stream(stp), nullptr, theta, true, true, bias, phase);
}

int clt_sparse_decoder_fwd_f32(const uint32_t* row_start, const int32_t* ent_col, const float* ent_val, const float* const* Wptr,
                               const float* bias, float* out, int B, int L, int F, int d, int cross_layer, uint32_t cap,
                               int w_bf16, long out_tok_stride, long out_layer_stride, int layer_lo, int layer_hi, void* stp) {
    if (B <= 0 || L <= 0 || F <= 0 || d <= 0) return CLT_ERR_SHAPE;
    if (layer_lo == 0 && layer_hi == 0) layer_hi = L;
    if (layer_lo < 0 || layer_hi > L || layer_lo >= layer_hi) return CLT_ERR_SHAPE;
    if (out_tok_stride == 0 && out_layer_stride == 0) { out_tok_stride = (long)L * d; out_layer_stride = d; }   // [B][L][d]
    const bool vec = d % 4 == 0 && aligned16(out) && out_tok_stride % 4 == 0 && out_layer_stride % 4 == 0;
#define FWDK(NT_, VEC_)                                                                                              \
    do {                                                                                                              \
        if (w_bf16 && clt_host_state.fp16)                                                                            \
            hipLaunchKernelGGL((sparse_decoder_fwd_kernel<NT_, VEC_, true, true>), dim3(B), dim3(TPB), 0, as_stream(stp), \
                               row_start, ent_col, ent_val, Wptr, bias, out, L, F, d, cross_layer, cap, out_tok_stride, out_layer_stride, layer_lo, layer_hi); \
        else if (w_bf16) hipLaunchKernelGGL((sparse_decoder_fwd_kernel<NT_, VEC_, true>), dim3(B), dim3(TPB), 0, as_stream(stp),  \
                                       row_start, ent_col, ent_val, Wptr, bias, out, L, F, d, cross_layer, cap, out_tok_stride, out_layer_stride, layer_lo, layer_hi); \
        else hipLaunchKernelGGL((sparse_decoder_fwd_kernel<NT_, VEC_, false>), dim3(B), dim3(TPB), 0, as_stream(stp), \
                                row_start, ent_col, ent_val, Wptr, bias, out, L, F, d, cross_layer, cap, out_tok_stride, out_layer_stride, layer_lo, layer_hi); \
    } while (0)
    const int chunks = (d + (vec ? 256 : 64) - 1) / (vec ? 256 : 64);
    if (vec) {
        if (chunks <= 1) FWDK(1, true); else if (chunks <= 2) FWDK(2, true); else if (chunks <= 3) FWDK(3, true);
        else if (chunks <= 4) FWDK(4, true); else if (chunks <= 8) FWDK(8, true); else if (chunks <= 16) FWDK(16, true);
        else return CLT_ERR_RANGE;
    } else {
        if (chunks <= 1) FWDK(1, false); else if (chunks <= 2) FWDK(2, false); else if (chunks <= 4) FWDK(4, false);
        else if (chunks <= 16) FWDK(16, false); else return CLT_ERR_RANGE;
    }
#undef FWDK
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_sparse_row_pass_f32(float* W, float* m, float* v, uint16_t* shadow, float* grad_out, int rows, int d, int F, int L,
                            int col0, int src_layer, const uint32_t* col_start, const int32_t* tok, const float* coef,
                            const float* src, const uint32_t* col_start2, const int32_t* tok2, const float* coef2,
                            const float* src2, float* dot_out, float* dot_out2, long dot_stride, long dot_stride2, int dot_slot,
                            float* colsum_out,
                            double lr,
                            double beta1, double beta2, double eps, double weight_decay, int adamw, int step, int src_bf16,
                            const float* extra, const int32_t* extra_map, const float* extra_colsum, const float* row_dnorm,
                            const float* row_sq, float* sq_out, const int32_t* hot_list, const uint32_t* hot_count, uint32_t hot_cap,
                            int hot_min_len, uint32_t hot_seg_cap, float* hot_partials, int hot_phase, void* stp) {
    if (rows <= 0 || d <= 0 || F <= 0 || L <= 0) return CLT_ERR_SHAPE;
    if (hot_list && (!hot_count || hot_cap == 0 || hot_min_len < 1)) return CLT_ERR_SHAPE;
    if (hot_phase < 0 || hot_phase > 2 || (hot_phase && !hot_list)) return CLT_ERR_SHAPE;
    if ((row_dnorm != nullptr) != (row_sq != nullptr)) return CLT_ERR_SHAPE;
    if (sq_out && (grad_out || d > 1024)) return CLT_ERR_SHAPE;   // the updated row's norm: Adam form, one wave per row
    if (!grad_out && (!m || !v || step < 1)) return CLT_ERR_SHAPE;
    RowPassArgs a = {};
    a.W = W; a.m = m; a.v = v; a.shadow = shadow; a.grad_out = grad_out; a.rows = rows; a.d = d; a.F = F; a.col0 = col0;
    a.src_bf16 = src_bf16;
    if (extra && !extra_map) return CLT_ERR_SHAPE;
    a.extra = extra; a.extra_map = extra_map; a.extra_colsum = extra ? extra_colsum : nullptr;
    a.row_dnorm = row_dnorm; a.row_sq = row_sq; a.sq_out = sq_out;
    a.src_layer = src_layer; a.L = L; a.col_start = col_start; a.tok = tok; a.coef = coef; a.src = src;
    a.col_start2 = col_start2; a.tok2 = tok2; a.coef2 = coef2; a.src2 = src2; a.dot_out = dot_out; a.dot_out2 = dot_out2; a.dot_stride = dot_stride; a.dot_stride2 = dot_stride2;
    a.dot_slot_base = dot_slot; a.colsum_out = colsum_out;
    const int s = step < 1 ? 1 : step;
    const double bc1 = 1.0 - pow(beta1, s), bc2 = 1.0 - pow(beta2, s);
    a.lr = (float)lr; a.omb1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.wd = (float)weight_decay; a.step_size = (float)(lr / bc1); a.isb2 = (float)(1.0 / sqrt(bc2)); a.adamw = adamw;
    const bool vec = d % 4 == 0 && aligned16(W) && (src_bf16 ? (reinterpret_cast<uintptr_t>(src) & 7) == 0 : aligned16(src)) && (!m || aligned16(m)) && (!v || aligned16(v)) &&
                     (!grad_out || aligned16(grad_out)) && (!src2 || (src_bf16 ? (reinterpret_cast<uintptr_t>(src2) & 7) == 0 : aligned16(src2))) &&
                     (!shadow || (reinterpret_cast<uintptr_t>(shadow) & 7) == 0) && (!extra || aligned16(extra));
    hipStream_t st = as_stream(stp);
    const int chunks = (d + (vec ? 256 : 64) - 1) / (vec ? 256 : 64);   // 64-lane iterations a whole row needs
    const dim3 g4((rows + 3) / 4), g1(rows);
    const bool hot = hot_list && vec && chunks <= 4;          // (one wave per row with 16-byte lanes: the hot-row form applies)
    a.hot_min_len = hot ? hot_min_len : 0;
    // hot_phase 1 / 2: the common launch alone / the hot rows' launches alone — the caller runs the two on different streams (they
    // touch disjoint rows and dot slots).  A shape the hot-row form does not apply to has no hot phase: everything is phase 1
    if (hot_phase == 2 && !hot) return CLT_OK;
    const bool common = hot_phase != 2, hot_rows = hot && hot_phase != 1;
#define RP(NT_, VEC_, CW_, GRID_)                                                                                          \
    do {                                                                                                                    \
        if (CW_ == 1 && sq_out) {                                                                                           \
            if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_kernel<NT_, VEC_, 1, true, true>), GRID_, dim3(TPB), 0, st, a); \
            else hipLaunchKernelGGL((row_pass_kernel<NT_, VEC_, 1, false, true>), GRID_, dim3(TPB), 0, st, a);              \
        } else if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_kernel<NT_, VEC_, CW_, true>), GRID_, dim3(CW_ == 1 ? TPB : 64 * CW_), 0, st, a); \
        else hipLaunchKernelGGL((row_pass_kernel<NT_, VEC_, CW_>), GRID_, dim3(CW_ == 1 ? TPB : 64 * CW_), 0, st, a);       \
    } while (0)
    if (!common) {
    } else if (vec) {
        if (chunks <= 1) RP(1, true, 1, g4);
        else if (chunks <= 2) RP(2, true, 1, g4);
        else if (chunks <= 3) RP(3, true, 1, g4);
        else if (chunks <= 4) RP(4, true, 1, g4);
        else if (chunks <= 8) RP(4, true, 2, g1);
        else if (chunks <= 16) RP(4, true, 4, g1);
        else if (chunks <= 32) RP(4, true, 8, g1);
        else if (chunks <= 64) RP(4, true, 16, g1);
        else return CLT_ERR_RANGE;
    } else {
        if (chunks <= 1) RP(1, false, 1, g4);
        else if (chunks <= 2) RP(2, false, 1, g4);
        else if (chunks <= 4) RP(4, false, 1, g4);
        else if (chunks <= 16) RP(16, false, 1, g4);
        else return CLT_ERR_RANGE;
    }
#undef RP
    if (hot_rows) {
        RowPassGroup none = {};
#define RPH(NT_)                                                                                                          \
    do {                                                                                                                    \
        if (hot_partials) {                                                                                                 \
            if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_seg_kernel<NT_, true>), dim3(HOT_GRID), dim3(64 * HotWaves<NT_>::value), 0, st, a, none, 0, hot_list, hot_count, hot_cap, hot_seg_cap, hot_partials); \
            else hipLaunchKernelGGL((row_pass_seg_kernel<NT_, false>), dim3(HOT_GRID), dim3(64 * HotWaves<NT_>::value), 0, st, a, none, 0, hot_list, hot_count, hot_cap, hot_seg_cap, hot_partials); \
        }                                                                                                                   \
        if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_hot_kernel<NT_, true>), dim3(HOT_GRID), dim3(TPB), 0, st, a, none, 0, hot_list, hot_count, hot_cap, hot_seg_cap, (const float*)hot_partials); \
        else hipLaunchKernelGGL((row_pass_hot_kernel<NT_, false>), dim3(HOT_GRID), dim3(TPB), 0, st, a, none, 0, hot_list, hot_count, hot_cap, hot_seg_cap, (const float*)hot_partials); \
    } while (0)
        if (chunks <= 1) RPH(1); else if (chunks <= 2) RPH(2); else if (chunks <= 3) RPH(3); else RPH(4);
#undef RPH
    }
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_sparse_hot_columns(const uint32_t* col_start, int ncols, int min_len, int32_t* hot_list, uint32_t hot_cap, uint32_t seg_cap,
                           uint32_t* hot_count, void* stp) {
    if (!col_start || ncols <= 0 || min_len < 1 || !hot_list || !hot_count || hot_cap == 0) return CLT_ERR_SHAPE;
    if (hipMemsetAsync(hot_count, 0, 2 * sizeof(uint32_t), as_stream(stp)) != hipSuccess) return CLT_ERR_LAUNCH;
    // (a column that finds no room leaves a hole in the segment table: holes read as hot index -1)
    if (seg_cap && hipMemsetAsync(hot_list + 2 * (size_t)hot_cap, 0xff, 2 * (size_t)seg_cap * sizeof(int32_t), as_stream(stp)) != hipSuccess)
        return CLT_ERR_LAUNCH;
    hipLaunchKernelGGL(hot_columns_kernel, dim3((ncols + TPB - 1) / TPB), dim3(TPB), 0, as_stream(stp), col_start, ncols, (uint32_t)min_len,
                       hot_list, hot_cap, seg_cap, hot_count);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_sparse_row_pass_group_f32(int nmat, float* const* W, float* const* m, float* const* v, uint16_t* const* shadow,
                                   const int* rows, const int* src_layer, const void* const* src, int d, int F, int L,
                                   const uint32_t* col_start, const int32_t* tok, const float* coef, float* dot_out, long dot_stride,
                                   double lr, double beta1, double beta2, double eps, double weight_decay, int adamw, int step,
                                   int src_bf16, const int32_t* hot_list, const uint32_t* hot_count, uint32_t hot_cap, int hot_min_len,
                                   uint32_t hot_seg_cap, float* hot_partials, void* stp) {
    if (hot_list && (!hot_count || hot_cap == 0 || hot_min_len < 1)) return CLT_ERR_SHAPE;
    if (nmat <= 0 || d <= 0 || F <= 0 || L <= 0 || step < 1 || !W || !m || !v || !rows || !src_layer || !src) return CLT_ERR_SHAPE;
    if (nmat > RP_GROUP_MAX) return 1;
    const int chunks = (d + 255) / 256;
    if (d % 4 || chunks > 4) return 1;                       // (one wave per row with 16-byte lanes only: the caller launches per matrix)
    RowPassArgs a = {};
    RowPassGroup g = {};
    g.n = nmat;
    long total = 0;
    for (int i = 0; i < nmat; ++i) {
        if (rows[i] <= 0 || src_layer[i] < 0 || !W[i] || !m[i] || !v[i] || !src[i]) return CLT_ERR_SHAPE;
        if (!aligned16(W[i]) || !aligned16(m[i]) || !aligned16(v[i]) || (shadow && shadow[i] && (reinterpret_cast<uintptr_t>(shadow[i]) & 7)) ||
            (src_bf16 ? (reinterpret_cast<uintptr_t>(src[i]) & 7) != 0 : !aligned16(src[i]))) return 1;
        total += rows[i];
        if (total > 0x7fffffffL) return CLT_ERR_RANGE;
        g.row_end[i] = (int)total;
        g.W[i] = W[i]; g.m[i] = m[i]; g.v[i] = v[i]; g.shadow[i] = shadow ? shadow[i] : nullptr;
        g.src[i] = static_cast<const float*>(src[i]); g.src_layer[i] = src_layer[i];
    }
    a.d = d; a.F = F; a.L = L; a.col0 = 0; a.col_start = col_start; a.tok = tok; a.coef = coef;
    a.dot_out = dot_out; a.dot_stride = dot_stride; a.dot_slot_base = -1; a.src_bf16 = src_bf16;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    a.lr = (float)lr; a.omb1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.wd = (float)weight_decay; a.step_size = (float)(lr / bc1); a.isb2 = (float)(1.0 / sqrt(bc2)); a.adamw = adamw;
    hipStream_t st = as_stream(stp);
    const dim3 grid((unsigned)((total + 3) / 4));
    a.hot_min_len = hot_list ? hot_min_len : 0;
#define RPG(NT_)                                                                                             \
    do {                                                                                                      \
        if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_group_kernel<NT_, true>), grid, dim3(TPB), 0, st, a, g); \
        else hipLaunchKernelGGL((row_pass_group_kernel<NT_, false>), grid, dim3(TPB), 0, st, a, g);           \
    } while (0)
    if (chunks <= 1) RPG(1); else if (chunks <= 2) RPG(2); else if (chunks <= 3) RPG(3); else RPG(4);
#undef RPG
    if (hot_list) {                                        // the hot rows of every matrix of the group: grid.y = matrix
#define RPGH(NT_)                                                                                              \
    do {                                                                                                        \
        if (hot_partials) {                                                                                     \
            if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_seg_kernel<NT_, true>), dim3(HOT_GRID / 4, nmat), dim3(64 * HotWaves<NT_>::value), 0, st, a, g, 1, hot_list, hot_count, hot_cap, hot_seg_cap, hot_partials); \
            else hipLaunchKernelGGL((row_pass_seg_kernel<NT_, false>), dim3(HOT_GRID / 4, nmat), dim3(64 * HotWaves<NT_>::value), 0, st, a, g, 1, hot_list, hot_count, hot_cap, hot_seg_cap, hot_partials); \
        }                                                                                                       \
        if (clt_host_state.fp16) hipLaunchKernelGGL((row_pass_hot_kernel<NT_, true>), dim3(HOT_GRID / 4, nmat), dim3(TPB), 0, st, a, g, 1, hot_list, hot_count, hot_cap, hot_seg_cap, (const float*)hot_partials); \
        else hipLaunchKernelGGL((row_pass_hot_kernel<NT_, false>), dim3(HOT_GRID / 4, nmat), dim3(TPB), 0, st, a, g, 1, hot_list, hot_count, hot_cap, hot_seg_cap, (const float*)hot_partials); \
    } while (0)
        if (chunks <= 1) RPGH(1); else if (chunks <= 2) RPGH(2); else if (chunks <= 3) RPGH(3); else RPGH(4);
#undef RPGH
    }
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_sparse_reduce_dots_f32(const float* dots, long stride, const uint32_t* col_start, int L, int F, int cross_layer,
                               float* dval, void* stp) {
    if (L <= 0 || F <= 0) return CLT_ERR_SHAPE;
    const int ncols = L * F;
    hipLaunchKernelGGL(reduce_dots_kernel, dim3(2048), dim3(TPB), 0, as_stream(stp), dots, stride, col_start,
                       ncols, F, L, cross_layer, dval);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_sparse_scatter_dense_f32(const uint32_t* col_start, const int32_t* tok, const float* val, int B, int ncols, float* out,
                                 void* stp) {
    if (B <= 0 || ncols <= 0) return CLT_ERR_SHAPE;
    if (hipMemsetAsync(out, 0, (size_t)B * ncols * sizeof(float), as_stream(stp)) != hipSuccess) return CLT_ERR_LAUNCH;
    hipLaunchKernelGGL(scatter_dense_kernel, dim3((ncols + TPB - 1) / TPB), dim3(TPB), 0, as_stream(stp), col_start, tok, val,
                       ncols, out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
}  // extern "C"
