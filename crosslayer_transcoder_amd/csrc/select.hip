// Exact top-k selection for the TopK / BatchTopK / PerLayerBatchTopK sparsifiers (gfx950).
// Replaces torch.topk + relu + scatter_ of the reference (model/topk.py:58-61, 84-91, 117-128).
//
// Method: MSB-first radix select (11 + 11 + 10 bits) on the order-preserving integer key of each
// fp32 pre-activation, so the k-th largest value T is found EXACTLY with streaming, coalesced reads
// and LDS histograms; everything > T is selected, and of the entries == T the lowest-index ones fill
// the remaining slots (canonical tie rule of the build).  Entries whose value is <= 0
// contribute relu(v) = 0, so ties at T <= 0 never change the output and are not ranked.
#include "select_common.h"

namespace {
// ------------------------------------------------------------------------------------------------
// mode 0: k of F per row, one block per row
// ------------------------------------------------------------------------------------------------
constexpr int MAXK_COMPACT = 2048;
__global__ __launch_bounds__(TPB) void topk_rows_kernel(const float* __restrict__ x, float* __restrict__ feat, int F, int k,
                                                        int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                        const uint8_t* __restrict__ col_mask, int L,
                                                        uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ col_cnt) {
    __shared__ uint32_t hist[NBIN];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t res[4];
    __shared__ int32_t cidx[MAXK_COMPACT];
    __shared__ float cval[MAXK_COMPACT];
    const long row = blockIdx.x;
    const float* xr = x + row * F;
    // optional column mask (where(dead, x, -inf) of clt_lightning.py:544-545 applied on the fly)
    const uint8_t* mk = col_mask ? col_mask + (row % L) * F : nullptr;
    auto ld = [&](int i) { return (mk && !mk[i]) ? -INFINITY : xr[i]; };
    uint32_t prefix = 0, mask = 0, krem = k, eq_total = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int nb = 1 << BITS[p];
        for (int i = threadIdx.x; i < NBIN; i += TPB) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < F; i += TPB) {
            const uint32_t key = f32_key(ld(i));
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> SHIFT[p]) & (nb - 1)], 1u);
        }
        __syncthreads();
        find_bin(hist, nb, krem, sh, res);
        prefix |= res[0] << SHIFT[p];
        mask |= (uint32_t)(nb - 1) << SHIFT[p];
        krem = res[1];
        eq_total = res[2];
        __syncthreads();
    }
    const uint32_t T = prefix;
    const bool rank_ties = (eq_total != krem) && (key_f32(T) > 0.0f);
    if (threadIdx.x == 0) res[3] = 0;       // compact-list cursor
    __syncthreads();
    uint32_t running = 0;
    float* fr = feat ? feat + row * F : nullptr;
    for (int base = 0; base < F; base += TPB) {
        const int i = base + threadIdx.x;
        float v = 0.0f; uint32_t key = 0;
        if (i < F) { v = ld(i); key = f32_key(v); }
        bool sel = (i < F) && key > T;
        const bool eq = (i < F) && key == T;
        if (rank_ties) {
            uint32_t tot;
            const uint32_t r = block_excl_scan(eq ? 1u : 0u, sh, &tot);
            sel |= eq && (running + r < krem);
            running += tot;
        } else sel |= eq;
        const float o = (sel && v > 0.0f) ? v : 0.0f;
        if (i < F && fr) fr[i] = o;
        if (idx_out && o > 0.0f) {
            const uint32_t s = atomicAdd(&res[3], 1u); cidx[s] = i; cval[s] = o;
            if (col_cnt) atomicAdd(&col_cnt[(row % L) * F + i], 1u);
        }
    }
    if (idx_out) {                          // order the <= k positive entries by feature index (rank sort)
        __syncthreads();
        const int n = (int)res[3];
        if (row_cnt && threadIdx.x == 0) row_cnt[row] = n;
        for (int e = threadIdx.x; e < k; e += TPB) {
            if (e < n) {
                const int me = cidx[e];
                int pos = 0;
                for (int q = 0; q < n; ++q) pos += cidx[q] < me;
                idx_out[row * k + pos] = me; val_out[row * k + pos] = cval[e];
            } else { idx_out[row * k + e] = -1; val_out[row * k + e] = 0.0f; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// modes 1/2: S segments (1 = whole batch, L = per layer).  Segment-row q of segment s is the
// contiguous row  (S == 1 ? q : q*L + s)  of F elements; block (c, s) owns rows [c*rpb, (c+1)*rpb).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long seg_row(int q, int s, int S, int L) { return S == 1 ? (long)q : (long)q * L + s; }

__global__ __launch_bounds__(TPB) void seg_hist_kernel(const float* __restrict__ x, int F, int L, int S, int RS, int rpb,
                                                       const SegState* __restrict__ state, uint32_t* __restrict__ ghist, int p, const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
    __shared__ uint32_t hist[NBIN];
    const int s = blockIdx.y, nb = 1 << BITS[p];
    const uint32_t prefix = state[s].prefix, mask = state[s].mask;
    for (int i = threadIdx.x; i < NBIN; i += TPB) hist[i] = 0;
    __syncthreads();
    const int q0 = blockIdx.x * rpb, q1 = min(RS, q0 + rpb);
    for (int q = q0; q < q1; ++q) {
        const float* xr = x + seg_row(q, s, S, L) * F;
        for (int i = threadIdx.x; i < F; i += TPB) {
            const uint32_t key = f32_key(xr[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> SHIFT[p]) & (nb - 1)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += TPB) { const uint32_t h = hist[i]; if (h) atomicAdd(&ghist[(long)s * NBIN + i], h); }
}

__global__ __launch_bounds__(TPB) void seg_count_eq_kernel(const float* __restrict__ x, int F, int L, int S, int RS, int rpb,
                                                           const SegState* __restrict__ state, uint32_t* __restrict__ eqcnt, int NC, const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
    __shared__ uint32_t sh[8];
    const int s = blockIdx.y;
    const SegState st = state[s];
    if (!need_rank(st)) return;
    uint32_t c = 0;
    const int q0 = blockIdx.x * rpb, q1 = min(RS, q0 + rpb);
    for (int q = q0; q < q1; ++q) {
        const float* xr = x + seg_row(q, s, S, L) * F;
        for (int i = threadIdx.x; i < F; i += TPB) c += f32_key(xr[i]) == st.prefix;
    }
    uint32_t tot;
    block_excl_scan(c, sh, &tot);
    if (threadIdx.x == 0) eqcnt[(long)s * NC + blockIdx.x] = tot;
}

// one block per segment: exclusive scan of the per-chunk tie counts (only if ranking is needed) + EMA
__global__ void seg_scan_kernel(const SegState* __restrict__ state, uint32_t* __restrict__ eqcnt, int NC,
                                float* __restrict__ threshold, float e, const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
    const int s = blockIdx.x;
    if (threadIdx.x != 0) return;
    const SegState st = state[s];
    if (need_rank(st)) {
        uint32_t run = 0;
        for (int c = 0; c < NC; ++c) { const uint32_t v = eqcnt[(long)s * NC + c]; eqcnt[(long)s * NC + c] = run; run += v; }
    }
    if (threadIdx.x == 0 && threshold) {
        const float kth = key_f32(st.prefix);
        threshold[s] = (1.0f - e) * threshold[s] + e * (kth > 0.0f ? kth : 0.0f);   // topk.py:94-95, 131-132
    }
}

__global__ __launch_bounds__(TPB) void seg_emit_kernel(const float* __restrict__ x, float* __restrict__ feat, int F, int L, int S,
                                                       int RS, int rpb, const SegState* __restrict__ state,
                                                       const uint32_t* __restrict__ eqoff, int NC, const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
    __shared__ uint32_t sh[8];
    const int s = blockIdx.y;
    const SegState st = state[s];
    const uint32_t T = st.prefix;
    const bool rank_ties = need_rank(st);
    uint32_t running = rank_ties ? eqoff[(long)s * NC + blockIdx.x] : 0;
    const int q0 = blockIdx.x * rpb, q1 = min(RS, q0 + rpb);
    for (int q = q0; q < q1; ++q) {
        const long ro = seg_row(q, s, S, L) * F;
        for (int base = 0; base < F; base += TPB) {
            const int i = base + threadIdx.x;
            float v = 0.0f; uint32_t key = 0;
            if (i < F) { v = x[ro + i]; key = f32_key(v); }
            bool sel = (i < F) && key > T;
            const bool eq = (i < F) && key == T;
            if (rank_ties) {
                uint32_t tot;
                const uint32_t r = block_excl_scan(eq ? 1u : 0u, sh, &tot);
                sel |= eq && (running + r < st.krem);
                running += tot;
            } else sel |= eq;
            if (i < F) feat[ro + i] = (sel && v > 0.0f) ? v : 0.0f;
        }
    }
}

__global__ void seg_init_kernel(SegState* state, int S, uint32_t K, const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) { SegState st; st.prefix = 0; st.mask = 0; st.krem = K; st.eq_total = 0; state[s] = st; }
}

struct SegPlan { int S, RS, rpb, NC; size_t off_hist, off_eq, total; };
SegPlan plan(int B, int L, int F, int per_layer) {
    SegPlan p;
    p.S = per_layer ? L : 1;
    p.RS = per_layer ? B : B * L;
    p.rpb = F >= 65536 ? 1 : (65536 / F);
    if (p.rpb > p.RS) p.rpb = p.RS;
    p.NC = (p.RS + p.rpb - 1) / p.rpb;
    p.off_hist = align_up((size_t)p.S * sizeof(SegState), 256);
    p.off_eq = p.off_hist + (size_t)p.S * NBIN * sizeof(uint32_t);
    p.total = p.off_eq + (size_t)p.S * p.NC * sizeof(uint32_t);
    return p;
}
}  // namespace

int clt_internal_topk_rows(const float* x, float* feat, int rows, int F, int k, int32_t* idx_out, float* val_out,
                           const uint8_t* col_mask, int L, uint32_t* row_cnt, uint32_t* col_cnt, hipStream_t st) {
    if (rows <= 0 || F <= 0 || k < 0 || L <= 0) return CLT_ERR_SHAPE;
    if (k > F) return CLT_ERR_K_GT_F;
    if ((idx_out != nullptr) != (val_out != nullptr)) return CLT_ERR_SHAPE;
    if (idx_out && k > MAXK_COMPACT) return CLT_ERR_RANGE;
    if (!feat && !idx_out) return CLT_ERR_SHAPE;
    if (k == 0) {
        if (feat && hipMemsetAsync(feat, 0, (size_t)rows * F * sizeof(float), st) != hipSuccess) return CLT_ERR_LAUNCH;
        if (row_cnt && hipMemsetAsync(row_cnt, 0, (size_t)rows * sizeof(uint32_t), st) != hipSuccess) return CLT_ERR_LAUNCH;
        return CLT_OK;
    }
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(TPB), 0, st, x, feat, F, k, idx_out, val_out, col_mask, L, row_cnt,
                       col_cnt);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_internal_topk_segmented(const float* x, float* feat, int B, int L, int F, int k, int per_layer, float* threshold,
                                float e, void* ws, size_t wsb, hipStream_t st, const int32_t* gate) {
    if (B <= 0 || L <= 0 || F <= 0 || k < 0) return CLT_ERR_SHAPE;
    if (k > F) return CLT_ERR_K_GT_F;
    const unsigned long long total = (unsigned long long)B * L * F;
    if (total >= 0xffffffffull) return CLT_ERR_RANGE;
    if (k == 0) {                                        // topk.py:86-87, 119-122: zeros, no threshold update
        if (gate) return CLT_ERR_SHAPE;                  // (the gated caller handles k == 0 itself)
        if (hipMemsetAsync(feat, 0, total * sizeof(float), st) != hipSuccess) return CLT_ERR_LAUNCH;
        return CLT_OK;
    }
    const SegPlan p = plan(B, L, F, per_layer);
    if (!ws || wsb < p.total) return CLT_ERR_WORKSPACE;
    SegState* state = (SegState*)ws;
    uint32_t* ghist = (uint32_t*)((char*)ws + p.off_hist);
    uint32_t* eqcnt = (uint32_t*)((char*)ws + p.off_eq);
    const uint32_t K = (uint32_t)((unsigned long long)k * B * (per_layer ? 1 : L));
    if (hipMemsetAsync(ghist, 0, (size_t)p.S * NBIN * sizeof(uint32_t), st) != hipSuccess) return CLT_ERR_LAUNCH;
    hipLaunchKernelGGL(seg_init_kernel, dim3((p.S + 63) / 64), dim3(64), 0, st, state, p.S, K, gate);
    const dim3 grid(p.NC, p.S);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(seg_hist_kernel, grid, dim3(TPB), 0, st, x, F, L, p.S, p.RS, p.rpb, state, ghist, pass, gate);
        hipLaunchKernelGGL(seg_find_kernel, dim3(p.S), dim3(TPB), 0, st, state, ghist, pass, gate, 1);
    }
    hipLaunchKernelGGL(seg_count_eq_kernel, grid, dim3(TPB), 0, st, x, F, L, p.S, p.RS, p.rpb, state, eqcnt, p.NC, gate);
    hipLaunchKernelGGL(seg_scan_kernel, dim3(p.S), dim3(64), 0, st, state, eqcnt, p.NC, threshold, e, gate);
    hipLaunchKernelGGL(seg_emit_kernel, grid, dim3(TPB), 0, st, x, feat, F, L, p.S, p.RS, p.rpb, state, eqcnt, p.NC, gate);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

extern "C" {
int clt_topk_rows_f32(const float* x, float* feat, int rows, int F, int k, int32_t* idx_out, float* val_out, void* st) {
    return clt_internal_topk_rows(x, feat, rows, F, k, idx_out, val_out, nullptr, 1, nullptr, nullptr, as_stream(st));
}

size_t clt_topk_segmented_workspace_bytes(int B, int L, int F, int per_layer) {
    if (B <= 0 || L <= 0 || F <= 0) return 0;
    return plan(B, L, F, per_layer).total;
}

int clt_topk_segmented_f32(const float* x, float* feat, int B, int L, int F, int k, int per_layer, float* threshold,
                           float e, void* ws, size_t wsb, void* stp) {
    return clt_internal_topk_segmented(x, feat, B, L, F, k, per_layer, threshold, e, ws, wsb, as_stream(stp), nullptr);
}
}  // extern "C"
