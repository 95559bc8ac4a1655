// Exact top-k selection for the TopK / BatchTopK / PerLayerBatchTopK sparsifiers (gfx950).
// Replaces torch.topk + relu + scatter_ of the reference (model/topk.py:58-61, 84-91, 117-128).
//
// Method: MSB-first radix select (11 + 11 + 10 bits) on the order-preserving integer key of each
// fp32 pre-activation, so the k-th largest value T is found EXACTLY with streaming, coalesced reads
// and LDS histograms; everything > T is selected, and of the entries == T the lowest-index ones fill
// the remaining slots (canonical tie rule of the build).  Entries whose value is <= 0
// contribute relu(v) = 0, so ties at T <= 0 never change the output and are not ranked.
#include <stdlib.h>
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
// mode 0, register-resident rows (F <= 32768, 16-byte aligned rows): one 1024-thread block per row reads the row ONCE
// (coalesced 16-byte loads) and keeps the order-preserving keys in registers.  The exact k-th largest key is found by a
// binary search on the key bits, two bits per step (three thresholds counted at once with wave ballots, one workgroup
// barrier per step) — first over the 1024 per-thread maxima, whose k-th largest is a lower bound of the row's k-th
// largest (k <= 1024), then over the few candidates >= that bound (compacted into LDS).  No LDS atomics on the data,
// so no contention on the handful of exponent bins that nearly all pre-activations share.
// ------------------------------------------------------------------------------------------------
constexpr int RT = 1024, RW = RT / 64, RCAP = 2048, RANK_TAIL_MAX = 256;
// block barrier that orders LDS traffic only (global loads / stores / DMA pieces stay in flight across it)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
extern __shared__ __attribute__((aligned(16))) unsigned char rowbuf_raw[];   // the next row, filled by LDS-DMA

// m-th largest of the 64 values one wave holds (one per lane): binary search on the key bits, two bits per step, counted
// with ballots — scalar work only, no LDS, no barrier.
__device__ __forceinline__ uint32_t wave_kth(uint32_t v, uint32_t m) {
    uint32_t t = 0;
    for (int b = 30; b >= 0; b -= 2) {
        const uint32_t c1 = (uint32_t)__popcll(__ballot(v >= (t | (1u << b))));
        const uint32_t c2 = (uint32_t)__popcll(__ballot(v >= (t | (2u << b))));
        const uint32_t c3 = (uint32_t)__popcll(__ballot(v >= (t | (3u << b))));
        if (c3 >= m) t |= 3u << b;
        else if (c2 >= m) t |= 2u << b;
        else if (c1 >= m) t |= 1u << b;
    }
    return t;
}

template <int NPT>
__global__ __launch_bounds__(RT) void topk_rows_reg_kernel(const float* __restrict__ x, float* __restrict__ feat, int F, int k,
                                                           int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                           const uint8_t* __restrict__ col_mask, int L,
                                                           uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ col_cnt, long nrows) {
    __shared__ uint2 red[2][RW];
    __shared__ uint32_t ckey[RCAP];
    __shared__ int32_t cidx[RCAP];                         // feature index of each candidate
    __shared__ uint32_t ccur;                              // candidate count (compaction cursor)
    __shared__ uint32_t wlb[RW];
    __shared__ uint32_t step_cnt[16][4];                   // block-wide search: per step the counts of 3 thresholds
    __shared__ uint32_t cursor, Tsh, cnt_sh[2];
    __shared__ int32_t oidx[1024];
    __shared__ float oval[1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(w);      // (scalar copy for the DMA's LDS base)
    // Persistent block: rows blockIdx.x, + gridDim.x, ...  While a row is processed out of registers the next one streams
    // into LDS by DMA (buffer_load ... lds, 16 bytes per lane, no registers), so the HBM read overlaps the selection work.
    float* rowbuf = reinterpret_cast<float*>(rowbuf_raw);
    // (MUBUF form of the DMA and LDS-only barriers: __syncthreads() carries a release fence, i.e. a vmcnt(0) that would wait
    // for the row just requested, and FLAT-encoded global_load_lds in flight turns every LDS wait into a full drain)
    auto prefetch = [&](long r) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + r * F), 0, F * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NPT / 4; ++j) {
            const int i0 = (j * RT + tid) * 4;
            if (i0 < F)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(rowbuf_raw + (size_t)(j * RT + wu * 64) * 16),
                                                         16, i0 * 4, 0, 0, 0);
        }
    };
    prefetch(blockIdx.x);
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
    const uint8_t* mk = col_mask ? col_mask + (row % L) * F : nullptr;
    int parity = 0;
    if (tid < 64) step_cnt[tid >> 2][tid & 3] = 0;
    if (tid == 0) { cursor = 0; ccur = 0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LDS_BARRIER();                                        // this row has landed in LDS (every wave's DMA pieces)
    // element e of this thread has index ((e / 4) * RT + tid) * 4 + e % 4; keys of positions >= F are 0 (below every real key)
    uint32_t key[NPT];
    uint32_t tmax = 0;
#pragma unroll
    for (int j = 0; j < NPT / 4; ++j) {
        const int i0 = (j * RT + tid) * 4;
        const bool in = i0 < F;
        const int ic = in ? i0 : 0;                            // branch-free: out-of-range groups re-read the row start
        const float4 v = *reinterpret_cast<const float4*>(rowbuf + ic);
        float a[4] = {v.x, v.y, v.z, v.w};
        if (mk) {
            const uchar4 m4 = *reinterpret_cast<const uchar4*>(mk + ic);
            a[0] = m4.x ? a[0] : -INFINITY; a[1] = m4.y ? a[1] : -INFINITY; a[2] = m4.z ? a[2] : -INFINITY; a[3] = m4.w ? a[3] : -INFINITY;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t kk = in ? f32_key(a[c]) : 0u; key[j * 4 + c] = kk; tmax = max(tmax, kk); }
    }
    LDS_BARRIER();                                        // every thread holds its keys: the buffer may be refilled
    if (row + gridDim.x < nrows) prefetch(row + gridDim.x);
    // lower bound of the k-th largest key: every wave holds at least m = ceil(k / 16) keys >= the m-th largest of its 64
    // per-thread maxima, so the minimum of those over the 16 waves is reached by at least k keys (k <= 1024)
    // (rows shorter than 4096: only the waves whose first group of 256 indices lies inside the row take part — an empty
    // wave would drag the bound to zero — and each of them then owes ceil(k / n_act) keys)
    uint32_t LB = 0;
    const int n_act = F >= 4 * RT ? RW : F / 256;
    if (k <= RT && n_act > 0) {
        const uint32_t mth = w < n_act ? wave_kth(tmax, (uint32_t)(k + n_act - 1) / n_act) : 0xffffffffu;
        if (lane == 0) wlb[w] = mth;
    }
    // candidates >= LB, compacted into LDS
    LDS_BARRIER();
    if (k <= RT && n_act > 0) {
        LB = wlb[0];
#pragma unroll
        for (int i = 1; i < RW; ++i) LB = min(LB, wlb[i]);
    }
    // (key, index) pairs of the candidates go to LDS in arrival order: per register slot one ballot — nearly always
    // empty, k << F — and one LDS atomic per wave that has any; the exact search does not depend on the order
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const bool c = key[e] >= LB;
        const unsigned long long m = __ballot(c);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ccur, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (c) {
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < (uint32_t)RCAP) { ckey[pos] = key[e]; cidx[pos] = ((e >> 2) * RT + tid) * 4 + (e & 3); }
        }
    }
    LDS_BARRIER();
    const uint32_t total = ccur;
    uint32_t T;
    // List output (no dense row): the row is finished from the candidate pairs alone.  Every candidate ranks itself in
    // the order (key descending, feature index ascending) — one thread per candidate, broadcast LDS reads of the others —
    // and the first k of that order are exactly the selection with the lowest-index tie rule; no search for the k-th key.
    // Quadratic in the candidates, so only for the usual short lists (k << F); longer ones take the search below.
    if (total <= (uint32_t)RANK_TAIL_MAX && !feat && idx_out) {
        for (uint32_t c = tid; c < total; c += RT) {
            const uint32_t mkey = ckey[c];
            const int mi = cidx[c];
            uint32_t rank = 0;
#pragma unroll 4
            for (uint32_t j = 0; j < total; ++j) {
                const uint32_t kj = ckey[j];
                const int ij = cidx[j];
                rank += (kj > mkey || (kj == mkey && ij < mi)) ? 1u : 0u;
            }
            const float v = key_f32(mkey);
            if (rank < (uint32_t)k && v > 0.0f) {
                const uint32_t sl = atomicAdd(&cursor, 1u);
                oidx[sl] = mi; oval[sl] = v;
                if (col_cnt) atomicAdd(&col_cnt[(row % L) * F + mi], 1u);
            }
        }
        LDS_BARRIER();
        const int n = (int)cursor;
        if (row_cnt && tid == 0) row_cnt[row] = n;
        for (int e = tid; e < k; e += RT) {                 // order the <= k positive entries by feature index (rank sort)
            if (e < n) {
                const int me = oidx[e];
                int pos = 0;
                for (int q = 0; q < n; ++q) pos += oidx[q] < me;
                idx_out[row * k + pos] = me; val_out[row * k + pos] = oval[e];
            } else { idx_out[row * k + e] = -1; val_out[row * k + e] = 0.0f; }
        }
        LDS_BARRIER();                                      // the per-row LDS state is reset at the top of the next row
        continue;
    }
    if (total <= (uint32_t)RCAP) {
        if (total <= 64u) {                                 // the usual case for k << F: one wave finishes the search alone
            if (w == 0) {
                const uint32_t ckl = (uint32_t)lane < total ? ckey[lane] : 0u;
                const uint32_t t = wave_kth(ckl, (uint32_t)k);
                const uint32_t g1 = (uint32_t)__popcll(__ballot(ckl > t)), e1 = (uint32_t)__popcll(__ballot(ckl == t && (uint32_t)lane < total));
                if (lane == 0) { Tsh = t; cnt_sh[0] = g1; cnt_sh[1] = e1; }     // every key >= T is a candidate (T >= LB)
            }
        } else {                                            // all waves count, one LDS atomic per wave, threshold and step
            uint32_t ck[RCAP / RT];
#pragma unroll
            for (int q = 0; q < RCAP / RT; ++q) { const uint32_t j = tid + q * RT; ck[q] = j < total ? ckey[j] : 0u; }
            uint32_t t = 0;
            for (int b = 30, st = 0; b >= 0; b -= 2, ++st) {
                uint32_t c[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < RCAP / RT; ++q) {
                    c[0] += (uint32_t)__popcll(__ballot(ck[q] >= (t | (1u << b)))); c[1] += (uint32_t)__popcll(__ballot(ck[q] >= (t | (2u << b))));
                    c[2] += (uint32_t)__popcll(__ballot(ck[q] >= (t | (3u << b))));
                }
                if (lane < 3) { const uint32_t mine = lane == 0 ? c[0] : (lane == 1 ? c[1] : c[2]); if (mine) atomicAdd(&step_cnt[st][lane], mine); }
                LDS_BARRIER();
                const uint32_t s1 = step_cnt[st][0], s2 = step_cnt[st][1], s3 = step_cnt[st][2];
                if (s3 >= (uint32_t)k) t |= 3u << b;
                else if (s2 >= (uint32_t)k) t |= 2u << b;
                else if (s1 >= (uint32_t)k) t |= 1u << b;
            }
            if (tid == 0) Tsh = t;
        }
        LDS_BARRIER();
        T = Tsh;
    } else {                                               // (k > 1024, or a row of mostly equal values): search all the keys,
        T = 0;                                             // one bit and one threshold per step
        for (int b = 31; b >= 0; --b) {
            const uint32_t cand = T | (1u << b);
            uint32_t c = 0;
#pragma unroll
            for (int e = 0; e < NPT; ++e) c += key[e] >= cand ? 1u : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            if (lane == 0) red[parity][w] = make_uint2(c, 0u);
            LDS_BARRIER();
            uint32_t sum = 0;
#pragma unroll
            for (int i = 0; i < RW; ++i) sum += red[parity][i].x;
            parity ^= 1;
            if (sum >= (uint32_t)k) T = cand;
        }
    }
    // counts above / at T (the single-wave search already has them)
    uint32_t ngt = 0, neq = 0;
    if (total <= 64u) { ngt = cnt_sh[0]; neq = cnt_sh[1]; }
    else {
        uint32_t cgt = 0, ceq = 0;
#pragma unroll
        for (int e = 0; e < NPT; ++e) { cgt += key[e] > T ? 1u : 0u; ceq += key[e] == T ? 1u : 0u; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cgt += __shfl_xor(cgt, o, 64); ceq += __shfl_xor(ceq, o, 64); }
        if (lane == 0) red[parity][w] = make_uint2(cgt, ceq);
        LDS_BARRIER();
#pragma unroll
        for (int i = 0; i < RW; ++i) { const uint2 r = red[parity][i]; ngt += r.x; neq += r.y; }
        parity ^= 1;
    }
    const uint32_t krem = (uint32_t)k - ngt;
    const bool rank_ties = (neq != krem) && (key_f32(T) > 0.0f);
    uint32_t running = 0;                                   // ties before this group of 4 * RT indices
    float* fr = feat ? feat + row * F : nullptr;
    // long lists (k > 64): the entries get their places in feature order from an ordered block scan per group of 4 * RT
    // indices (index order = group, thread, element) instead of the quadratic rank sort below
    const bool ordered = idx_out && k > 64;
    uint32_t run_out = 0;
#pragma unroll
    for (int j = 0; j < NPT / 4; ++j) {
        const int i0 = (j * RT + tid) * 4;
        // list output only (no dense row to write): a wave whose 256 keys of this group are all below T has nothing to do
        if (!fr && !rank_ties && !ordered &&
            !__any(key[j * 4] >= T || key[j * 4 + 1] >= T || key[j * 4 + 2] >= T || key[j * 4 + 3] >= T)) continue;
        uint32_t before = 0;                                // ties with a lower index inside this group
        if (rank_ties) {                                    // (block-uniform branch) ordered scan: thread order, then c
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) mine += key[j * 4 + c] == T ? 1u : 0u;
            uint32_t sc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(sc, o, 64); if (lane >= o) sc += t; }
            if (lane == 63) red[parity][w] = make_uint2(sc, 0u);
            LDS_BARRIER();
            uint32_t pb = 0, pt = 0;
#pragma unroll
            for (int i = 0; i < RW; ++i) { const uint32_t s2 = red[parity][i].x; if (i < w) pb += s2; pt += s2; }
            parity ^= 1;
            before = running + pb + sc - mine;
            running += pt;
        }
        float o4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t kk = key[j * 4 + c];
            bool sel = kk > T;
            if (kk == T) { sel = rank_ties ? before < krem : true; ++before; }
            const float v = key_f32(kk);
            o4[c] = (sel && v > 0.0f && i0 + c < F) ? v : 0.0f;
            if (idx_out && !ordered && o4[c] > 0.0f) {
                const uint32_t sl = atomicAdd(&cursor, 1u);
                oidx[sl] = i0 + c; oval[sl] = o4[c];
                if (col_cnt) atomicAdd(&col_cnt[(row % L) * F + i0 + c], 1u);
            }
        }
        if (ordered) {                                      // (block-uniform branch)
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) mine += o4[c] > 0.0f ? 1u : 0u;
            uint32_t sc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(sc, o, 64); if (lane >= o) sc += t; }
            if (lane == 63) red[parity][w] = make_uint2(sc, 0u);
            LDS_BARRIER();
            uint32_t pb = 0, pt = 0;
#pragma unroll
            for (int i = 0; i < RW; ++i) { const uint32_t s2 = red[parity][i].x; if (i < w) pb += s2; pt += s2; }
            parity ^= 1;
            uint32_t p = run_out + pb + sc - mine;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (o4[c] > 0.0f) {
                    idx_out[row * k + p] = i0 + c; val_out[row * k + p] = o4[c]; ++p;
                    if (col_cnt) atomicAdd(&col_cnt[(row % L) * F + i0 + c], 1u);
                }
            run_out += pt;
        }
        if (fr && i0 < F) *reinterpret_cast<float4*>(fr + i0) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
    if (ordered) {
        if (row_cnt && tid == 0) row_cnt[row] = run_out;
        for (int e = (int)run_out + tid; e < k; e += RT) { idx_out[row * k + e] = -1; val_out[row * k + e] = 0.0f; }
    } else if (idx_out) {                   // order the <= k positive entries by feature index (rank sort)
        LDS_BARRIER();
        const int n = (int)cursor;
        if (row_cnt && tid == 0) row_cnt[row] = n;
        for (int e = tid; e < k; e += RT) {
            if (e < n) {
                const int me = oidx[e];
                int pos = 0;
                for (int q = 0; q < n; ++q) pos += oidx[q] < me;
                idx_out[row * k + pos] = me; val_out[row * k + pos] = oval[e];
            } else { idx_out[row * k + e] = -1; val_out[row * k + e] = 0.0f; }
        }
    }
    LDS_BARRIER();                                          // the per-row LDS state is reset at the top of the next row
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

// ------------------------------------------------------------------------------------------------
// mode 0, short rows (F <= 6144: the compacted dead columns of the TopK-aux pass, k = 512; a rank's slice of the features
// under sharding; small dictionaries), dense and / or list output: one WAVE per
// row, the row's keys in registers, no LDS and no block barrier.  The exact k-th largest key comes from a bit-by-bit search
// (one threshold per step: per-lane compare-and-count over the registers, one wave reduction); a block that owns a row
// pays ~20 barriers per row for the same search, which at 20 KB per row is all of its time.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
// (LIST: list output compiled in.  The register budget is set for 3-4 waves per SIMD — left to itself the compiler spends 270
// registers on the fully unrolled row and one wave per SIMD then waits out every cross-lane reduction on its own)
template <int J, bool LIST>                            // float4 groups per lane: F <= 256 * J
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(J >= 24 ? 3 : 4)))
void topk_rows_wave_kernel(const float* __restrict__ x, float* __restrict__ feat, int F, int k,
                                                             int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                             const uint8_t* __restrict__ col_mask, int L,
                                                             uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ col_cnt, long nrows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;                          // (whole waves leave; nothing below synchronises the block)
    if (!LIST) idx_out = nullptr;
    const float* xr = x + row * F;
    const uint8_t* mk = col_mask ? col_mask + (row % L) * F : nullptr;
    // (every load of the row is requested before the first is used: a mask test between the loads would put each in its own
    // block behind a full wait, two loads in flight per wave instead of J)
    uint32_t key[J * 4];
    {
        float4 raw[J];
#pragma unroll
        for (int j = 0; j < J; ++j) { const int i0 = (j * 64 + lane) * 4; raw[j] = *reinterpret_cast<const float4*>(xr + (i0 < F ? i0 : 0)); }
        if (mk) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int i0 = (j * 64 + lane) * 4;
                const uchar4 m4 = *reinterpret_cast<const uchar4*>(mk + (i0 < F ? i0 : 0));
                raw[j].x = m4.x ? raw[j].x : -INFINITY; raw[j].y = m4.y ? raw[j].y : -INFINITY;
                raw[j].z = m4.z ? raw[j].z : -INFINITY; raw[j].w = m4.w ? raw[j].w : -INFINITY;
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool in = (j * 64 + lane) * 4 < F;
            key[j * 4 + 0] = in ? f32_key(raw[j].x) : 0u; key[j * 4 + 1] = in ? f32_key(raw[j].y) : 0u;
            key[j * 4 + 2] = in ? f32_key(raw[j].z) : 0u; key[j * 4 + 3] = in ? f32_key(raw[j].w) : 0u;
        }
    }
    // Exact k-th largest key, bit by bit.  Long rows (J >= 8) search the top WP1 bits over the registers, then compact the keys
    // that share that prefix — the only ones the remaining bits can tell apart, a few per cent of the row — into a wave-private
    // LDS list and finish the search on those (`above` = keys beyond the prefix range, known from the last rejected step):
    // 12 x 4J + 20 x ~2 compares per lane instead of 32 x 4J.  Rows whose list would not fit (mostly equal values) carry on
    // over the registers.
    constexpr int WP1 = J >= 8 ? 12 : 32, WR = 32 - WP1, WCAP = 512;
    __shared__ uint32_t wbuf[J >= 8 ? 4 * WCAP : 1];
    uint32_t T = 0, above = 0;
#pragma unroll 1
    for (int b = 31; b >= WR; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t c = 0;
#pragma unroll
        for (int e = 0; e < J * 4; ++e) c += key[e] >= cand ? 1u : 0u;
        c = wave_sum_u32(c);
        if (c >= (uint32_t)k) T = cand; else above = c;
    }
    if (J >= 8) {
        uint32_t* buf = wbuf + (threadIdx.x >> 6) * WCAP;
        const uint32_t pre = T >> WR;
        uint32_t cnt = 0;
#pragma unroll
        for (int e = 0; e < J * 4; ++e) {
            const bool m = (key[e] >> WR) == pre;
            const unsigned long long bal = __ballot(m);
            if (bal == 0ull) continue;
            if (m) {
                const uint32_t pos = cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (pos < (uint32_t)WCAP) buf[pos] = key[e];
            }
            cnt += (uint32_t)__popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();                 // (the list is this wave's own: LDS serves a wave's accesses in order)
        if (cnt <= (uint32_t)WCAP) {
            uint32_t ck[WCAP / 64];
#pragma unroll
            for (int q = 0; q < WCAP / 64; ++q) ck[q] = (uint32_t)(q * 64 + lane) < cnt ? buf[q * 64 + lane] : 0u;
            const int nq = (int)((cnt + 63u) >> 6);
#pragma unroll 1
            for (int b = WR - 1; b >= 0; --b) {
                const uint32_t cand = T | (1u << b);
                uint32_t c = 0;
#pragma unroll
                for (int q = 0; q < WCAP / 64; ++q) if (q < nq) c += ck[q] >= cand ? 1u : 0u;
                if (above + wave_sum_u32(c) >= (uint32_t)k) T = cand;
            }
        } else {
#pragma unroll 1
            for (int b = WR - 1; b >= 0; --b) {
                const uint32_t cand = T | (1u << b);
                uint32_t c = 0;
#pragma unroll
                for (int e = 0; e < J * 4; ++e) c += key[e] >= cand ? 1u : 0u;
                if (wave_sum_u32(c) >= (uint32_t)k) T = cand;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t cgt = 0, ceq = 0;
#pragma unroll
    for (int e = 0; e < J * 4; ++e) { cgt += key[e] > T ? 1u : 0u; ceq += key[e] == T ? 1u : 0u; }
    const uint32_t ngt = wave_sum_u32(cgt), neq = wave_sum_u32(ceq), krem = (uint32_t)k - ngt;
    const bool rank_ties = neq != krem && key_f32(T) > 0.0f;      // more ties than places: the lower indices win
    uint32_t running = 0, run_out = 0;
    float* fr = feat ? feat + row * F : nullptr;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int i0 = (j * 64 + lane) * 4;
        uint32_t before = 0;
        if (rank_ties) {                               // (wave-uniform) index order = group, lane, element
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) mine += key[j * 4 + c] == T ? 1u : 0u;
            uint32_t sc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)sc, o, 64); if (lane >= o) sc += t; }
            before = running + sc - mine;
            running += (uint32_t)__shfl((int)sc, 63, 64);
        }
        float o4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t kk = key[j * 4 + c];
            bool sel = kk > T;
            if (kk == T) { sel = rank_ties ? before < krem : true; ++before; }
            const float v = key_f32(kk);
            o4[c] = (sel && v > 0.0f) ? v : 0.0f;
        }
        if (idx_out) {                                 // list output: places in feature order from an ordered wave scan
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) mine += o4[c] > 0.0f ? 1u : 0u;
            uint32_t sc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)sc, o, 64); if (lane >= o) sc += t; }
            uint32_t p = run_out + sc - mine;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (o4[c] > 0.0f) {
                    idx_out[row * k + p] = i0 + c; val_out[row * k + p] = o4[c]; ++p;
                    if (col_cnt) atomicAdd(&col_cnt[(row % L) * F + i0 + c], 1u);
                }
            run_out += (uint32_t)__shfl((int)sc, 63, 64);
        }
        if (fr && i0 < F) *reinterpret_cast<float4*>(fr + i0) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
    if (idx_out) {
        for (int e = (int)run_out + lane; e < k; e += 64) { idx_out[row * k + e] = -1; val_out[row * k + e] = 0.0f; }
        if (row_cnt && lane == 0) row_cnt[row] = run_out;
    }
}

// ------------------------------------------------------------------------------------------------
// mode 0, long rows, k <= 64, list output: two streaming passes of independent WAVES — no block barrier anywhere, so the
// chip is filled by occupancy instead of one persistent block per CU waiting at a dozen barriers per row.
//   pass A  one wave per 2048-element chunk of a row (32 keys per lane, one coalesced read of the row): the chunk's own top k
//           in the selection's order (key descending, index ascending), positive values only -> k (key, index) slots.
//           Lower bound = k-th largest of the 64 per-lane maxima (ballot search, scalar work only); the ~1.5 k candidates
//           above it go one per lane through a 512-byte wave-private LDS list, and their exact k-th key is a second ballot
//           search.  Chunks with more than 64 candidates (heavy ties) extract their maxima one by one from the registers.
//   pass B  one wave per row: exact top k of the row's chunks * k slots (bit search over <= 16 registers per lane), ties by
//           index, written in ascending feature order as the compact row the list construction reads.
// The row's top k is contained in the union of its chunks' top k under the same total order.
// ------------------------------------------------------------------------------------------------
constexpr int CHK = 2048;
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// k-th largest of the 64 lane values, searched on the top 16 key bits only (lower bits zero): a LOWER bound of it — at
// least m lanes hold values >= the result — at half the steps of the exact search
__device__ __forceinline__ uint32_t wave_kth_hi16(uint32_t v, uint32_t m) {
    uint32_t t = 0;
    for (int b = 30; b >= 16; b -= 2) {
        const uint32_t c1 = (uint32_t)__popcll(__ballot(v >= (t | (1u << b))));
        const uint32_t c2 = (uint32_t)__popcll(__ballot(v >= (t | (2u << b))));
        const uint32_t c3 = (uint32_t)__popcll(__ballot(v >= (t | (3u << b))));
        if (c3 >= m) t |= 3u << b;
        else if (c2 >= m) t |= 2u << b;
        else if (c1 >= m) t |= 1u << b;
    }
    return t;
}

__global__ __launch_bounds__(256) void topk_chunk_kernel(const float* __restrict__ x, int F, int k, const uint8_t* __restrict__ col_mask,
                                                         int L, long nrows, int chunks, uint32_t* __restrict__ okey,
                                                         int32_t* __restrict__ oidx) {
    __shared__ float lval[4][64];
    __shared__ int32_t lidx[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wv;
    if (gw >= nrows * chunks) return;                     // (whole waves leave; nothing below synchronises the block)
    const long row = gw / chunks;
    const int base = (int)(gw - row * chunks) * CHK;
    const float* xr = x + row * F;
    const uint8_t* mk = col_mask ? col_mask + (row % L) * F : nullptr;
    // (the values stay floats: only positive ones can be emitted, and among those float order = key order)
    float a[32];
    float tmax = -INFINITY;
    {   // (all eight loads requested before the first is used; the mask, when there is one, in a second sweep)
        float4 raw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int i0 = base + (j * 64 + lane) * 4; raw[j] = *reinterpret_cast<const float4*>(xr + (i0 < F ? i0 : base)); }
        if (mk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i0 = base + (j * 64 + lane) * 4;
                const uchar4 m4 = *reinterpret_cast<const uchar4*>(mk + (i0 < F ? i0 : base));
                raw[j].x = m4.x ? raw[j].x : -INFINITY; raw[j].y = m4.y ? raw[j].y : -INFINITY;
                raw[j].z = m4.z ? raw[j].z : -INFINITY; raw[j].w = m4.w ? raw[j].w : -INFINITY;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = base + (j * 64 + lane) * 4 < F;
            const float q[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float vv = in ? q[c] : -INFINITY; a[j * 4 + c] = vv; tmax = fmaxf(tmax, vv); }
        }
    }
    constexpr uint32_t POS = 0x80000001u;                 // smallest key of a value > 0: nothing else is ever emitted
    // at least k values of the chunk (k distinct lanes' maxima; k <= 64) are >= LB
    const float LB = key_f32(max(wave_kth_hi16(f32_key(tmax), (uint32_t)k), POS));
    uint32_t cnt = 0;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const bool c = a[e] >= LB;
        const unsigned long long bal = __ballot(c);
        if (bal == 0ull) continue;
        if (c) {
            const uint32_t pos = cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (pos < 64u) { lval[wv][pos] = a[e]; lidx[wv][pos] = base + ((e >> 2) * 64 + lane) * 4 + (e & 3); }
        }
        cnt += (uint32_t)__popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();                      // (the list is this wave's own: LDS serves a wave's accesses in order)
    uint32_t* ok = okey + gw * k;
    int32_t* oi = oidx + gw * k;
    if (cnt <= 64u) {
        // one candidate per lane: each ranks itself in the selection's order (value descending, index ascending) against the
        // others, read lane by lane through scalar registers; the first k of that order are the chunk's selection
        const bool have = (uint32_t)lane < cnt;
        const float cv = have ? lval[wv][lane] : -INFINITY;
        const int ci = have ? lidx[wv][lane] : 0x7fffffff;
        uint32_t rank = 0;
        if (cnt > (uint32_t)k) {
            for (uint32_t j = 0; j < cnt; ++j) {
                const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), (int)j));
                const int ij = __builtin_amdgcn_readlane(ci, (int)j);
                rank += (vj > cv || (vj == cv && ij < ci)) ? 1u : 0u;
            }
        }
        const bool sel = have && rank < (uint32_t)k;
        const unsigned long long sb = __ballot(sel);
        const int n = __popcll(sb);
        if (sel) { const int pos = __popcll(sb & ((1ull << lane) - 1ull)); ok[pos] = f32_key(cv); oi[pos] = ci; }
        if (lane >= n && lane < k) { ok[lane] = 0u; oi[lane] = -1; }
    } else {
        int s = 0;
        for (; s < k; ++s) {                               // (wave-uniform loop) the next entry in the selection's order
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 32; ++e) m = fmaxf(m, a[e]);
            m = key_f32(wave_max_u32(f32_key(m)));
            if (!(m > 0.0f)) break;
            int mi = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < 32; ++e) if (a[e] == m) mi = min(mi, base + ((e >> 2) * 64 + lane) * 4 + (e & 3));
            mi = wave_min_i32(mi);
            if (lane == 0) { ok[s] = f32_key(m); oi[s] = mi; }
#pragma unroll
            for (int e = 0; e < 32; ++e) if (a[e] == m && base + ((e >> 2) * 64 + lane) * 4 + (e & 3) == mi) a[e] = -INFINITY;
        }
        if (lane >= s && lane < k) { ok[lane] = 0u; oi[lane] = -1; }
    }
}

template <int NQ>                                      // slots per lane: chunks * k <= 64 * NQ
__global__ __launch_bounds__(256) void topk_merge_kernel(const uint32_t* __restrict__ ckey, const int32_t* __restrict__ cidx, int n, int k,
                                                         int F, int L, long nrows, int32_t* __restrict__ idx_out,
                                                         float* __restrict__ val_out, uint32_t* __restrict__ row_cnt,
                                                         uint32_t* __restrict__ col_cnt, const uint32_t* __restrict__ slot_cnt = nullptr,
                                                         const int32_t* __restrict__ gate = nullptr) {
    // slot_cnt (the fused encoder + per-row selection): only the first min(slot_cnt[row], n) slots of a row are filled, in
    // arrival order (the rest of the block is stale); gate != 0: the attempt is void, every row stays empty
    __shared__ int32_t sidx[4][64];
    __shared__ uint32_t skey[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wv;
    if (row >= nrows) return;
    if (gate && *gate) {
        if (row_cnt && lane == 0) row_cnt[row] = 0;
        return;
    }
    const int nr = slot_cnt ? (int)min(slot_cnt[row], (uint32_t)n) : n;
    uint32_t kq[NQ];
    int iq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = q * 64 + lane;
        kq[q] = p < nr ? ckey[row * n + p] : 0u;
        iq[q] = p < nr ? cidx[row * n + p] : 0x7fffffff;
    }
    uint32_t T = 0;                                       // exact k-th largest slot key (0: fewer than k slots are filled)
#pragma unroll 1
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) c += kq[q] >= cand ? 1u : 0u;
        if (wave_sum_u32(c) >= (uint32_t)k) T = cand;
    }
    uint32_t cgt = 0, ceq = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { cgt += kq[q] > T ? 1u : 0u; ceq += (kq[q] == T && kq[q] != 0u) ? 1u : 0u; }
    const uint32_t ngt = wave_sum_u32(cgt), neq = wave_sum_u32(ceq), krem = (uint32_t)k - ngt;
    uint32_t rank[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rank[q] = 0;
    if (T != 0u && neq > krem) {                          // (wave-uniform) more ties than places: the lower indices win
#pragma unroll
        for (int q2 = 0; q2 < NQ; ++q2) {
            unsigned long long tb = __ballot(kq[q2] == T);
            while (tb) {
                const int l = __builtin_ctzll(tb);
                tb &= tb - 1ull;
                const int o = __shfl(iq[q2], l, 64);
#pragma unroll
                for (int q = 0; q < NQ; ++q) rank[q] += (kq[q] == T && o < iq[q]) ? 1u : 0u;
            }
        }
    }
    uint32_t nsel = 0;                                    // selected slots (<= k) into the wave's list, arrival order
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const bool sel = kq[q] != 0u && (kq[q] > T || (kq[q] == T && rank[q] < krem)) && key_f32(kq[q]) > 0.0f;
        const unsigned long long sb = __ballot(sel);
        if (sb == 0ull) continue;
        if (sel) { const uint32_t pos = nsel + (uint32_t)__popcll(sb & ((1ull << lane) - 1ull)); sidx[wv][pos] = iq[q]; skey[wv][pos] = kq[q]; }
        nsel += (uint32_t)__popcll(sb);
    }
    __builtin_amdgcn_wave_barrier();
    const bool have = (uint32_t)lane < nsel;
    const int me = have ? sidx[wv][lane] : 0x7fffffff;
    const uint32_t mk = have ? skey[wv][lane] : 0u;
    int pos = 0;                                          // ascending feature index (rank among the <= k selected)
    for (uint32_t j = 0; j < nsel; ++j) pos += __shfl(me, (int)j, 64) < me ? 1 : 0;
    if (have) {
        idx_out[row * k + pos] = me; val_out[row * k + pos] = key_f32(mk);
        if (col_cnt) atomicAdd(&col_cnt[(row % L) * F + me], 1u);
    } else if (lane < k) { idx_out[row * k + lane] = -1; val_out[row * k + lane] = 0.0f; }
    if (row_cnt && lane == 0) row_cnt[row] = nsel;
}

size_t clt_internal_topk_rows_workspace_bytes(long rows, int F, int k) {
    if (k < 1 || k > 64 || F <= CHK || F % 4) return 0;
    const long chunks = (F + CHK - 1) / CHK;
    if (chunks * k > 1024) return 0;
    return (size_t)rows * chunks * k * 8;
}

int clt_internal_topk_rows(const float* x, float* feat, int rows, int F, int k, int32_t* idx_out, float* val_out,
                           const uint8_t* col_mask, int L, uint32_t* row_cnt, uint32_t* col_cnt, hipStream_t st, void* ws, size_t wsb) {
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
    // list output of long rows with a workspace at hand: the two streaming passes
    const size_t need = clt_internal_topk_rows_workspace_bytes(rows, F, k);
    if (!feat && idx_out && ws && need && wsb >= need && aligned16(x) && (!col_mask || (reinterpret_cast<uintptr_t>(col_mask) & 3) == 0) &&
        !diag_env("CLT_TOPK_NO_CHUNKS")) {
        const int chunks = (F + CHK - 1) / CHK, n = chunks * k;
        const long waves = (long)rows * chunks;
        if ((waves + 3) / 4 <= 0x7fffffffL) {
            uint32_t* ck = (uint32_t*)ws;
            int32_t* ci = (int32_t*)(ck + (size_t)waves * k);
            hipLaunchKernelGGL(topk_chunk_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, F, k, col_mask, L, (long)rows, chunks, ck, ci);
            const dim3 mg((unsigned)((rows + 3) / 4));
#define TKM(NQ_) hipLaunchKernelGGL((topk_merge_kernel<NQ_>), mg, dim3(256), 0, st, (const uint32_t*)ck, (const int32_t*)ci, n, k, F, L, (long)rows, \
                                    idx_out, val_out, row_cnt, col_cnt)
            if (n <= 256) TKM(4); else if (n <= 512) TKM(8); else TKM(16);
#undef TKM
            CLT_LAUNCH_CHECK(); return CLT_OK;
        }
    }
    const bool reg_ok = F % 4 == 0 && F <= 32 * RT && (!idx_out || k <= 1024) && aligned16(x) && (!feat || aligned16(feat)) &&
                        (!col_mask || (reinterpret_cast<uintptr_t>(col_mask) & 3) == 0) && !diag_env("CLT_TOPK_ROWS_V1");
    // (measured, 49152 rows: F = 5632 k = 512: 2.07 -> 1.30 ms; F = 2048 k = 16: 4.40 -> 0.31 ms; at F = 8192 the 128 keys per
    // lane leave one wave per SIMD and the block kernel is faster)
    // (list output, F = 4096 k = 16: the block kernel's rank-counting tail is faster, 1.0 vs 1.17 ms; at F <= 2048 its fixed cost
    // per row dominates: 4.5 vs 0.36 ms)
    if (reg_ok && (F <= 2048 || (!idx_out && F <= 6144)) && !diag_env("CLT_TOPK_NO_WAVE")) {
        const dim3 grid((unsigned)((rows + 3) / 4));
#define TKW(J_) do { if (idx_out) hipLaunchKernelGGL((topk_rows_wave_kernel<J_, true>), grid, dim3(256), 0, st, x, feat, F, k, idx_out, val_out, col_mask, L, \
                                                     row_cnt, col_cnt, (long)rows); \
                     else hipLaunchKernelGGL((topk_rows_wave_kernel<J_, false>), grid, dim3(256), 0, st, x, feat, F, k, idx_out, val_out, col_mask, L, \
                                             row_cnt, col_cnt, (long)rows); } while (0)
        const int groups = (F + 255) / 256;
        // (list output only reaches here with F <= 2048: the longer instantiations are dense-only)
#define TKWD(J_) hipLaunchKernelGGL((topk_rows_wave_kernel<J_, false>), grid, dim3(256), 0, st, x, feat, F, k, idx_out, val_out, col_mask, L, \
                                    row_cnt, col_cnt, (long)rows)
        if (groups <= 4) TKW(4); else if (groups <= 8) TKW(8); else if (idx_out) return CLT_ERR_RANGE; else if (groups <= 16) TKWD(16); else TKWD(24);
#undef TKWD
#undef TKW
        CLT_LAUNCH_CHECK(); return CLT_OK;
    }
    if (reg_ok) {
        static int cus = 0;
        if (!cus) {
            int dev = 0; hipDeviceProp_t pr;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        }
        const int grid = rows < cus ? rows : cus;
        const size_t dyn = (size_t)((F + 4095) / 4096) * 4096 * 4;           // whole 16 KB DMA rounds
#define TKR(NPT_) do { if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_reg_kernel<NPT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return CLT_ERR_LAUNCH; \
        hipLaunchKernelGGL((topk_rows_reg_kernel<NPT_>), dim3(grid), dim3(RT), dyn, st, x, feat, F, k, idx_out, val_out, col_mask, L, row_cnt, col_cnt, (long)rows); } while (0)
        const int per = (F + RT - 1) / RT;
        if (per <= 4) TKR(4); else if (per <= 8) TKR(8); else if (per <= 12) TKR(12); else if (per <= 16) TKR(16);
        else if (per <= 20) TKR(20); else if (per <= 24) TKR(24); else TKR(32);
#undef TKR
    } else
        hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(TPB), 0, st, x, feat, F, k, idx_out, val_out, col_mask, L, row_cnt,
                           col_cnt);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

// Lower bound of every row's j-th largest value (the fused encoder + per-row selection's sample pass, select_lists.hip): the
// chunk pass alone — a chunk's j largest positive values; the bound of a row = the largest, over its chunks, of a chunk's j-th
// largest (that chunk alone holds j values >= it; one chunk: the row's exact j-th largest).  A row none of whose chunks holds j
// positive values gets the smallest positive float: all its positive elements become candidates.
__global__ void chunk_bound_kernel(const uint32_t* __restrict__ okey, long rows, int chunks, int j, float* __restrict__ lb) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint32_t best = 1u;                                   // (bit pattern of the smallest positive float)
    for (int c = 0; c < chunks; ++c) {
        const uint32_t* k = okey + (r * chunks + c) * j;
        uint32_t m = 0xffffffffu;
        for (int e = 0; e < j; ++e) m = min(m, k[e]);     // (unfilled places hold key 0)
        if (m > 0x80000000u) best = max(best, m & 0x7fffffffu);      // key of a positive value = its bits | sign bit
    }
    lb[r] = __uint_as_float(best);
}

size_t clt_internal_row_bounds_workspace_bytes(long rows, int F, int j) {
    if (j < 1 || j > 64 || F % 4) return 0;
    const long chunks = (F + CHK - 1) / CHK;
    return (size_t)rows * chunks * j * 8;
}
// returns 1 when the streaming form does not apply (the caller then takes the general per-row selection)
int clt_internal_row_bounds(const float* x, long rows, int F, int j, float* lb, void* ws, size_t wsb, hipStream_t st) {
    const size_t need = clt_internal_row_bounds_workspace_bytes(rows, F, j);
    if (!need || !ws || wsb < need || !aligned16(x)) return 1;
    const int chunks = (F + CHK - 1) / CHK;
    const long waves = rows * chunks;
    if ((waves + 3) / 4 > 0x7fffffffL) return 1;
    uint32_t* ck = (uint32_t*)ws;
    int32_t* ci = (int32_t*)(ck + (size_t)waves * j);
    hipLaunchKernelGGL(topk_chunk_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, F, j, (const uint8_t*)nullptr, 1, rows, chunks, ck, ci);
    hipLaunchKernelGGL(chunk_bound_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (const uint32_t*)ck, rows, chunks, j, lb);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

// Exact per-row top k (ties: lowest index) among each row's slot block — the second half of the streaming rows selection on its
// own, for candidates that the encoder GEMM's row-select epilogue collected (select_lists.hip: clt_select_rows_ext_*).
int clt_internal_topk_merge_slots(const uint32_t* slot_key, const int32_t* slot_idx, const uint32_t* slot_cnt, int slots, int k, int F,
                                  int L, long rows, int32_t* idx_out, float* val_out, uint32_t* row_cnt, uint32_t* col_cnt,
                                  const int32_t* gate, hipStream_t st) {
    if (slots < 1 || slots > 1024 || k < 1 || k > 64 || rows <= 0) return CLT_ERR_RANGE;
    const dim3 mg((unsigned)((rows + 3) / 4));
#define TKM(NQ_) hipLaunchKernelGGL((topk_merge_kernel<NQ_>), mg, dim3(256), 0, st, slot_key, slot_idx, slots, k, F, L, rows, idx_out, val_out, \
                                    row_cnt, col_cnt, slot_cnt, gate)
    if (slots <= 256) TKM(4); else if (slots <= 512) TKM(8); else TKM(16);
#undef TKM
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
    return clt_internal_topk_rows(x, feat, rows, F, k, idx_out, val_out, nullptr, 1, nullptr, nullptr, as_stream(st), nullptr, 0);
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
