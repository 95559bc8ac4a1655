// Device helpers shared by the selection kernels (select.hip, select_lists.hip): block scans and the
// MSB-first radix-select bin search on order-preserving fp32 keys (11 + 11 + 10 bits).
#pragma once
#include "common.h"

namespace {
constexpr int TPB = 256, NBIN = 2048;
constexpr int SHIFT[3] = {21, 10, 0};
constexpr int BITS[3] = {11, 11, 10};

struct SegState { uint32_t prefix, mask, krem, eq_total; };

// Exclusive prefix sum of v over the 256 threads of the block (thread order); *total = block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sh /*>=5*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();                       // protect sh from the previous call's readers
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint32_t s = sh[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}

// Given hist[NBIN] in LDS (only the low `nb` bins used) find, scanning from the TOP bin, the bin that
// holds the krem-th largest matching element.  Results broadcast through res[0..2] (LDS).
__device__ __forceinline__ void find_bin(const uint32_t* hist, int nb, uint32_t krem, uint32_t* sh, uint32_t* res) {
    const int per = nb / TPB;                       // 8 or 4 bins per thread
    const int chunk = TPB - 1 - threadIdx.x;        // thread 0 owns the top bins
    uint32_t mine = 0;
    for (int b = 0; b < per; ++b) mine += hist[chunk * per + b];
    uint32_t tot;
    uint32_t cum = block_excl_scan(mine, sh, &tot); // elements in strictly higher chunks
    if (cum < krem && krem <= cum + mine) {
        for (int b = per - 1; b >= 0; --b) {
            const uint32_t h = hist[chunk * per + b];
            if (krem <= cum + h) { res[0] = chunk * per + b; res[1] = krem - cum; res[2] = h; break; }
            cum += h;
        }
    }
    __syncthreads();
}


__device__ __forceinline__ bool need_rank(const SegState& st) { return st.eq_total != st.krem && key_f32(st.prefix) > 0.0f; }

// one block per segment: consume the global histogram of radix pass p, narrow the segment's prefix
__global__ __launch_bounds__(TPB) void seg_find_kernel(SegState* __restrict__ state, uint32_t* __restrict__ ghist, int p,
                                                       const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ uint32_t hist[NBIN];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t res[4];
    const int s = blockIdx.x, nb = 1 << BITS[p];
    for (int i = threadIdx.x; i < NBIN; i += TPB) { hist[i] = ghist[(long)s * NBIN + i]; ghist[(long)s * NBIN + i] = 0; }
    __syncthreads();
    find_bin(hist, nb, state[s].krem, sh, res);
    if (threadIdx.x == 0) {
        SegState st = state[s];
        st.prefix |= res[0] << SHIFT[p];
        st.mask |= (uint32_t)(nb - 1) << SHIFT[p];
        st.krem = res[1];
        st.eq_total = res[2];
        state[s] = st;
    }
}

}  // namespace
