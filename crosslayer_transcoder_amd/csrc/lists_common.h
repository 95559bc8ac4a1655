// List-construction kernels shared by sparse.hip (lists from a dense feature tensor) and select_lists.hip
// (lists straight from the selection).  Every kernel takes a trailing `gate`: NULL = always run, otherwise the
// kernel runs only when (*gate != 0) == gate_on (device-side choice between the fast path and the exact fallback).
#pragma once
#include "common.h"

namespace {
constexpr int LTPB = 256;
// ------------------------------------------------------------------------------------------------
// list construction
// ------------------------------------------------------------------------------------------------
// An entry of the lists is a position with feat > 0 — and, when `theta` ([L*F], JumpReLU) is given, feat > theta as
// well: the lists of JumpReLU(feat) straight from the pre-activations (jumprelu.py:33-37: x * [(x > theta) & (x > 0)]).
__device__ __forceinline__ bool list_entry(float x, const float* __restrict__ th, int col) { return x > 0.0f && (!th || x > th[col]); }

// Element type of the dense tensor the lists are built from: XT = 0 fp32, 1 bf16, 2 fp16 (the dense 16-bit regime's
// pre-activations, clt_gemm_bf16_nt_out16); entries are stored widened to fp32 either way.
template <int XT> __device__ __forceinline__ float4 rows_ld4(const void* __restrict__ xr, int i) {
    if (XT == 0) return *reinterpret_cast<const float4*>(static_cast<const float*>(xr) + i);
    const ushort4 h = *reinterpret_cast<const ushort4*>(static_cast<const uint16_t*>(xr) + i);
    return make_float4(cvt16<XT == 2>(h.x), cvt16<XT == 2>(h.y), cvt16<XT == 2>(h.z), cvt16<XT == 2>(h.w));
}
template <int XT> __device__ __forceinline__ float rows_ld1(const void* __restrict__ xr, int i) {
    if (XT == 0) return static_cast<const float*>(xr)[i];
    return cvt16<XT == 2>(static_cast<const uint16_t*>(xr)[i]);
}
// Segment of the row that wave w of the block owns, for both list-build kernels: a quarter of the row, whole 16-byte groups of
// 64 lanes in the vector form.  wave_cnt (optional, [rows][4]): the per-wave counts, so that fill_rows_kernel need not count again.
__device__ __forceinline__ bool rows_vec_form(const void* xr, const float* th, int F, int xbytes) {
    return F % 4 == 0 && (reinterpret_cast<uintptr_t>(xr) & (uintptr_t)(4 * xbytes - 1)) == 0 && (reinterpret_cast<uintptr_t>(th) & 15) == 0;
}
__device__ __forceinline__ int rows_seg(int F, bool vec) { return vec ? (F / 4 + 255) / 256 * 256 : ((F + 3) / 4 + 63) / 64 * 64; }

template <int XT = 0>
__global__ __launch_bounds__(LTPB) void count_rows_kernel(const void* __restrict__ feat, int F, int L, const float* __restrict__ theta,
                                                         uint32_t* __restrict__ row_cnt, const int32_t* __restrict__ gate, int gate_on,
                                                         uint32_t* __restrict__ wave_cnt = nullptr) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ uint32_t sh[4];
    constexpr int XB = XT == 0 ? 4 : 2;
    const void* xr = static_cast<const char*>(feat) + (long)blockIdx.x * F * XB;
    const float* th = theta ? theta + (long)(blockIdx.x % L) * F : nullptr;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool vec = rows_vec_form(xr, th, F, XB);
    const int seg = rows_seg(F, vec);
    const int s0 = min(F, w * seg), s1 = min(F, s0 + seg);
    uint32_t c = 0;
    if (vec) {                                             // 16-byte loads: the pass is a plain HBM stream
        for (int i = s0 + lane * 4; i < s1; i += 256) {
            const float4 x = rows_ld4<XT>(xr, i);
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (th) t = *reinterpret_cast<const float4*>(th + i);
            c += (x.x > 0.0f && x.x > t.x) + (x.y > 0.0f && x.y > t.y) + (x.z > 0.0f && x.z > t.z) + (x.w > 0.0f && x.w > t.w);
        }
    } else {
        for (int i = s0 + lane; i < s1; i += 64) c += list_entry(rows_ld1<XT>(xr, i), th, i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) { sh[w] = c; if (wave_cnt) wave_cnt[(long)blockIdx.x * 4 + w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) row_cnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// Lists in the vector form, ROWS_PER_BLOCK rows of ONE layer per block (tokens b0 .. b0 + 3): four row loads in flight per
// lane, and the layer's thresholds (TH) are read once for four rows.  With one row per block the fp32 thresholds re-read from L2 for every row are twice the bytes
// of a 16-bit row: count + fill took 0.76 + 0.79 ms for 3.2 GB of pre-activations against 0.55 + 0.55 without thresholds
// (scripts/archive/build_probe.py).  Same per-wave segments, counts and entry order as count_rows_kernel / fill_rows_kernel.
constexpr int ROWS_PER_BLOCK = 4;
__device__ __forceinline__ uint32_t rows_hits(const float4 x, const float4 t) {
    return (uint32_t)(x.x > 0.0f && x.x > t.x) + (uint32_t)(x.y > 0.0f && x.y > t.y) + (uint32_t)(x.z > 0.0f && x.z > t.z) +
           (uint32_t)(x.w > 0.0f && x.w > t.w);
}
// bias ([L*F], optional): added to every element in fp32 before the comparison and in the stored value — the 16-bit
// pre-activation tensor of the 16-bit modes holds the rounded encoder product alone (autocast order, clt.py:157-165).
__device__ __forceinline__ float4 rows_add4(const float4 x, const float4 b) { return make_float4(x.x + b.x, x.y + b.y, x.z + b.z, x.w + b.w); }
template <int XT, bool TH>
__global__ __launch_bounds__(LTPB) void count_rows_th_kernel(const void* __restrict__ feat, int B, int F, int L, const float* __restrict__ theta,
                                                            const float* __restrict__ bias, uint32_t* __restrict__ row_cnt, const int32_t* __restrict__ gate, int gate_on,
                                                            uint32_t* __restrict__ wave_cnt) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    constexpr int R = ROWS_PER_BLOCK, XB = XT == 0 ? 4 : 2;
    __shared__ uint32_t sh[R][4];
    const int l = blockIdx.x % L, b0 = (blockIdx.x / L) * R;
    const int nb = min(R, B - b0);                         // >= 1 by the grid
    const float* th = TH ? theta + (long)l * F : nullptr;
    const float* bi = bias ? bias + (long)l * F : nullptr;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int seg = rows_seg(F, true);
    const int s0 = min(F, w * seg), s1 = min(F, s0 + seg);
    const char* xr[R];
#pragma unroll
    for (int j = 0; j < R; ++j) xr[j] = static_cast<const char*>(feat) + ((long)(b0 + (j < nb ? j : 0)) * L + l) * F * XB;   // (rows past B: row b0 again, not counted)
    uint32_t c[R];
#pragma unroll
    for (int j = 0; j < R; ++j) c[j] = 0;
    for (int i = s0 + lane * 4; i < s1; i += 256) {
        float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (TH) t = *reinterpret_cast<const float4*>(th + i);
        float4 x[R];
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = rows_ld4<XT>(xr[j], i);
        if (bi) {
            const float4 bv = *reinterpret_cast<const float4*>(bi + i);
#pragma unroll
            for (int j = 0; j < R; ++j) x[j] = rows_add4(x[j], bv);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) c[j] += rows_hits(x[j], t);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c[j] += __shfl_xor(c[j], o, 64);
        if (lane == 0 && j < nb) { sh[j][w] = c[j]; wave_cnt[((long)(b0 + j) * L + l) * 4 + w] = c[j]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) row_cnt[(long)(b0 + threadIdx.x) * L + l] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

template <int XT, bool TH>
__global__ __launch_bounds__(LTPB) void fill_rows_th_kernel(const void* __restrict__ feat, int B, int F, int L, const float* __restrict__ theta,
                                                           const float* __restrict__ bias, const uint32_t* __restrict__ row_start, int32_t* __restrict__ ent_col,
                                                           float* __restrict__ ent_val, uint32_t cap, int32_t* __restrict__ overflow,
                                                           const int32_t* __restrict__ gate, int gate_on, const uint32_t* __restrict__ wave_cnt) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    constexpr int R = ROWS_PER_BLOCK, XB = XT == 0 ? 4 : 2;
    const int l = blockIdx.x % L, b0 = (blockIdx.x / L) * R;
    const int nb = min(R, B - b0);
    const float* th = TH ? theta + (long)l * F : nullptr;
    const float* bi = bias ? bias + (long)l * F : nullptr;
    const int colbase = l * F;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int seg = rows_seg(F, true);
    const int s0 = min(F, w * seg), s1 = min(F, s0 + seg);
    const char* xr[R];
    uint32_t off[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long row = (long)(b0 + (j < nb ? j : 0)) * L + l;
        xr[j] = static_cast<const char*>(feat) + row * F * XB;
        off[j] = row_start[row];
        for (int k = 0; k < w; ++k) off[j] += wave_cnt[row * 4 + k];
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = s0; base < s1; base += 256) {
        const int i = base + lane * 4;
        const bool in = i < s1;
        const int ii = in ? i : 0;                         // (column 0 is always readable; F % 4 == 0, so i + 3 < F when in)
        float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (TH) t = *reinterpret_cast<const float4*>(th + ii);
        float4 x[R];
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = rows_ld4<XT>(xr[j], ii);
        if (bi) {
            const float4 bv = *reinterpret_cast<const float4*>(bi + ii);
#pragma unroll
            for (int j = 0; j < R; ++j) x[j] = rows_add4(x[j], bv);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j >= nb) continue;                         // (block-uniform)
            const bool p[4] = {in && x[j].x > 0.0f && x[j].x > t.x, in && x[j].y > 0.0f && x[j].y > t.y,
                               in && x[j].z > 0.0f && x[j].z > t.z, in && x[j].w > 0.0f && x[j].w > t.w};
            const unsigned long long b0m = __ballot(p[0]), b1m = __ballot(p[1]), b2m = __ballot(p[2]), b3m = __ballot(p[3]);
            if ((b0m | b1m | b2m | b3m) == 0ull) continue;
            uint32_t pos = off[j] + (uint32_t)(__popcll(b0m & lt) + __popcll(b1m & lt) + __popcll(b2m & lt) + __popcll(b3m & lt));
            const float xv[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p[q]) {
                    if (pos < cap) { ent_col[pos] = colbase + i + q; ent_val[pos] = xv[q]; }
                    else *overflow = 1;
                    ++pos;
                }
            off[j] += (uint32_t)(__popcll(b0m) + __popcll(b1m) + __popcll(b2m) + __popcll(b3m));
        }
    }
}

// exclusive scan of cnt[0..n) into start[0..n], start[n] = total; single block of 1024 threads.  Per iteration the
// block takes 16384 items as four 4096-item sub-chunks (thread t: 4 consecutive items of each, one 16-byte load when
// aligned, so a wave reads 1 KB contiguous per load); the next iteration's loads are issued before this iteration's
// barrier, so the single barrier per 16384 items (per-wave totals double buffered, running carry in a register of every
// thread) never waits on memory.
__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t* __restrict__ cnt, int n, uint32_t* __restrict__ start,
                                                     uint32_t* __restrict__ total, uint8_t* __restrict__ fired, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ uint32_t sh[2][4][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool vec = (reinterpret_cast<uintptr_t>(cnt) & 15) == 0;
    auto load = [&](long base, uint32_t (*v)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long i0 = base + s * 4096 + (long)threadIdx.x * 4;
            v[s][0] = v[s][1] = v[s][2] = v[s][3] = 0u;
            if (vec && i0 + 3 < n) { const uint4 t = *reinterpret_cast<const uint4*>(cnt + i0); v[s][0] = t.x; v[s][1] = t.y; v[s][2] = t.z; v[s][3] = t.w; }
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (i0 + j < n) v[s][j] = cnt[i0 + j];
            }
        }
    };
    uint32_t carry = 0;
    int par = 0;
    uint32_t nxt[4][4];
    load(0, nxt);
    for (long base = 0; base < n; base += 16384, par ^= 1) {
        uint32_t v[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[s][j] = nxt[s][j];
        if (base + 16384 < n) load(base + 16384, nxt);
        uint32_t sum[4], inc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sum[s] = v[s][0] + v[s][1] + v[s][2] + v[s][3];
            inc[s] = sum[s];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc[s], o, 64); if (lane >= o) inc[s] += t; }
            if (lane == 63) sh[par][s][w] = inc[s];
        }
        __syncthreads();
        uint32_t sub = carry;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint32_t wbase = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const uint32_t x = sh[par][s][k]; if (k < w) wbase += x; tot += x; }
            uint32_t run = sub + wbase + inc[s] - sum[s];
            const long i0 = base + s * 4096 + (long)threadIdx.x * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < n) { start[i0 + j] = run; if (fired) fired[i0 + j] = v[s][j] > 0; run += v[s][j]; }
            sub += tot;
        }
        carry = sub;
    }
    if (threadIdx.x == 0) { start[n] = carry; if (total) *total = carry; }
}

// The same scan over many workgroups, for long arrays (one CU alone streams only ~20 GB/s): chunk totals first, then
// every chunk scans itself behind the sum of the totals before it.  `scratch`: ceil(n / 4096) words.
__global__ __launch_bounds__(1024) void scan_totals_kernel(const uint32_t* __restrict__ cnt, int n, uint32_t* __restrict__ scratch,
                                                            const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ uint32_t sh[16];
    const long i0 = (long)blockIdx.x * 4096 + (long)threadIdx.x * 4;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i0 + j < n) s += cnt[i0 + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < 16; ++k) t += sh[k]; scratch[blockIdx.x] = t; }
}

__global__ __launch_bounds__(1024) void scan_chunks_kernel(const uint32_t* __restrict__ cnt, int n, const uint32_t* __restrict__ scratch,
                                                            uint32_t* __restrict__ start, uint32_t* __restrict__ total,
                                                            uint8_t* __restrict__ fired, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ uint32_t sh[16], shb[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t before = 0;                                  // sum of the totals of the chunks before this one
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += 1024) before += scratch[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    const long i0 = (long)blockIdx.x * 4096 + (long)threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + j < n ? cnt[i0 + j] : 0u;
    const uint32_t sum = v[0] + v[1] + v[2] + v[3];
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) sh[w] = inc;
    if (lane == 0) shb[w] = before;
    __syncthreads();
    uint32_t wbase = 0, tot = 0, carry = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t x = sh[k]; if (k < w) wbase += x; tot += x; carry += shb[k]; }
    uint32_t run = carry + wbase + inc - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < n) { start[i0 + j] = run; if (fired) fired[i0 + j] = v[j] > 0; run += v[j]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { start[n] = carry + tot; if (total) *total = carry + tot; }
}

// scratch == nullptr or a short array: the single-workgroup kernel
static inline void launch_scan(const uint32_t* cnt, int n, uint32_t* start, uint32_t* total, uint8_t* fired, const int32_t* gate,
                               int gate_on, uint32_t* scratch, hipStream_t st) {
    if (!scratch || n <= 8192) {              // (49152 items: one workgroup 21 us, twelve chunks 5 + 5 us)
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, cnt, n, start, total, fired, gate, gate_on);
        return;
    }
    const int chunks = (n + 4095) / 4096;
    hipLaunchKernelGGL(scan_totals_kernel, dim3(chunks), dim3(1024), 0, st, cnt, n, scratch, gate, gate_on);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(chunks), dim3(1024), 0, st, cnt, n, (const uint32_t*)scratch, start, total, fired, gate, gate_on);
}

// ordered compaction of the positive entries of each row; wave w owns a contiguous quarter of the row
template <int XT = 0>
__global__ __launch_bounds__(LTPB) void fill_rows_kernel(const void* __restrict__ feat, int F, int L, const float* __restrict__ theta,
                                                        const uint32_t* __restrict__ row_start, int32_t* __restrict__ ent_col,
                                                        float* __restrict__ ent_val, uint32_t cap, int32_t* __restrict__ overflow, const int32_t* __restrict__ gate, int gate_on,
                                                        const uint32_t* __restrict__ wave_cnt = nullptr) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ uint32_t wcnt[4];
    const long row = blockIdx.x;
    constexpr int XB = XT == 0 ? 4 : 2;
    const void* xr = static_cast<const char*>(feat) + row * F * XB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* th = theta ? theta + (row % L) * F : nullptr;
    const int colbase = (int)(row % L) * F;
    const bool vec = rows_vec_form(xr, th, F, XB);
    if (vec) {
        // 16-byte loads, a wave owns a quarter of the row (a multiple of 256 columns): its entries are placed in column
        // order — lane l's four consecutive columns before lane l + 1's (four ballots give the offsets) — behind the entries
        // of the waves before it, whose counts count_rows_kernel left in wave_cnt (without them: one counting sweep first).
        // (The scalar form below read the row twice, 4 bytes per lane: 4.9 ms for 6.4 GB.)
        const int seg = rows_seg(F, true);
        const int s0 = min(F, w * seg), s1 = min(F, s0 + seg);
        auto flags = [&](int i, bool in, float4& x, bool p[4]) {
            const int ii = in ? i : 0;                     // (column 0 is always readable; F % 4 == 0, so i + 3 < F when in)
            x = rows_ld4<XT>(xr, ii);
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (th) t = *reinterpret_cast<const float4*>(th + ii);
            p[0] = in && x.x > 0.0f && x.x > t.x; p[1] = in && x.y > 0.0f && x.y > t.y;
            p[2] = in && x.z > 0.0f && x.z > t.z; p[3] = in && x.w > 0.0f && x.w > t.w;
        };
        uint32_t off = row_start[row];
        if (wave_cnt) {
            for (int k = 0; k < w; ++k) off += wave_cnt[row * 4 + k];
        } else {
            uint32_t c = 0;
            for (int base = s0; base < s1; base += 256) {
                const int i = base + lane * 4;
                float4 x; bool p[4];
                flags(i, i < s1, x, p);
                c += (uint32_t)p[0] + (uint32_t)p[1] + (uint32_t)p[2] + (uint32_t)p[3];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            if (lane == 0) wcnt[w] = c;
            __syncthreads();
            for (int k = 0; k < w; ++k) off += wcnt[k];
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int base = s0; base < s1; base += 256) {
            const int i = base + lane * 4;
            float4 x; bool p[4];
            flags(i, i < s1, x, p);
            const unsigned long long b0 = __ballot(p[0]), b1 = __ballot(p[1]), b2 = __ballot(p[2]), b3 = __ballot(p[3]);
            if ((b0 | b1 | b2 | b3) == 0ull) continue;
            uint32_t pos = off + (uint32_t)(__popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt) + __popcll(b3 & lt));
            const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p[q]) {
                    if (pos < cap) { ent_col[pos] = colbase + i + q; ent_val[pos] = xv[q]; }
                    else *overflow = 1;
                    ++pos;
                }
            off += (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
        }
        return;
    }
    const int seg = ((F + 3) / 4 + 63) / 64 * 64;
    const int s0 = w * seg, s1 = min(F, s0 + seg);
    uint32_t c = 0;
    for (int i = s0 + lane; i < s1; i += 64) c += list_entry(rows_ld1<XT>(xr, i), th, i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wcnt[w] = c;
    __syncthreads();
    uint32_t off = row_start[row];
    for (int k = 0; k < w; ++k) off += wcnt[k];
    for (int base = s0; base < s1; base += 64) {
        const int i = base + lane;
        const float v = i < s1 ? rows_ld1<XT>(xr, i) : 0.0f;
        const bool p = i < s1 && list_entry(v, th, i);
        const unsigned long long m = __ballot(p);
        if (p) {
            const uint32_t pos = off + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) { ent_col[pos] = colbase + i; ent_val[pos] = v; }
            else *overflow = 1;
        }
        off += __popcll(m);
    }
}

__global__ void count_cols_kernel(const int32_t* __restrict__ ent_col, const uint32_t* __restrict__ total, uint32_t cap,
                                  uint32_t* __restrict__ col_cnt, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    const uint32_t n = min(*total, cap);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x)
        atomicAdd(&col_cnt[ent_col[e]], 1u);
}

// one wave per CSR row: scatter its entries into the CSC lists (order inside a column fixed up by sort)
__global__ __launch_bounds__(LTPB) void fill_cols_kernel(const uint32_t* __restrict__ row_start, const int32_t* __restrict__ ent_col,
                                                        const float* __restrict__ ent_val, int rows, int L, uint32_t cap,
                                                        const uint32_t* __restrict__ col_start, uint32_t* __restrict__ cursor,
                                                        int32_t* __restrict__ csc_tok, float* __restrict__ csc_val,
                                                        int32_t* __restrict__ csc_ent, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint32_t e0 = row_start[row], e1 = min(row_start[row + 1], cap);
    const int tok = row / L;
    for (uint32_t e = e0 + (threadIdx.x & 63); e < e1; e += 64) {
        const int c = ent_col[e];
        const uint32_t pos = col_start[c] + atomicAdd(&cursor[c], 1u);
        csc_tok[pos] = tok; csc_val[pos] = ent_val[e]; csc_ent[pos] = (int32_t)e;
    }
}

// order each column's entries by token.  Short columns: one thread, insertion sort.  Longer columns are queued for
// sort_mid_cols_kernel (up to MIDCOL_MAX entries: one wave per column, rank counting) / sort_long_cols_kernel (a block, bitonic in LDS).
constexpr int SHORT_COL = 24, MIDCOL_MAX = 1024, LONGCOL_MAX = 4096;
__global__ void sort_cols_kernel(const uint32_t* __restrict__ col_start, int ncols, int32_t* __restrict__ tok,
                                 float* __restrict__ val, int32_t* __restrict__ ent, int32_t* __restrict__ long_list,
                                 uint32_t* __restrict__ long_count, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const uint32_t s = col_start[c], n = col_start[c + 1] - s;
    if (n < 2) return;
    if (n > SHORT_COL) { long_list[atomicAdd(long_count, 1u)] = c; return; }
    for (uint32_t i = 1; i < n; ++i) {
        const int32_t t = tok[s + i], en = ent ? ent[s + i] : 0; const float v = val[s + i];
        uint32_t j = i;
        while (j > 0 && tok[s + j - 1] > t) { tok[s + j] = tok[s + j - 1]; val[s + j] = val[s + j - 1]; if (ent) ent[s + j] = ent[s + j - 1]; --j; }
        tok[s + j] = t; val[s + j] = v; if (ent) ent[s + j] = en;
    }
}

// columns of SHORT_COL < n <= MIDCOL_MAX entries: one WAVE each, no block barrier — the keys of a list are distinct (a token / a
// column occurs once), so an entry's place is the number of smaller keys.  Up to 64 entries: one entry per lane, the keys passed
// round by v_readlane (no LDS at all); beyond: every lane counts for its (up to 16) entries against the list, read key by key
// from the wave's LDS copy.  Entries go straight to their places.  (16 KB of LDS per block: ten blocks per CU — the column
// queue of a thresholding sparsifier at 0.5 % active holds most of the L*F columns, and a wave's work per column is one
// round trip to memory.)
// (one wave; s, n wave-uniform, 2 <= n <= MIDCOL_MAX; wtok: MIDCOL_MAX words of LDS of this wave's own)
__device__ __forceinline__ void wave_sort_list(uint32_t s, uint32_t n, int32_t* __restrict__ tok, float* __restrict__ val,
                                               int32_t* __restrict__ ent, int32_t* wtok, int lane) {
    if (n <= 64) {
        const bool in = (uint32_t)lane < n;
        const int32_t t = in ? tok[s + lane] : 0x7fffffff;
        const float v = in ? val[s + lane] : 0.0f;
        const int32_t en = (in && ent) ? ent[s + lane] : 0;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += __builtin_amdgcn_readlane(t, j) < t ? 1u : 0u;
        if (in) { tok[s + rank] = t; val[s + rank] = v; if (ent) ent[s + rank] = en; }
        return;
    }
    const int E = (int)((n + 63) >> 6);
    int32_t t[MIDCOL_MAX / 64], en[MIDCOL_MAX / 64];
    float v[MIDCOL_MAX / 64];
    uint32_t rank[MIDCOL_MAX / 64];
#pragma unroll
    for (int e = 0; e < MIDCOL_MAX / 64; ++e) {
        const uint32_t i = (uint32_t)e * 64 + lane;
        rank[e] = 0;
        if (e < E && i < n) { t[e] = tok[s + i]; v[e] = val[s + i]; en[e] = ent ? ent[s + i] : 0; wtok[i] = t[e]; }
        else t[e] = 0x7fffffff;
    }
    __builtin_amdgcn_wave_barrier();               // (the copy is this wave's own: LDS serves a wave's accesses in order)
    for (uint32_t j = 0; j < n; ++j) {
        const int32_t tj = wtok[j];
#pragma unroll
        for (int e = 0; e < MIDCOL_MAX / 64; ++e) if (e < E) rank[e] += tj < t[e] ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < MIDCOL_MAX / 64; ++e) {
        const uint32_t i = (uint32_t)e * 64 + lane;
        if (e < E && i < n) { tok[s + rank[e]] = t[e]; val[s + rank[e]] = v[e]; if (ent) ent[s + rank[e]] = en[e]; }
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(LTPB) void sort_mid_cols_kernel(const uint32_t* __restrict__ col_start, int32_t* __restrict__ tok,
                                                            float* __restrict__ val, int32_t* __restrict__ ent,
                                                            const int32_t* __restrict__ long_list,
                                                            const uint32_t* __restrict__ long_count, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ int32_t wtok_all[(LTPB / 64) * MIDCOL_MAX];
    const uint32_t nl = *long_count;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t q = blockIdx.x * (LTPB / 64) + wv; q < nl; q += gridDim.x * (LTPB / 64)) {
        const int c = __builtin_amdgcn_readfirstlane(long_list[q]);
        const uint32_t s = __builtin_amdgcn_readfirstlane(col_start[c]);
        const uint32_t n = __builtin_amdgcn_readfirstlane(col_start[c + 1]) - s;
        if (n > MIDCOL_MAX) continue;
        wave_sort_list(s, n, tok, val, ent, wtok_all + wv * MIDCOL_MAX, lane);
    }
}

// the rows of a token-major list (a dozen or more entries each, unlike most columns): one wave per row straight away — a
// thread per row walks its own 100-odd bytes and nothing coalesces.  Rows beyond MIDCOL_MAX are queued for sort_long_cols_kernel
__global__ __launch_bounds__(LTPB) void sort_rows_kernel(const uint32_t* __restrict__ row_start, int rows, int32_t* __restrict__ col,
                                                        float* __restrict__ val, int32_t* __restrict__ long_list,
                                                        uint32_t* __restrict__ long_count, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ int32_t wtok_all[(LTPB / 64) * MIDCOL_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * (LTPB / 64) + wv;
    if (row >= rows) return;
    const uint32_t s = __builtin_amdgcn_readfirstlane(row_start[row]);
    const uint32_t n = __builtin_amdgcn_readfirstlane(row_start[row + 1]) - s;
    if (n < 2) return;
    if (n > MIDCOL_MAX) { if (lane == 0) long_list[atomicAdd(long_count, 1u)] = row; return; }
    wave_sort_list(s, n, col, val, (int32_t*)nullptr, wtok_all + wv * MIDCOL_MAX, lane);
}

// columns of MIDCOL_MAX < n <= LONGCOL_MAX entries: a block each (bitonic in LDS).  The queue is read LTPB entries at a time by
// the whole block, which keeps the few columns of that length in a small LDS queue: a block does not pay a dependent
// round trip to memory per queue entry that is not its business
__global__ __launch_bounds__(LTPB) void sort_long_cols_kernel(const uint32_t* __restrict__ col_start, int32_t* __restrict__ tok,
                                                             float* __restrict__ val, int32_t* __restrict__ ent,
                                                             const int32_t* __restrict__ long_list,
                                                             const uint32_t* __restrict__ long_count, const int32_t* __restrict__ gate, int gate_on) {
    if (gate && (*gate != 0) != (gate_on != 0)) return;
    __shared__ unsigned long long key[LONGCOL_MAX];     // (token << 32) | position within the column
    __shared__ float sval[LONGCOL_MAX];
    __shared__ int32_t sent[LONGCOL_MAX];
    __shared__ int32_t mine[LTPB];
    __shared__ uint32_t nmine;
    const uint32_t nl = *long_count;
    // (block b reads the queue entries congruent to b modulo the grid: hot columns are neighbours in the queue — a Zipf
    // dictionary's frequent features have neighbouring indices — and one block must not end up with all of them)
    for (uint32_t base = 0; base < nl; base += gridDim.x * LTPB) {
        if (threadIdx.x == 0) nmine = 0;
        __syncthreads();
        const uint32_t q = base + threadIdx.x * gridDim.x + blockIdx.x;
        if (q < nl) {
            const int c = long_list[q];
            const uint32_t n = col_start[c + 1] - col_start[c];
            if (n > MIDCOL_MAX && n <= LONGCOL_MAX) mine[atomicAdd(&nmine, 1u)] = c;    // (beyond LONGCOL_MAX: left in arrival
        }                                                                               //  order — still correct, not bit-reproducible)
        __syncthreads();
        const uint32_t m = nmine;
        for (uint32_t k0 = 0; k0 < m; ++k0) {
            const int c = mine[k0];
            const uint32_t s = col_start[c], n = col_start[c + 1] - s;
            uint32_t np2 = 1; while (np2 < n) np2 <<= 1;
            for (uint32_t i = threadIdx.x; i < np2; i += LTPB) {
                key[i] = i < n ? (((unsigned long long)(uint32_t)tok[s + i] << 32) | i) : ~0ull;
                if (i < n) { sval[i] = val[s + i]; sent[i] = ent ? ent[s + i] : 0; }
            }
            __syncthreads();
            for (uint32_t k = 2; k <= np2; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = threadIdx.x; i < np2; i += LTPB) {
                        const uint32_t ixj = i ^ j;
                        if (ixj > i) {
                            const bool up = (i & k) == 0;
                            const unsigned long long a = key[i], b = key[ixj];
                            if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            for (uint32_t i = threadIdx.x; i < n; i += LTPB) {
                const uint32_t src = (uint32_t)(key[i] & 0xffffffffu);
                tok[s + i] = (int32_t)(key[i] >> 32); val[s + i] = sval[src]; if (ent) ent[s + i] = sent[src];
            }
            __syncthreads();
        }
    }
}

// rows of a token-major list in column order
static inline void launch_sort_rows(hipStream_t st, const uint32_t* row_start, int rows, int32_t* col, float* val, int32_t* long_list,
                                    uint32_t* long_count, const int32_t* gate, int gate_on) {
    hipLaunchKernelGGL(sort_rows_kernel, dim3((rows + LTPB / 64 - 1) / (LTPB / 64)), dim3(LTPB), 0, st, row_start, rows, col, val, long_list,
                       long_count, gate, gate_on);
    hipLaunchKernelGGL(sort_long_cols_kernel, dim3(512), dim3(LTPB), 0, st, row_start, col, val, (int32_t*)nullptr, long_list, long_count,
                       gate, gate_on);
}

// the queue of sort_cols_kernel: mid columns by waves, long columns by blocks
static inline void launch_sort_queued_cols(hipStream_t st, const uint32_t* col_start, int32_t* tok, float* val, int32_t* ent,
                                           const int32_t* long_list, const uint32_t* long_count, const int32_t* gate, int gate_on) {
    hipLaunchKernelGGL(sort_mid_cols_kernel, dim3(2048), dim3(LTPB), 0, st, col_start, tok, val, ent, long_list, long_count, gate, gate_on);
    hipLaunchKernelGGL(sort_long_cols_kernel, dim3(512), dim3(LTPB), 0, st, col_start, tok, val, ent, long_list, long_count, gate, gate_on);
}


// rows mode: pack the [rows][k] compact rows (ascending feature id, valid count in row_cnt) into the CSR
__global__ __launch_bounds__(LTPB) void compact_rows_kernel(const int32_t* __restrict__ cidx, const float* __restrict__ cval, int k, int F, int L,
                                                           int rows, const uint32_t* __restrict__ row_start, uint32_t cap,
                                                           int32_t* __restrict__ ent_col, float* __restrict__ ent_val,
                                                           int32_t* __restrict__ overflow) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint32_t s = row_start[row], n = row_start[row + 1] - s;
    const int colbase = (row % L) * F;
    for (uint32_t e = threadIdx.x & 63; e < n; e += 64) {
        if (s + e < cap) { ent_col[s + e] = colbase + cidx[(long)row * k + e]; ent_val[s + e] = cval[(long)row * k + e]; }
        else *overflow = 1;
    }
}

}  // namespace
