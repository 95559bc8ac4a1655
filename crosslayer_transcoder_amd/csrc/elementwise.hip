// HBM-bound elementwise / reduction kernels of the crosslayer-transcoder training step (gfx950).
// One pass over each [B, L*F] or [B, L*d] tensor per kernel; every reduction is two-stage and
// fixed-order, so results are bitwise reproducible run to run.
#include <math.h>
#include <mutex>
#include "common.h"

namespace {
constexpr int TPB = 256;
inline int col_chunks(int rows) { int c = (rows + 63) / 64; return c < 1 ? 1 : (c > 32 ? 32 : c); }
inline int blocks_for(long count, int per_block, int cap = 8192) {
    long b = (count + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- K1 -------------------------------------------------------------------------------------
// 64 columns x 16 row phases per block: phase p takes rows p, p + 16, ... (4 loads in flight), the 16 partial sums of a
// column are added in phase order — fp64 throughout, two passes (mean, then squared deviations) as torch.std does
__global__ __launch_bounds__(1024) void column_stats_kernel(const float* __restrict__ x, long rs, int rows, int n, float* __restrict__ mean,
                                                            float* __restrict__ stdv) {
    __shared__ double part[16][64];
    __shared__ double mu_s[64];
    const int tx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    const bool in = j < n;
    auto column_sum = [&](auto f) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (in) {
            int r = ph;
            for (; r + 48 < rows; r += 64) {
                const float a = x[(long)r * rs + j], b = x[(long)(r + 16) * rs + j], c = x[(long)(r + 32) * rs + j], d = x[(long)(r + 48) * rs + j];
                s0 += f(a); s1 += f(b); s2 += f(c); s3 += f(d);
            }
            for (; r < rows; r += 16) s0 += f(x[(long)r * rs + j]);
        }
        part[ph][tx] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        double s = 0.0;
        if (ph == 0) for (int p = 0; p < 16; ++p) s += part[p][tx];
        __syncthreads();
        return s;                                    // (valid in phase 0)
    };
    const double s = column_sum([](float v) { return (double)v; });
    if (ph == 0) mu_s[tx] = s / rows;
    __syncthreads();
    const double mu = mu_s[tx];
    const double q = column_sum([mu](float v) { const double dlt = (double)v - mu; return dlt * dlt; });
    if (ph == 0 && in) {
        // torch.std: unbiased (N-1); a single row gives nan in torch, as here (0/0)
        const float sd = (float)sqrt(q / (double)(rows - 1));
        mean[j] = (float)mu;
        stdv[j] = sd < 1e-8f ? 1e-8f : sd;    // clamp_(min=1e-8); nan stays nan as in torch.clamp
    }
}

template <bool VEC, bool F16>
__global__ void standardize_kernel(const float* __restrict__ x, long rs, const float* __restrict__ mean,
                                   const float* __restrict__ stdv, int rows, int n, float* __restrict__ o32,
                                   uint16_t* __restrict__ o16) {
    constexpr int W = VEC ? 4 : 1;
    const int nq = n / W;
    const long total = (long)rows * nq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / nq), j = (int)(i - (long)r * nq) * W;
        float v[W];
        if (VEC) {
            const float4 a = *reinterpret_cast<const float4*>(x + (long)r * rs + j);
            const float4 m = *reinterpret_cast<const float4*>(mean + j);
            const float4 s = *reinterpret_cast<const float4*>(stdv + j);
            v[0] = (a.x - m.x) / s.x; v[1 % W] = (a.y - m.y) / s.y; v[2 % W] = (a.z - m.z) / s.z; v[3 % W] = (a.w - m.w) / s.w;
        } else {
            v[0] = (x[(long)r * rs + j] - mean[j]) / stdv[j];
        }
        const long o = (long)r * n + j;
        if (o32) {
            if (VEC) *reinterpret_cast<float4*>(o32 + o) = make_float4(v[0], v[1 % W], v[2 % W], v[3 % W]);
            else o32[o] = v[0];
        }
        if (o16) {
            if (VEC) {
                ushort4 h; h.x = to16<F16>(v[0]); h.y = to16<F16>(v[1 % W]); h.z = to16<F16>(v[2 % W]); h.w = to16<F16>(v[3 % W]);
                *reinterpret_cast<ushort4*>(o16 + o) = h;
            } else o16[o] = to16<F16>(v[0]);
        }
    }
}

__global__ void destandardize_kernel(const float* __restrict__ r, const float* __restrict__ mean,
                                     const float* __restrict__ stdv, long total, int n, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % n);
        out[i] = r[i] * stdv[j] + mean[j];
    }
}

// ---- K3 -------------------------------------------------------------------------------------
__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ o, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        o[i] = v > 0.0f ? v : (v != v ? v : 0.0f);
    }
}
__global__ void relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = x[i] > 0.0f ? g[i] : 0.0f;
}
__global__ void mask_positive_kernel(const float* __restrict__ f, const float* __restrict__ g, float* __restrict__ dx, long n,
                                     int accumulate) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = f[i] > 0.0f ? g[i] : 0.0f;
        dx[i] = accumulate ? dx[i] + v : v;
    }
}
__global__ void jumprelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ theta, float* __restrict__ o,
                                    long total, int n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        o[i] = (v > theta[i % n] && v > 0.0f) ? v : v * 0.0f;   // mask * input (keeps nan/inf semantics)
    }
}

__device__ __forceinline__ float rect_window(float x, float th, float bw) {
    const float u = (x - th) / bw;                         // jumprelu.py:9-14, 47
    return ((u + 0.5f > 0.0f) ? 1.0f : 0.0f) - ((u - 0.5f > 0.0f) ? 1.0f : 0.0f);
}

// (bf16 regime: the division by the bandwidth as a multiplication by its reciprocal)
__device__ __forceinline__ float rect_window_rcp(float x, float th, float inv_bw) {
    const float u = (x - th) * inv_bw;
    return ((u + 0.5f > 0.0f) ? 1.0f : 0.0f) - ((u - 0.5f > 0.0f) ? 1.0f : 0.0f);
}

// grid: (ceil(n/256), chunks).  Each thread owns one column j and a contiguous chunk of rows.
__global__ void jumprelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ theta, float bw,
                                    const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ part,
                                    int rows, int n, int rows_per_chunk) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    const float th = theta[j], coef = -(th / bw);
    float acc = 0.0f;
    for (int r = r0; r < r1; ++r) {
        const long i = (long)r * n + j;
        const float v = x[i], gg = g[i];
        dx[i] = v < 0.0f ? 0.0f : gg;
        acc += coef * rect_window(v, th, bw) * gg;
    }
    part[(long)blockIdx.y * n + j] = acc;
}

__global__ void reduce_cols_kernel(const float* __restrict__ part, int chunks, int n, float* __restrict__ out, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.0f;
    for (int c = 0; c < chunks; ++c) s += part[(long)c * n + j];
    out[j] = accumulate ? out[j] + s : s;
}

__global__ void column_sum_kernel(const float* __restrict__ x, float* __restrict__ part, int rows, int n, int rows_per_chunk) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float acc = 0.0f;
    for (int r = r0; r < r1; ++r) acc += x[(long)r * n + j];
    part[(long)blockIdx.y * n + j] = acc;
}

// sums nb x k block partials (double) in block order; one block of 64 threads per output.
// (256 threads, four partials in flight per thread: one wave walking tens of thousands of block partials one dependent load
// at a time took 84 us behind the dense JumpReLU loss pass)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ part, int nb, int k, double* __restrict__ out) {
    __shared__ double sh[4];
    const int o = blockIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = threadIdx.x;
    for (; b + 768 < nb; b += 1024) {
        s0 += part[(long)b * k + o]; s1 += part[(long)(b + 256) * k + o];
        s2 += part[(long)(b + 512) * k + o]; s3 += part[(long)(b + 768) * k + o];
    }
    for (; b < nb; b += 256) s0 += part[(long)b * k + o];
    const double s = block_sum_d((s0 + s1) + (s2 + s3), sh);
    if (threadIdx.x == 0) out[o] = s;
}

// ---- K8 -------------------------------------------------------------------------------------
__global__ void mse_kernel(const float* __restrict__ r, const float* __restrict__ y, long yrs, const float* __restrict__ mean,
                           const float* __restrict__ stdv, int rows, int n, float gs, float* __restrict__ g,
                           float* __restrict__ ys_out, double* __restrict__ part) {
    __shared__ double sh[4];
    const long total = (long)rows * n;
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int rr = (int)(i / n), j = (int)(i - (long)rr * n);
        const float ys = (y[(long)rr * yrs + j] - mean[j]) / stdv[j];
        const float dlt = r[i] - ys;
        if (g) g[i] = gs * dlt;
        if (ys_out) ys_out[i] = ys;
        acc += (double)(dlt * dlt);
    }
    const double s = block_sum_d(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// The base-loss step of the list regime in bf16 mode wants three things of the reconstruction error and nothing else: the
// loss sum, d(loss)/d(recons) as the bf16 rows the weight pass gathers, and its column sums (the decoder bias gradient).
// One pass instead of mse + cast + column_sum, and no fp32 gradient tensor.  Block = 64 rows x 256 columns (a wave per
// row, 4 columns per lane); part_c: [gridDim.y][n] column partials, part_l: [gridDim.x * gridDim.y] loss partials.
// Guarded 16-byte load, branch-free: an out-of-range element reads the tensor's first element instead and is replaced by zero
// (see the tile passes below: a load inside `if (in_range)` is not hoisted, and the wave then has two loads in flight, not sixteen)
// (`safe`: an index that is always readable — 0 unless `base` is a rebased pointer whose element 0 lies outside the tensor)
__device__ __forceinline__ float4 ld4z(const float* __restrict__ base, long idx, bool ok, long safe = 0) {
    const float4 v = *reinterpret_cast<const float4*>(base + (ok ? idx : safe));
    return make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
}
constexpr int MF_R = 64;                                // (32 / 16 / 8 rows per block measured equal or slower on a cold cache)
constexpr int MF_U = MF_R / 4 < 8 ? MF_R / 4 : 8;      // rows a thread has in flight
template <bool F16>
__global__ __launch_bounds__(256) void mse_bf16_colsum_kernel(const float* __restrict__ r, const float* __restrict__ y, long yrs,
                                                              const float* __restrict__ mean, const float* __restrict__ stdv, int rows,
                                                              int n, float gs, uint16_t* __restrict__ g16, float* __restrict__ part_c,
                                                              double* __restrict__ part_l, int layer_rows) {
    __shared__ float red[4][256];
    __shared__ double sh[4];
    const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + 4 * tx;
    const int r0 = blockIdx.y * MF_R;
    const bool cin = col < n;
    // layer_rows > 0: r / g16 rows are layer-major (row = layer * layer_rows + token, layer_rows % 64 == 0, so a block's
    // rows share one layer); the target row of (layer, token) is y + token * yrs + layer * n and mean / std are [layers][n]
    const int layer = layer_rows > 0 ? r0 / layer_rows : 0;
    if (layer_rows > 0) { y += (long)layer * n - (long)layer * layer_rows * yrs; mean += (long)layer * n; stdv += (long)layer * n; }
    float4 mu = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sd = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (cin) { mu = *reinterpret_cast<const float4*>(mean + col); sd = *reinterpret_cast<const float4*>(stdv + col); }
    float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    double acc = 0.0;
    // (8 rows = 16 loads of 16 bytes in flight per thread)
    for (int i0 = 0; i0 < MF_R / 4; i0 += MF_U) {
        float4 rv[MF_U], yv[MF_U];
#pragma unroll
        for (int u = 0; u < MF_U; ++u) {
            const int row = r0 + w + 4 * (i0 + u);
            // (y was rebased by the layer offset above: its element 0 may lie outside the batch — the block's first row is safe)
            rv[u] = ld4z(r, (long)row * n + col, cin && row < rows, (long)r0 * n);
            yv[u] = ld4z(y, (long)row * yrs + col, cin && row < rows, (long)r0 * yrs);
        }
        // (the bf16 rows leave by 16-byte stores where the shape allows: neighbouring lanes trade the halves of two rows — the
        // even lane stores 8 columns of row u, the odd lane 8 columns of row u + 1; 8-byte stores run at about half the rate)
        const bool wide = (n & 7) == 0 && (rows & (MF_R - 1)) == 0 && (reinterpret_cast<uintptr_t>(g16) & 15) == 0;
#pragma unroll
        for (int u = 0; u < MF_U; u += 2) {
            uint32_t q[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int row = r0 + w + 4 * (i0 + u + e);
                q[e][0] = q[e][1] = 0;
                if (!cin || row >= rows) continue;
                const float4 rr = rv[u + e], yy = yv[u + e];
                // (IEEE division kept — a multiply by 1 / std would save 6 of 46 us on a 6-layer slab, but the bf16 rows are
                // defined as the rounding of clt_mse_f32's gradient, bit for bit)
                const float d0 = rr.x - (yy.x - mu.x) / sd.x, d1 = rr.y - (yy.y - mu.y) / sd.y;
                const float d2 = rr.z - (yy.z - mu.z) / sd.z, d3 = rr.w - (yy.w - mu.w) / sd.w;
                const float g0 = gs * d0, g1 = gs * d1, g2 = gs * d2, g3 = gs * d3;
                cs[0] += g0; cs[1] += g1; cs[2] += g2; cs[3] += g3;
                acc += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
                q[e][0] = pack16<F16>(g0, g1); q[e][1] = pack16<F16>(g2, g3);
                if (!wide) *reinterpret_cast<uint2*>(g16 + (long)row * n + col) = make_uint2(q[e][0], q[e][1]);
            }
            if (wide) {
                const bool odd = tx & 1;
                const uint32_t x0 = __shfl_xor(odd ? q[0][0] : q[1][0], 1, 64), x1 = __shfl_xor(odd ? q[0][1] : q[1][1], 1, 64);
                const long row = r0 + w + 4 * (i0 + u + (odd ? 1 : 0));
                if (cin) {
                    if (odd) *reinterpret_cast<uint4*>(g16 + row * n + col - 4) = make_uint4(x0, x1, q[1][0], q[1][1]);
                    else *reinterpret_cast<uint4*>(g16 + row * n + col) = make_uint4(q[0][0], q[0][1], x0, x1);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[w][4 * tx + k] = cs[k];
    const double s = block_sum_d(acc, sh);                 // (contains the barrier that also publishes red[])
    if (threadIdx.x == 0) part_l[blockIdx.y * gridDim.x + blockIdx.x] = s;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) part_c[(long)blockIdx.y * n + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[g][c] (+)= sum over the chunks of group g (blockIdx.y; nchunks consecutive chunks each) of part[chunk][c], fixed order
// The block after the column blocks (blockIdx.x == gridDim.x - 1, group 0) adds the nb_l loss partials into loss_sum[0]
// (when part_l is given): one launch finishes both reductions of the pass.
__global__ void sum_chunks_kernel(const float* __restrict__ part, int nchunks, int n, float* __restrict__ out, int accumulate,
                                  const double* __restrict__ part_l, int nb_l, double* __restrict__ loss_sum) {
    if (part_l && blockIdx.x == gridDim.x - 1) {
        if (blockIdx.y == 0 && threadIdx.x < 64) {
            double s = 0.0;
            for (int b = threadIdx.x; b < nb_l; b += 64) s += part_l[b];
            s = wave_sum_d(s);
            if (threadIdx.x == 0) loss_sum[0] = s;
        }
        return;
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    part += (long)blockIdx.y * nchunks * n;
    out += (long)blockIdx.y * n;
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < nchunks; ++k) s += part[(long)k * n + c];          // (fixed order; the loads of 8 chunks in flight)
    out[c] = accumulate ? out[c] + s : s;
}

__global__ void aux_mse_kernel(const float* __restrict__ ar, const float* __restrict__ ys, const float* __restrict__ r,
                               long total, float gs, float* __restrict__ ga, float* __restrict__ gm,
                               double* __restrict__ part) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float e = ar[i] - (ys[i] - r[i]);
        const float e2 = e * e;
        const bool fin = isfinite(e2);
        // nan_to_num(e2, 0.0): nan -> 0, +inf -> FLT_MAX; its backward passes grad only where finite
        acc += (e2 != e2) ? 0.0 : (fin ? (double)e2 : (double)3.4028234663852886e38f);
        const float gv = fin ? gs * e : 0.0f;
        ga[i] = gv;
        gm[i] += gv;
    }
    const double s = block_sum_d(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- K9 / K10 ---------------------------------------------------------------------------------
// one wave per row
__global__ void row_sqnorm_kernel(const float* __restrict__ W, int rows, int d, float* __restrict__ sq, int accumulate) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = W + (long)row * d;
    float s = 0.0f;
    if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {          // 16-byte loads (rows stay 16-byte aligned)
        for (int j = lane * 4; j < d; j += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + j);
            s += sq4(v.x, v.y, v.z, v.w);
        }
    } else {
        for (int j = lane; j < d; j += 64) { const float v = p[j]; s = __fmaf_rn(v, v, s); }
    }
    s = wave_sum(s);
    if (lane == 0) sq[row] = accumulate ? sq[row] + s : s;
}

__global__ void rowscale_add_kernel(const float* __restrict__ W, const float* __restrict__ dnorm, const float* __restrict__ sq,
                                    float* __restrict__ dW, int rows, int d) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float coef = dnorm[row] / sqrtf(sq[row]);
    const long o = (long)row * d;
    for (int j = lane; j < d; j += 64) dW[o + j] += coef * W[o + j];
}

// grid: (ceil(n/256), chunks); part layout: [2][chunks][n] floats then [nblocks][3] doubles
__global__ void jumprelu_loss_kernel(const float* __restrict__ x, const float* __restrict__ theta, const float* __restrict__ sq,
                                     const float* __restrict__ dfm, float bw, float lamB, float c, int use_tanh,
                                     float coefBL, float* __restrict__ dpre, float* __restrict__ part_th,
                                     float* __restrict__ part_dn, double* __restrict__ part_s, int rows, int n,
                                     int rows_per_chunk) {
    __shared__ double sh[4];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double s_t = 0.0, s_p = 0.0, s_w = 0.0;
    if (j < n) {
        const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
        const float th = theta[j], nrm = sqrtf(sq[j]), thc = -(th / bw);
        float a_th = 0.0f, a_dn = 0.0f;
        for (int r = r0; r < r1; ++r) {
            const long i = (long)r * n + j;
            const float v = x[i];
            const float f = (v > th && v > 0.0f) ? v : 0.0f;
            const float w = f * nrm * c;
            const float t = use_tanh ? tanhf(w) : w;
            const float dt = use_tanh ? (1.0f - t * t) : 1.0f;
            s_t += (double)t; s_w += (double)w;
            s_p += (double)(v < 0.0f ? -v : 0.0f);
            if (dfm) {
                const float df = dfm[i] + lamB * dt * c * nrm;
                dpre[i] = (v < 0.0f ? 0.0f : df) - (v < 0.0f ? coefBL : 0.0f);
                a_th += thc * rect_window(v, th, bw) * df;
                a_dn += lamB * dt * c * f;
            }
        }
        if (dfm) { part_th[(long)blockIdx.y * n + j] = a_th; part_dn[(long)blockIdx.y * n + j] = a_dn; }
    }
    const long b = (long)blockIdx.y * gridDim.x + blockIdx.x;
    double s;
    s = block_sum_d(s_t, sh); if (threadIdx.x == 0) part_s[b * 3 + 0] = s;
    s = block_sum_d(s_p, sh); if (threadIdx.x == 0) part_s[b * 3 + 1] = s;
    s = block_sum_d(s_w, sh); if (threadIdx.x == 0) part_s[b * 3 + 2] = s;
}

// ---- K12 --------------------------------------------------------------------------------------
__global__ void column_any_positive_kernel(const float* __restrict__ f, int rows, int n, uint8_t* __restrict__ fired) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = j < n;
    bool hit = false;
    for (int r0 = 0; r0 < rows; r0 += 8) {
        if (in && !hit) {
#pragma unroll
            for (int u = 0; u < 8; ++u) if (r0 + u < rows) hit |= f[(long)(r0 + u) * n + j] > 0.0f;
        }
        if (__all(hit || !in)) break;
    }
    if (in) fired[j] = hit ? 1 : 0;
}

__global__ void dead_update_kernel(const uint8_t* __restrict__ fired, int n, long add, int64_t* __restrict__ la, long ttd,
                                   uint8_t* __restrict__ dead, int32_t* __restrict__ dead_count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    int dflag = 0;
    if (j < n) {
        const int64_t v = (fired[j] ? 0 : la[j]) + add;
        la[j] = v;
        dflag = v > ttd ? 1 : 0;
        dead[j] = (uint8_t)dflag;
    }
    const unsigned long long m = __ballot(dflag);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(dead_count, __popcll(m));   // integer: order-independent
}

__global__ void mask_dead_kernel(const float* __restrict__ x, const uint8_t* __restrict__ dead, float* __restrict__ o,
                                 long total, int n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        o[i] = dead[i % n] ? x[i] : -INFINITY;
}

__global__ void threshold_fwd_kernel(const float* __restrict__ x, const float* __restrict__ thr, int per_layer,
                                     float* __restrict__ o, long total, int L, int F) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int l = per_layer ? (int)((i / F) % L) : 0;
        const float t = fmaxf(thr[l], 0.0f), v = x[i];
        o[i] = v > t ? v : 0.0f;
    }
}

// n8 groups of 8 elements (two 16-byte loads, one 16-byte store per thread; block b owns a contiguous range), then the tail
template <bool F16>
__global__ void cast_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, long n8, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n8) {
        const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
        reinterpret_cast<uint4*>(out)[i] = make_uint4(pack16<F16>(a.x, a.y), pack16<F16>(a.z, a.w), pack16<F16>(b.x, b.y), pack16<F16>(b.z, b.w));
    }
    for (long j = n8 * 8 + i; j < n; j += (long)gridDim.x * blockDim.x) out[j] = to16<F16>(in[j]);
}

// 64x64 tile transpose through LDS: in [rows][cols] (ld_in) -> out [cols][rows] (ld_out), bf16
template <bool F16>
__global__ void cast_transpose_kernel(const float* __restrict__ in, long ldi, uint16_t* __restrict__ out, long ldo, int rows,
                                      int cols, long s_in, long s_out) {
    __shared__ float tile[64][65];
    in += blockIdx.z * s_in; out += blockIdx.z * s_out;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? in[(long)(r0 + r) * ldi + c0 + tx] : 0.0f;
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) out[(long)(c0 + c) * ldo + r0 + tx] = to16<F16>(tile[tx][c]);
}

// ---- fused tile passes of the dense bf16 regime (JumpReLU / ReLU modules) ---------------------------------------
// The dense step needs the feature matrix twice as a bf16 GEMM operand — [tokens][L*F] for the decoder forward and
// transposed [L*F][tokens] for the decoder weight gradient — and the same for d(loss)/d(pre-activations) of the
// encoder weight gradient.  These kernels read the fp32 pre-activations ONCE per direction and write the bf16 operands
// directly (no fp32 feature / dpre tensor, no separate cast, transpose, column-sum or any-positive passes).
// Tile: 64 rows (tokens) x 256 columns; a thread owns 4 consecutive columns of the row PAIRS w, w+4, ... (rows 2p, 2p+1).
// The transposed copy goes through LDS as [32 pairs][256] dwords, a dword = (row 2p, row 2p+1) of one column — one packed
// conversion produces it, and the transposed read-out takes 8 rows of a column with 4 b32 reads (pitch 258 dwords: the 8 row
// groups x 8 columns of a wave-instruction fall on 64 distinct banks).
constexpr int FT_R = 64, FT_C = 256, FT_PITCH = 258;      // pitch in dwords

template <bool F16>
__device__ __forceinline__ void ft_store_pair(uint32_t (*tile)[FT_PITCH], int p, int tx, const float a[4], const float b[4]) {
    uint2* q = reinterpret_cast<uint2*>(&tile[p][4 * tx]);
    q[0] = make_uint2(pack16<F16>(a[0], b[0]), pack16<F16>(a[1], b[1]));
    q[1] = make_uint2(pack16<F16>(a[2], b[2]), pack16<F16>(a[3], b[3]));
}
// TILE_NT (bit 0 loads, bit 1 stores): the activation streams of the tile passes as non-temporal accesses.  Measured SLOWER
// (PLT JumpReLU dense 19.9 ms -> 20.0 with loads, 20.4 with stores, scripts/archive/gpu_tilent_ab.sh): what these passes write is the
// next GEMM's operand and what they read the last GEMM's result, and part of it is still in the memory-side cache; unlike the
// row pass's weight streams and the GEMMs' C tiles, where the hint pays.  Off.
#ifndef TILE_NT
#define TILE_NT 0
#endif
typedef unsigned int tp_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int tp_u2 __attribute__((ext_vector_type(2)));
typedef float tp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tp_st16(void* q, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if TILE_NT & 2
    const tp_u4 t = {a, b, c, d}; __builtin_nontemporal_store(t, reinterpret_cast<tp_u4*>(q));
#else
    *reinterpret_cast<uint4*>(q) = make_uint4(a, b, c, d);
#endif
}
// transposed write-out of a [64][256] bf16 tile: column c gets 64 consecutive rows (128 B); 8 threads x 16 B per column
__device__ __forceinline__ void ft_write_transposed(const uint32_t (*tile)[FT_PITCH], uint16_t* __restrict__ outT, long c0, int r0,
                                                    int rows, int n) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 256 + threadIdx.x, c = idx >> 3, rg = idx & 7;
        const uint4 q = make_uint4(tile[rg * 4][c], tile[rg * 4 + 1][c], tile[rg * 4 + 2][c], tile[rg * 4 + 3][c]);
        if (c0 + c < n && r0 + rg * 8 < rows)
            tp_st16(outT + (c0 + c) * (long)rows + r0 + rg * 8, q.x, q.y, q.z, q.w);
    }
}

// Row-major 16-bit copy of a row pair (a = row, b = row + 1; this lane's 4 columns at `col`): 8-byte stores run at about half
// the rate of 16-byte ones, so neighbouring lanes trade halves — the even lane stores 8 columns of the first row, the odd lane
// 8 columns of the second (wide: n % 8 == 0, so the partner's columns exist and the rows are 16-byte aligned).
template <bool F16>
__device__ __forceinline__ void store_rowpair16(uint16_t* __restrict__ out, long row, int n, long col, int tx, const float a[4],
                                                const float b[4], bool in, bool wide) {
    const uint32_t a0 = pack16<F16>(a[0], a[1]), a1 = pack16<F16>(a[2], a[3]);
    const uint32_t b0 = pack16<F16>(b[0], b[1]), b1 = pack16<F16>(b[2], b[3]);
    if (wide) {
        const bool odd = tx & 1;
        const uint32_t r0 = __shfl_xor(odd ? a0 : b0, 1, 64), r1 = __shfl_xor(odd ? a1 : b1, 1, 64);
        if (in) {
            if (odd) tp_st16(out + (row + 1) * n + col - 4, r0, r1, b0, b1);
            else tp_st16(out + row * n + col, a0, a1, r0, r1);
        }
    } else if (in) {
        *reinterpret_cast<uint2*>(out + row * n + col) = make_uint2(a0, a1);
        *reinterpret_cast<uint2*>(out + (row + 1) * n + col) = make_uint2(b0, b1);
    }
}

// four consecutive elements of an activation tensor held as fp32 (BF = false) or bf16 (BF = true: the dense bf16 regime's
// GEMMs can leave pre-activations and feature gradients as bf16, clt_gemm_bf16_nt_out16)
template <bool BF, bool F16> __device__ __forceinline__ float4 act_load4(const void* __restrict__ base, long idx) {
    if (BF) {           // 16-bit activations in the active operand format
#if TILE_NT & 1
        const tp_u2 h = __builtin_nontemporal_load(reinterpret_cast<const tp_u2*>(static_cast<const uint16_t*>(base) + idx));
        return make_float4(cvt16<F16>((uint16_t)(h[0] & 0xffffu)), cvt16<F16>((uint16_t)(h[0] >> 16)), cvt16<F16>((uint16_t)(h[1] & 0xffffu)),
                           cvt16<F16>((uint16_t)(h[1] >> 16)));
#else
        const ushort4 h = *reinterpret_cast<const ushort4*>(static_cast<const uint16_t*>(base) + idx);
        return make_float4(cvt16<F16>(h.x), cvt16<F16>(h.y), cvt16<F16>(h.z), cvt16<F16>(h.w));
#endif
    }
#if TILE_NT & 1
    const tp_f4 t = __builtin_nontemporal_load(reinterpret_cast<const tp_f4*>(static_cast<const float*>(base) + idx));
    return make_float4(t[0], t[1], t[2], t[3]);
#else
    return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
#endif
}

// Guarded loads of the tile passes, branch-free: an out-of-range element reads the tensor's first element instead and is
// replaced by zero.  (A load inside `if (in_range)` cannot be hoisted by the compiler: every guarded load pair then sits in its
// own exec-masked block followed by s_waitcnt vmcnt(0), and a wave has two loads in flight instead of sixteen.)
template <bool BF, bool F16> __device__ __forceinline__ float4 act_load4z(const void* __restrict__ base, long idx, bool ok) {
    const float4 v = act_load4<BF, F16>(base, ok ? idx : 0);
    return make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
}

// ... with the per-column bias b added to the in-range elements: the 16-bit pre-activation tensor of the 16-bit modes holds the
// encoder product alone, rounded to the operand format, and the fp32 bias joins it here in fp32 — the reference's autocast
// order (clt.py:157-165: `pre_actvs + self.b.to(acts_norm.dtype)` after the 16-bit einsum)
template <bool BF, bool F16> __device__ __forceinline__ float4 act_load4zb(const void* __restrict__ base, long idx, bool ok, const float4 b) {
    const float4 v = act_load4<BF, F16>(base, ok ? idx : 0);
    return make_float4(ok ? v.x + b.x : 0.0f, ok ? v.y + b.y : 0.0f, ok ? v.z + b.z : 0.0f, ok ? v.w + b.w : 0.0f);
}

// grid (ceil(n/256), ceil(rows/64)); rows % 8 == 0, n % 4 == 0
template <bool XB, bool F16>
__global__ __launch_bounds__(256) void jumprelu_fwd_bf16_kernel(const void* __restrict__ x, const float* __restrict__ theta,
                                                                const float* __restrict__ bias, int rows,
                                                                int n, uint16_t* __restrict__ feat, uint16_t* __restrict__ featT,
                                                                uint8_t* __restrict__ fired) {
    __shared__ uint32_t tile[FT_R / 2][FT_PITCH];
    const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * FT_C;
    const int r0 = blockIdx.y * FT_R;
    const long col = c0 + 4 * tx;
    const bool cin = col < n;
    float4 th = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (theta && cin) th = *reinterpret_cast<const float4*>(theta + col);
    float4 bb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (bias && cin) bb = *reinterpret_cast<const float4*>(bias + col);
    bool any[4] = {false, false, false, false};
// (variants measured at B*L*F = 1.61 G, bf16 in, scripts/archive/jf_probe.py: 2 / 4 / 8 row pairs ahead 1966 / 1996 / 1987 us, 16-byte
// row stores by lane exchange 1997 us: the pass writes twice what it reads and sits at ~4.9 TB/s either way)
#ifndef JF_AHEAD
#define JF_AHEAD 2
#endif
#ifndef JF_WIDE
#define JF_WIDE 0
#endif
    for (int i0 = 0; i0 < 8; i0 += JF_AHEAD) {             // JF_AHEAD row pairs loaded ahead of their arithmetic
        float4 vv[2 * JF_AHEAD];
#pragma unroll
        for (int u = 0; u < 2 * JF_AHEAD; ++u) {
            const int row = r0 + 2 * (w + 4 * (i0 + (u >> 1))) + (u & 1);
            vv[u] = act_load4zb<XB, F16>(x, (long)row * n + col, cin && row < rows, bb);
        }
#pragma unroll
        for (int u = 0; u < JF_AHEAD; ++u) {
            const int pr = w + 4 * (i0 + u), row = r0 + 2 * pr;
            float f[2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 v = vv[2 * u + e];
                f[e][0] = (v.x > th.x && v.x > 0.0f) ? v.x : v.x * 0.0f; f[e][1] = (v.y > th.y && v.y > 0.0f) ? v.y : v.y * 0.0f;
                f[e][2] = (v.z > th.z && v.z > 0.0f) ? v.z : v.z * 0.0f; f[e][3] = (v.w > th.w && v.w > 0.0f) ? v.w : v.w * 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) any[k] |= f[e][k] > 0.0f;
                // (8-byte stores: trading halves with the neighbour lane for 16-byte ones, as weight_prep_kernel does, measured
                // 6 % slower here — this kernel is closer to instruction issue than to the store rate)
                if (!JF_WIDE && cin && row + e < rows)
                    *reinterpret_cast<uint2*>(feat + (long)(row + e) * n + col) = make_uint2(pack16<F16>(f[e][0], f[e][1]), pack16<F16>(f[e][2], f[e][3]));
            }
            if (JF_WIDE) store_rowpair16<F16>(feat, row, n, col, tx, f[0], f[1], cin && row < rows, (n & 7) == 0);
            if (featT) ft_store_pair<F16>(tile, pr, tx, f[0], f[1]);
        }
    }
    if (cin) {                                         // fired[] is zeroed by the host wrapper; equal-value stores may race
#pragma unroll
        for (int k = 0; k < 4; ++k) if (any[k]) fired[col + k] = 1;
    }
    // (featT == nullptr: the row-major copy alone, for a consumer that takes the features k-major.  A streaming form of that case —
    // no tile, 8 columns per lane, 16-byte loads and stores — measured the same 1.20 ms for 3.2 GB read + 3.2 GB written)
    if (!featT) return;
    __syncthreads();
    ft_write_transposed(tile, featT, c0, r0, rows, n);
}

// Per-step preparation of a dense-regime decoder weight tensor W [nb][rows][n] fp32 in ONE read: the bf16 copy (feature-
// gradient GEMM operand), the bf16 transposed copy [nb][n][rows] (forward GEMM operand: contraction index contiguous) and the
// squared row norms sq[b*rows + r] (+)= sum_j W^2 (dec_norms of the sparsity penalty, clt_lightning.py:438-448) — three
// separate passes over 7.85 GB of decoder weights otherwise.  grid (ceil(rows/64), nb); a block walks the column tiles of
// its 64 rows, so a row's norm is one wave's fixed-order sum.  rows % 8 == 0, n % 4 == 0.
// GATHER: output row r of group g is input row ridx[g * rows + r] of the group's src_rows input rows (< 0: a zero row) — the
// compacted TopK-aux pass preps the dead rows of the decoder matrices without a gathered fp32 copy in between.
// (occupancy hints, measured as variants — scripts/archive/gpu_waves_ab.sh: weight_prep at 2 / 3 / 4 waves per SIMD 617 / 795 / 1070 us
// per PLT decoder, the tighter register budgets spill; the JumpReLU loss pass at 3 / 4: PLT step 21.5 / 22.1 ms)
#ifndef WP_WAVES
#define WP_WAVES 2
#endif
#ifndef JL_WAVES
#define JL_WAVES 3
#endif
template <bool F16, bool GATHER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WP_WAVES))) void weight_prep_kernel(const float* __restrict__ W, int rows, int n, uint16_t* __restrict__ op,
                                                          uint16_t* __restrict__ opT, float* __restrict__ sq, int accumulate,
                                                          const int32_t* __restrict__ ridx = nullptr, int src_rows = 0) {
    __shared__ uint32_t tile[FT_R / 2][FT_PITCH];
    const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = blockIdx.x * FT_R;
    const long boff = (long)blockIdx.y * rows * n;
    W += GATHER ? (long)blockIdx.y * src_rows * n : boff;
    if (op) op += boff;
    if (opT) opT += boff;
    float acc[16];
    int srow[GATHER ? 16 : 1];                                  // GATHER: input row of each of this thread's 16 output rows (-1: none)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = 0.0f;
        if (GATHER) {
            const int row = r0 + 2 * (w + 4 * (i >> 1)) + (i & 1);
            srow[i] = row < rows ? ridx[(long)blockIdx.y * rows + row] : -1;
        }
    }
    for (long c0 = 0; c0 < n; c0 += FT_C) {
        const long col = c0 + 4 * tx;
        const bool cin = col < n;
        float4 vv[16];                                      // all 16 rows of this thread requested before any is used
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = r0 + 2 * (w + 4 * (i >> 1)) + (i & 1);
            const long sr = GATHER ? (long)srow[GATHER ? i : 0] : (long)row;
            vv[i] = ld4z(W, sr * n + col, cin && (GATHER ? sr >= 0 : row < rows));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                       // row pair w + 4 i: rows 2 (w + 4 i), + 1
            const int pr = w + 4 * i, row = r0 + 2 * pr;
            float f[2][4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 v = vv[2 * i + e];
                acc[2 * i + e] += sq4(v.x, v.y, v.z, v.w);       // (explicit FMAs: the same bits as row_sqnorm_kernel and adam_prep_kernel)
                f[e][0] = v.x; f[e][1] = v.y; f[e][2] = v.z; f[e][3] = v.w;
            }
            if (op) store_rowpair16<F16>(op, row, n, col, tx, f[0], f[1], cin && row < rows, (n & 7) == 0);
            ft_store_pair<F16>(tile, pr, tx, f[0], f[1]);
        }
        if (opT) {
            __syncthreads();
            ft_write_transposed(tile, opT, c0, r0, rows, n);
            __syncthreads();
        }
    }
    if (sq) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float s = wave_sum(acc[i]);
            const int row = r0 + 2 * (w + 4 * (i >> 1)) + (i & 1);
            if (tx == 0 && row < rows) { float* q = sq + (long)blockIdx.y * rows + row; *q = accumulate ? *q + s : s; }
        }
    }
}

// grid (ceil(n/256), ceil(rows/256)): a block walks 4 row tiles of its 256 columns, keeping the column partial sums in
// registers.  part: [3][gridDim.y][n] floats (dtheta, dnorm, dbias partials) then [nblocks][3] doubles.
template <bool XB, bool GB, bool TANH, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JL_WAVES))) void jumprelu_loss_bf16_kernel(const void* __restrict__ x, const float* __restrict__ theta,
                                                                 const float* __restrict__ bias, const float* __restrict__ sq, const void* __restrict__ dfm, float bw,
                                                                 float lamB, float c, float coefBL,
                                                                 uint16_t* __restrict__ dpreT, float* __restrict__ part,
                                                                 double* __restrict__ part_s, int rows, int n) {
    // (the kernel is bound by instruction issue, not by HBM: the element arithmetic is branch-free — out-of-range elements
    // are loaded as zeros and masked where a zero input would still contribute —, per-column factors are hoisted, tanh runs
    // on the hardware exp2 / rcp, and the loss sums of a lane's 64 elements of a row tile are taken in fp32 before they join
    // the fp64 totals)
    __shared__ uint32_t tile[FT_R / 2][FT_PITCH];
    __shared__ double sh[4];
    float (*red)[4][FT_C] = reinterpret_cast<float (*)[4][FT_C]>(&tile[0][0]);      // [3][4][256] floats: reuses the tile after the loop
    static_assert(sizeof(float) * 3 * 4 * FT_C <= sizeof(uint32_t) * (FT_R / 2) * FT_PITCH, "red fits the tile");
    const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * FT_C;
    const long col = c0 + 4 * tx;
    const bool cin = col < n;
    float th[4] = {0, 0, 0, 0}, nrm[4] = {0, 0, 0, 0};
    if (cin) {
        const float4 t4 = *reinterpret_cast<const float4*>(theta + col), s4 = *reinterpret_cast<const float4*>(sq + col);
        th[0] = t4.x; th[1] = t4.y; th[2] = t4.z; th[3] = t4.w;
        nrm[0] = sqrtf(s4.x); nrm[1] = sqrtf(s4.y); nrm[2] = sqrtf(s4.z); nrm[3] = sqrtf(s4.w);
    }
    float4 bb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (bias && cin) bb = *reinterpret_cast<const float4*>(bias + col);
    float thc[4], nc[4], nc2[4], lcn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        thc[k] = -(th[k] / bw);
        nc[k] = nrm[k] * c;                                  // weighted feature = f * nc
        nc2[k] = nc[k] * 2.885390081777927f;                 // 2 log2(e): exp(2 wv) = exp2(f * nc2)
        lcn[k] = lamB * c * nrm[k];                          // d(sparsity)/d(feature) = lcn * tanh'
    }
    const float hb = 0.5f * bw, ncoef = -coefBL;
    float a_th[4] = {0, 0, 0, 0}, a_dn[4] = {0, 0, 0, 0}, a_db[4] = {0, 0, 0, 0};
    double s_t = 0.0, s_p = 0.0, s_w = 0.0;
    for (int rt = 0; rt < 4; ++rt) {
        const int r0 = (blockIdx.y * 4 + rt) * FT_R;
        if (r0 >= rows) break;
        float f_t = 0.0f, f_p = 0.0f, f_w = 0.0f;
        for (int i0 = 0; i0 < 8; i0 += 2) {                  // two row pairs (four rows) loaded ahead of their arithmetic
            float4 v4[4], g4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + 2 * (w + 4 * (i0 + (u >> 1))) + (u & 1);
                v4[u] = act_load4zb<XB, F16>(x, (long)row * n + col, cin && row < rows, bb);
                g4[u] = act_load4z<GB, F16>(dfm, (long)row * n + col, cin && row < rows);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pr = w + 4 * (i0 + u);
                float dpv[2][4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float inm = (cin && r0 + 2 * pr + e < rows) ? 1.0f : 0.0f;
                    const float v[4] = {v4[2 * u + e].x, v4[2 * u + e].y, v4[2 * u + e].z, v4[2 * u + e].w};
                    const float gg[4] = {g4[2 * u + e].x, g4[2 * u + e].y, g4[2 * u + e].z, g4[2 * u + e].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float f = (v[k] > th[k] && v[k] > 0.0f) ? v[k] : 0.0f;
                        const float wv = f * nc[k];
                        float t = wv, dt = 1.0f;
                        if (TANH) {                          // tanh(wv), wv >= 0, = 1 - 2 / (exp(2 wv) + 1); about 1e-6 absolute
                            t = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(f * nc2[k]) + 1.0f), 1.0f);
                            dt = fmaf(-t, t, 1.0f);
                        }
                        f_t += t; f_w += wv; f_p += fmaxf(-v[k], 0.0f);
                        const float df = fmaf(dt, lcn[k], gg[k]) * inm;
                        const float dp = v[k] < 0.0f ? ncoef : df;
                        const float d = v[k] - th[k];        // rectangle window of the threshold's STE: -bw/2 < d <= bw/2
                        a_th[k] = fmaf((d > -hb && d <= hb) ? thc[k] : 0.0f, df, a_th[k]);
                        a_dn[k] = fmaf(dt, f, a_dn[k]);      // (times lamB * c after the loop)
                        a_db[k] += dp;
                        dpv[e][k] = dp;
                    }
                }
                ft_store_pair<F16>(tile, pr, tx, dpv[0], dpv[1]);
            }
        }
        s_t += (double)f_t; s_w += (double)f_w; s_p += (double)f_p;
        __syncthreads();
        ft_write_transposed(tile, dpreT, c0, r0, rows, n);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a_dn[k] *= lamB * c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][w][4 * tx + k] = a_th[k]; red[1][w][4 * tx + k] = a_dn[k]; red[2][w][4 * tx + k] = a_db[k]; }
    __syncthreads();
    {
        const int j = threadIdx.x;
        if (c0 + j < n) {
            const long stride = (long)gridDim.y * n;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                part[q * stride + (long)blockIdx.y * n + c0 + j] = ((red[q][0][j] + red[q][1][j]) + red[q][2][j]) + red[q][3][j];
        }
    }
    const long b = (long)blockIdx.y * gridDim.x + blockIdx.x;
    double s;
    s = block_sum_d(s_t, sh); if (threadIdx.x == 0) part_s[b * 3 + 0] = s;
    s = block_sum_d(s_p, sh); if (threadIdx.x == 0) part_s[b * 3 + 1] = s;
    s = block_sum_d(s_w, sh); if (threadIdx.x == 0) part_s[b * 3 + 2] = s;
}

// Backward of a selection's relu + scatter (topk.py:61-65: the gradient reaches the kept, positive entries only) for the dense
// compacted TopK-aux pass in the 16-bit modes: dpre = feat > 0 ? dfeat : 0, written ONCE as the transposed 16-bit operand
// [n][rows] of the encoder weight-gradient GEMM, with its column sums (the encoder bias gradient) — instead of a masked fp32
// tensor, a transposing cast of it and a column-sum pass over it.  grid (ceil(n/256), ceil(rows/256)); part: [gridDim.y][n].
template <bool F16>
__global__ __launch_bounds__(256) void mask_positive_t_kernel(const float* __restrict__ feat, const float* __restrict__ dfeat, int rows,
                                                              int n, uint16_t* __restrict__ dpreT, float* __restrict__ part) {
    __shared__ uint32_t tile[FT_R / 2][FT_PITCH];
    float (*red)[FT_C] = reinterpret_cast<float (*)[FT_C]>(&tile[0][0]);      // [4][256] floats: reuses the tile after the loop
    const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * FT_C;
    const long col = c0 + 4 * tx;
    const bool cin = col < n;
    float a_db[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int rt = 0; rt < 4; ++rt) {
        const int r0 = (blockIdx.y * 4 + rt) * FT_R;
        if (r0 >= rows) break;
        for (int i0 = 0; i0 < 8; i0 += 2) {                  // two row pairs (four rows) loaded ahead of their arithmetic
            float4 v4[4], g4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + 2 * (w + 4 * (i0 + (u >> 1))) + (u & 1);
                v4[u] = ld4z(feat, (long)row * n + col, cin && row < rows);
                g4[u] = ld4z(dfeat, (long)row * n + col, cin && row < rows);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float dpv[2][4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v[4] = {v4[2 * u + e].x, v4[2 * u + e].y, v4[2 * u + e].z, v4[2 * u + e].w};
                    const float gg[4] = {g4[2 * u + e].x, g4[2 * u + e].y, g4[2 * u + e].z, g4[2 * u + e].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float dp = v[k] > 0.0f ? gg[k] : 0.0f; a_db[k] += dp; dpv[e][k] = dp; }
                }
                ft_store_pair<F16>(tile, w + 4 * (i0 + u), tx, dpv[0], dpv[1]);
            }
        }
        __syncthreads();
        ft_write_transposed(tile, dpreT, c0, r0, rows, n);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[w][4 * tx + k] = a_db[k];
    __syncthreads();
    const int j = threadIdx.x;
    if (c0 + j < n) part[(long)blockIdx.y * n + c0 + j] = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
}

// ---- logging side outputs (clt_lightning.py:131-199): per-column sums over the batch, fp64 ----------------------
// out[j][0..5] = sum_r (yn - r)^2, sum yn, sum yn^2, sum (r*std + mean - y)^2, sum r, sum r^2   (yn = (y - mean)/std)
__global__ __launch_bounds__(256) void recon_metrics_kernel(const float* __restrict__ r, const float* __restrict__ y, long yrs,
                                                            const float* __restrict__ mean, const float* __restrict__ stdv, int rows,
                                                            int n, double* __restrict__ out) {
    __shared__ double sh[4][64][6];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, j = blockIdx.x * 64 + tx;
    double a[6] = {0, 0, 0, 0, 0, 0};
    if (j < n) {
        const float mu = mean[j], sd = stdv[j];
        for (int i = ty; i < rows; i += 4) {
            const float rv = r[(long)i * n + j], yv = y[(long)i * yrs + j];
            const float yn = (yv - mu) / sd;
            const float e = yn - rv, er = (rv * sd + mu) - yv;
            a[0] += (double)e * e; a[1] += (double)yn; a[2] += (double)yn * yn; a[3] += (double)er * er;
            a[4] += (double)rv; a[5] += (double)rv * rv;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) sh[ty][tx][q] = a[q];
    __syncthreads();
    if (ty == 0 && j < n)
#pragma unroll
        for (int q = 0; q < 6; ++q) out[(long)j * 6 + q] = ((sh[0][tx][q] + sh[1][tx][q]) + sh[2][tx][q]) + sh[3][tx][q];
}

// counts[j] = #{r : feat[r][j] > 0}
__global__ void count_positive_kernel(const float* __restrict__ f, int rows, int n, int32_t* __restrict__ counts) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int c = 0;
    for (int r = 0; r < rows; ++r) c += f[(long)r * n + j] > 0.0f ? 1 : 0;
    counts[j] = c;
}

// out[r*ld_out + j] = bias[j] + sum_s slabs[s*slab_stride + r*n + j]   (split-K partial results of the dense
// cross-layer decoder: one [rows][n] slab per source layer, summed in source-layer order)
__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nslab, long slab_stride, long total, int n, long ld_out,
                                 const float* __restrict__ bias, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / n; const int j = (int)(i - r * n);
        float acc = 0.0f;
        for (int s = 0; s < nslab; ++s) acc += slabs[(long)s * slab_stride + i];
        out[r * ld_out + j] = acc + (bias ? bias[j] : 0.0f);
    }
}

// ---- K13 --------------------------------------------------------------------------------------
#ifndef ADAM_VARIANT
#define ADAM_VARIANT 0x12
#endif
#define ADAM_UN (1 << (ADAM_VARIANT & 3))
template <bool F16>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n4, long n, float lr, float omb1, float b2, float omb2, float eps, float wd, int adamw,
                            float step_size, float inv_sqrt_bc2, uint16_t* __restrict__ sh, const float* __restrict__ rdn,
                            const float* __restrict__ rsq, int rlen) {
    // optional per-row gradient term g += (rdn[row] / sqrt(rsq[row])) * p, row = element / rlen: the gradient that reaches
    // the decoder weights through dec_norms (clt_lightning.py:438-446), folded in here instead of a pass of its own.
    // Loss scaling (clt_set_loss_scaling): gradients arrive multiplied by the loss scale and are unscaled here; a step
    // whose gradients overflowed is skipped as a whole (GradScaler.step)
    if (clt_skip_update()) return;
    const float unscale = clt_dev_state.grad_unscale;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= unscale;
        if (wd != 0.0f) { if (adamw) pp *= (1.0f - lr * wd); else gg += wd * pp; }
        // (explicit fused multiply-adds, the same sequence as the Adam inside row_pass_kernel: both regimes update alike)
        mm = __fmaf_rn(gg - mm, omb1, mm);
        vv = __fmaf_rn(vv, b2, (omb2 * gg) * gg);
        const float denom = __fmaf_rn(sqrtf(vv), inv_sqrt_bc2, eps);
        pp = __fmaf_rn(-step_size, mm / denom, pp);
    };
    const long stride = (long)gridDim.x * blockDim.x;
#ifndef ADAM_VARIANT
#define ADAM_VARIANT 0x12
#endif
    // vector part: block b owns the ADAM_UN consecutive 4 KB chunks [b * UN, (b + 1) * UN) of every stream (the launch covers
    // n4 exactly — contiguous chunks per block instead of a grid-stride walk: 4.85 -> 5.9 TB/s on 302 M parameters), all
    // loads of a thread issued before its arithmetic; the streams are touched once per step: non-temporal
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
#if ADAM_VARIANT & 0x10
    auto ald = [](const float* q, long idx) { const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(q) + idx); return make_float4(t[0], t[1], t[2], t[3]); };
    auto ast = [](float* q, long idx, float4 val) { const nt_f4 t = {val.x, val.y, val.z, val.w}; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(q) + idx); };
#else
    auto ald = [](const float* q, long idx) { return reinterpret_cast<const float4*>(q)[idx]; };
    auto ast = [](float* q, long idx, float4 val) { reinterpret_cast<float4*>(q)[idx] = val; };
#endif
    constexpr int UN = ADAM_UN;
    {
        float4 P[UN], M[UN], V[UN], G[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = ((long)blockIdx.x * UN + u) * blockDim.x + threadIdx.x;
            if (i < n4) { P[u] = ald(p, i); M[u] = ald(m, i); V[u] = ald(v, i); G[u] = ald(g, i); }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = ((long)blockIdx.x * UN + u) * blockDim.x + threadIdx.x;
            if (i >= n4) break;
            if (rdn) { const long row = i * 4 / rlen; const float cf = rdn[row] / sqrtf(rsq[row]); G[u].x = __fmaf_rn(cf, P[u].x, G[u].x); G[u].y = __fmaf_rn(cf, P[u].y, G[u].y); G[u].z = __fmaf_rn(cf, P[u].z, G[u].z); G[u].w = __fmaf_rn(cf, P[u].w, G[u].w); }
            upd(P[u].x, G[u].x, M[u].x, V[u].x); upd(P[u].y, G[u].y, M[u].y, V[u].y); upd(P[u].z, G[u].z, M[u].z, V[u].z); upd(P[u].w, G[u].w, M[u].w, V[u].w);
            ast(p, i, P[u]); ast(m, i, M[u]); ast(v, i, V[u]);
            if (sh) reinterpret_cast<uint2*>(sh)[i] = make_uint2(pack16<F16>(P[u].x, P[u].y), pack16<F16>(P[u].z, P[u].w));
        }
    }
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
        float P = p[i], M = m[i], V = v[i];
        float G = g[i];
        if (rdn) { const long row = i / rlen; G = __fmaf_rn(rdn[row] / sqrtf(rsq[row]), P, G); }
        upd(P, G, M, V);
        p[i] = P; m[i] = M; v[i] = V;
        if (sh) sh[i] = to16<F16>(P);
    }
}

// Adam and the dense regime's weight preparation in ONE pass over a decoder matrix W [nb][rows][n] (round 3): every element is
// updated as adam_kernel does (same FMA sequence, same row term, same loss-scaling rules) and the UPDATED values leave as the
// fp32 weight, the 16-bit copy op, the transposed 16-bit copy opT [nb][n][rows] and the squared row norms sq[b * rows + r]
// (overwritten, not accumulated) — what weight_prep_kernel would derive from the updated matrix at the next step, bit for
// bit (same tiles, same summation order), without reading the 4 B/parameter again.  A skipped step (overflow flag) writes
// nothing: op / opT / sq then still describe the unchanged weights.  grid (ceil(rows / 64), nb), rows % 8 == 0, n % 4 == 0.
#ifndef APK_PAIRS
#define APK_PAIRS 1
#endif
template <bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void adam_prep_kernel(float* __restrict__ W, const float* __restrict__ G,
                                                                 float* __restrict__ M, float* __restrict__ V, int rows, int n,
                                                                 uint16_t* __restrict__ op, uint16_t* __restrict__ opT, float* __restrict__ sq,
                                                                 float lr, float omb1, float b2, float omb2, float eps, float wd, int adamw,
                                                                 float step_size, float inv_sqrt_bc2, const float* __restrict__ rdn,
                                                                 const float* __restrict__ rsq) {
    if (clt_skip_update()) return;
    const float unscale = clt_dev_state.grad_unscale;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= unscale;
        if (wd != 0.0f) { if (adamw) pp *= (1.0f - lr * wd); else gg += wd * pp; }
        mm = __fmaf_rn(gg - mm, omb1, mm);
        vv = __fmaf_rn(vv, b2, (omb2 * gg) * gg);
        const float denom = __fmaf_rn(sqrtf(vv), inv_sqrt_bc2, eps);
        pp = __fmaf_rn(-step_size, mm / denom, pp);
    };
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    auto ast = [](float* q, uint32_t idx, const float v[4]) { const nt_f4 t = {v[0], v[1], v[2], v[3]}; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(q + idx)); };
    __shared__ uint32_t tile[FT_R / 2][FT_PITCH];
    const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = blockIdx.x * FT_R;
    const long boff = (long)blockIdx.y * rows * n;
    W += boff; G += boff; M += boff; V += boff;
    if (op) op += boff;
    if (opT) opT += boff;
    float acc[16], cf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = 0.0f;
        const int row = r0 + 2 * (w + 4 * (i >> 1)) + (i & 1);
        cf[i] = (rdn && row < rows) ? rdn[(long)blockIdx.y * rows + row] / sqrtf(rsq[(long)blockIdx.y * rows + row]) : 0.0f;
    }
    for (long c0 = 0; c0 < n; c0 += FT_C) {
        const long col = c0 + 4 * tx;
        const bool cin = col < n;
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += APK_PAIRS) {          // APK_PAIRS row pairs of the four streams in flight
            float4 P[2 * APK_PAIRS], Gv[2 * APK_PAIRS], Mv[2 * APK_PAIRS], Vv[2 * APK_PAIRS];
#pragma unroll
            for (int u = 0; u < 2 * APK_PAIRS; ++u) {
                // (one 32-bit element offset per row for all four streams: with 64-bit addresses per stream and row the
                // loop-invariant address arithmetic alone took 128 registers and the kernel spilled)
                const int row = r0 + 2 * (w + 4 * (i0 + (u >> 1))) + (u & 1);
                const bool ok = cin && row < rows;
                const uint32_t o = ok ? (uint32_t)row * (uint32_t)n + (uint32_t)col : 0u;
                const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                P[u] = *reinterpret_cast<const float4*>(W + o); Gv[u] = *reinterpret_cast<const float4*>(G + o);
                Mv[u] = *reinterpret_cast<const float4*>(M + o); Vv[u] = *reinterpret_cast<const float4*>(V + o);
                if (!ok) { P[u] = z; Gv[u] = z; Mv[u] = z; Vv[u] = z; }
            }
#pragma unroll
            for (int q = 0; q < APK_PAIRS; ++q) {
                const int i = i0 + q, pr = w + 4 * i, row = r0 + 2 * pr;
                float f[2][4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = 2 * q + e;
                    float pv[4] = {P[u].x, P[u].y, P[u].z, P[u].w}, gv[4] = {Gv[u].x, Gv[u].y, Gv[u].z, Gv[u].w};
                    float mv[4] = {Mv[u].x, Mv[u].y, Mv[u].z, Mv[u].w}, vv[4] = {Vv[u].x, Vv[u].y, Vv[u].z, Vv[u].w};
                    const float c = cf[2 * i + e];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { if (rdn) gv[k] = __fmaf_rn(c, pv[k], gv[k]); upd(pv[k], gv[k], mv[k], vv[k]); f[e][k] = pv[k]; }
                    if (cin && row + e < rows) {
                        const uint32_t o = (uint32_t)(row + e) * (uint32_t)n + (uint32_t)col;
                        ast(W, o, pv); ast(M, o, mv); ast(V, o, vv);
                    }
                    acc[2 * i + e] += sq4(pv[0], pv[1], pv[2], pv[3]);
                }
                if (op) store_rowpair16<F16>(op, row, n, col, tx, f[0], f[1], cin && row < rows, (n & 7) == 0);
                ft_store_pair<F16>(tile, pr, tx, f[0], f[1]);
            }
            asm volatile("" ::: "memory");                   // (the next rows' loads stay behind these rows' stores)
        }
        if (opT) {
            __syncthreads();
            ft_write_transposed(tile, opT, c0, r0, rows, n);
            __syncthreads();
        }
    }
    if (sq) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float s = wave_sum(acc[i]);
            const int row = r0 + 2 * (w + 4 * (i >> 1)) + (i & 1);
            if (tx == 0 && row < rows) sq[(long)blockIdx.y * rows + row] = s;
        }
    }
}
}  // namespace

CltState clt_host_state = {0, 1.0f, nullptr};
CLT_DEFINE_TU_STATE_SETTER(clt_internal_set_state_elementwise)

// =============================================== C ABI ===============================================
extern "C" {
const char* clt_arch(void) { return "gfx950"; }
int clt_abi_version(void) { return 20; }

int clt_set_loss_scaling(int fp16_operands, float loss_scale, int32_t* found_inf) {
    if (!(loss_scale > 0.0f) || (fp16_operands && !found_inf)) return CLT_ERR_SHAPE;
    // (work already queued read the old state: drain it first; the state changes at configuration time and when the loss
    // scale moves — every growth interval, or after an overflow)
    if (hipDeviceSynchronize() != hipSuccess) return CLT_ERR_LAUNCH;
    const CltState s = {fp16_operands ? 1 : 0, 1.0f / loss_scale, reinterpret_cast<int*>(found_inf)};
    int rc = clt_internal_set_state_elementwise(&s);
    if (rc == CLT_OK) rc = clt_internal_set_state_sparse(&s);
    if (rc == CLT_OK) rc = clt_internal_set_state_gemm_nt(&s);
    if (rc == CLT_OK) rc = clt_internal_set_state_gemm(&s);
    if (rc == CLT_OK) clt_host_state = s;
    return rc;
}
int clt_operand_format(void) { return clt_host_state.fp16; }

int clt_column_stats_f32(const float* x, long rs, int rows, int n, float* mean, float* stdv, void* st) {
    if (rows <= 0 || n <= 0 || rs < n) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(column_stats_kernel, dim3((n + 63) / 64), dim3(1024), 0, as_stream(st), x, rs, rows, n, mean, stdv);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_standardize_f32(const float* x, long rs, const float* mean, const float* stdv, int rows, int n, float* o32,
                        uint16_t* o16, void* st) {
    if (rows <= 0 || n <= 0 || rs < n) return CLT_ERR_SHAPE;
    const bool vec = n % 4 == 0 && rs % 4 == 0 && aligned16(x) && aligned16(mean) && aligned16(stdv) &&
                     (!o32 || aligned16(o32)) && (!o16 || (reinterpret_cast<uintptr_t>(o16) & 7) == 0);
    const long work = (long)rows * n / (vec ? 4 : 1);
    const int nb = blocks_for(work, TPB);
#define STD(V_, F_) hipLaunchKernelGGL((standardize_kernel<V_, F_>), dim3(nb), dim3(TPB), 0, as_stream(st), x, rs, mean, stdv, rows, n, o32, o16)
    if (clt_host_state.fp16) { if (vec) STD(true, true); else STD(false, true); }
    else { if (vec) STD(true, false); else STD(false, false); }
#undef STD
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_destandardize_f32(const float* r, const float* mean, const float* stdv, int rows, int n, float* out, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    const long total = (long)rows * n;
    hipLaunchKernelGGL(destandardize_kernel, dim3(blocks_for(total, TPB)), dim3(TPB), 0, as_stream(st), r, mean, stdv, total, n, out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_relu_fwd_f32(const float* x, float* o, long n, void* st) {
    if (n <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(blocks_for(n, TPB)), dim3(TPB), 0, as_stream(st), x, o, n);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_relu_bwd_f32(const float* x, const float* g, float* dx, long n, void* st) {
    if (n <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(n, TPB)), dim3(TPB), 0, as_stream(st), x, g, dx, n);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_mask_positive_f32(const float* f, const float* g, float* dx, long n, int accumulate, void* st) {
    if (n <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(mask_positive_kernel, dim3(blocks_for(n, TPB)), dim3(TPB), 0, as_stream(st), f, g, dx, n, accumulate);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_jumprelu_fwd_f32(const float* x, const float* theta, float* o, int rows, int n, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    const long total = (long)rows * n;
    hipLaunchKernelGGL(jumprelu_fwd_kernel, dim3(blocks_for(total, TPB)), dim3(TPB), 0, as_stream(st), x, theta, o, total, n);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_jumprelu_bwd_workspace_bytes(int rows, int n) { return (size_t)col_chunks(rows) * n * sizeof(float); }
int clt_jumprelu_bwd_f32(const float* x, const float* theta, float bw, const float* g, float* dx, float* dtheta,
                         int rows, int n, void* ws, size_t wsb, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    if (!ws || wsb < clt_jumprelu_bwd_workspace_bytes(rows, n)) return CLT_ERR_WORKSPACE;
    const int ch = col_chunks(rows), rpc = (rows + ch - 1) / ch;
    hipLaunchKernelGGL(jumprelu_bwd_kernel, dim3((n + TPB - 1) / TPB, ch), dim3(TPB), 0, as_stream(st), x, theta, bw, g,
                       dx, (float*)ws, rows, n, rpc);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, as_stream(st), (const float*)ws, ch, n, dtheta, 0);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_column_sum_workspace_bytes(int rows, int n) { return (size_t)col_chunks(rows) * n * sizeof(float); }
int clt_column_sum_f32(const float* x, int rows, int n, float* out, int accumulate, void* ws, size_t wsb, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    if (!ws || wsb < clt_column_sum_workspace_bytes(rows, n)) return CLT_ERR_WORKSPACE;
    const int ch = col_chunks(rows), rpc = (rows + ch - 1) / ch;
    hipLaunchKernelGGL(column_sum_kernel, dim3((n + TPB - 1) / TPB, ch), dim3(TPB), 0, as_stream(st), x, (float*)ws, rows, n, rpc);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, as_stream(st), (const float*)ws, ch, n, out, accumulate);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_reduce_workspace_bytes(long count) { return (size_t)blocks_for(count, TPB * 8, 2048) * sizeof(double); }
int clt_mse_f32(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n,
                float gs, float* g, float* ys_out, double* loss_sum, void* ws, size_t wsb, void* st) {
    if (rows <= 0 || n <= 0 || yrs < n) return CLT_ERR_SHAPE;
    const long total = (long)rows * n;
    if (!ws || wsb < clt_reduce_workspace_bytes(total)) return CLT_ERR_WORKSPACE;
    const int nb = blocks_for(total, TPB * 8, 2048);
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(TPB), 0, as_stream(st), r, y, yrs, mean, stdv, rows, n, gs, g, ys_out, (double*)ws);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, as_stream(st), (const double*)ws, nb, 1, loss_sum);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_mse_bf16_colsum_workspace_bytes(int rows, int n) {
    const size_t chunks = (size_t)(rows + MF_R - 1) / MF_R, ctiles = (size_t)(n + 255) / 256;
    return chunks * ctiles * sizeof(double) + chunks * (size_t)n * sizeof(float);
}
static int mse_bf16_colsum_impl(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n, float gs,
                                uint16_t* g16, float* colsum, int accumulate, double* loss_sum, void* ws, size_t wsb, void* st,
                                int layer_rows) {
    if (rows <= 0 || n <= 0 || yrs < n || !g16 || !colsum || !loss_sum) return CLT_ERR_SHAPE;
    if (layer_rows < 0 || (layer_rows > 0 && (layer_rows % MF_R || rows % layer_rows))) return CLT_ERR_SHAPE;
    if (n % 4 || yrs % 4 || !aligned16(r) || !aligned16(y) || !aligned16(mean) || !aligned16(stdv) || (reinterpret_cast<uintptr_t>(g16) & 7))
        return CLT_ERR_ALIGN;
    if (!ws || wsb < clt_mse_bf16_colsum_workspace_bytes(rows, n)) return CLT_ERR_WORKSPACE;
    const int chunks = (rows + MF_R - 1) / MF_R, ctiles = (n + 255) / 256;
    double* part_l = (double*)ws;
    float* part_c = (float*)(part_l + (size_t)chunks * ctiles);
    if (clt_host_state.fp16) hipLaunchKernelGGL((mse_bf16_colsum_kernel<true>), dim3(ctiles, chunks), dim3(256), 0, as_stream(st), r, y, yrs, mean, stdv, rows, n, gs, g16,
                       part_c, part_l, layer_rows);
    else hipLaunchKernelGGL((mse_bf16_colsum_kernel<false>), dim3(ctiles, chunks), dim3(256), 0, as_stream(st), r, y, yrs, mean, stdv, rows, n, gs, g16,
                       part_c, part_l, layer_rows);
    const int groups = layer_rows > 0 ? rows / layer_rows : 1;        // column sums per layer: colsum is [layers][n]
    hipLaunchKernelGGL(sum_chunks_kernel, dim3((n + 255) / 256 + 1, groups), dim3(256), 0, as_stream(st), (const float*)part_c, chunks / groups, n,
                       colsum, accumulate, (const double*)part_l, chunks * ctiles, loss_sum);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_mse_bf16_colsum_f32(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n, float gs,
                            uint16_t* g16, float* colsum, int accumulate, double* loss_sum, void* ws, size_t wsb, void* st) {
    return mse_bf16_colsum_impl(r, y, yrs, mean, stdv, rows, n, gs, g16, colsum, accumulate, loss_sum, ws, wsb, st, 0);
}
// The same pass over a layer-major slab: r / g16 are [layers][layer_rows][n] (rows = layers * layer_rows, layer_rows % 64 == 0),
// the target row of (layer, token) is y + token * y_row_stride + layer * n (the [B][2][L][d] batch), mean / std / colsum are
// [layers][n].  One launch per group of layers of the feature-sharded step's pipelined exchange.
int clt_mse_bf16_colsum_layers_f32(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n,
                                   int layer_rows, float gs, uint16_t* g16, float* colsum, int accumulate, double* loss_sum, void* ws,
                                   size_t wsb, void* st) {
    if (layer_rows <= 0) return CLT_ERR_SHAPE;
    return mse_bf16_colsum_impl(r, y, yrs, mean, stdv, rows, n, gs, g16, colsum, accumulate, loss_sum, ws, wsb, st, layer_rows);
}
// ---- clt_mse_bf16_colsum_noalloc_f32: the same pass for callers that cannot allocate ------------------------------------
// The partial sums of clt_mse_bf16_colsum_f32 (per-block loss partials + per-row-chunk column partials) live in a
// file-scope device array instead of a caller workspace.  `scratch` is the HOST-side stand-in of that array: its device
// address is resolved once per device with hipGetSymbolAddress (each GPU holds its own copy at its own address) and only
// that address is ever handed to a kernel.
constexpr size_t NOALLOC_SCRATCH_BYTES = 4u << 20;
constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static void* noalloc_scratch_ptr[NOALLOC_MAX_DEVICES];        // per device: device address of `scratch`, nullptr = not resolved yet
static std::mutex noalloc_mu;

static int noalloc_scratch(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) return CLT_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(noalloc_mu);
    if (!noalloc_scratch_ptr[dev]) {             // first call on this device (the device is current): resolve and keep
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return CLT_ERR_LAUNCH;
        noalloc_scratch_ptr[dev] = p;
    }
    *out = noalloc_scratch_ptr[dev];
    return CLT_OK;
}
size_t clt_mse_bf16_colsum_noalloc_max_bytes(void) { return NOALLOC_SCRATCH_BYTES; }
int clt_mse_bf16_colsum_noalloc_f32(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n,
                                    float gs, uint16_t* g16, float* colsum, int accumulate, double* loss_sum, void* st) {
    if (rows <= 0 || n <= 0 || yrs < n || !g16 || !colsum || !loss_sum) return CLT_ERR_SHAPE;
    if (n % 4 || yrs % 4 || !aligned16(r) || !aligned16(y) || !aligned16(mean) || !aligned16(stdv) || (reinterpret_cast<uintptr_t>(g16) & 7))
        return CLT_ERR_ALIGN;
    if (clt_mse_bf16_colsum_workspace_bytes(rows, n) > NOALLOC_SCRATCH_BYTES) return CLT_ERR_WORKSPACE;   // beyond the fixed scratch
    void* ptr = nullptr;
    const int rc = noalloc_scratch(&ptr);
    if (rc != CLT_OK) return rc;
    // Every partial the two reduction kernels read is written by the first kernel of THIS call (chunks * ctiles loss
    // partials, chunks * n column partials), so nothing has to start at zero and the previous call's contents are inert.
    return clt_mse_bf16_colsum_f32(r, y, yrs, mean, stdv, rows, n, gs, g16, colsum, accumulate, loss_sum, ptr, NOALLOC_SCRATCH_BYTES, st);
}
int clt_aux_mse_f32(const float* ar, const float* ys, const float* r, long total, float gs, float* ga, float* gm,
                    double* aux_sum, void* ws, size_t wsb, void* st) {
    if (total <= 0) return CLT_ERR_SHAPE;
    if (!ws || wsb < clt_reduce_workspace_bytes(total)) return CLT_ERR_WORKSPACE;
    const int nb = blocks_for(total, TPB * 8, 2048);
    hipLaunchKernelGGL(aux_mse_kernel, dim3(nb), dim3(TPB), 0, as_stream(st), ar, ys, r, total, gs, ga, gm, (double*)ws);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, as_stream(st), (const double*)ws, nb, 1, aux_sum);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_row_sqnorm_f32(const float* W, int rows, int d, float* sq, int accumulate, void* st) {
    if (rows <= 0 || d <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((rows + 3) / 4), dim3(TPB), 0, as_stream(st), W, rows, d, sq, accumulate);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_rowscale_add_f32(const float* W, const float* dnorm, const float* sq, float* dW, int rows, int d, void* st) {
    if (rows <= 0 || d <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(rowscale_add_kernel, dim3((rows + 3) / 4), dim3(TPB), 0, as_stream(st), W, dnorm, sq, dW, rows, d);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_jumprelu_loss_workspace_bytes(int rows, int n) {
    const int ch = col_chunks(rows);
    const size_t nblk = (size_t)((n + TPB - 1) / TPB) * ch;
    return align_up((size_t)2 * ch * n * sizeof(float), 16) + nblk * 3 * sizeof(double);
}
int clt_jumprelu_loss_bwd_f32(const float* x, const float* theta, const float* sq, const float* dfm, float bw,
                              float lamB, float c, int use_tanh, float coefBL, float* dpre, float* dtheta,
                              float* dnorm, double* sums, int rows, int n, void* ws, size_t wsb, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    if (!ws || wsb < clt_jumprelu_loss_workspace_bytes(rows, n)) return CLT_ERR_WORKSPACE;
    if (dfm && (!dpre || !dtheta || !dnorm)) return CLT_ERR_SHAPE;
    const int ch = col_chunks(rows), rpc = (rows + ch - 1) / ch, gx = (n + TPB - 1) / TPB;
    float* pth = (float*)ws;
    float* pdn = pth + (size_t)ch * n;
    double* ps = (double*)((char*)ws + align_up((size_t)2 * ch * n * sizeof(float), 16));
    hipLaunchKernelGGL(jumprelu_loss_kernel, dim3(gx, ch), dim3(TPB), 0, as_stream(st), x, theta, sq, dfm, bw, lamB, c,
                       use_tanh, coefBL, dpre, pth, pdn, ps, rows, n, rpc);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(3), dim3(256), 0, as_stream(st), (const double*)ps, gx * ch, 3, sums);
    if (dfm) {
        hipLaunchKernelGGL(reduce_cols_kernel, dim3(gx), dim3(TPB), 0, as_stream(st), (const float*)pth, ch, n, dtheta, 0);
        hipLaunchKernelGGL(reduce_cols_kernel, dim3(gx), dim3(TPB), 0, as_stream(st), (const float*)pdn, ch, n, dnorm, 0);
    }
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_jumprelu_fwd_bf16(const void* x, int x_bf16, const float* theta, const float* bias, int rows, int n, uint16_t* feat,
                          uint16_t* featT, uint8_t* fired, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    if (rows % 8 || n % 4 || !aligned16(x) || !aligned16(feat) || (featT && !aligned16(featT)) || (theta && !aligned16(theta)) ||
        (bias && !aligned16(bias))) return CLT_ERR_ALIGN;
    if (hipMemsetAsync(fired, 0, (size_t)n, as_stream(st)) != hipSuccess) return CLT_ERR_LAUNCH;
    const dim3 grid((n + FT_C - 1) / FT_C, (rows + FT_R - 1) / FT_R);
#define JF(XB_, F_) hipLaunchKernelGGL((jumprelu_fwd_bf16_kernel<XB_, F_>), grid, dim3(256), 0, as_stream(st), x, theta, bias, rows, n, feat, featT, fired)
    if (clt_host_state.fp16) { if (x_bf16) JF(true, true); else JF(false, true); }
    else if (x_bf16) JF(true, false);
    else JF(false, false);
#undef JF
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_jumprelu_loss_bf16_workspace_bytes(int rows, int n) {
    const size_t gy = (rows + 4 * FT_R - 1) / (4 * FT_R), gx = (n + FT_C - 1) / FT_C;
    return align_up(3 * gy * (size_t)n * sizeof(float), 16) + gx * gy * 3 * sizeof(double);
}
int clt_jumprelu_loss_bwd_bf16(const void* x, int x_bf16, const float* theta, const float* bias, const float* sq, const void* dfm, int dfm_bf16, float bw,
                               float lamB, float c, int use_tanh, float coefBL, uint16_t* dpreT, float* dbias, float* dtheta,
                               float* dnorm, double* sums, int rows, int n, void* ws, size_t wsb, void* st) {
    if (rows <= 0 || n <= 0 || !x || !theta || !sq || !dfm || !dpreT || !dbias || !dtheta || !dnorm || !sums) return CLT_ERR_SHAPE;
    if (rows % 8 || n % 4 || !aligned16(x) || !aligned16(theta) || !aligned16(sq) || !aligned16(dfm) || !aligned16(dpreT) ||
        (bias && !aligned16(bias))) return CLT_ERR_ALIGN;
    if (!ws || wsb < clt_jumprelu_loss_bf16_workspace_bytes(rows, n)) return CLT_ERR_WORKSPACE;
    const int gy = (rows + 4 * FT_R - 1) / (4 * FT_R), gx = (n + FT_C - 1) / FT_C, gc = (n + TPB - 1) / TPB;
    float* part = (float*)ws;
    double* ps = (double*)((char*)ws + align_up((size_t)3 * gy * n * sizeof(float), 16));
#define JL2(XB_, GB_, TH_, F_) hipLaunchKernelGGL((jumprelu_loss_bf16_kernel<XB_, GB_, TH_, F_>), dim3(gx, gy), dim3(256), 0, as_stream(st), x, theta, bias, sq, \
                                                 dfm, bw, lamB, c, coefBL, dpreT, part, ps, rows, n)
#define JL(XB_, GB_, F_) do { if (use_tanh) JL2(XB_, GB_, true, F_); else JL2(XB_, GB_, false, F_); } while (0)
#define JLF(F_) do { if (x_bf16) { if (dfm_bf16) JL(true, true, F_); else JL(true, false, F_); } \
                     else { if (dfm_bf16) JL(false, true, F_); else JL(false, false, F_); } } while (0)
    if (clt_host_state.fp16) JLF(true); else JLF(false);
#undef JLF
#undef JL2
#undef JL
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(3), dim3(256), 0, as_stream(st), (const double*)ps, gx * gy, 3, sums);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(gc), dim3(TPB), 0, as_stream(st), (const float*)part, gy, n, dtheta, 0);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(gc), dim3(TPB), 0, as_stream(st), (const float*)(part + (size_t)gy * n), gy, n, dnorm, 0);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(gc), dim3(TPB), 0, as_stream(st), (const float*)(part + (size_t)2 * gy * n), gy, n, dbias, 0);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
size_t clt_mask_positive_t_workspace_bytes(int rows, int n) {
    return (size_t)((rows + 4 * FT_R - 1) / (4 * FT_R)) * (size_t)n * sizeof(float);
}
int clt_mask_positive_t_f32(const float* feat, const float* dfeat, int rows, int n, uint16_t* dpreT, float* colsum, void* ws, size_t wsb,
                            void* st) {
    if (rows <= 0 || n <= 0 || !feat || !dfeat || !dpreT || !colsum) return CLT_ERR_SHAPE;
    if (rows % 8 || n % 4 || !aligned16(feat) || !aligned16(dfeat) || !aligned16(dpreT)) return CLT_ERR_ALIGN;
    if (!ws || wsb < clt_mask_positive_t_workspace_bytes(rows, n)) return CLT_ERR_WORKSPACE;
    const int gy = (rows + 4 * FT_R - 1) / (4 * FT_R), gx = (n + FT_C - 1) / FT_C, gc = (n + TPB - 1) / TPB;
    if (clt_host_state.fp16) hipLaunchKernelGGL((mask_positive_t_kernel<true>), dim3(gx, gy), dim3(256), 0, as_stream(st), feat, dfeat, rows, n, dpreT, (float*)ws);
    else hipLaunchKernelGGL((mask_positive_t_kernel<false>), dim3(gx, gy), dim3(256), 0, as_stream(st), feat, dfeat, rows, n, dpreT, (float*)ws);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(gc), dim3(TPB), 0, as_stream(st), (const float*)ws, gy, n, colsum, 0);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_recon_metrics_f32(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n,
                          double* out, void* st) {
    if (rows <= 0 || n <= 0 || yrs < n) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(recon_metrics_kernel, dim3((n + 63) / 64), dim3(256), 0, as_stream(st), r, y, yrs, mean, stdv, rows, n, out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_count_positive_f32(const float* feat, int rows, int n, int32_t* counts, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(count_positive_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, as_stream(st), feat, rows, n, counts);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_column_any_positive_f32(const float* f, int rows, int n, uint8_t* fired, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(column_any_positive_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, as_stream(st), f, rows, n, fired);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_dead_update(const uint8_t* fired, int n, long tokens, int64_t* la, long ttd, uint8_t* dead, int32_t* dead_count,
                    void* st) {
    if (n <= 0) return CLT_ERR_SHAPE;
    if (hipMemsetAsync(dead_count, 0, sizeof(int32_t), as_stream(st)) != hipSuccess) return CLT_ERR_LAUNCH;
    hipLaunchKernelGGL(dead_update_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, as_stream(st), fired, n, tokens, la, ttd, dead, dead_count);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_mask_dead_f32(const float* x, const uint8_t* dead, float* o, int rows, int n, void* st) {
    if (rows <= 0 || n <= 0) return CLT_ERR_SHAPE;
    const long total = (long)rows * n;
    hipLaunchKernelGGL(mask_dead_kernel, dim3(blocks_for(total, TPB)), dim3(TPB), 0, as_stream(st), x, dead, o, total, n);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_threshold_fwd_f32(const float* x, const float* thr, int per_layer, float* o, int B, int L, int F, void* st) {
    if (B <= 0 || L <= 0 || F <= 0) return CLT_ERR_SHAPE;
    const long total = (long)B * L * F;
    hipLaunchKernelGGL(threshold_fwd_kernel, dim3(blocks_for(total, TPB)), dim3(TPB), 0, as_stream(st), x, thr, per_layer, o, total, L, F);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_cast_bf16(const float* in, uint16_t* out, long n, void* st) {
    if (n <= 0) return CLT_ERR_SHAPE;
    const long n8 = (aligned16(in) && aligned16(out)) ? n / 8 : 0;
    const long nb = n8 > 0 ? (n8 + TPB - 1) / TPB : blocks_for(n, TPB);
    if (nb > 0x7fffffffL) return CLT_ERR_RANGE;
    if (clt_host_state.fp16) hipLaunchKernelGGL((cast_bf16_kernel<true>), dim3(nb), dim3(TPB), 0, as_stream(st), in, out, n8, n);
    else hipLaunchKernelGGL((cast_bf16_kernel<false>), dim3(nb), dim3(TPB), 0, as_stream(st), in, out, n8, n);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_weight_prep_bf16(const float* W, int nb, int rows, int n, uint16_t* op, uint16_t* opT, float* sq, int accumulate, void* st) {
    if (nb <= 0 || rows <= 0 || n <= 0 || !W) return CLT_ERR_SHAPE;
    if (rows % 8 || n % 4 || !aligned16(W) || (op && !aligned16(op)) || (opT && !aligned16(opT))) return CLT_ERR_ALIGN;
    if (clt_host_state.fp16) hipLaunchKernelGGL((weight_prep_kernel<true>), dim3((rows + FT_R - 1) / FT_R, nb), dim3(256), 0, as_stream(st), W, rows, n, op, opT, sq, accumulate, (const int32_t*)nullptr, 0);
    else hipLaunchKernelGGL((weight_prep_kernel<false>), dim3((rows + FT_R - 1) / FT_R, nb), dim3(256), 0, as_stream(st), W, rows, n, op, opT, sq, accumulate, (const int32_t*)nullptr, 0);
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}

int clt_weight_prep_gather_bf16(const float* W, const int32_t* ridx, int nb, int src_rows, int rows, int n, uint16_t* op, uint16_t* opT,
                                void* st) {
    if (nb <= 0 || rows <= 0 || src_rows <= 0 || n <= 0 || !W || !ridx) return CLT_ERR_SHAPE;
    if (rows % 8 || n % 4 || !aligned16(W) || (op && !aligned16(op)) || (opT && !aligned16(opT))) return CLT_ERR_ALIGN;
    const dim3 grid((rows + FT_R - 1) / FT_R, nb);
    if (clt_host_state.fp16) hipLaunchKernelGGL((weight_prep_kernel<true, true>), grid, dim3(256), 0, as_stream(st), W, rows, n, op, opT, (float*)nullptr, 0, ridx, src_rows);
    else hipLaunchKernelGGL((weight_prep_kernel<false, true>), grid, dim3(256), 0, as_stream(st), W, rows, n, op, opT, (float*)nullptr, 0, ridx, src_rows);
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}

int clt_cast_transpose_bf16(const float* in, long ldi, uint16_t* out, long ldo, int rows, int cols, int batch, long s_in,
                            long s_out, void* st) {
    if (rows <= 0 || cols <= 0 || batch <= 0 || ldi < cols || ldo < rows) return CLT_ERR_SHAPE;
    if (clt_host_state.fp16) hipLaunchKernelGGL((cast_transpose_kernel<true>), dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(TPB), 0, as_stream(st),
                       in, ldi, out, ldo, rows, cols, s_in, s_out);
    else hipLaunchKernelGGL((cast_transpose_kernel<false>), dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(TPB), 0, as_stream(st),
                       in, ldi, out, ldo, rows, cols, s_in, s_out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_sum_slabs_f32(const float* slabs, int nslab, long slab_stride, int rows, int n, long ld_out, const float* bias,
                      float* out, void* st) {
    if (nslab <= 0 || rows <= 0 || n <= 0 || ld_out < n || slab_stride < (long)rows * n) return CLT_ERR_SHAPE;
    const long total = (long)rows * n;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks_for(total, TPB)), dim3(TPB), 0, as_stream(st), slabs, nslab, slab_stride, total,
                       n, ld_out, bias, out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
int clt_adam_f32(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps, double wd,
                 int adamw, int step, uint16_t* shadow, const float* row_dnorm, const float* row_sq, int row_len, void* st) {
    if (n <= 0 || step < 1) return CLT_ERR_SHAPE;
    if (row_dnorm && (!row_sq || row_len <= 0 || n % row_len)) return CLT_ERR_SHAPE;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const float step_size = (float)(lr / bc1), isb2 = (float)(1.0 / sqrt(bc2));
    const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && (!shadow || (reinterpret_cast<uintptr_t>(shadow) & 7) == 0);
    const long n4 = (vec && (!row_dnorm || row_len % 4 == 0)) ? n / 4 : 0;
    // blocks cover the vector part exactly (ADAM_UN chunks of 256 float4 each); the scalar remainder (n - 4 n4 elements: all of
    // them when the layout is not vectorisable) is walked grid-stride by the same launch
    const long nbv = (n4 + (long)TPB * ADAM_UN - 1) / ((long)TPB * ADAM_UN);
    const long adam_blocks = nbv > 0 ? nbv : blocks_for(n, TPB, 16384);
    if (adam_blocks > 0x7fffffffL) return CLT_ERR_RANGE;
    if (clt_host_state.fp16) hipLaunchKernelGGL((adam_kernel<true>), dim3(adam_blocks), dim3(TPB), 0, as_stream(st), p, g, m, v, n4, n,
                       (float)lr, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, adamw, step_size, isb2, shadow, row_dnorm, row_sq, row_len);
    else hipLaunchKernelGGL((adam_kernel<false>), dim3(adam_blocks), dim3(TPB), 0, as_stream(st), p, g, m, v, n4, n,
                       (float)lr, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, adamw, step_size, isb2, shadow, row_dnorm, row_sq, row_len);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_adam_prep_f32(float* p, const float* g, float* m, float* v, int nb, int rows, int n, double lr, double b1, double b2, double eps,
                      double wd, int adamw, int step, uint16_t* op, uint16_t* opT, float* sq, const float* row_dnorm,
                      const float* row_sq, void* st) {
    if (nb <= 0 || rows <= 0 || n <= 0 || step < 1 || !p || !g || !m || !v) return CLT_ERR_SHAPE;
    if ((row_dnorm != nullptr) != (row_sq != nullptr)) return CLT_ERR_SHAPE;
    if (rows % 8 || n % 4 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (op && !aligned16(op)) || (opT && !aligned16(opT)))
        return CLT_ERR_ALIGN;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const float step_size = (float)(lr / bc1), isb2 = (float)(1.0 / sqrt(bc2));
    const dim3 grid((rows + FT_R - 1) / FT_R, nb);
    if (clt_host_state.fp16) hipLaunchKernelGGL((adam_prep_kernel<true>), grid, dim3(256), 0, as_stream(st), p, g, m, v, rows, n, op, opT, sq, (float)lr,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, adamw, step_size, isb2, row_dnorm, row_sq);
    else hipLaunchKernelGGL((adam_prep_kernel<false>), grid, dim3(256), 0, as_stream(st), p, g, m, v, rows, n, op, opT, sq, (float)lr,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, adamw, step_size, isb2, row_dnorm, row_sq);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
}  // extern "C"
