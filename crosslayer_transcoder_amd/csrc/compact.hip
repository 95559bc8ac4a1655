// Column compaction for the TopK-aux ("dead feature") pass of the crosslayer-transcoder step on gfx950.
//
// The aux loss (reference model/clt_lightning.py:539-553) runs a SECOND decoder pass over the top k_aux = 512
// pre-activations among the dead features of every (token, layer) row.  With 10-20 % of the features dead that pass is
// far too dense for the activation-list kernels (512 entries per row, hundreds of tokens per feature) yet touches only
// the dead columns.  It is therefore run as a dense MFMA sub-problem on COMPACTED operands: the dead columns of the
// pre-activations and the dead rows of every decoder / encoder matrix are gathered into [.., n_pad, ..] tensors
// (n_pad = the largest per-layer dead count, rounded up to 64), the ordinary dense kernels run on those, and the compact
// weight gradients re-enter the Adam row pass through a row -> compact-row map.
#include <math.h>
#include "common.h"

namespace {
constexpr int CT = 1024;

// one block per layer.  idx == nullptr: counts only.
__global__ __launch_bounds__(CT) void compact_columns_kernel(const uint8_t* __restrict__ dead, int F, int n_pad,
                                                             int32_t* __restrict__ counts, int32_t* __restrict__ idx,
                                                             int32_t* __restrict__ map) {
    __shared__ int wsum[CT / 64];
    __shared__ int total;
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (F + CT - 1) / CT, i0 = min(F, tid * per), i1 = min(F, i0 + per);
    const uint8_t* dl = dead + (long)l * F;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += dl[i] ? 1 : 0;
    int incl = cnt;                                             // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (tid == 0) { int s = 0; for (int k = 0; k < CT / 64; ++k) { const int v = wsum[k]; wsum[k] = s; s += v; } total = s; }
    __syncthreads();
    if (tid == 0) counts[l] = total;
    if (!idx) return;
    int pos = wsum[w] + incl - cnt;
    for (int i = i0; i < i1; ++i) {
        if (dl[i]) {
            if (pos < n_pad) { idx[(long)l * n_pad + pos] = i; map[(long)l * F + i] = l * n_pad + pos; }
            else map[(long)l * F + i] = -1;
            ++pos;
        } else map[(long)l * F + i] = -1;
    }
    for (int j = total + tid; j < n_pad; j += CT) idx[(long)l * n_pad + j] = -1;
}

// out[r][j] = x[r][idx[l][j]] (l = r % L), `fill` where idx < 0
__global__ void gather_cols_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx, int L, int F, int n_pad, float fill,
                                   float* __restrict__ out) {
    const long r = blockIdx.x;
    const int l = (int)(r % L);
    const float* xr = x + r * F;
    const int32_t* il = idx + (long)l * n_pad;
    for (int j = threadIdx.x; j < n_pad; j += blockDim.x) {
        const int c = il[j];
        out[r * n_pad + j] = c >= 0 ? xr[c] : fill;
    }
}

// out[g][j][:] = W[g][idx[g % L... see host][j]][:]  — one wave per output row; zero rows where idx < 0
__global__ void gather_rows_kernel(const float* __restrict__ W, const int32_t* __restrict__ idx, int groups, int F, int n_pad, int d,
                                   int vec, float* __restrict__ out) {
    const long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (orow >= (long)groups * n_pad) return;
    const int lane = threadIdx.x & 63;
    const int gq = (int)(orow / n_pad);
    const int c = idx[orow];
    const float* src = W + ((long)gq * F + (c >= 0 ? c : 0)) * d;
    float* dst = out + orow * d;
    if (vec) {
        for (int o = lane * 4; o < d; o += 256) {
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (c >= 0) v = *reinterpret_cast<const float4*>(src + o);
            *reinterpret_cast<float4*>(dst + o) = v;
        }
    } else {
        for (int o = lane; o < d; o += 64) dst[o] = c >= 0 ? src[o] : 0.0f;
    }
}
}  // namespace

extern "C" {

int clt_compact_columns(const uint8_t* dead, int L, int F, int n_pad, int32_t* counts, int32_t* idx, int32_t* map, void* st) {
    if (L <= 0 || F <= 0 || !dead || !counts) return CLT_ERR_SHAPE;
    if (idx && (n_pad <= 0 || !map)) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(compact_columns_kernel, dim3(L), dim3(CT), 0, as_stream(st), dead, F, n_pad, counts, idx, map);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_gather_cols_f32(const float* x, const int32_t* idx, int rows, int L, int F, int n_pad, float fill, float* out, void* st) {
    if (rows <= 0 || L <= 0 || F <= 0 || n_pad <= 0 || rows % L) return CLT_ERR_SHAPE;
    hipLaunchKernelGGL(gather_cols_kernel, dim3(rows), dim3(256), 0, as_stream(st), x, idx, L, F, n_pad, fill, out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

int clt_gather_rows_f32(const float* W, const int32_t* idx, int groups, int F, int n_pad, int d, float* out, void* st) {
    if (groups <= 0 || F <= 0 || n_pad <= 0 || d <= 0) return CLT_ERR_SHAPE;
    const long orows = (long)groups * n_pad;
    if ((orows + 3) / 4 > 0x7fffffffL) return CLT_ERR_RANGE;
    const int vec = d % 4 == 0 && aligned16(W) && aligned16(out);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((orows + 3) / 4)), dim3(256), 0, as_stream(st), W, idx, groups, F, n_pad, d,
                       vec, out);
    CLT_LAUNCH_CHECK(); return CLT_OK;
}

}  // extern "C"
