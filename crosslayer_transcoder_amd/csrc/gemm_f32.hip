// Exact-fp32 strided/batched GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
// Replaces the aten::bmm calls behind einops.einsum in the reference's Encoder / Decoder /
// CrosslayerDecoder forward and their autograd (reference model/clt.py:157-165, 241-248, 334-343).
// The f32-input MFMA is bit-for-bit a k-ordered fmaf chain, so this path meets the fp32 parity bar
// (rtol 1e-5); the bf16 path (gemm_bf16.hip) is the throughput path.
//
// Block tile 128x128x16, 256 threads = 4 waves as 2(M) x 2(N); each wave owns 64x64 = 2x2 MFMA tiles
// (64 accumulator registers).  Operand tiles are staged k-major in LDS ([BK][128+4]) whatever the
// global layout, so every fragment read is a conflict-free ds_read_b32 of 32 consecutive words; the
// next tile's global loads are issued before the MFMAs of the current one.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BM = 128, BN = 128, BK = 16, LDSW = BM + 4;

// Load this thread's 8 elements of a 128(r) x 16(k) operand tile.
// KC = true : matrix stored [R][K] (k contiguous);  KC = false: stored [K][R] (r contiguous).
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int k0, int R, int K,
                                          int tid, float (&reg)[8]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        int row, col, nrow, ncol;      // row: strided index, col: first of 4 contiguous elements
        if (KC) { row = r0 + (idx >> 2); col = k0 + (idx & 3) * 4; nrow = R; ncol = K; }
        else    { row = k0 + (idx >> 5); col = r0 + (idx & 31) * 4; nrow = K; ncol = R; }
        int valid = (row < nrow) ? (ncol - col) : 0;
        valid = valid < 0 ? 0 : (valid > 4 ? 4 : valid);
        const float* src = P + (long)row * ld + col;
        if (VEC && valid == 4) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            reg[j * 4 + 0] = v.x; reg[j * 4 + 1] = v.y; reg[j * 4 + 2] = v.z; reg[j * 4 + 3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) reg[j * 4 + c] = (c < valid) ? src[c] : 0.0f;
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const float (&reg)[8]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        if (KC) {
            const int r = idx >> 2, kq = (idx & 3) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) S[(kq + c) * LDSW + r] = reg[j * 4 + c];
        } else {
            const int kr = idx >> 5, rq = (idx & 31) * 4;
            *reinterpret_cast<float4*>(&S[kr * LDSW + rq]) =
                make_float4(reg[j * 4 + 0], reg[j * 4 + 1], reg[j * 4 + 2], reg[j * 4 + 3]);
        }
    }
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K, long lda,
                                                       long ldb, long ldc, long sA, long sB, long sC,
                                                       const float* __restrict__ bias, long sBias, int accumulate) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDSW];
    float* As = lds;
    float* Bs = lds + BK * LDSW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long bz = blockIdx.z;
    A += bz * sA; B += bz * sB; C += bz * sC;
    if (bias) bias += bz * sBias;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    float ra[8], rb[8];
    // A tile: rows = M.  transA == 0 -> stored [M][K] (k contiguous).
    load_tile<!TA, VEC>(A, lda, m0, 0, M, K, tid, ra);
    // B tile: rows = N.  transB == 0 -> stored [K][N] (n contiguous).
    load_tile<TB, VEC>(B, ldb, n0, 0, N, K, tid, rb);
    const int nt = (K + BK - 1) / BK;
    for (int t = 0; t < nt; ++t) {
        store_tile<!TA>(As, tid, ra);
        store_tile<TB>(Bs, tid, rb);
        __syncthreads();
        if (t + 1 < nt) {
            load_tile<!TA, VEC>(A, lda, m0, (t + 1) * BK, M, K, tid, ra);
            load_tile<TB, VEC>(B, ldb, n0, (t + 1) * BK, N, K, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float* ar = As + (kk + kh) * LDSW + wm * 64 + li;
            const float* br = Bs + (kk + kh) * LDSW + wn * 64 + li;
            const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + li;
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) {
                    float* dst = C + (long)row * ldc + col;
                    float v = acc[mi][ni][r] + bv;
                    if (accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <bool VEC>
int launch(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int tA,
           int tB, int batch, long sA, long sB, long sC, const float* bias, long sBias, int acc, hipStream_t st) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch), block(256);
#define GO(TA_, TB_) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, VEC>), grid, block, 0, st, A, B, C, M, N, K, lda, \
                                        ldb, ldc, sA, sB, sC, bias, sBias, acc)
    if (!tA && !tB) GO(false, false);
    else if (!tA && tB) GO(false, true);
    else if (tA && !tB) GO(true, false);
    else GO(true, true);
#undef GO
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}
}  // namespace

extern "C" int clt_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
                            long ldc, int transA, int transB, int batch, long sA, long sB, long sC,
                            const float* bias, long sBias, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return CLT_ERR_SHAPE;
    if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return CLT_ERR_SHAPE;
    if ((long)((M + BM - 1) / BM) > 65535 || batch > 65535) return CLT_ERR_RANGE;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (sA % 4 == 0) &&
                     (sB % 4 == 0);
    return vec ? launch<true>(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, batch, sA, sB, sC, bias, sBias,
                              accumulate, as_stream(stream))
               : launch<false>(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, batch, sA, sB, sC, bias, sBias,
                               accumulate, as_stream(stream));
}
