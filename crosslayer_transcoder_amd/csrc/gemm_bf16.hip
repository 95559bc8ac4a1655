// bf16-operand / fp32-accumulate strided batched GEMM on the gfx950 matrix cores
// (v_mfma_f32_32x32x16_bf16): the throughput path for the encoder / decoder contractions
// (reference model/clt.py:157-165, 241-248, 334-343 under `precision: bf16-mixed`).
//
// Block tile 128x128x32, 256 threads = 4 waves as 2(M) x 2(N), 64x64 per wave (2x2 MFMA tiles).
// Both operand tiles live in LDS as [128 rows][32 k] bf16 with an 80-byte row pitch, so a lane's
// 8-element k-fragment is one aligned ds_read_b128.  Operands whose global layout is k-contiguous
// ("NT" form: transA = 0, transB = 1) are staged with 16-byte loads/stores; the other layouts are
// transposed on the way into LDS (functional, slower) — the host keeps hot operands in NT form.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 8;   // PITCH in bf16 elements (80 B)

union Pack8 { uint4 v; uint16_t h[8]; };

// 128(r) x 32(k) tile, 4096 elements = 512 groups of 8 -> 2 groups per thread.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(const uint16_t* __restrict__ P, long ld, int r0, int k0, int R, int K, int tid,
                                          Pack8 (&reg)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        int row, col, nrow, ncol;
        if (KC) { row = r0 + (idx >> 2); col = k0 + (idx & 3) * 8; nrow = R; ncol = K; }
        else    { row = k0 + (idx >> 4); col = r0 + (idx & 15) * 8; nrow = K; ncol = R; }
        int valid = (row < nrow) ? (ncol - col) : 0;
        valid = valid < 0 ? 0 : (valid > 8 ? 8 : valid);
        const uint16_t* src = P + (long)row * ld + col;
        if (VEC && valid == 8) reg[j].v = *reinterpret_cast<const uint4*>(src);
        else {
#pragma unroll
            for (int c = 0; c < 8; ++c) reg[j].h[c] = (c < valid) ? src[c] : (uint16_t)0;
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(uint16_t* __restrict__ S, int tid, const Pack8 (&reg)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        if (KC) {
            const int r = idx >> 2, kq = (idx & 3) * 8;
            *reinterpret_cast<uint4*>(&S[r * PITCH + kq]) = reg[j].v;
        } else {
            const int kr = idx >> 4, rq = (idx & 15) * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) S[(rq + c) * PITCH + kr] = reg[j].h[c];
        }
    }
}

// F16: the operands are fp16 instead of bf16 (clt_set_loss_scaling) — same tiles, the f16 MFMA of the same shape
template <bool TA, bool TB, bool VEC, bool F16>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                        float* __restrict__ C, int M, int N, int K, long lda, long ldb,
                                                        long ldc, long sA, long sB, long sC, const float* __restrict__ bias,
                                                        long sBias, int accumulate) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * BM * PITCH];
    uint16_t* As = lds;
    uint16_t* Bs = lds + BM * PITCH;
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

    Pack8 ra[2], rb[2];
    load_tile<!TA, VEC>(A, lda, m0, 0, M, K, tid, ra);
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
        for (int ks = 0; ks < BK; ks += 16) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 32 + li) * PITCH + ks + kh * 8]);
                b[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + i * 32 + li) * PITCH + ks + kh * 8]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    if constexpr (F16)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[mi]), __builtin_bit_cast(f16x8, b[ni]), acc[mi][ni], 0, 0, 0);
                    else
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }
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
                    float v = ((accumulate & CLT_GEMM_ROUND_ACC) ? round16<F16>(acc[mi][ni][r]) : acc[mi][ni][r]) + bv;
                    if (accumulate & CLT_GEMM_ACCUMULATE) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <bool VEC>
int launch(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int tA,
           int tB, int batch, long sA, long sB, long sC, const float* bias, long sBias, int acc, hipStream_t st) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch), block(256);
#define GO2(TA_, TB_, F_) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, VEC, F_>), grid, block, 0, st, A, B, C, M, N, K, lda, \
                                             ldb, ldc, sA, sB, sC, bias, sBias, acc)
#define GO(TA_, TB_) do { if (clt_host_state.fp16) GO2(TA_, TB_, true); else GO2(TA_, TB_, false); } while (0)
    if (!tA && !tB) GO(false, false);
    else if (!tA && tB) GO(false, true);
    else if (tA && !tB) GO(true, false);
    else GO(true, true);
#undef GO
#undef GO2
    CLT_LAUNCH_CHECK();
    return CLT_OK;
}
}  // namespace

CLT_DEFINE_TU_STATE_SETTER(clt_internal_set_state_gemm)

extern "C" int clt_gemm_bf16(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, long lda, long ldb,
                             long ldc, int transA, int transB, int batch, long sA, long sB, long sC,
                             const float* bias, long sBias, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return CLT_ERR_SHAPE;
    if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return CLT_ERR_SHAPE;
    if ((long)((M + BM - 1) / BM) > 65535 || batch > 65535) return CLT_ERR_RANGE;
    {   // the LDS-DMA / 16x16x32 fast path: k-contiguous operands (A as [M][K], B as [N][K]) staged as they lie, k-major ones
        // (transA: A as [K][M]; !transB: B as [K][N]) through the transposed LDS reads of gemm_nt_bf16.hip
        const int rc = clt_internal_gemm_nt_bf16(A, B, C, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, bias, sBias, accumulate,
                                                 as_stream(stream), nullptr, 0, nullptr, false, (transA ? 1 : 0) | (transB ? 0 : 2));
        if (rc <= 0) return rc;
    }
    const bool vec = aligned16(A) && aligned16(B) && (lda % 8 == 0) && (ldb % 8 == 0) && (sA % 8 == 0) && (sB % 8 == 0);
    return vec ? launch<true>(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, batch, sA, sB, sC, bias, sBias,
                              accumulate, as_stream(stream))
               : launch<false>(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, batch, sA, sB, sC, bias, sBias,
                               accumulate, as_stream(stream));
}
