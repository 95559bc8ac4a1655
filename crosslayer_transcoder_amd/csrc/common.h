// Shared helpers for the gfx950 kernels of libclt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clt_hip.h"

#define CLT_LAUNCH_CHECK() \
    do { if (hipGetLastError() != hipSuccess) return CLT_ERR_LAUNCH; } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// Ablation switches read from the environment exist only in a `make DIAG=1` build; the shipped library ignores them.
#ifdef CLT_DIAG
#include <stdlib.h>
static inline bool diag_env(const char* name) { return getenv(name) != nullptr; }
#else
static inline bool diag_env(const char*) { return false; }
#endif

// fp32 -> bf16, round to nearest even, by gfx950's packed conversion (v_cvt_pk_bf16_f32): lo in bits 0..15, hi in 16..31
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 clt_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float clt_f32x2 __attribute__((ext_vector_type(2)));
    const clt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, clt_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)pack_bf16x2(f, 0.0f); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// ---- 16-bit operand format and loss-scaling state (clt_set_loss_scaling, reference: Lightning precision "16-mixed" =
// torch.autocast(float16) + GradScaler around clt_lightning.py:290-312) ------------------------------------------------
// Process-wide library state, like a BLAS math mode.  The host copy (clt_host_state) picks the MFMA instantiation of the
// GEMMs; the kernels that convert to / from the 16-bit operand format or apply Adam read the copies in constant memory of
// their own translation unit (CLT_DEFINE_TU_STATE_SETTER), on the device that was current when the state was set.
struct CltState {
    int fp16;             // 0: bf16 operands (default), 1: fp16 operands
    float grad_unscale;   // Adam multiplies every gradient by this (1 / loss scale; 1 when no scaling)
    int* inf_flag;        // device int: set to 1 by any fp32 -> fp16 conversion that yields inf / NaN; Adam kernels leave
                          // parameters and moments untouched when it is set (GradScaler's skipped step).  nullptr: off
};
extern CltState clt_host_state;
__attribute__((unused)) static __constant__ CltState clt_dev_state = {0, 1.0f, nullptr};
#define CLT_DEFINE_TU_STATE_SETTER(fn) \
    int fn(const CltState* s) { return hipMemcpyToSymbol(HIP_SYMBOL(clt_dev_state), s, sizeof(CltState)) == hipSuccess ? CLT_OK : CLT_ERR_LAUNCH; }

// fp32 -> fp16, round to nearest even; an inf / NaN result raises the loss scaler's flag
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    typedef _Float16 clt_f16x2 __attribute__((ext_vector_type(2)));
    typedef float clt_f32x2 __attribute__((ext_vector_type(2)));
    const clt_f32x2 v = {lo, hi};
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, clt_f16x2));
    if ((((u & 0x7c00u) == 0x7c00u) || ((u & 0x7c000000u) == 0x7c000000u)) && clt_dev_state.inf_flag) *clt_dev_state.inf_flag = 1;
    return u;
}
// the same without the flag store: the caller ORs `bad` over many conversions and raises the flag once (GEMM epilogues)
__device__ __forceinline__ uint32_t pack_f16x2_track(float lo, float hi, uint32_t& bad) {
    typedef _Float16 clt_f16x2 __attribute__((ext_vector_type(2)));
    typedef float clt_f32x2 __attribute__((ext_vector_type(2)));
    const clt_f32x2 v = {lo, hi};
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, clt_f16x2));
    bad |= (uint32_t)((u & 0x7c00u) == 0x7c00u) | (uint32_t)((u & 0x7c000000u) == 0x7c000000u);
    return u;
}
__device__ __forceinline__ void clt_raise_inf_flag(uint32_t bad) { if (bad && clt_dev_state.inf_flag) *clt_dev_state.inf_flag = 1; }
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// 16-bit operand conversions with the format as a template parameter (kernels are instantiated per format and picked at
// launch from clt_host_state.fp16: the hot loops must not pay a per-element select)
template <bool F16> __device__ __forceinline__ float cvt16(uint16_t h) { return F16 ? f16_to_f32(h) : bf16_to_f32(h); }
template <bool F16> __device__ __forceinline__ uint32_t pack16(float lo, float hi) { return F16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
template <bool F16> __device__ __forceinline__ uint16_t to16(float f) { return (uint16_t)pack16<F16>(f, 0.0f); }
// Round-trip through the 16-bit operand format (what a contraction RETURNS under torch.autocast: clt.py:157-165 hands the
// encoder's einsum result on as a 16-bit tensor and adds the fp32 bias afterwards)
template <bool F16> __device__ __forceinline__ void round16x2(float& a, float& b) {
    const uint32_t u = pack16<F16>(a, b);
    if (F16) { a = f16_to_f32((uint16_t)(u & 0xffffu)); b = f16_to_f32((uint16_t)(u >> 16)); }
    else { a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u); }
}
template <bool F16> __device__ __forceinline__ float round16(float a) { float z = 0.0f; round16x2<F16>(a, z); return a; }
// Adam kernels: true when the step is to be skipped (an overflow was seen while this step's gradients were formed)
__device__ __forceinline__ bool clt_skip_update() { return clt_dev_state.inf_flag && *clt_dev_state.inf_flag != 0; }

// Order-preserving key of an fp32 value (ascending key == ascending value; -0 folded onto +0).
__device__ __forceinline__ uint32_t f32_key(float f) {
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// 64-lane wave reductions.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// squared length of four consecutive row elements in ONE fixed operation order: shared by row_sqnorm_kernel and the
// row norms the Adam row pass leaves behind (row_pass_kernel's sq_out), so that both give the same bits for a row
__device__ __forceinline__ float sq4(float x, float y, float z, float w) {
    return __fmaf_rn(w, w, __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x))));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Block (256 threads = 4 waves) sum in double; result valid in thread 0.  `sh` has >= 4 doubles.
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}

// Internal (cross-file) entry points with a device-side gate; see select_lists.hip.
// (x16: feat holds 16-bit elements in the active operand format instead of fp32)
int clt_internal_sparse_build(const void* feat, int B, int L, int F, uint32_t cap, uint32_t* row_start, int32_t* ent_col,
                              float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val, int32_t* csc_ent,
                              uint8_t* fired, uint32_t* total, int32_t* overflow, void* ws, size_t wsb, hipStream_t st,
                              const int32_t* gate, const float* theta = nullptr, bool theta_gate = false, bool x16 = false,
                              const float* bias = nullptr, int phase = 0);
int clt_internal_topk_segmented(const float* x, float* feat, int B, int L, int F, int k, int per_layer, float* threshold,
                                float e, void* ws, size_t wsb, hipStream_t st, const int32_t* gate);
int clt_internal_topk_rows(const float* x, float* feat, int rows, int F, int k, int32_t* idx_out, float* val_out,
                           const uint8_t* col_mask, int L, uint32_t* row_cnt, uint32_t* col_cnt, hipStream_t st,
                           void* ws = nullptr, size_t wsb = 0);
int clt_internal_topk_merge_slots(const uint32_t* slot_key, const int32_t* slot_idx, const uint32_t* slot_cnt, int slots, int k, int F,
                                  int L, long rows, int32_t* idx_out, float* val_out, uint32_t* row_cnt, uint32_t* col_cnt,
                                  const int32_t* gate, hipStream_t st);
size_t clt_internal_row_bounds_workspace_bytes(long rows, int F, int j);
int clt_internal_row_bounds(const float* x, long rows, int F, int j, float* lb, void* ws, size_t wsb, hipStream_t st);
// scratch the list-output form of long rows wants (two streaming passes, select.hip); 0: that form does not apply
size_t clt_internal_topk_rows_workspace_bytes(long rows, int F, int k);
int clt_internal_gemm_nt_bf16(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, long lda, long ldb, long ldc,
                              int batch, long sA, long sB, long sC, const float* bias, long sBias, int accumulate,
                              hipStream_t st, const uint16_t* const* Bseg = nullptr, int kts = 0, const void* groups = nullptr, bool out16 = false,
                              int km = 0);
int clt_internal_set_state_elementwise(const CltState* s);
int clt_internal_set_state_sparse(const CltState* s);
int clt_internal_set_state_gemm_nt(const CltState* s);
int clt_internal_set_state_gemm(const CltState* s);
