/*
 * clt_hip.h — C ABI of libclt_hip.so: the MI355X (gfx950) kernels of the crosslayer-transcoder
 * training step.  This is the drop-in boundary: plain device pointers, sizes and a hipStream_t
 * (passed as void*); no torch types.  The reference has no FFI of its own on this path — it
 * reaches stock ATen kernels through einops/torch — so each entry point cites the reference
 * Python op sequence (file:line, relative to the reference repo root) that it replaces.
 *
 * Rules for every entry point
 *   - stream-ordered, asynchronous, no host synchronisation, no allocation (hipMalloc/hipFree):
 *     scratch comes in through `workspace` (size from the matching *_workspace_bytes query);
 *   - returns CLT_OK (0) or a negative CLT_ERR_* code; nothing is launched on error;
 *   - all tensors are dense row-major unless a leading dimension / row stride is given (in ELEMENTS);
 *   - float = IEEE fp32, bf16 = the upper 16 bits of fp32 (round-to-nearest-even), carried as uint16_t.
 */
#ifndef CLT_HIP_H
#define CLT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLT_OK 0
#define CLT_ERR_SHAPE (-1)      /* non-positive / inconsistent dimension                           */
#define CLT_ERR_K_GT_F (-2)     /* k > d_features: the host raises AssertionError (topk.py:56,80,113) */
#define CLT_ERR_WORKSPACE (-3)  /* workspace NULL or too small                                      */
#define CLT_ERR_ALIGN (-4)      /* pointer / stride alignment the kernel needs is not met           */
#define CLT_ERR_LAUNCH (-5)     /* hipGetLastError() != hipSuccess after the launch                 */
#define CLT_ERR_RANGE (-6)      /* element count exceeds the 32-bit index range of the kernel       */

/* Library / build identification ("gfx950"), and the ABI version checked by the Python loader. */
const char* clt_arch(void);
int clt_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * K1  standardizers — model/standardize.py:39-59, :80-108
 * ------------------------------------------------------------------------------------------- */
/* mean[n], std[n] (unbiased, clamped at 1e-8) over `rows` samples of x[r*row_stride + j]. */
int clt_column_stats_f32(const float* x, long row_stride, int rows, int n, float* mean, float* std,
                         void* stream);
/* out[r,j] = (x[r*row_stride+j] - mean[j]) / std[j]; out_f32 and/or out_bf16 may be NULL. */
int clt_standardize_f32(const float* x, long row_stride, const float* mean, const float* std, int rows,
                        int n, float* out_f32, uint16_t* out_bf16, void* stream);
/* out[r,j] = r_in[r,j]*std[j] + mean[j]  (standardize.py:96). */
int clt_destandardize_f32(const float* r_in, const float* mean, const float* std, int rows, int n,
                          float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2/K5/K6/K7  dense contractions — clt.py:157-165 (encoder einsum + bias), :334-343 (cross-layer
 * decoder loop), :241-248 (per-layer decoder) and their autograd (bmm backward).
 *
 * C[b] (M x N, ldc) (+)= op(A[b]) (M x K) * op(B[b]) (K x N) + bias[b][N]
 *   transA = 0: A stored [M][K] (lda >= K);   1: A stored [K][M] (lda >= M)
 *   transB = 0: B stored [K][N] (ldb >= N);   1: B stored [N][K] (ldb >= K)
 * `batch` problems at element strides sA/sB/sC/sBias.  bias may be NULL.  accumulate != 0: C += .
 * f32: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), the parity path (rtol 1e-5 vs the fp32 reference).
 * bf16, which kernel: shapes with M*N >= 256^2 and 8-element alignment go through the persistent LDS-DMA kernel
 *       (gemm_nt_bf16.hip) in every layout — k-contiguous operands (transA = 0, transB = 1) staged as they lie, k-major ones
 *       (transA = 1: A as [K][M]; transB = 0: B as [K][N]) through transposed LDS reads, same k order per element in all four
 *       forms (bit-identical results; the k-major forms run 5-15 % slower, profiles/r03_gemm_ab.md section 11); anything else
 *       through the generic tile kernel.
 * bf16: bf16 operands (uint16_t), fp32 accumulate and fp32 C  (v_mfma_f32_32x32x16_bf16).
 *       `accumulate` is a flag word there: bit 0 (CLT_GEMM_ACCUMULATE) C +=; bit 1 (CLT_GEMM_ROUND_ACC): the product is
 *       rounded to the 16-bit operand format BEFORE the fp32 bias is added — torch.autocast's order for the encoder
 *       (clt.py:157-165: the einsum returns a 16-bit tensor, `+ self.b.to(acts_norm.dtype)` promotes back to fp32), so the
 *       sparsifier thresholds / selects the values the reference's 16-bit run sees (pinned by tests/golden/ac_*.pt).
 * ------------------------------------------------------------------------------------------- */
#define CLT_GEMM_ACCUMULATE 1
#define CLT_GEMM_ROUND_ACC 2
int clt_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
                 long ldc, int transA, int transB, int batch, long sA, long sB, long sC,
                 const float* bias, long sBias, int accumulate, void* stream);
int clt_gemm_bf16(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, long lda, long ldb,
                  long ldc, int transA, int transB, int batch, long sA, long sB, long sC,
                  const float* bias, long sBias, int accumulate, void* stream);
/* out[r*ld_out + j] = bias[j] + sum_{s < nslab} slabs[s*slab_stride + r*n + j]: combines the per-source-layer
 * partial products of the dense cross-layer decoder (split-K over source layers, clt.py:334-343). bias may be NULL. */
int clt_sum_slabs_f32(const float* slabs, int nslab, long slab_stride, int rows, int n, long ld_out, const float* bias,
                      float* out, void* stream);
/* fp32 -> bf16 (RNE) copy, optionally transposing a [rows][cols] matrix (ld_in) to [cols][rows] (ld_out). */
int clt_cast_bf16(const float* in, uint16_t* out, long n, void* stream);
int clt_cast_transpose_bf16(const float* in, long ld_in, uint16_t* out, long ld_out, int rows, int cols,
                            int batch, long s_in, long s_out, void* stream);
/* One read of a dense-regime decoder weight tensor W [nb][rows][n] (rows % 8 == 0, n % 4 == 0, 16-byte aligned) for all
 * three things the step derives from it: op = bf16 copy, opT = bf16 [nb][n][rows] transposed copy (autocast's casts +
 * the transposes of the bmm operands, clt.py:334-343 under bf16-mixed), sq[b*rows + r] (+)= squared row norms (dec_norms,
 * clt_lightning.py:438-448).  op / opT / sq may each be NULL. */
int clt_weight_prep_bf16(const float* W, int nb, int rows, int n, uint16_t* op, uint16_t* opT, float* sq, int accumulate,
                         void* stream);
/* The same pass reading its rows through an index: output row r of group g = input row ridx[g*rows + r] of that group's
 * src_rows input rows (W: [nb][src_rows][n]; ridx < 0: a zero row).  The compacted TopK-aux pass (clt_lightning.py:539-553)
 * preps the dead rows of every decoder matrix without a gathered fp32 copy. */
int clt_weight_prep_gather_bf16(const float* W, const int32_t* ridx, int nb, int src_rows, int rows, int n, uint16_t* op,
                                uint16_t* opT, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  ReLU / JumpReLU — model/jumprelu.py:17-22, :31-48
 * x, out, g: [rows][n] (n = L*F); theta: [n]
 * ------------------------------------------------------------------------------------------- */
int clt_relu_fwd_f32(const float* x, float* out, long count, void* stream);
int clt_relu_bwd_f32(const float* x, const float* g, float* dx, long count, void* stream);
int clt_jumprelu_fwd_f32(const float* x, const float* theta, float* out, int rows, int n, void* stream);
/* dx = g where x >= 0 else 0;  dtheta[j] = sum_r -(theta/bw)*rect((x-theta)/bw)*g.  Deterministic
 * two-stage column reduction through `workspace`. */
size_t clt_jumprelu_bwd_workspace_bytes(int rows, int n);
int clt_jumprelu_bwd_f32(const float* x, const float* theta, float bandwidth, const float* g, float* dx,
                         float* dtheta, int rows, int n, void* workspace, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  TopK / BatchTopK / PerLayerBatchTopK selection — model/topk.py:53-63, :76-96, :109-133
 * Exact radix select on order-preserving fp32 keys; ties at the k-th value go to the LOWER index.
 * x: [B][L][F] contiguous fp32.  feat_out (dense, same shape) = relu(x) at selected positions, else 0.
 * ------------------------------------------------------------------------------------------- */
/* mode 0: k of F per (b,l) row.  idx_out/val_out: optional [B*L][k] compact copies (int32 feature id,
 * relu'd value; unused slots of a row hold idx = -1, val = 0 — only when k-th value <= 0). */
int clt_topk_rows_f32(const float* x, float* feat_out, int rows, int F, int k, int32_t* idx_out,
                      float* val_out, void* stream);
/* mode 1 (per_layer = 0): global top k*B*L of B*L*F.  mode 2 (per_layer = 1): per layer l, top k*B of
 * the B*F entries x[:, l, :].  threshold[S] (S = 1 or L) is updated in place:
 * thr <- (1-e)*thr + e*max(kth_value, 0); pass threshold = NULL to skip.  k == 0 -> zeros, no update. */
size_t clt_topk_segmented_workspace_bytes(int B, int L, int F, int per_layer);
int clt_topk_segmented_f32(const float* x, float* feat_out, int B, int L, int F, int k, int per_layer,
                           float* threshold, float ema_e, void* workspace, size_t ws_bytes, void* stream);
/* backward of scatter(relu(gather)): dx (+)= g * [feat > 0]. */
int clt_mask_positive_f32(const float* feat, const float* g, float* dx, long count, int accumulate,
                          void* stream);
/* The same backward for the dense compacted TopK-aux pass in the 16-bit modes (clt_lightning.py:539-553 under autocast):
 * dpre = feat > 0 ? dfeat : 0 written once as the TRANSPOSED 16-bit operand dpreT [n][rows] of the encoder weight-gradient
 * GEMM (the transposes autograd makes for that bmm), plus colsum[n] = its column sums (the encoder bias gradient).
 * feat / dfeat: fp32 [rows][n]; rows % 8 == 0, n % 4 == 0. */
size_t clt_mask_positive_t_workspace_bytes(int rows, int n);
int clt_mask_positive_t_f32(const float* feat, const float* dfeat, int rows, int n, uint16_t* dpreT, float* colsum,
                            void* workspace, size_t ws_bytes, void* stream);
/* inference path (topk.py:71-74, 104-107): out = x > max(thr[seg],0) ? x : 0, seg = 0 or layer. */
int clt_threshold_fwd_f32(const float* x, const float* threshold, int per_layer, float* out, int B, int L,
                          int F, void* stream);
/* where(dead[l,f], x, -inf)  (clt_lightning.py:544-545); dead: uint8 [n]. */
int clt_mask_dead_f32(const float* x, const uint8_t* dead, float* out, int rows, int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K8  reconstruction MSE + its gradient — clt_lightning.py:303-304, 434, 536
 * diff = r - (y - mean)/std ;  loss_sum[0] = sum diff^2 ;  g = grad_scale * diff  (grad_scale = 2/(B*L*d)).
 * y rows are strided (batch[:,1]).  g may be NULL.
 * ------------------------------------------------------------------------------------------- */
size_t clt_reduce_workspace_bytes(long count);
int clt_mse_f32(const float* r, const float* y, long y_row_stride, const float* mean, const float* std,
                int rows, int n, float grad_scale, float* g, float* ys_out, double* loss_sum,
                void* workspace, size_t ws_bytes, void* stream);
/* The same loss, with the gradient leaving only as what the list regime's backward reads in bf16 mode: g16 = bf16 rows of
 * 2/(B*L*d) * (r - standardize(y)) and colsum[n] (+)= their fp32 column sums (the decoder bias gradient) — mse + cast +
 * column sum in one pass, no fp32 gradient tensor.  n % 4 == 0, 16-byte aligned rows. */
size_t clt_mse_bf16_colsum_workspace_bytes(int rows, int n);
int clt_mse_bf16_colsum_f32(const float* r, const float* y, long y_row_stride, const float* mean, const float* std, int rows,
                            int n, float grad_scale, uint16_t* g16, float* colsum, int accumulate, double* loss_sum,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same pass over a layer-major slab (feature-sharded step, pipelined exchange: one launch per group of layers whose
 * all-reduce has completed): r / g16 are [layers][layer_rows][n] (rows = layers * layer_rows, layer_rows % 64 == 0), the target
 * row of (layer, token) is y + token * y_row_stride + layer * n (a [B][2][L][d] batch), mean / std / colsum are [layers][n]. */
int clt_mse_bf16_colsum_layers_f32(const float* r, const float* y, long y_row_stride, const float* mean, const float* std, int rows,
                                   int n, int layer_rows, float grad_scale, uint16_t* g16, float* colsum, int accumulate,
                                   double* loss_sum, void* workspace, size_t workspace_bytes, void* stream);
/* clt_mse_bf16_colsum_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference: same arithmetic and same results as the entry
 * point above, but the partial sums live in a fixed 4 MiB device array owned by the library (one per GPU, held until the
 * process ends) instead of a caller workspace; the call allocates nothing.  Returns CLT_ERR_WORKSPACE when
 * clt_mse_bf16_colsum_workspace_bytes(rows, n) exceeds clt_mse_bf16_colsum_noalloc_max_bytes().  The first call on a
 * device resolves the array's address (make it outside graph capture).  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (different streams) race. */
size_t clt_mse_bf16_colsum_noalloc_max_bytes(void);
int clt_mse_bf16_colsum_noalloc_f32(const float* r, const float* y, long y_row_stride, const float* mean, const float* std,
                                    int rows, int n, float grad_scale, uint16_t* g16, float* colsum, int accumulate,
                                    double* loss_sum, void* stream);
/* TopK aux loss — clt_lightning.py:548-553: e = aux_r - (ys - r); aux_sum = sum nan_to_num(e^2);
 * g_aux = grad_scale*e*[isfinite(e^2)];  g_main += g_aux (recons_err is not detached). */
int clt_aux_mse_f32(const float* aux_r, const float* ys, const float* r, long count, float grad_scale,
                    float* g_aux, float* g_main, double* aux_sum, void* workspace, size_t ws_bytes,
                    void* stream);
/* column sums: out[j] (+)= sum_r x[r][j]  (bias gradients). */
size_t clt_column_sum_workspace_bytes(int rows, int n);
int clt_column_sum_f32(const float* x, int rows, int n, float* out, int accumulate, void* workspace,
                       size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K9/K10  JumpReLU-module sparsity + pre-activation losses — clt_lightning.py:438-468
 * ------------------------------------------------------------------------------------------- */
/* sq[row] (+)= sum_d W[row][d]^2   (dec_norms accumulation, one call per W_l, rows = (l+1)*F). */
int clt_row_sqnorm_f32(const float* W, int rows, int d, float* sq, int accumulate, void* stream);
/* Fused elementwise pass of the JumpReLU training step, x = pre [rows][n], n = L*F:
 *   f = jumprelu(x, theta); w = f*sqrt(sq[j])*c; t = tanh(w) | w
 *   sums[0] = sum t ; sums[1] = sum relu(-x) ; sums[2] = sum w   (loss terms, as raw sums; overwritten)
 *   dfeat   = dfeat_main + lam_over_B * dt * c * norm           (dt = 1-t^2 | 1)
 *   dpre    = (x < 0 ? 0 : dfeat) - coef_over_BL*[x < 0]
 *   dtheta[j] = sum_r -(theta/bw)*rect((x-theta)/bw)*dfeat
 *   dnorm[j]  = sum_r lam_over_B * dt * c * f
 * dfeat_main may be NULL (loss terms only; dpre/dtheta/dnorm then skipped). */
size_t clt_jumprelu_loss_workspace_bytes(int rows, int n);
int clt_jumprelu_loss_bwd_f32(const float* x, const float* theta, const float* sq, const float* dfeat_main,
                              float bandwidth, float lam_over_B, float c, int use_tanh, float coef_over_BL,
                              float* dpre, float* dtheta, float* dnorm, double* sums, int rows, int n,
                              void* workspace, size_t ws_bytes, void* stream);
/* Dense bf16 regime, fused tile passes (jumprelu.py:31-48 + the operand casts/transposes that the reference's autocast
 * and autograd do implicitly).  x: [rows][n] fp32 (rows = tokens, n = layers*features); rows % 8 == 0, n % 4 == 0,
 * 16-byte aligned pointers (else CLT_ERR_ALIGN: use the unfused entry points).
 *   clt_jumprelu_fwd_bf16:  feat[r][j] = bf16(jumprelu(x, theta)) ; featT[j][r] = the same, transposed ;
 *                           fired[j] = any_r feat > 0 (K12).  theta may be NULL (ReLU).  No fp32 feature tensor.  featT may be
 *                           NULL (a weight-gradient GEMM that takes the features k-major needs no transposed copy).
 *   clt_jumprelu_loss_bwd_bf16: the math of clt_jumprelu_loss_bwd_f32, but d(loss)/d(pre) leaves only as the transposed
 *                           bf16 GEMM operand dpreT[j][r] plus its fp32 column sums dbias[j] (encoder bias gradient). */
/* x (and dfeat_main) may be held as bf16 (x_bf16 / dfeat_bf16 != 0): what clt_gemm_bf16_nt_out16 leaves.  bias (optional,
 * [n] fp32): added to x in fp32 before anything else — under the reference's autocast the encoder einsum returns a 16-bit
 * tensor and the fp32 bias is added to it afterwards (clt.py:157-165), so the 16-bit tensor holds the product alone. */
int clt_jumprelu_fwd_bf16(const void* x, int x_bf16, const float* theta, const float* bias, int rows, int n, uint16_t* feat,
                          uint16_t* featT, uint8_t* fired, void* stream);
size_t clt_jumprelu_loss_bf16_workspace_bytes(int rows, int n);
int clt_jumprelu_loss_bwd_bf16(const void* x, int x_bf16, const float* theta, const float* bias, const float* sq, const void* dfeat_main,
                               int dfeat_bf16, float bandwidth, float lam_over_B, float c, int use_tanh, float coef_over_BL,
                               uint16_t* dpreT, float* dbias, float* dtheta, float* dnorm, double* sums, int rows, int n,
                               void* workspace, size_t ws_bytes, void* stream);
/* dW[row][:] += (dnorm[row]/sqrt(sq[row])) * W[row][:]   (gradient through dec_norms). */
int clt_rowscale_add_f32(const float* W, const float* dnorm, const float* sq, float* dW, int rows, int d,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * K12  dead-feature bookkeeping — clt_lightning.py:282-288
 * fired[j] = any_r feat[r][j] > 0   (features.sum(0) > 0 on non-negative features)
 * last_active[j] = (fired[j] ? 0 : last_active[j]) + tokens ; dead[j] = last_active[j] > tokens_till_dead ;
 * dead_count[0] = sum_j dead[j]   (logged as metrics/dead_features = dead_count / n).
 * ------------------------------------------------------------------------------------------- */
/* Logging side outputs (clt_lightning.py:131-199 log_training_metrics, metrics/dead_features.py:40-42): what the
 * reference computes from [B,L,F] / [B,L,d] tensors copied to the host.
 *   clt_recon_metrics_f32: out[j][0..5] (fp64) = sum_r (yn-r)^2, sum yn, sum yn^2, sum (r*std+mean-y)^2, sum r, sum r^2
 *                          per column j of recons_norm r [rows][n]; yn = (y-mean)/std, y with row stride yrs.
 *   clt_count_positive_f32: counts[j] = #{r : feat[r][j] > 0}  (per-feature fire counts: L0 per layer, DeadFeatures). */
int clt_recon_metrics_f32(const float* r, const float* y, long yrs, const float* mean, const float* stdv, int rows, int n,
                          double* out, void* stream);
int clt_count_positive_f32(const float* feat, int rows, int n, int32_t* counts, void* stream);
int clt_column_any_positive_f32(const float* feat, int rows, int n, uint8_t* fired, void* stream);
int clt_dead_update(const uint8_t* fired, int n, long tokens, int64_t* last_active, long tokens_till_dead,
                    uint8_t* dead, int32_t* dead_count, void* stream);

/* 16-bit operand format and loss scaling — process-wide library state, like a BLAS math mode (reference: Lightning
 * `precision: "16-mixed"`, config/default.yaml:13 = torch.autocast(float16) + torch.amp.GradScaler around
 * clt_lightning.py:290-312).  fp16_operands = 1: every kernel that writes or reads a 16-bit GEMM operand (casts, shadows,
 * gathered sources, the MFMA GEMMs — all entry points named *_bf16 included) uses IEEE fp16 instead of bf16.
 * loss_scale: the factor the caller multiplied the loss gradient by; clt_adam_f32 and the Adam inside
 * clt_sparse_row_pass_f32 multiply gradients by 1 / loss_scale.  found_inf (device int32, required with fp16 operands):
 * set to 1 by any fp32 -> fp16 conversion whose result is inf or NaN; while it is set the Adam kernels leave parameters,
 * moments and shadows untouched (GradScaler's skipped step).  The caller clears it before a step and reads it after.
 * Synchronises the device; applies to the device current at the call.  Default: (0, 1.0, NULL). */
int clt_set_loss_scaling(int fp16_operands, float loss_scale, int32_t* found_inf);
int clt_operand_format(void);     /* 0 bf16, 1 fp16 */

/* ---------------------------------------------------------------------------------------------
 * K13  fused Adam / AdamW — clt_lightning.py:339-352 (torch.optim semantics, eps outside sqrt)
 * step counts from 1; hyper-parameters are doubles (torch derives 1-beta and the bias corrections in
 * double before rounding to fp32).  bf16_shadow (optional) receives the updated weights rounded to bf16.
 * row_dnorm / row_sq / row_len (optional): the gradient gets the per-row term (row_dnorm[r] / sqrt(row_sq[r])) * p
 * (r = element / row_len) before the update — what clt_rowscale_add_f32 would have added to g in a pass of its own
 * (the gradient through dec_norms, clt_lightning.py:438-446).
 * ------------------------------------------------------------------------------------------- */
int clt_adam_f32(float* p, const float* g, float* m, float* v, long count, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int adamw, int step, uint16_t* bf16_shadow,
                 const float* row_dnorm, const float* row_sq, int row_len, void* stream);
/* clt_adam_f32 and clt_weight_prep_bf16 in one pass over a decoder matrix p [nb][rows][n] of the dense 16-bit regime (no
 * counterpart in the reference: there torch.optim.Adam steps and the next forward's autocast casts / bmm transposes / W**2
 * read the weights again): the updated values leave as fp32 p, as the 16-bit copy op, the transposed 16-bit copy opT
 * [nb][n][rows] and the squared row norms sq[b * rows + r] (overwritten) — bit for bit what clt_weight_prep_bf16 derives from
 * the updated matrix.  row_dnorm / row_sq: the row term of clt_adam_f32, indexed b * rows + r.  A step skipped under loss
 * scaling writes nothing.  rows % 8 == 0, n % 4 == 0, 16-byte aligned (CLT_ERR_ALIGN otherwise). */
int clt_adam_prep_f32(float* p, const float* g, float* m, float* v, int nb, int rows, int n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int adamw, int step, uint16_t* op, uint16_t* opT, float* sq,
                      const float* row_dnorm, const float* row_sq, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse decoder path (K5b/K6 sparse, fused with K13) — replaces the dense bmm's of clt.py:334-343 and
 * their autograd when the sparsifier leaves few active features (TopK family; JumpReLU at low L0).
 *
 * Lists (built from the dense feature tensor feat [B][L][F], entries = positions with feat > 0):
 *   token-major CSR : row_start[B*L+1]; ent_col[cap] (= l*F+i), ent_val[cap]; rows ordered (b,l), entries by i
 *   feature-major CSC: col_start[L*F+1]; csc_tok[cap] (token b), csc_val[cap], csc_ent[cap] (CSR position),
 *                      entries of a column ordered by token
 *   fired[L*F] = column non-empty;  total[0] = number of entries;  overflow[0] = 1 if total > cap.
 * ------------------------------------------------------------------------------------------- */
size_t clt_sparse_build_workspace_bytes(int B, int L, int F);
int clt_sparse_build_f32(const float* feat, int B, int L, int F, uint32_t cap, uint32_t* row_start, int32_t* ent_col,
                         float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val, int32_t* csc_ent,
                         uint8_t* fired, uint32_t* total, int32_t* overflow, void* workspace, size_t ws_bytes,
                         void* stream);
/* The same lists straight from the pre-activations x [B][L][F] of a thresholding sparsifier — JumpReLU (theta [L*F],
 * jumprelu.py:33-37: entries = positions with x > theta and x > 0, value x) or ReLU (theta == NULL: x > 0) — without a
 * dense feature tensor.  When more than `cap` entries exist, overflow[0] = 1 and the lists are incomplete: `cap` is the
 * caller's break-even between the list regime and the dense GEMM regime (total[0] always holds the true count).
 * phase: 0 = the whole build; 1 = counts, row offsets and overflow[0] (final here: the host may fetch it while the lists are
 * filled); 2 = the token-major lists (row_start / ent_col / ent_val: all the decoder's gather forward reads); 3 = the feature-major
 * lists (col_start / csc_* / fired) from what phases 1-2 left in the lists and the workspace — a caller may queue phase 3 on a
 * second stream beside the forward. */
int clt_sparse_build_threshold_f32(const float* x, const float* theta, int B, int L, int F, uint32_t cap, uint32_t* row_start,
                                   int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val,
                                   int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow, void* workspace,
                                   size_t ws_bytes, int phase, void* stream);
/* The same from 16-bit pre-activations (the active operand format: what clt_gemm_bf16_nt_out16 leaves — under the reference's
 * autocast the encoder einsum returns a 16-bit tensor too).  bias (optional, [L*F] fp32, 16-byte aligned): the encoder bias,
 * added in fp32 to the widened value before the comparison (autocast's order, clt.py:165); entry values are widened x (+ bias).
 * F % 4 == 0. */
int clt_sparse_build_threshold_x16(const uint16_t* x, const float* theta, const float* bias, int B, int L, int F, uint32_t cap, uint32_t* row_start,
                                   int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val,
                                   int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow, void* workspace,
                                   size_t ws_bytes, int phase, void* stream);
/* Selection straight to lists (replaces topk.py:58-61, 84-91, 117-128 + the list build above when the dense
 * feature tensor is not needed).  mode 0: k of F per (b,l) row (k <= 2048); 1: global k*B*L; 2: per layer k*B.
 * x: [B][L][F] fp32 pre-activations; col_mask (optional, uint8 [L*F]): positions with mask 0 count as -inf
 * (where(dead, x, -inf), clt_lightning.py:544-545).  threshold[S] EMA as in clt_topk_segmented_f32 (modes 1/2).
 * Lists: exactly those clt_sparse_build_f32 would build from the dense selection result.  dense_scratch
 * ([B][L][F] floats, modes 1/2): scratch of the exhaustive fallback, which runs — chosen on the device, no host
 * sync — only when the sampled candidate set cannot contain the answer; used_fallback[0] (optional) reports it. */
size_t clt_select_lists_workspace_bytes(int B, int L, int F, int k, int mode);
int clt_select_lists_f32(const float* x, const uint8_t* col_mask, int B, int L, int F, int k, int mode, float* threshold,
                         float ema_e, uint32_t cap, uint32_t* row_start, int32_t* ent_col, float* ent_val,
                         uint32_t* col_start, int32_t* csc_tok, float* csc_val, int32_t* csc_ent, uint8_t* fired,
                         uint32_t* total, int32_t* overflow, float* dense_scratch, int32_t* used_fallback,
                         void* workspace, size_t ws_bytes, void* stream);
/* Feature-sharded form of the batch modes (1, 2) for one process per GPU: this rank holds columns
 * [f_off, f_off + F) of F_global features per layer; k counts per (b,l) row over ALL F_global features.  The
 * selection runs as stages stage_lo..stage_hi with a collective (issued by the host, torch.distributed / RCCL)
 * on a region of the workspace between stages:
 *   after 0: MIN  over ranks of the lower-bound keys   (word 0 of each 16-byte record at exchange offset 0, S records)
 *   after 1, 2, 3: SUM over ranks of the uint32 array at exchange offset 1 (length: info 2): histograms | counts
 *   after 4: ALL-GATHER of the per-rank tie blocks (offset 3, info 4 words each; rank r's block is block r)
 *   5: emit + list build; used_fallback[0] != 0 afterwards means the candidate set was insufficient on some rank
 *      (nothing was emitted): the host re-runs all stages with exhaustive = 1 (every element is a candidate).
 * exhaustive = 9 (1 | 8) is the form without tie lists, exact for ANY number of entries tied at the k-th value (the tie
 * blocks hold 4096 per shard and segment): stages 0..3 as above, then
 *   after 4, 5, 6, 7: SUM over ranks of the same uint32 array (256-bin histograms of a radix select on the GLOBAL flat
 *      index among the tied entries: the lowest indices win);  8: emit + list build.  The host runs it when the
 *      exhaustive = 1 attempt reports a tie-list overflow.
 * Replaces the DTensor plumbing of model/parallel.py, which cannot shard TopK modules at all (parallel.py:268-274). */
size_t clt_select_lists_sharded_workspace_bytes(int B, int L, int F, int k, int mode, int world, int exhaustive);
size_t clt_select_lists_exchange_info(int B, int L, int F, int k, int mode, int world, int exhaustive, int which);
int clt_select_lists_sharded_f32(int stage_lo, int stage_hi, int world, int rank, int f_off, int F_global, int exhaustive,
                                 const float* x, const uint8_t* col_mask, int B, int L, int F, int k, int mode,
                                 float* threshold, float ema_e, uint32_t cap, uint32_t* row_start, int32_t* ent_col,
                                 float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val, int32_t* csc_ent,
                                 uint8_t* fired, uint32_t* total, int32_t* overflow, int32_t* used_fallback,
                                 void* workspace, size_t ws_bytes, void* stream);
/* NT GEMM with a K-SEGMENTED B operand: C[M][N] (+)= sum_s A[:, s*kseg : (s+1)*kseg] * Bseg[s][N][kseg]^T (+ bias[N]).
 * Bseg: DEVICE array of nseg device pointers (bf16 matrices, row stride ldb, 16-byte aligned).  Replaces the chain of
 * accumulating bmm's of the cross-layer decoder's feature gradient (autograd of clt.py:334-343: d feat[:, l'] =
 * sum_{l >= l'} g[:, l] @ W_l[l']^T, one weight tensor per target layer): the contraction runs over (target layer,
 * d_acts) in one pass and the output is written once.  Returns 1 when the shape does not qualify (kseg % 64, alignment). */
int clt_gemm_bf16_nt_kseg(const uint16_t* A, const uint16_t* const* Bseg, float* C, int M, int N, int nseg, int kseg, long lda,
                          long ldb, long ldc, const float* bias, int accumulate, void* stream);
/* NT GEMMs whose result leaves in the 16-bit operand format (bf16; fp16 under clt_set_loss_scaling) (C: [M][N]; ldc, sC in
 * 16-bit elements; N % 8 == 0, ldc % 8 == 0): for the dense
 * bf16 regime's pre-activations (clt.py:157-165 under bf16 autocast: a bf16 tensor in the reference too) and feature
 * gradients, which are consumed once, elementwise.  Halves the bytes and the store instructions of the K = 768
 * contractions' epilogue, which bounds them.  Fast kernel only: returns 1 when the shape does not qualify. */
int clt_gemm_bf16_nt_out16(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K, long lda, long ldb, long ldc,
                           int batch, long sA, long sB, long sC, const float* bias, long sBias, void* stream);
int clt_gemm_bf16_nt_kseg_out16(const uint16_t* A, const uint16_t* const* Bseg, uint16_t* C, int M, int N, int nseg, int kseg,
                                long lda, long ldb, long ldc, void* stream);
/* Test / A-B hook (no counterpart in the reference): which form of the NT kernel the store-epilogue entry points launch —
 * 1: every wave issues operand DMA pieces and C stores (rounds 1-2); 2: waves 0-3 issue all operand DMA pieces, waves 4-7 all C
 * stores (their partner's accumulators handed over through LDS), so no wave waits for a store before it can see an operand
 * piece land; 0 (default): by shape, form 2 for 2048 <= K <= 8192 (profiles/r03_gemm_ab.md).  `reserved` is ignored.  Results
 * are bit-identical between the forms (same k order per element).  Returns the previous form. */
int clt_gemm_nt_variant(int variant, int reserved);
/* Test / probe hook (no counterpart in the reference; no product path sets it): > 0 = the persistent grid of every NT-kernel
 * launch made while it is set takes at most `blocks` workgroups (rounded down to a multiple of 8; one workgroup fills a CU),
 * leaving the other CUs to a kernel that runs beside it on another stream.  Used by scripts/archive/overlap_probe.py, which
 * measured what co-scheduling the next step's encoder GEMM with the optimizer's weight passes would gain (0.26 of 12.68 ms at
 * best: profiles/r03_overlap_probe.txt).  0 = one workgroup per CU (default); negative = query only.  Results do not depend on
 * it (same k order per element).  Process-wide host state.  Returns the previous value. */
int clt_gemm_nt_grid_limit(int blocks);
/* Grouped NT GEMM: ngroups problems of one shape in ONE persistent launch, C_g[M][N] = (A + aoff_g)[M][K] * B_g[N][K]^T.
 * groups: DEVICE array of records {int64 aoff (elements into A); const uint16_t* B; float* C}.  Used for the dense
 * cross-layer decoder forward (clt.py:334-343): all (target layer, source layer) pairs — 78 at L = 12 — fill the chip
 * together instead of 12 launches of 48..576 tiles.  kmajor: bit 0 = every A_g is stored [K][M] (row stride lda), bit 1 = every
 * B_g is stored [K][N] (row stride ldb) — the operand as it lies when the contraction runs over the outer index of a row-major
 * tensor (decoder forward from the [F][d] weights, weight gradients from [B][F] features and [B][d] gradients): no transposed
 * copy is made, the kernel transposes in its LDS reads.  Returns 1 when the shape does not qualify. */
int clt_gemm_bf16_nt_grouped(const uint16_t* A, const void* groups, int ngroups, int M, int N, int K, long lda, long ldb,
                             long ldc, int kmajor, void* stream);
/* Encoder GEMM fused with the candidate filter of the batch-mode selection (bf16 NT fast kernel only; returns 1 when
 * the shape does not qualify): C = A[M][K] * B[N][K]^T + bias is NOT stored; every element whose key is >= the lower
 * bound of its segment is appended to the candidate list of a selection workspace prepared by stage 0 of
 * clt_select_lists_ext_f32.  ldc / sC define the flat index batch*sC + row*ldc + col of an element.
 * ctl / cand_key / cand_idx / capS: workspace + clt_select_lists_exchange_info(..., flags = 2, which = 0 / 7 / 8 / 9).
 * flags: CLT_GEMM_ROUND_ACC as in clt_gemm_bf16 (the candidates are then the values the reference's autocast run selects from). */
int clt_gemm_bf16_select(const uint16_t* A, const uint16_t* B, int M, int N, int K, long lda, long ldb, long ldc, int batch,
                         long sA, long sB, long sC, const float* bias, long sBias, const uint32_t* ctl, uint32_t* cand_key,
                         uint32_t* cand_idx, uint32_t capS, int seg_per_batch, int flags, void* stream);
/* Batch-mode selection whose candidates come from clt_gemm_bf16_select.  The lower bound is margin * (the previous
 * step's exact k-th value, prev_T[S] as order-preserving keys: word 0 of the SegState records, exchange info 6).
 * Stage 0 prepares the workspace (then run the GEMM), stages 1..4 as clt_select_lists_sharded_f32 (workspace /
 * exchange queries with flags = 2); stage 5 merges the outcome over shards and PUBLISHES used_fallback[0], stage 6 emits
 * the selection and builds the token-major lists (row_start / ent_col / ent_val: all the decoder's gather forward reads),
 * stage 7 the feature-major ones (col_start / csc_* / fired) — so the host can fetch the flag (side stream) while stage 6
 * runs, and may queue stage 7 on a second stream beside the forward (it reads what stage 6 wrote and the workspace).
 * used_fallback[0] != 0: the attempt is void (bound too high, overflow, too many ties, previous value not positive):
 * nothing is emitted, the caller runs the ordinary path.
 * drift (optional, [2 S] floats the caller keeps from step to step, initially zeros: stage 0 tracks the relative
 * change of the k-th value in it and tightens the bound from `margin` (the floor) to 1 - 2.5 drift - 0.005, at most 0.98. */
int clt_select_lists_ext_f32(int stage_lo, int stage_hi, int world, int rank, int f_off, int F_global, const uint32_t* prev_T,
                             float margin, float* drift, int B, int L, int F, int k, int mode, float* threshold, float ema_e, uint32_t cap,
                             uint32_t* row_start, int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok,
                             float* csc_val, int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow,
                             int32_t* used_fallback, void* workspace, size_t ws_bytes, void* stream);
/* PerLayerTopK fused with the encoder GEMM (topk.py:53-63 on clt.py:157-165's result, which is never written): per-row lower
 * bounds from a sample of the row's own pre-activations, candidates filed per row by the GEMM epilogue, exact per-row
 * selection (ties: lowest index) on the candidates, lists as clt_select_lists_f32(mode 0) builds them — bit-identical lists
 * whenever the attempt holds.  One GPU, k <= 64, slots (candidate places per row) <= 1024.
 *   1. the caller runs the encoder GEMM on every stride-th feature (clt_gemm_bf16 with ldb * stride) into pre_s [B*L][Fs];
 *   2. clt_select_rows_ext_f32 stage 0: clears the counters; bound of a row = the j-th largest of its Fs sampled values when
 *      all j are positive, else 0;
 *   3. clt_gemm_bf16_select_rows: the full-width GEMM; element (layer l, token b, feature f) >= bound[b * L + l] goes into slot
 *      block b * L + l (row_lb / slot_cnt / slot_key / slot_idx = workspace + clt_select_rows_ext_info(..., 0 / 1 / 2 / 3));
 *   4. stage 1: used_fallback[0] = 1 when some row's block overflowed or holds fewer than k candidates under a positive bound
 *      (the sample's j-th largest was above the row's k-th value: probability C(k-1, j) / stride^j per row) — the attempt is
 *      void, stage 2 then emits empty lists and the caller reruns the unfused GEMM + selection; stage 2: selection + the
 *      token-major lists (what the decoder's gather forward reads); stage 3: the feature-major lists (col_start / csc_* / fired),
 *      which may be queued on a second stream beside the forward (it reads what stage 2 wrote and the workspace). */
size_t clt_select_rows_ext_workspace_bytes(int B, int L, int F, int k, int slots, int Fs, int j);
size_t clt_select_rows_ext_info(int B, int L, int F, int k, int slots, int Fs, int j, int which);
int clt_select_rows_ext_f32(int stage_lo, int stage_hi, const float* pre_s, int B, int L, int F, int k, int slots, int Fs, int j,
                            uint32_t cap, uint32_t* row_start, int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok,
                            float* csc_val, int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow,
                            int32_t* used_fallback, void* workspace, size_t ws_bytes, void* stream);
int clt_gemm_bf16_select_rows(const uint16_t* A, const uint16_t* B, int M, int N, int K, long lda, long ldb, int batch, long sA,
                              long sB, const float* bias, long sBias, const float* row_lb, uint32_t* slot_cnt, uint32_t* slot_key,
                              uint32_t* slot_idx, int slots, int flags, void* stream);
/* Feature-sharded per-row TopK (mode 0 under sharding; k <= 2048): local top-k rows, then the exact global k-th value
 * of every row by 8 rounds of 4-bit radix counts.  Stages stage_lo..stage_hi with host collectives in between:
 *   after stages 0..7: SUM over ranks of the uint32 array at exchange offset 0 (length info 1 = rows*16)
 *   after stage 8:     ALL-GATHER of the [world][rows] uint32 tie counts at exchange offset 2 (rank r fills block r)
 *   stage 9: keep/drop (ties in global feature order = rank order, then index order) + rank-local list build. */
size_t clt_select_rows_sharded_workspace_bytes(int B, int L, int F, int k, int world);
size_t clt_select_rows_sharded_exchange_info(int B, int L, int F, int k, int world, int which);
int clt_select_rows_sharded_f32(int stage_lo, int stage_hi, int world, int rank, int F_global, const float* x,
                                const uint8_t* col_mask, int B, int L, int F, int k, uint32_t cap, uint32_t* row_start,
                                int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val,
                                int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow, void* workspace,
                                size_t ws_bytes, void* stream);
/* out[b,l,:] = bias[l] + sum over entries (l',i,val) of token b with l' <= l (cross_layer) or l' == l of
 * val * Wptr[l][row,:], row = l'*F+i (cross_layer: W_l is [(l+1)F][d]) or i (per-layer: Wptr[l] = W + l*F*d).
 * Wptr: DEVICE array of L device pointers (to fp32 rows, or — w_bf16 != 0 — to the bf16 shadow copies that the
 * Adam row pass maintains: half the gathered bytes in bf16 mode).  out[b,l,:] starts at out + b*out_tok_stride +
 * l*out_layer_stride (elements; both 0 = the contiguous [B][L][d] layout): the feature-sharded step writes the partial
 * reconstruction layer-major ([L][B][d]) so that every layer's slab is one contiguous all-reduce.  Only the target layers
 * [layer_lo, layer_hi) are computed (0, 0 = all): that step launches the forward layer group by layer group. */
int clt_sparse_decoder_fwd_f32(const uint32_t* row_start, const int32_t* ent_col, const float* ent_val,
                               const float* const* Wptr, const float* bias, float* out, int B, int L, int F, int d,
                               int cross_layer, uint32_t cap, int w_bf16, long out_tok_stride, long out_layer_stride,
                               int layer_lo, int layer_hi, void* stream);
/* One streaming pass over `rows` weight rows W[r][0:d] (row r <-> CSC column col0 + r):
 *   grad[r,:]  = sum_{p in column} coef[p] * src[tok[p], layer, :]   (+ the same over the optional 2nd list)
 *   dot_out[slot*dot_stride + p] = src[tok[p], layer, :] . W[r,:]    (pre-update weights; dot_out2: the same for
 *                                   the 2nd list; slot = dot_slot, or layer - (col/F) when dot_slot < 0)
 *   colsum_out[col] = sum_p coef[p]
 *   grad_out != NULL: grad is written there and nothing else is touched; otherwise the Adam/AdamW update
 *   (same semantics as clt_adam_f32) is applied to W, m, v in place (+ optional bf16 shadow of W).
 * layer = src_layer, or (col0 + r) / F when src_layer < 0; src tensors are [B][L][d] fp32, or bf16 (uint16_t) when
 * src_bf16 != 0 (bf16 mode: halves the gather traffic; products still accumulate in fp32).
 * extra (optional, with extra_map): a dense gradient contribution for a subset of the rows — the compacted TopK-aux
 * pass below —: grad[r,:] += extra[extra_map[col0 + r]][0:d] and colsum_out[col] += extra_colsum[extra_map[col]]
 * wherever extra_map[...] >= 0.  row_dnorm / row_sq (optional, [L*F], indexed by column): grad[r,:] +=
 * (row_dnorm[col] / sqrt(row_sq[col])) * W[r,:], the gradient through dec_norms (clt_lightning.py:438-448).
 * sq_out (optional, [rows]; Adam form and d <= 1024 only, CLT_ERR_SHAPE otherwise): sq_out[r] = |W[r,:]|^2 of the
 * UPDATED row, bit-identical to clt_row_sqnorm_f32 on the updated matrix — the next step's dec_norms
 * (clt_lightning.py:440-444) without another read of the weights.
 * hot_list / hot_count / hot_cap / hot_min_len (optional; from clt_sparse_hot_columns on the SAME col_start with the same
 * min_len): rows whose list holds more than hot_min_len entries are left out by the common launch and processed by two more
 * (their lists in equal segments, a block per segment; then a wave per row that adds the segment sums and updates the row) —
 * real dictionaries are heavy-tailed: a feature active in 10 % of the tokens is a 400-entry list.  Applies to d <= 1024
 * with 16-byte-aligned operands (ignored otherwise: one wave, or CW waves per long row, as before).
 * The hot block: hot_list is int32 [2 * hot_cap + 2 * seg_cap] — [0, cap) the hot columns, [cap, 2 cap) per hot column its
 * first segment slot (-1: the table was full, the column is walked whole by one wave), then the segment table (seg_cap pairs:
 * hot index, segment) — and hot_count uint32 [2] = {hot columns, segments asked for}; hot_cap >= entries / (min_len + 1) + 1.
 * Every hot column is cut into segments of 64 entries.  With hot_partials (scratch of seg_cap * (ceil(d / 256) * 256 + 64)
 * floats — times nmat for the group form; not to be shared by launches that may run concurrently) every segment is walked by
 * a block of its own and the row's wave adds the segments' sums in segment order: equal work items whatever the column
 * lengths (a feature that fires in half of the tokens — 2000 entries — no longer keeps one block busy while the chip waits).
 * Without hot_partials a hot row is walked by one wave of the second launch.
 * hot_phase (clt_sparse_row_pass_f32; needs hot_list): 0 = all launches on `stream`; 1 = the common launch alone; 2 = the hot
 * rows' launches alone (nothing when the hot-row form does not apply to the shape: phase 1 then covers every row).  The two
 * phases touch disjoint rows and dot slots, so a caller may queue them on two streams and let the short, latency-bound hot
 * launches run under the common pass instead of behind it. */
int clt_sparse_hot_columns(const uint32_t* col_start, int ncols, int min_len, int32_t* hot_list, uint32_t hot_cap,
                           uint32_t seg_cap, uint32_t* hot_count, void* stream);
int clt_sparse_row_pass_f32(float* W, float* m, float* v, uint16_t* shadow, float* grad_out, int rows, int d, int F,
                            int L, int col0, int src_layer, const uint32_t* col_start, const int32_t* tok,
                            const float* coef, const float* src, const uint32_t* col_start2, const int32_t* tok2,
                            const float* coef2, const float* src2, float* dot_out, float* dot_out2, long dot_stride,
                            long dot_stride2, int dot_slot, float* colsum_out, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int adamw, int step, int src_bf16, const float* extra,
                            const int32_t* extra_map, const float* extra_colsum, const float* row_dnorm, const float* row_sq,
                            float* sq_out, const int32_t* hot_list, const uint32_t* hot_count, uint32_t hot_cap, int hot_min_len,
                            uint32_t hot_seg_cap, float* hot_partials, int hot_phase, void* stream);
/* Several weight matrices of one decoder in ONE launch of the same pass (Adam in place; no gradient output, second list,
 * compact contribution or row term): matrix i has rows[i] rows of d floats (row r <-> CSC column r), gathers from src[i]
 * (a [B][L][d] tensor, layer src_layer[i]; fp32, or 16-bit when src_bf16) and writes its dots to slot
 * src_layer[i] - r / F of dot_out.  W / m / v / shadow / rows / src_layer / src: HOST arrays of nmat entries (nmat <= 16).
 * For feature shards, whose per-matrix launches are too short to reach the HBM rate.  Returns 1 when the shape does not
 * qualify (d % 4, d > 1024, alignment): the caller then launches clt_sparse_row_pass_f32 per matrix. */
int clt_sparse_row_pass_group_f32(int nmat, float* const* W, float* const* m, float* const* v, uint16_t* const* shadow,
                                  const int* rows, const int* src_layer, const void* const* src, int d, int F, int L,
                                  const uint32_t* col_start, const int32_t* tok, const float* coef, float* dot_out, long dot_stride,
                                  double lr, double beta1, double beta2, double eps, double weight_decay, int adamw, int step,
                                  int src_bf16, const int32_t* hot_list, const uint32_t* hot_count, uint32_t hot_cap, int hot_min_len,
                                  uint32_t hot_seg_cap, float* hot_partials, void* stream);

/* Column compaction for the TopK-aux pass (clt_lightning.py:539-553: top k_aux among the DEAD features, second decoder
 * pass): the dead columns are gathered so that the pass runs as a dense sub-problem of n_pad columns per layer.
 *   clt_compact_columns: counts[l] = number of dead features of layer l; when idx != NULL also
 *       idx[l][j] = j-th dead feature of layer l (ascending; -1 beyond counts[l]; n_pad entries per layer) and
 *       map[l*F + i] = l*n_pad + position of feature i among layer l's dead features, or -1.
 *   clt_gather_cols_f32: out[r][j] = x[r][idx[r % L][j]] (x: [rows][F], rows = tokens*L), `fill` where idx < 0.
 *   clt_gather_rows_f32: out[g][j][0:d] = W[g][idx[g][j]][0:d] (W: [groups][F][d]; group g uses idx row g), 0 where idx < 0. */
int clt_compact_columns(const uint8_t* dead, int L, int F, int n_pad, int32_t* counts, int32_t* idx, int32_t* map, void* stream);
int clt_gather_cols_f32(const float* x, const int32_t* idx, int rows, int L, int F, int n_pad, float fill, float* out,
                        void* stream);
int clt_gather_rows_f32(const float* W, const int32_t* idx, int groups, int F, int n_pad, int d, float* out, void* stream);
/* dval[p] = sum_{s < (cross_layer ? L - col/F : 1)} dots[s*stride + p]  — d(loss)/d(feature) per CSC entry. */
int clt_sparse_reduce_dots_f32(const float* dots, long stride, const uint32_t* col_start, int L, int F,
                               int cross_layer, float* dval, void* stream);
/* out [B][ncols] = 0 except out[tok[p]][col] = val[p] (dense view of CSC-ordered values). */
int clt_sparse_scatter_dense_f32(const uint32_t* col_start, const int32_t* tok, const float* val, int B, int ncols,
                                 float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLT_HIP_H */
