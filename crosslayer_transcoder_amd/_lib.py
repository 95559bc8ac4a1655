"""ctypes binding of libclt_hip.so (the C ABI declared in include/clt_hip.h).

The product path has NO fallback: if the shared library is missing or does not export a declared
symbol, importing this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C crosslayer_transcoder_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclt_hip.so")
if os.environ.get("CLT_HIP_LIBRARY"):           # kernel-variant builds for A/B runs (scripts/build_variants.sh)
    LIB_PATH = os.environ["CLT_HIP_LIBRARY"]

ERRORS = {-1: "CLT_ERR_SHAPE", -2: "CLT_ERR_K_GT_F", -3: "CLT_ERR_WORKSPACE", -4: "CLT_ERR_ALIGN",
          -5: "CLT_ERR_LAUNCH", -6: "CLT_ERR_RANGE"}

P, I, L, F32, SZ = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/clt_hip.h one to one
SIGNATURES = {
    "clt_arch": (C.c_char_p, []),
    "clt_abi_version": (I, []),
    "clt_column_stats_f32": (I, [P, L, I, I, P, P, P]),
    "clt_standardize_f32": (I, [P, L, P, P, I, I, P, P, P]),
    "clt_destandardize_f32": (I, [P, P, P, I, I, P, P]),
    "clt_gemm_f32": (I, [P, P, P, I, I, I, L, L, L, I, I, I, L, L, L, P, L, I, P]),
    "clt_gemm_bf16": (I, [P, P, P, I, I, I, L, L, L, I, I, I, L, L, L, P, L, I, P]),
    "clt_sum_slabs_f32": (I, [P, I, L, I, I, L, P, P, P]),
    "clt_cast_bf16": (I, [P, P, L, P]),
    "clt_cast_transpose_bf16": (I, [P, L, P, L, I, I, I, L, L, P]),
    "clt_weight_prep_bf16": (I, [P, I, I, I, P, P, P, I, P]),
    "clt_weight_prep_gather_bf16": (I, [P, P, I, I, I, I, P, P, P]),
    "clt_relu_fwd_f32": (I, [P, P, L, P]),
    "clt_relu_bwd_f32": (I, [P, P, P, L, P]),
    "clt_jumprelu_fwd_f32": (I, [P, P, P, I, I, P]),
    "clt_jumprelu_bwd_workspace_bytes": (SZ, [I, I]),
    "clt_jumprelu_bwd_f32": (I, [P, P, F32, P, P, P, I, I, P, SZ, P]),
    "clt_topk_rows_f32": (I, [P, P, I, I, I, P, P, P]),
    "clt_topk_segmented_workspace_bytes": (SZ, [I, I, I, I]),
    "clt_topk_segmented_f32": (I, [P, P, I, I, I, I, I, P, F32, P, SZ, P]),
    "clt_mask_positive_f32": (I, [P, P, P, L, I, P]),
    "clt_mask_positive_t_workspace_bytes": (SZ, [I, I]),
    "clt_mask_positive_t_f32": (I, [P, P, I, I, P, P, P, SZ, P]),
    "clt_threshold_fwd_f32": (I, [P, P, I, P, I, I, I, P]),
    "clt_mask_dead_f32": (I, [P, P, P, I, I, P]),
    "clt_reduce_workspace_bytes": (SZ, [L]),
    "clt_mse_f32": (I, [P, P, L, P, P, I, I, F32, P, P, P, P, SZ, P]),
    "clt_mse_bf16_colsum_workspace_bytes": (SZ, [I, I]),
    "clt_mse_bf16_colsum_f32": (I, [P, P, L, P, P, I, I, F32, P, P, I, P, P, SZ, P]),
    "clt_mse_bf16_colsum_layers_f32": (I, [P, P, L, P, P, I, I, I, F32, P, P, I, P, P, SZ, P]),
    "clt_mse_bf16_colsum_noalloc_max_bytes": (SZ, []),
    "clt_mse_bf16_colsum_noalloc_f32": (I, [P, P, L, P, P, I, I, F32, P, P, I, P, P]),
    "clt_aux_mse_f32": (I, [P, P, P, L, F32, P, P, P, P, SZ, P]),
    "clt_column_sum_workspace_bytes": (SZ, [I, I]),
    "clt_column_sum_f32": (I, [P, I, I, P, I, P, SZ, P]),
    "clt_row_sqnorm_f32": (I, [P, I, I, P, I, P]),
    "clt_jumprelu_loss_workspace_bytes": (SZ, [I, I]),
    "clt_jumprelu_loss_bwd_f32": (I, [P, P, P, P, F32, F32, F32, I, F32, P, P, P, P, I, I, P, SZ, P]),
    "clt_jumprelu_fwd_bf16": (I, [P, I, P, P, I, I, P, P, P, P]),
    "clt_jumprelu_loss_bf16_workspace_bytes": (SZ, [I, I]),
    "clt_jumprelu_loss_bwd_bf16": (I, [P, I, P, P, P, P, I, F32, F32, F32, I, F32, P, P, P, P, P, I, I, P, SZ, P]),
    "clt_rowscale_add_f32": (I, [P, P, P, P, I, I, P]),
    "clt_recon_metrics_f32": (I, [P, P, L, P, P, I, I, P, P]),
    "clt_count_positive_f32": (I, [P, I, I, P, P]),
    "clt_column_any_positive_f32": (I, [P, I, I, P, P]),
    "clt_dead_update": (I, [P, I, L, P, L, P, P, P]),
    "clt_sparse_build_workspace_bytes": (SZ, [I, I, I]),
    "clt_sparse_build_f32": (I, [P, I, I, I, C.c_uint32, P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    "clt_sparse_build_threshold_f32": (I, [P, P, I, I, I, C.c_uint32, P, P, P, P, P, P, P, P, P, P, P, SZ, I, P]),
    "clt_sparse_build_threshold_x16": (I, [P, P, P, I, I, I, C.c_uint32, P, P, P, P, P, P, P, P, P, P, P, SZ, I, P]),
    "clt_select_lists_workspace_bytes": (SZ, [I, I, I, I, I]),
    "clt_select_lists_f32": (I, [P, P, I, I, I, I, I, P, F32, C.c_uint32, P, P, P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    "clt_select_lists_sharded_workspace_bytes": (SZ, [I, I, I, I, I, I, I]),
    "clt_select_lists_exchange_info": (SZ, [I, I, I, I, I, I, I, I]),
    "clt_select_lists_sharded_f32": (I, [I, I, I, I, I, I, I, P, P, I, I, I, I, I, P, F32, C.c_uint32, P, P, P, P, P, P, P, P,
                                         P, P, P, P, SZ, P]),
    "clt_select_rows_sharded_workspace_bytes": (SZ, [I, I, I, I, I]),
    "clt_select_rows_sharded_exchange_info": (SZ, [I, I, I, I, I, I]),
    "clt_select_rows_sharded_f32": (I, [I, I, I, I, I, P, P, I, I, I, I, C.c_uint32, P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    "clt_gemm_bf16_nt_kseg": (I, [P, P, P, I, I, I, I, L, L, L, P, I, P]),
    "clt_set_loss_scaling": (I, [I, F32, P]),
    "clt_operand_format": (I, []),
    "clt_gemm_bf16_nt_out16": (I, [P, P, P, I, I, I, L, L, L, I, L, L, L, P, L, P]),
    "clt_gemm_bf16_nt_kseg_out16": (I, [P, P, P, I, I, I, I, L, L, L, P]),
    "clt_gemm_nt_variant": (I, [I, I]),
    "clt_gemm_nt_grid_limit": (I, [I]),
    "clt_gemm_bf16_nt_grouped": (I, [P, P, I, I, I, I, L, L, L, I, P]),
    "clt_gemm_bf16_select": (I, [P, P, I, I, I, L, L, L, I, L, L, L, P, L, P, P, P, C.c_uint32, I, I, P]),
    "clt_select_rows_ext_workspace_bytes": (SZ, [I, I, I, I, I, I, I]),
    "clt_select_rows_ext_info": (SZ, [I, I, I, I, I, I, I, I]),
    "clt_select_rows_ext_f32": (I, [I, I, P, I, I, I, I, I, I, I, C.c_uint32, P, P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    "clt_gemm_bf16_select_rows": (I, [P, P, I, I, I, L, L, I, L, L, P, L, P, P, P, P, I, I, P]),
    "clt_select_lists_ext_f32": (I, [I, I, I, I, I, I, P, F32, P, I, I, I, I, I, P, F32, C.c_uint32, P, P, P, P, P, P, P, P, P, P,
                                     P, P, SZ, P]),
    "clt_sparse_decoder_fwd_f32": (I, [P, P, P, P, P, P, I, I, I, I, I, C.c_uint32, I, L, L, I, I, P]),
    "clt_sparse_row_pass_group_f32": (I, [I, P, P, P, P, P, P, P, I, I, I, P, P, P, P, L, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, I, I, I, P, P, C.c_uint32, I, C.c_uint32, P, P]),
    "clt_sparse_row_pass_f32": (I, [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, L, L, I, P,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I, I, I, P, P, P, P, P, P,
                                    P, P, C.c_uint32, I, C.c_uint32, P, I, P]),
    "clt_sparse_hot_columns": (I, [P, I, I, P, C.c_uint32, C.c_uint32, P, P]),
    "clt_compact_columns": (I, [P, I, I, I, P, P, P, P]),
    "clt_gather_cols_f32": (I, [P, P, I, I, I, I, F32, P, P]),
    "clt_gather_rows_f32": (I, [P, P, I, I, I, I, P, P]),
    "clt_sparse_reduce_dots_f32": (I, [P, L, P, I, I, I, P, P]),
    "clt_sparse_scatter_dense_f32": (I, [P, P, P, I, I, P, P]),
    "clt_adam_f32": (I, [P, P, P, P, L, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I, I, P, P, P, I, P]),
    "clt_adam_prep_f32": (I, [P, P, P, P, I, I, I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I, I, P, P, P, P, P, P]),
}

ABI_VERSION = 20


class CltError(RuntimeError):
    def __init__(self, fn, code):
        super().__init__(f"{fn} failed: {ERRORS.get(code, code)}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first (__graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.clt_abi_version() != ABI_VERSION:
        raise ImportError(f"libclt_hip.so ABI {lib.clt_abi_version()} != expected {ABI_VERSION}")
    return lib


lib = _load()


def call(name, *args):
    """Invoke an int-returning entry point; negative status -> CltError (AssertionError for k > F,
    matching the reference's `assert self.k <= d_features`, model/topk.py:56,80,113)."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        if rc == -2:
            raise AssertionError("k must be <= d_features")
        raise CltError(name, rc)
