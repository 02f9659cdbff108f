"""ctypes binding of liblitcoder_hip.so (C ABI declared in include/litcoder_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails this module
raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m litcoder_core_amd.build``.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblitcoder_hip.so")

LC_F32, LC_F64, LC_I32 = 0, 1, 2
LC_SCORE_CORR, LC_SCORE_R2 = 0, 1
LC_NB, LC_MB = 64, 32
COL_TILE = 128          # voxel-axis padding granule of the MFMA kernels
K_TILE = 32             # contraction-axis padding granule of the MFMA kernels

_ptr = c_void_p
# name -> (restype, argtypes); mirrors include/litcoder_hip.h one to one
SIGNATURES = {
    "lc_version": (c_int, []),
    "lc_last_error": (c_char_p, []),
    "lc_check_device": (c_int, [c_int]),
    "lc_stream_create_cu_mask": (c_int, [_ptr, c_int, _ptr]),
    "lc_stream_destroy": (c_int, [_ptr]),
    "lc_timing_enable": (c_int, [c_int]),
    "lc_timing_enable_slots": (c_int, [c_uint64]),
    "lc_timing_slots": (c_int, []),
    "lc_timing_name": (c_char_p, [c_int]),
    "lc_timing_read": (c_int, [c_int, POINTER(c_double), POINTER(c_int)]),
    "lc_fir_delay": (c_int, [_ptr, c_int, c_int64, c_int64, c_int64, POINTER(c_int64), c_int, c_int, _ptr, c_int64, _ptr]),
    "lc_lanczos_interp": (c_int, [_ptr, c_int, c_int64, c_int64, c_int64, _ptr, _ptr, c_int64, c_double, c_double,
                                  c_int, _ptr, c_int64, _ptr]),
    "lc_sinc_interp": (c_int, [_ptr, c_int, c_int64, c_int64, c_int64, _ptr, _ptr, c_int64, c_double, c_double, c_int,
                               c_int, _ptr, c_int64, _ptr]),
    "lc_segment_reduce": (c_int, [_ptr, c_int, c_int64, c_int64, _ptr, _ptr, c_int64, c_int, _ptr, c_int64, _ptr]),
    "lc_cast_f64_f32": (c_int, [_ptr, c_int64, _ptr, c_int64, c_int64, c_int64, _ptr]),
    "lc_gather_f32": (c_int, [_ptr, c_int64, _ptr, c_int64, _ptr, c_int64, _ptr, c_int64, _ptr, _ptr]),
    "lc_scatter_axpy_f32": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int64, c_float, _ptr, c_int64, _ptr]),
    "lc_scatter_cols": (c_int, [_ptr, c_int64, c_int64, c_int, _ptr, c_int64, _ptr, c_int64, _ptr]),
    "lc_gemv_cols_f32": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, c_int64, _ptr, _ptr, c_int, c_int32, _ptr, c_int64, _ptr]),
    "lc_invert_perm": (c_int, [_ptr, c_int64, c_int32, _ptr, _ptr]),
    "lc_combine_folds_f32": (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), POINTER(c_float), c_int, c_int64,
                                     c_int64, _ptr, c_int64, _ptr]),
    "lc_col_mean_std_f32": (c_int, [_ptr, c_int64, _ptr, c_int64, c_int64, _ptr, _ptr, _ptr]),
    "lc_col_normalize_f32": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, _ptr, c_float, _ptr]),
    "lc_zscore_story_f64": (c_int, [_ptr, c_int64, c_int64, c_int64, c_int, _ptr, c_int64, _ptr]),
    "lc_pearson_cols": (c_int, [_ptr, c_int64, _ptr, c_int64, c_int64, c_int64, _ptr, _ptr]),
    "lc_pearson_cols_gather": (c_int, [_ptr, c_int64, _ptr, _ptr, _ptr, c_int64, c_int64, c_int64, _ptr, _ptr]),
    "lc_pearson_pvalues": (c_int, [_ptr, c_int64, c_int64, _ptr, _ptr]),
    "lc_gram_f64": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, c_int64, _ptr]),
    "lc_gram_f64_mfma": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, _ptr, c_int64, _ptr]),
    "lc_lambda_max": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int, c_int, c_int, _ptr, _ptr, _ptr]),
    "lc_lambda_max_masked": (c_int, [_ptr, c_int64, c_int, _ptr, c_int, c_int, c_double, _ptr, _ptr, c_int, _ptr]),
    "lc_penalties": (c_int, [_ptr, c_int, _ptr, c_int, c_int, _ptr, _ptr]),
    "lc_batch_assemble": (c_int, [_ptr, c_int64, _ptr, _ptr, _ptr, _ptr, c_int, c_int, c_int, c_int, _ptr, _ptr]),
    "lc_batch_assemble_sel": (c_int, [_ptr, c_int64, c_int64, _ptr, _ptr, _ptr, _ptr, _ptr, c_int, c_int, c_int, c_int, _ptr,
                                      _ptr]),
    "lc_gather_transpose_f32": (c_int, [_ptr, c_int64, _ptr, c_int, c_int, c_int, c_int, _ptr, _ptr]),
    "lc_gram_blocks_f64": (c_int, [_ptr, c_int64, c_int, c_int, c_int, _ptr, _ptr]),
    "lc_gather_rows_f64": (c_int, [_ptr, c_int64, _ptr, c_int, c_int, c_int, c_int, _ptr, _ptr]),
    "lc_primal_pad": (c_int, [c_int]),
    "lc_xty_f64": (c_int, [_ptr, c_int64, c_int, _ptr, c_int64, c_int64, _ptr, c_int, _ptr, _ptr, c_int, c_int, c_int, _ptr,
                           _ptr]),
    "lc_primal_set_stats": (c_int, [_ptr, c_int64, c_int, _ptr, c_int, _ptr, c_int, _ptr, _ptr]),
    "lc_primal_gsys": (c_int, [_ptr, _ptr, c_int, c_int, _ptr, _ptr]),
    "lc_primal_inverse": (c_int, [_ptr, _ptr, c_int, c_int, c_int, _ptr, _ptr, _ptr]),
    "lc_primal_scores": (c_int, [_ptr, c_int, c_int, _ptr, _ptr, _ptr, c_int64, c_int64, _ptr, _ptr, _ptr, c_int, c_int,
                                 c_int, _ptr, c_int64, _ptr]),
    "lc_primal_refit": (c_int, [_ptr, c_int, c_int, _ptr, _ptr, _ptr, c_int64, c_int64, c_int, c_int, _ptr, _ptr, _ptr,
                                c_int, c_float, _ptr, c_int64, _ptr, _ptr]),
    "lc_fill_argmax": (c_int, [_ptr, c_int, _ptr, c_int64, _ptr]),
    "lc_accumulate_f64": (c_int, [_ptr, _ptr, c_int64, _ptr]),
    "lc_fold_pack": (c_int, [_ptr, _ptr, _ptr, c_int64, _ptr, c_int64, _ptr, c_int, _ptr, c_int, _ptr, c_int64, _ptr]),
    "lc_fold_pack_at": (c_int, [_ptr, _ptr, _ptr, c_int64, _ptr, c_int64, _ptr, c_int, _ptr, c_int, _ptr, c_int64, c_int64, c_int,
                                _ptr]),
    "lc_memcpy2d_async": (c_int, [_ptr, c_int64, _ptr, c_int64, c_int64, c_int64, c_int, _ptr]),
    "lc_fill2d_bytes": (c_int, [_ptr, c_int64, c_int, c_int64, c_int64, _ptr]),
    "lc_host_cast_f64_f32": (c_int, [_ptr, c_int64, _ptr, c_int64, c_int64, c_int64]),
    "lc_host_copy_f32": (c_int, [_ptr, c_int64, _ptr, c_int64, c_int64, c_int64]),
    "lc_upload_start": (c_int, [_ptr, c_int, _ptr, c_int, c_int64, c_int, c_int, _ptr, _ptr]),
    "lc_upload_start_staged": (c_int, [_ptr, c_int, _ptr, _ptr, c_int, c_int64, c_int, c_int, _ptr, _ptr]),
    "lc_host_zscore_story": (c_int, [_ptr, c_int, c_int64, c_int64, c_int64, _ptr, c_int64]),
    "lc_lanczos_interp_stories": (c_int, [_ptr, c_int, c_int64, c_int64, _ptr, _ptr, c_int64, _ptr, _ptr, c_int, c_double,
                                          c_int, _ptr, c_int64, _ptr]),
    "lc_story_design_f32": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int, POINTER(c_int64), c_int, _ptr, c_int64, _ptr]),
    "lc_upload_wait": (c_int, [_ptr, c_int, _ptr]),
    "lc_upload_finish": (c_int, [_ptr]),
    "lc_upload_free": (c_int, [_ptr]),
    "lc_batch_eigh_work_bytes": (c_int64, [c_int, c_int]),
    "lc_batch_eigh_jacobi": (c_int, [_ptr, c_int, c_int, _ptr, _ptr, _ptr, _ptr, c_int64, c_int, c_double, POINTER(c_int32), _ptr]),
    "lc_batch_spectral_work_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "lc_batch_spectral_apply": (c_int, [_ptr, _ptr, c_int, c_int, _ptr, c_int, _ptr, c_int, c_double, _ptr, _ptr, c_int64, _ptr,
                                        POINTER(c_int32), _ptr, _ptr]),
    "lc_fold_unpack": (c_int, [_ptr, c_int, c_int64, _ptr, c_int64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "lc_batch_chol_solve": (c_int, [_ptr, c_int, c_int, c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "lc_batch_chol_inverse": (c_int, [_ptr, c_int, c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "lc_lambda_max_dense": (c_int, [_ptr, c_int64, c_int64, c_int, c_int, c_int, c_int, c_double, _ptr, _ptr, _ptr]),
    "lc_batch_series_hat": (c_int, [_ptr, c_int64, _ptr, _ptr, c_int, c_int, c_int, _ptr, _ptr, _ptr, c_int, c_int, c_int,
                                    _ptr, _ptr, _ptr]),
    "lc_batch_series_terms": (c_int, [_ptr, c_int64, _ptr, _ptr, c_int, c_int, c_int, _ptr, c_int, _ptr, _ptr, _ptr, c_int,
                                      _ptr]),
    "lc_fisher_combine": (c_int, [_ptr, c_int, c_int64, _ptr, _ptr]),
    "lc_bh_fdr_work_bytes": (c_int64, [c_int64]),
    "lc_bh_fdr": (c_int, [_ptr, c_int64, c_double, _ptr, _ptr, _ptr, c_int64, _ptr]),
    "lc_bh_reject": (c_int, [_ptr, c_int64, c_double, _ptr, _ptr, _ptr]),
    "lc_fill_bytes": (c_int, [_ptr, c_int, c_int64, _ptr]),
    "lc_gather_sub_f32_strided": (c_int, [_ptr, c_int64, _ptr, _ptr, c_int, c_int, c_int, _ptr, _ptr, c_int64, c_int64, c_int64,
                                          _ptr]),
    "lc_series_place": (c_int, [_ptr, c_int, c_int, c_int, c_int, _ptr, _ptr, c_int, _ptr]),
    "lc_scale_cast_f64_f32": (c_int, [_ptr, _ptr, _ptr, c_int64, _ptr]),
    "lc_combine_terms_f32": (c_int, [POINTER(c_void_p), POINTER(c_float), c_int, _ptr, c_int64, _ptr]),
    "lc_combine_terms_colmax_f32": (c_int, [POINTER(c_void_p), POINTER(c_float), c_int, _ptr, c_int64, c_int64, c_int64, _ptr,
                                            _ptr]),
    "lc_col_scales_from_max": (c_int, [_ptr, c_int64, _ptr, _ptr]),
    "lc_combine_terms_f64": (c_int, [POINTER(c_void_p), POINTER(c_double), c_int, _ptr, c_int64, _ptr]),
    "lc_gather_sub_f64": (c_int, [_ptr, c_int64, _ptr, _ptr, c_int, c_int, c_int, _ptr, _ptr]),
    "lc_gather_sub_f32": (c_int, [_ptr, c_int64, _ptr, _ptr, c_int, c_int, c_int, _ptr, _ptr, _ptr]),
    "lc_alpha_sweep_scores_f16x3_folds": (c_int, [_ptr, _ptr, c_int, c_int, c_int, c_int, _ptr, _ptr, _ptr, c_int64,
                                                  POINTER(c_int32), _ptr, _ptr, c_int, _ptr, _ptr, c_int, c_int64,
                                                  POINTER(c_int64), POINTER(c_int64), c_int, _ptr, _ptr]),
    "lc_alpha_sweep_finalize_folds": (c_int, [_ptr, _ptr, _ptr, c_int, c_int, c_int, POINTER(c_int32), c_int64, c_int, _ptr, c_int,
                                              _ptr, _ptr]),
    "lc_series_sweep_finalize_folds": (c_int, [_ptr, _ptr, _ptr, c_int, c_int, POINTER(c_int32), c_int64, _ptr, _ptr, c_int, _ptr,
                                               c_int, _ptr, _ptr]),
    "lc_series_sweep_scores_f16x3_folds": (c_int, [_ptr, _ptr, c_int, c_int, POINTER(c_int32), c_int64, _ptr, _ptr, c_int64,
                                                   _ptr, c_int64, _ptr, _ptr, _ptr, _ptr, c_int, _ptr, _ptr, c_int,
                                                   c_int64, POINTER(c_int64), POINTER(c_int64), c_int, _ptr, _ptr]),
    "lc_kappa_sums": (c_int, [_ptr, c_int64, c_int64, _ptr, _ptr]),
    "lc_undecided_cols": (c_int, [_ptr, c_int, c_int64, c_int64, c_float, _ptr, c_int64, _ptr, _ptr, _ptr, c_int, _ptr, _ptr]),
    "lc_series_scores": (c_int, [_ptr, c_int64, c_int, c_int, c_int, c_int64, _ptr, _ptr, _ptr, _ptr, c_int, _ptr, _ptr,
                                 c_int, _ptr]),
    "lc_transpose_rows_f64": (c_int, [_ptr, c_int64, _ptr, c_int, c_int64, _ptr, _ptr]),
    "lc_val_stats_folds": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int, c_int, POINTER(c_int32), _ptr, _ptr, _ptr, _ptr, _ptr]),
    "lc_val_stats": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int, c_int, _ptr, _ptr, _ptr, _ptr]),
    "lc_alpha_sweep_scores": (c_int, [_ptr, c_int, c_int, c_int, _ptr, c_int64, c_int64, _ptr, _ptr, c_int, _ptr, _ptr,
                                      c_int, _ptr, _ptr, c_int, _ptr]),
    "lc_split_rows_f16": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, _ptr, _ptr]),
    "lc_split_rows_f16_groups": (c_int, [_ptr, c_int64, c_int, c_int64, c_int64, _ptr, _ptr, _ptr]),
    "lc_mean_operator_image_f16": (c_int, [POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), c_int, c_float, c_int64,
                                           c_int64, _ptr, _ptr, _ptr]),
    "lc_mean_operator_images_f16": (c_int, [_ptr, c_int, POINTER(c_int64), POINTER(c_void_p), c_int, c_float, c_int64, c_int64,
                                            _ptr, _ptr, _ptr]),
    "lc_split_rows_f16_alphas": (c_int, [_ptr, c_int64, c_int, c_int, c_int64, c_int64, _ptr, _ptr, _ptr]),
    "lc_split_rows_f16_alphas_sel": (c_int, [_ptr, c_int64, c_int, c_int, c_int, c_int64, c_int64, _ptr, _ptr, _ptr]),
    "lc_col_scales_f16": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, _ptr, _ptr]),
    "lc_col_scales_f16_flags": (c_int, [_ptr, c_int64, c_int64, c_int64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "lc_split_cols_f16": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int, _ptr, _ptr, _ptr, _ptr]),
    "lc_permute_cols_f16": (c_int, [_ptr, _ptr, c_int64, c_int, _ptr, _ptr]),
    "lc_gemm_grouped_f16x3": (c_int, [_ptr, _ptr, c_int64, _ptr, _ptr, _ptr, c_int64, c_int64, c_int64,
                                      POINTER(c_int32), c_int, _ptr, c_int64, c_int64, c_int64, _ptr]),
    "lc_gemm_grouped_f16x3_pearson": (c_int, [_ptr, _ptr, c_int64, _ptr, _ptr, c_int64, c_int64, POINTER(c_int32), c_int,
                                              _ptr, c_int64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "lc_select_alpha": (c_int, [_ptr, c_int, c_int64, _ptr, _ptr, _ptr]),
    "lc_group_by_alpha": (c_int, [_ptr, c_int64, c_int, c_int, _ptr, _ptr, _ptr]),
    "lc_group_by_alpha_range": (c_int, [_ptr, c_int64, c_int, c_int, c_int, _ptr, _ptr, _ptr]),
    "lc_gemm_grouped_f32": (c_int, [_ptr, c_int64, c_int64, _ptr, c_int64, _ptr, _ptr, c_int64, c_int64, c_int64,
                                    c_int64, POINTER(c_int32), c_int, _ptr]),
    "lc_comm_unique_id_bytes": (c_int, []),
    "lc_comm_unique_id": (c_int, [_ptr, c_int]),
    "lc_comm_create": (c_int, [_ptr, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "lc_comm_destroy": (c_int, [_ptr]),
    "lc_allgather_f32": (c_int, [_ptr, _ptr, _ptr, c_int64, _ptr]),
    "lc_allgather_bytes": (c_int, [_ptr, _ptr, _ptr, c_int64, _ptr]),
    "lc_allreduce": (c_int, [_ptr, _ptr, c_int64, c_int, c_int, _ptr]),
    "lc_allreduce_sum_f32": (c_int, [_ptr, _ptr, c_int64, _ptr]),
}

_lib = None


class LitcoderHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LitcoderHipError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "litcoder_core_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().lc_last_error()
        msg = msg.decode() if msg else ""
        if rc == -2 or rc == -1:
            raise ValueError(f"{what}: {msg} (code {rc})")
        raise LitcoderHipError(f"{what}: {msg} (code {rc})")


def call(name, *args):
    check(getattr(load(), name)(*args), name)
