/* Diagnostic entry points of tools/bin/liblitcoder_debug.so (tools/debug_kernels/build.py): NOT part of the product's
 * C ABI (include/litcoder_hip.h) -- until round 4 they shipped inside liblitcoder_hip.so. */
#pragma once
#include "../../include/litcoder_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The fp16x3 sweep (score mode) with s_memtime stamps at its phase boundaries; d_stamps (32 x uint64, caller-zeroed)
 * receives per wave-group sums: [main loop cycles, -, -, -, -, K-tiles, prologue, epilogue, catch-up barrier, epilogue step 0,
 * steps 1-6, step 7, store drain, step 0 repeated, main loop 100 MHz ticks, -].  Results of the kernel are not meaningful. */
int lc_debug_sweep16_stamps(const void* d_ht, const float* d_rowscale_inv, int A, int M, int N,
                            const void* d_yt, const float* d_cscale_inv, const float* d_yv,
                            int64_t V, int n_val, const float* d_ystat, float* d_part,
                            unsigned long long* d_stamps, lc_stream_t stream);

/* The single-group plain contraction of lc_gemm_grouped_f16x3 on v_mfma_f32_16x16x32_f16 instead of 32x32x16. */
int lc_debug_gemm_f16x3_wide(const void* d_at, const float* d_rowscale_inv, int64_t Mrows, const void* d_bt,
                             const float* d_cscale_inv, float* d_c, int64_t ldc, int64_t Ncols, int64_t K,
                             lc_stream_t stream);

#ifdef __cplusplus
}
#endif
