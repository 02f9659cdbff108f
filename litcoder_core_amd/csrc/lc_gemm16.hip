// Split-precision variant of the fused alpha sweep: the same contraction  pred_a = H_a . Y[train rows]
// and the same epilogue as lc_gemm.hip, but each fp32 operand is carried as TWO fp16 numbers
// (x*2^e = hi + lo, 22 significant bits after an exact power-of-two pre-scale of every H row and every
// Y column) and every product as THREE fp16 MFMAs with fp32 accumulation:
//       h*y  ~=  hi_h*hi_y + hi_h*lo_y + lo_h*hi_y          (dropped lo*lo term <= 2^-22 relative)
// v_mfma_f32_32x32x16_f16 runs at 16x the rate of the f32-input MFMA, so three of them are ~5x faster
// than one f32 MFMA step at fp32-level accuracy (measured against fp64: profiles/, tests).
//
// Operands are pre-tiled by the split kernels so that a K-tile (16 k) of either operand is ONE contiguous
// 16 KB chunk whose order is exactly the LDS image:  [plane hi|lo][k-group of 8][row or column 0..255][8 x f16].
// A lane's MFMA fragment (8 consecutive k of one row / column) is then one conflict-free ds_read_b128.
//
// Tile 256 x 256 x 16, 512 threads = 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA blocks, four 32 KB LDS
// stages filled by LDS-DMA three to four K-tiles ahead, and the two waves of every SIMD run half an iteration
// apart (one reads fragments / issues DMA while the other issues MFMAs).
#include "lc_gemm16_kernel.h"


// defined in lc_gemm.hip: combines the per-block partial moments into scores (F folds: fp32 sum in fold order)
int lc_score_finalize_launch(const float* d_part, const float* d_ystat, const float* d_yblk, int A, int M,
                             const int* h_n_val, int F, long long V, int mode, float* d_scores, int accumulate,
                             hipStream_t s, const int* d_live_cols);

// B view from the C-ABI triple (rows of the image, first row of the gap, rows of the gap; all multiples of 16)
static int make_bview(const char* who, int64_t K, int64_t b_rows, int64_t gap_begin, int64_t gap_rows, BView* bv) {
    if (b_rows <= 0) { b_rows = K; gap_begin = K; gap_rows = 0; }
    LC_REQUIRE(b_rows % TK == 0 && gap_begin % TK == 0 && gap_rows % TK == 0 && gap_rows >= 0 && gap_begin >= 0 &&
                   gap_begin <= K && K + gap_rows <= b_rows,
               LC_E_SHAPE, "%s: B view needs b_rows, gap_begin, gap_rows multiples of %d with K + gap_rows <= b_rows", who,
               TK);
    bv->kt_total = (int)(b_rows / TK);
    bv->cut = (int)(gap_begin / TK);
    bv->skip = (int)(gap_rows / TK);
    return LC_OK;
}

extern "C" int lc_split_rows_f16_groups(const float* d_h, int64_t ld, int groups, int64_t rows, int64_t K, void* d_tiled,
                                        float* d_rowscale_inv, lc_stream_t stream) {
    LC_REQUIRE(d_h && d_tiled && d_rowscale_inv, LC_E_BADARG, "lc_split_rows_f16: null pointer");
    LC_REQUIRE(groups > 0 && rows > 0 && K > 0 && K % TK == 0 && ld % 4 == 0 && ld >= K, LC_E_SHAPE,
               "lc_split_rows_f16: need K %% %d == 0 and ld %% 4 == 0", TK);
    const long long rows_pad = lc::ceil_div<long long>(rows, TM) * TM;
    LC_REQUIRE(rows_pad * groups < (1ll << 31), LC_E_SHAPE, "lc_split_rows_f16: too many rows");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_split_rows_f16, dim3((unsigned)(rows_pad * groups / 4)), dim3(256), 0, lc::as_stream(stream), d_h,
                       (long long)ld, (int)rows, (int)K, (uint4*)d_tiled, d_rowscale_inv, (int)rows_pad, groups, 0, (int)rows);
    return lc::launched("k_split_rows_f16");
}

extern "C" int lc_split_rows_f16_alphas_sel(const float* d_h, int64_t ld, int groups, int A_src, int A, int64_t M, int64_t K,
                                            void* d_tiled, float* d_rowscale_inv, lc_stream_t stream) {
    LC_REQUIRE(d_h && d_tiled && d_rowscale_inv, LC_E_BADARG, "lc_split_rows_f16_alphas: null pointer");
    LC_REQUIRE(groups > 0 && A > 0 && A_src >= A && M > 0 && M % LC_MB == 0 && K > 0 && K % TK == 0 && ld % 4 == 0 && ld >= K,
               LC_E_SHAPE, "lc_split_rows_f16_alphas: need 0 < A <= A_src, M %% %d == 0, K %% %d == 0 and ld %% 4 == 0", LC_MB, TK);
    const long long rows = (long long)A * M;
    const long long rows_pad = lc::ceil_div<long long>(rows, TM) * TM;
    LC_REQUIRE(rows_pad * groups < (1ll << 31) && (long long)A_src * M < (1ll << 31), LC_E_SHAPE,
               "lc_split_rows_f16_alphas: too many rows");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_split_rows_f16, dim3((unsigned)(rows_pad * groups / 4)), dim3(256), 0, lc::as_stream(stream), d_h,
                       (long long)ld, (int)rows, (int)K, (uint4*)d_tiled, d_rowscale_inv, (int)rows_pad, groups, A,
                       (int)((long long)A_src * M));
    return lc::launched("k_split_rows_f16");
}

extern "C" int lc_split_rows_f16_alphas(const float* d_h, int64_t ld, int groups, int A, int64_t M, int64_t K, void* d_tiled,
                                        float* d_rowscale_inv, lc_stream_t stream) {
    return lc_split_rows_f16_alphas_sel(d_h, ld, groups, A, A, M, K, d_tiled, d_rowscale_inv, stream);
}

extern "C" int lc_split_rows_f16(const float* d_h, int64_t ld, int64_t rows, int64_t K, void* d_tiled,
                                 float* d_rowscale_inv, lc_stream_t stream) {
    return lc_split_rows_f16_groups(d_h, ld, 1, rows, K, d_tiled, d_rowscale_inv, stream);
}

extern "C" int lc_mean_operator_image_f16(const float* const* h_mats, const int64_t* h_ld, const int32_t* const* h_maps,
                                          int n_folds, float scale, int64_t rows, int64_t K, void* d_tiled,
                                          float* d_rowscale_inv, lc_stream_t stream) {
    LC_REQUIRE(h_mats && h_ld && h_maps && d_tiled && d_rowscale_inv, LC_E_BADARG, "lc_mean_operator_image_f16: null pointer");
    LC_REQUIRE(n_folds >= 1 && n_folds <= MO_MAX_FOLDS, LC_E_SHAPE, "lc_mean_operator_image_f16: 1..%d folds", MO_MAX_FOLDS);
    LC_REQUIRE(rows > 0 && K > 0 && K % TK == 0 && K <= 512 * MO_UNITS, LC_E_SHAPE,
               "lc_mean_operator_image_f16: need K %% %d == 0 and K <= %d", TK, 512 * MO_UNITS);
    MeanOpArgs a{};
    for (int f = 0; f < n_folds; ++f) {
        LC_REQUIRE(h_mats[f] && h_maps[f], LC_E_BADARG, "lc_mean_operator_image_f16: null operator / map");
        LC_REQUIRE((reinterpret_cast<uintptr_t>(h_maps[f]) & 15) == 0, LC_E_BADARG,
                   "lc_mean_operator_image_f16: maps must be 16-byte aligned");
        a.m[f] = h_mats[f];
        a.ld[f] = (long long)h_ld[f];
        a.map[f] = h_maps[f];
    }
    a.nf = n_folds;
    a.scale = scale;
    const long long rows_pad = lc::ceil_div<long long>(rows, TM) * TM;
    LC_REQUIRE(rows_pad < (1ll << 31), LC_E_SHAPE, "lc_mean_operator_image_f16: too many rows");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_mean_operator_image<false>, dim3((unsigned)(rows_pad / 4)), dim3(256), 0, lc::as_stream(stream), a,
                       (const long long*)nullptr, (int)rows, (int)K, (uint4*)d_tiled, d_rowscale_inv, (int)rows_pad);
    return lc::launched("k_mean_operator_image");
}

extern "C" int lc_mean_operator_images_f16(const int64_t* d_table, int n_images, const int64_t* h_ld,
                                           const int32_t* const* h_maps, int n_folds, float scale, int64_t rows, int64_t K,
                                           void* d_tiled, float* d_rowscale_inv, lc_stream_t stream) {
    LC_REQUIRE(d_table && h_ld && h_maps && d_tiled && d_rowscale_inv, LC_E_BADARG, "lc_mean_operator_images_f16: null pointer");
    LC_REQUIRE(n_folds >= 1 && n_folds <= MO_MAX_FOLDS, LC_E_SHAPE, "lc_mean_operator_images_f16: 1..%d folds", MO_MAX_FOLDS);
    LC_REQUIRE(n_images >= 0 && n_images <= 65535, LC_E_SHAPE, "lc_mean_operator_images_f16: 0..65535 images per launch");
    LC_REQUIRE(rows > 0 && K > 0 && K % TK == 0 && K <= 512 * MO_UNITS, LC_E_SHAPE,
               "lc_mean_operator_images_f16: need K %% %d == 0 and K <= %d", TK, 512 * MO_UNITS);
    if (n_images == 0) return LC_OK;
    MeanOpArgs a{};
    for (int f = 0; f < n_folds; ++f) {
        LC_REQUIRE(h_maps[f] && (reinterpret_cast<uintptr_t>(h_maps[f]) & 15) == 0, LC_E_BADARG,
                   "lc_mean_operator_images_f16: null or misaligned map");
        a.ld[f] = (long long)h_ld[f];
        a.map[f] = h_maps[f];
    }
    a.nf = n_folds;
    a.scale = scale;
    const long long rows_pad = lc::ceil_div<long long>(rows, TM) * TM;
    LC_REQUIRE(rows_pad < (1ll << 31), LC_E_SHAPE, "lc_mean_operator_images_f16: too many rows");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_mean_operator_image<true>, dim3((unsigned)(rows_pad / 4), (unsigned)n_images), dim3(256), 0,
                       lc::as_stream(stream), a, (const long long*)d_table, (int)rows, (int)K, (uint4*)d_tiled, d_rowscale_inv,
                       (int)rows_pad);
    return lc::launched("k_mean_operator_image");
}

extern "C" int lc_col_scales_f16_flags(const float* d_y, int64_t ldy, int64_t T, int64_t V, float* d_cscale,
                                       int32_t* d_flag, uint8_t* d_colflag, const int32_t* d_live_cols, lc_stream_t stream) {
    LC_REQUIRE(d_y && d_cscale, LC_E_BADARG, "lc_col_scales_f16: null pointer");     // d_flag may be NULL: scales only
    LC_REQUIRE(T > 0 && V > 0 && ldy >= V, LC_E_SHAPE, "lc_col_scales_f16: bad shape");
    LC_REQUIRE(d_flag || !d_colflag, LC_E_BADARG, "lc_col_scales_f16: per-column flags need the flag pass (d_flag)");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_col_scales, dim3((unsigned)lc::ceil_div<long long>(V, 64)), dim3(64, CS_RG), 0,
                       lc::as_stream(stream), d_y, (long long)ldy, (int)T, (long long)V, d_cscale, d_flag, d_colflag, d_live_cols);
    return lc::launched("k_col_scales");
}

extern "C" int lc_col_scales_f16(const float* d_y, int64_t ldy, int64_t T, int64_t V, float* d_cscale,
                                 int32_t* d_flag, lc_stream_t stream) {
    return lc_col_scales_f16_flags(d_y, ldy, T, V, d_cscale, d_flag, nullptr, nullptr, stream);
}

extern "C" int lc_combine_terms_colmax_f32(const float* const* h_terms, const float* h_coef, int terms, float* d_out,
                                          int64_t ld, int64_t rows, int64_t cols, uint32_t* d_colmax, lc_stream_t stream) {
    LC_REQUIRE(h_terms && h_coef && d_out && terms >= 1 && terms <= 4, LC_E_BADARG,
               "lc_combine_terms_colmax_f32: need 1..4 terms");
    LC_REQUIRE(rows >= 0 && rows < (1ll << 31) && cols >= 0 && cols % 4 == 0 && ld >= cols && ld % 4 == 0, LC_E_SHAPE,
               "lc_combine_terms_colmax_f32: need cols %% 4 == 0, ld %% 4 == 0, ld >= cols");
    LC_REQUIRE((reinterpret_cast<uintptr_t>(d_out) & 15) == 0, LC_E_BADARG, "lc_combine_terms_colmax_f32: d_out not 16-byte aligned");
    for (int j = 0; j < terms; ++j)
        LC_REQUIRE(h_terms[j] && (reinterpret_cast<uintptr_t>(h_terms[j]) & 15) == 0, LC_E_BADARG,
                   "lc_combine_terms_colmax_f32: null or misaligned term");
    if (rows == 0 || cols == 0) return LC_OK;
    const float* t[4] = {h_terms[0], terms > 1 ? h_terms[1] : nullptr, terms > 2 ? h_terms[2] : nullptr,
                         terms > 3 ? h_terms[3] : nullptr};
    const float c[4] = {h_coef[0], terms > 1 ? h_coef[1] : 0.f, terms > 2 ? h_coef[2] : 0.f, terms > 3 ? h_coef[3] : 0.f};
    const long long cols4 = cols / 4;
    const long long gy = lc::ceil_div<long long>(rows, CC_ROWS);
    LC_REQUIRE(gy <= 65535, LC_E_SHAPE, "lc_combine_terms_colmax_f32: too many rows");
    dim3 grid((unsigned)lc::ceil_div<long long>(cols4, 64), (unsigned)gy);
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_combine_colmax, grid, dim3(256), 0, lc::as_stream(stream), t[0], t[1], t[2], t[3], c[0], c[1],
                       c[2], c[3], terms, d_out, (long long)ld, (int)rows, cols4, d_colmax);
    return lc::launched("k_combine_colmax");
}

extern "C" int lc_col_scales_from_max(const uint32_t* d_colmax, int64_t V, float* d_cscale, lc_stream_t stream) {
    LC_REQUIRE(d_colmax && d_cscale, LC_E_BADARG, "lc_col_scales_from_max: null pointer");
    LC_REQUIRE(V > 0, LC_E_SHAPE, "lc_col_scales_from_max: bad shape");
    hipLaunchKernelGGL(k_scales_from_max, dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0,
                       lc::as_stream(stream), d_colmax, (long long)V, d_cscale);
    return lc::launched("k_scales_from_max");
}

extern "C" int lc_split_cols_f16(const float* d_y, int64_t ldy, int64_t V, const int32_t* d_rows, int K,
                                 const float* d_cscale, void* d_tiled, const int32_t* d_live_cols, lc_stream_t stream) {
    LC_REQUIRE(d_y && d_rows && d_cscale && d_tiled, LC_E_BADARG, "lc_split_cols_f16: null pointer");
    LC_REQUIRE(V > 0 && K > 0 && K % TK == 0 && ldy >= V, LC_E_SHAPE, "lc_split_cols_f16: need K %% %d == 0", TK);
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    dim3 grid((unsigned)lc::ceil_div<long long>(V, TN), (unsigned)(K / 8));
    hipLaunchKernelGGL(k_split_cols_f16, grid, dim3(256), 0, lc::as_stream(stream), d_y, (long long)ldy, (long long)V,
                       d_rows, K, d_cscale, (uint4*)d_tiled, d_live_cols);
    return lc::launched("k_split_cols_f16");
}

extern "C" int lc_permute_cols_f16(const void* d_tiled, const int32_t* d_perm, int64_t Vs, int K, void* d_out,
                                   lc_stream_t stream) {
    LC_REQUIRE(d_tiled && d_perm && d_out, LC_E_BADARG, "lc_permute_cols_f16: null pointer");
    LC_REQUIRE(Vs > 0 && Vs % TN == 0 && K > 0 && K % TK == 0, LC_E_SHAPE,
               "lc_permute_cols_f16: need Vs %% %d == 0, K %% %d == 0", TN, TK);
    const int rows16 = (K / TK) * 2 * KG;                 // 16-byte unit rows per column tile
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    dim3 grid((unsigned)(Vs / TN), (unsigned)lc::ceil_div(rows16, 8));
    hipLaunchKernelGGL(k_permute_cols_f16, grid, dim3(256), 0, lc::as_stream(stream), (const uint4*)d_tiled, d_perm,
                       (long long)Vs, rows16, (uint4*)d_out);
    return lc::launched("k_permute_cols_f16");
}

static int fill_fold_views(const char* who, int F, int64_t K, int64_t b_rows, const int64_t* h_gap_begin,
                           const int64_t* h_gap_rows, const int32_t* h_n_val, int M, BView* bv, FoldViews* fv) {
    LC_REQUIRE(F >= 1 && F <= MAX_FOLDS16, LC_E_SHAPE, "%s: 1 <= F <= %d inner folds per launch", who, MAX_FOLDS16);
    for (int f = 0; f < F; ++f) {
        BView one;
        if (int rc = make_bview(who, K, b_rows, h_gap_begin ? h_gap_begin[f] : 0, h_gap_rows ? h_gap_rows[f] : 0, &one))
            return rc;
        LC_REQUIRE(h_n_val[f] > 0 && h_n_val[f] <= M, LC_E_SHAPE, "%s: need 0 < n_val <= M", who);
        *bv = one;
        fv->cut[f] = one.cut;
        fv->skip[f] = one.skip;
        fv->n_val[f] = h_n_val[f];
    }
    return LC_OK;
}

extern "C" int lc_alpha_sweep_scores_f16x3_folds(const void* d_ht, const float* d_rowscale_inv, int F, int A, int M, int N,
                                                 const void* d_yt, const float* d_cscale_inv, const float* d_yv,
                                                 int64_t V, const int32_t* h_n_val, const float* d_ystat,
                                                 const float* d_yblk, int mode, float* d_part, float* d_scores,
                                                 int accumulate, int64_t b_rows, const int64_t* h_gap_begin,
                                                 const int64_t* h_gap_rows, int terms, const int32_t* d_live_cols,
        lc_stream_t stream) {
    LC_REQUIRE(d_ht && d_rowscale_inv && d_yt && d_cscale_inv && d_yv && d_ystat && d_yblk && d_part && d_scores &&
                   h_n_val, LC_E_BADARG, "lc_alpha_sweep_scores_f16x3: null pointer");
    LC_REQUIRE(terms == 3 || ((terms == 1 || terms == 101) && N % (4 * TK) == 0 && mode == LC_SCORE_CORR), LC_E_BADARG,
               "lc_alpha_sweep_scores_f16x3: terms must be 3, or 1 / 101 (screening: correlation scores, N %% %d == 0)", 4 * TK);
    LC_REQUIRE(A > 0 && M > 0 && M % LC_MB == 0 && N > 0 && N % (2 * TK) == 0, LC_E_SHAPE,
               "lc_alpha_sweep_scores_f16x3: need M %% %d == 0, N %% %d == 0", LC_MB, 2 * TK);
    LC_REQUIRE(V > 0 && V % 128 == 0, LC_E_SHAPE, "lc_alpha_sweep_scores_f16x3: V must be a multiple of 128");
    LC_REQUIRE(mode == LC_SCORE_CORR || mode == LC_SCORE_R2, LC_E_BADARG, "lc_alpha_sweep_scores_f16x3: bad mode");
    const void* kern = terms == 1     ? reinterpret_cast<const void*>(k_sweep_hi2<true>)
                       : terms == 101 ? reinterpret_cast<const void*>(k_sweep_f16x3<true, false, false, false, false, true>)
                                      : reinterpret_cast<const void*>(k_sweep_f16x3<true, false>);
    if (int rc = lc::ensure_dynamic_lds(kern, terms == 1 ? H2_LDS_BYTES : LDS16_BYTES)) return rc;
    hipStream_t s = lc::as_stream(stream);
    BView bv;
    FoldViews fv{};
    if (int rc = fill_fold_views("lc_alpha_sweep_scores_f16x3", F, N, b_rows, h_gap_begin, h_gap_rows, h_n_val, M, &bv, &fv))
        return rc;
    const int Mrows = A * M;
    const int Mtiles = lc::ceil_div(Mrows, TM);            // per fold: the folds' images are stacked tile-aligned
    const long long Ntiles = lc::ceil_div<long long>(V, TN);
    LC_REQUIRE((long long)F * Mtiles * Ntiles * 2 < (1ll << 31), LC_E_SHAPE, "lc_alpha_sweep_scores_f16x3: grid too large");
    fv.mt_per_fold = Mtiles;
    fv.part_stride = (long long)(Mrows / LC_MB) * 4 * V;
    Score16Args sa{d_yv, d_ystat, d_rowscale_inv, d_cscale_inv, d_part, (long long)V, M, 0, mode, Mrows, A, d_live_cols};
    {
        lc::ScopedTimer timer_(lc::T_SWEEP_GEMM, s);
        Plain16Args pa{};
        if (terms == 1)                    // screening, two 4-wave workgroups per CU on 256 x 128 tiles
            hipLaunchKernelGGL((k_sweep_hi2<true>), dim3((unsigned)(F * Mtiles * Ntiles * 2)), dim3(H2_THREADS), H2_LDS_BYTES,
                               s, (const uint4*)d_ht, (const uint4*)d_yt, N / TK, F * Mtiles, sa, pa, bv, fv);
        else if (terms == 101)             // screening, the one-workgroup-per-CU form (kept for A/B measurements)
            hipLaunchKernelGGL((k_sweep_f16x3<true, false, false, false, false, true>), dim3((unsigned)(F * Mtiles * Ntiles)),
                               dim3(512), LDS16_BYTES, s, (const uint4*)d_ht, (const uint4*)d_yt, N / TK, F * Mtiles, sa, pa,
                               bv, fv);
        else
            hipLaunchKernelGGL((k_sweep_f16x3<true, false>), dim3((unsigned)(F * Mtiles * Ntiles)), dim3(512), LDS16_BYTES, s,
                               (const uint4*)d_ht, (const uint4*)d_yt, N / TK, F * Mtiles, sa, pa, bv, fv);
    }
    if (int rc = lc::launched("k_sweep_f16x3")) return rc;
    if (accumulate == 2) return LC_OK;                    // the contraction alone: the caller finalises several folds at once
    return lc_score_finalize_launch(d_part, d_ystat, d_yblk, A, M, h_n_val, F, (long long)V, mode, d_scores, accumulate, s,
                                    d_live_cols);
}

// The second halves of the two sweeps alone, for F folds whose contractions were launched one by one with accumulate = 2
// (each into its slice of the (F, ...) stacks): one pass over all folds' partials instead of one small launch per fold.
extern "C" int lc_alpha_sweep_finalize_folds(const float* d_part, const float* d_ystat, const float* d_yblk, int F, int A, int M,
                                             const int32_t* h_n_val, int64_t V, int mode, float* d_scores, int accumulate,
                                             const int32_t* d_live_cols, lc_stream_t stream) {
    LC_REQUIRE(d_part && d_ystat && d_yblk && d_scores && h_n_val, LC_E_BADARG, "lc_alpha_sweep_finalize_folds: null pointer");
    LC_REQUIRE(F >= 1 && F <= MAX_FOLDS16 && A > 0 && M > 0 && M % LC_MB == 0 && V > 0 && V % 128 == 0, LC_E_SHAPE,
               "lc_alpha_sweep_finalize_folds: bad shape");
    return lc_score_finalize_launch(d_part, d_ystat, d_yblk, A, M, h_n_val, F, (long long)V, mode, d_scores, accumulate ? 1 : 0,
                                    lc::as_stream(stream), d_live_cols);
}

extern "C" int lc_gemm_grouped_f16x3(const void* d_at, const float* d_rowscale_inv, int64_t Mrows, const void* d_bt,
                                     const float* d_cscale_inv, float* d_c, int64_t ldc, int64_t Ncols, int64_t K,
                                     const int32_t* h_group_tiles, int G, const uint8_t* d_slab_light,
                                     int64_t b_rows, int64_t b_gap_begin, int64_t b_gap_rows, lc_stream_t stream) {
    LC_REQUIRE(d_at && d_rowscale_inv && d_bt && d_cscale_inv && d_c && h_group_tiles, LC_E_BADARG,
               "lc_gemm_grouped_f16x3: null pointer");
    LC_REQUIRE(G >= 1 && G <= MAX_GROUPS16, LC_E_SHAPE, "lc_gemm_grouped_f16x3: G must be in 1..%d", MAX_GROUPS16);
    // ldc < Ncols: only the first ldc columns are stored (an output whose width is not a multiple of the column tile)
    LC_REQUIRE(Mrows > 0 && K > 0 && K % (2 * TK) == 0 && Ncols > 0 && Ncols % TN == 0 && ldc > Ncols - TN, LC_E_SHAPE,
               "lc_gemm_grouped_f16x3: need K %% %d == 0, Ncols %% %d == 0, ldc > Ncols - %d", 2 * TK, TN, TN);
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_sweep_f16x3<false, false>), LDS16_BYTES)) return rc;
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_sweep_f16x3<false, false, true>), LDS16_BYTES))
        return rc;
    const int Mtiles = (int)lc::ceil_div<long long>(Mrows, TM);
    const long long Ntiles = Ncols / TN;
    BView bv;
    if (int rc = make_bview("lc_gemm_grouped_f16x3", K, b_rows, b_gap_begin, b_gap_rows, &bv)) return rc;
    Plain16Args pa{};
    pa.c = d_c;
    pa.ldc = ldc;
    pa.rs_inv = d_rowscale_inv;
    pa.cs_inv = d_cscale_inv;
    pa.Mrows = (int)Mrows;
    pa.G = G;
    pa.slab_light = d_slab_light;
    pa.col_limit = ldc < Ncols ? ldc : Ncols;
    for (int g = 0; g <= G; ++g) {
        pa.start[g] = h_group_tiles[g];
        LC_REQUIRE(g == 0 ? pa.start[0] == 0 : pa.start[g] >= pa.start[g - 1], LC_E_SHAPE,
                   "lc_gemm_grouped_f16x3: group tile offsets must start at 0 and be non-decreasing");
    }
    LC_REQUIRE(pa.start[G] == Ntiles, LC_E_SHAPE, "lc_gemm_grouped_f16x3: last group offset %d != %lld column tiles",
               pa.start[G], Ntiles);
    LC_REQUIRE((long long)Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_gemm_grouped_f16x3: grid too large");
    Score16Args sa{};
    sa.A = 1;
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_GROUPED_GEMM, s);
    if (d_slab_light)
        hipLaunchKernelGGL((k_sweep_f16x3<false, false, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512), LDS16_BYTES,
                           s, (const uint4*)d_at, (const uint4*)d_bt, (int)(K / TK), Mtiles, sa, pa, bv, FoldViews{});
    else
        hipLaunchKernelGGL((k_sweep_f16x3<false, false>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512), LDS16_BYTES, s,
                           (const uint4*)d_at, (const uint4*)d_bt, (int)(K / TK), Mtiles, sa, pa, bv, FoldViews{});
    return lc::launched("k_sweep_f16x3<plain>");
}

namespace {
// Pearson r per column from the slabs' partials (PEARSON epilogue): the pairwise update of means and centred sums.
__global__ void __launch_bounds__(256) k_pearson_from_parts(const double* __restrict__ part, int slabs, long long ncols,
                                                            double* __restrict__ r_out) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    double n = 0.0, ma = 0.0, mb = 0.0, qa = 0.0, qb = 0.0, qab = 0.0;
    for (int s = 0; s < slabs; ++s) {
        const double* p = part + (long long)s * 6 * ncols + c;
        const double ns = p[0];
        if (!(ns > 0.0)) continue;
        const double mas = p[ncols], mbs = p[2 * ncols];
        if (n == 0.0) {
            n = ns; ma = mas; mb = mbs; qa = p[3 * ncols]; qb = p[4 * ncols]; qab = p[5 * ncols];
        } else {
            const double tot = n + ns, da = mas - ma, db = mbs - mb, w = n * ns / tot;
            qa += p[3 * ncols] + da * da * w;
            qb += p[4 * ncols] + db * db * w;
            qab += p[5 * ncols] + da * db * w;
            ma += da * (ns / tot);
            mb += db * (ns / tot);
            n = tot;
        }
    }
    double r = qab / (sqrt(qa) * sqrt(qb));
    if (r > 1.0) r = 1.0;
    if (r < -1.0) r = -1.0;
    r_out[c] = r;
}
}  // namespace

extern "C" int lc_gemm_grouped_f16x3_pearson(const void* d_at, const float* d_rowscale_inv, int64_t Mrows, const void* d_bt,
                                             const float* d_cscale_inv, int64_t Ncols, int64_t K,
                                             const int32_t* h_group_tiles, int G, const float* d_y, int64_t ldy,
                                             const int32_t* d_y_rows, const int32_t* d_y_cols, double* d_part, double* d_r,
                                             lc_stream_t stream) {
    LC_REQUIRE(d_at && d_rowscale_inv && d_bt && d_cscale_inv && h_group_tiles && d_y && d_part && d_r, LC_E_BADARG,
               "lc_gemm_grouped_f16x3_pearson: null pointer");
    LC_REQUIRE(G >= 1 && G <= MAX_GROUPS16, LC_E_SHAPE, "lc_gemm_grouped_f16x3_pearson: G must be in 1..%d", MAX_GROUPS16);
    LC_REQUIRE(Mrows > 0 && K > 0 && K % (2 * TK) == 0 && Ncols > 0 && Ncols % TN == 0 && ldy > 0, LC_E_SHAPE,
               "lc_gemm_grouped_f16x3_pearson: need K %% %d == 0, Ncols %% %d == 0", 2 * TK, TN);
    const void* kern = reinterpret_cast<const void*>(k_sweep_f16x3<false, false, false, false, true>);
    if (int rc = lc::ensure_dynamic_lds(kern, LDS16_BYTES)) return rc;
    const int Mtiles = (int)lc::ceil_div<long long>(Mrows, TM);
    const long long Ntiles = Ncols / TN;
    BView bv;
    if (int rc = make_bview("lc_gemm_grouped_f16x3_pearson", K, 0, 0, 0, &bv)) return rc;
    Plain16Args pa{};
    pa.rs_inv = d_rowscale_inv;
    pa.cs_inv = d_cscale_inv;
    pa.Mrows = (int)Mrows;
    pa.G = G;
    pa.col_limit = Ncols;
    pa.pr_y = d_y;
    pa.pr_ldy = ldy;
    pa.pr_rows = d_y_rows;
    pa.pr_cols = d_y_cols;
    pa.pr_part = d_part;
    pa.pr_ncols = Ncols;
    for (int g = 0; g <= G; ++g) {
        pa.start[g] = h_group_tiles[g];
        LC_REQUIRE(g == 0 ? pa.start[0] == 0 : pa.start[g] >= pa.start[g - 1], LC_E_SHAPE,
                   "lc_gemm_grouped_f16x3_pearson: group tile offsets must start at 0 and be non-decreasing");
    }
    LC_REQUIRE(pa.start[G] == Ntiles, LC_E_SHAPE, "lc_gemm_grouped_f16x3_pearson: last group offset %d != %lld column tiles",
               pa.start[G], Ntiles);
    LC_REQUIRE((long long)Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_gemm_grouped_f16x3_pearson: grid too large");
    Score16Args sa{};
    sa.A = 1;
    hipStream_t s = lc::as_stream(stream);
    {
        lc::ScopedTimer timer_(lc::T_GROUPED_GEMM, s);
        hipLaunchKernelGGL((k_sweep_f16x3<false, false, false, false, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512),
                           LDS16_BYTES, s, (const uint4*)d_at, (const uint4*)d_bt, (int)(K / TK), Mtiles, sa, pa, bv,
                           FoldViews{});
    }
    if (int rc = lc::launched("k_sweep_f16x3<pearson>")) return rc;
    lc::ScopedTimer timer_(lc::T_PEARSON, s);
    hipLaunchKernelGGL(k_pearson_from_parts, dim3((unsigned)lc::ceil_div<long long>(Ncols, 256)), dim3(256), 0, s, d_part,
                       (int)lc::ceil_div<long long>(Mrows, 128), (long long)Ncols, d_r);     // (slabs that hold rows)
    return lc::launched("k_pearson_from_parts");
}

// defined in lc_gemm.hip
int lc_series_finalize_launch(const float* d_part, const float* d_ystat, const float* d_yblk, int M, const int* h_n_val,
                              int F, long long V, const double* d_coef, const int* d_aidx, int S, float* d_scores,
                              int accumulate, hipStream_t s, const int* d_live_cols);

extern "C" int lc_series_sweep_scores_f16x3_folds(const void* d_pt, const float* d_rowscale_inv, int F, int M,
                                                  const int32_t* h_n_val, int64_t K, const void* d_yt,
                                                  const float* d_cscale_inv, int64_t Ncols, const float* d_yv, int64_t V,
                                                  const float* d_ystat, const float* d_yblk, const double* d_coef,
                                                  const int32_t* d_aidx, int S, float* d_part, float* d_scores,
                                                  int accumulate, int64_t b_rows, const int64_t* h_gap_begin,
                                                  const int64_t* h_gap_rows, int terms, const int32_t* d_live_cols,
        lc_stream_t stream) {
    LC_REQUIRE(d_pt && d_rowscale_inv && d_yt && d_cscale_inv && d_yv && d_ystat && d_yblk && d_coef && d_aidx &&
                   d_part && d_scores && h_n_val, LC_E_BADARG, "lc_series_sweep_scores_f16x3: null pointer");
    LC_REQUIRE(terms == 3 || ((terms == 1 || terms == 101) && K % (4 * TK) == 0), LC_E_BADARG,
               "lc_series_sweep_scores_f16x3: terms must be 3, or 1 / 101 (screening: K %% %d == 0)", 4 * TK);
    LC_REQUIRE(M > 0 && M % LC_MB == 0 && K > 0 && K % (2 * TK) == 0 && S > 0, LC_E_SHAPE,
               "lc_series_sweep_scores_f16x3: need M %% %d == 0, K %% %d == 0", LC_MB, 2 * TK);
    LC_REQUIRE(V > 0 && V % 128 == 0 && Ncols >= V && Ncols % TN == 0, LC_E_SHAPE,
               "lc_series_sweep_scores_f16x3: V must be a multiple of 128, Ncols >= V a multiple of %d", TN);
    const void* kern = terms == 1     ? reinterpret_cast<const void*>(k_sweep_hi2<false>)
                       : terms == 101 ? reinterpret_cast<const void*>(k_sweep_f16x3<false, false, true, true, false, true>)
                                      : reinterpret_cast<const void*>(k_sweep_f16x3<false, false, true, true>);
    if (int rc = lc::ensure_dynamic_lds(kern, terms == 1 ? H2_LDS_BYTES : LDS16_BYTES)) return rc;
    hipStream_t s = lc::as_stream(stream);
    BView bv;
    FoldViews fv{};
    if (int rc = fill_fold_views("lc_series_sweep_scores_f16x3", F, K, b_rows, h_gap_begin, h_gap_rows, h_n_val, M, &bv, &fv))
        return rc;
    for (int f = 0; f < F; ++f) LC_REQUIRE(h_n_val[f] > 1, LC_E_SHAPE, "lc_series_sweep_scores_f16x3: need n_val > 1");
    const int nblk = M / LC_MB;
    const int Mtiles = (nblk + 1) / 2;                     // two 32-row validation blocks x four terms per tile
    const long long Ntiles = Ncols / TN;
    LC_REQUIRE((long long)F * Mtiles * Ntiles * 2 < (1ll << 31), LC_E_SHAPE, "lc_series_sweep_scores_f16x3: grid too large");
    fv.mt_per_fold = Mtiles;
    fv.part_stride = (long long)nblk * lc::EPI_SERIES_PARTS * V;
    Score16Args sa{d_yv, d_ystat, nullptr, nullptr, d_part, (long long)V, M, 0, LC_SCORE_CORR, Mtiles * TM, 1, d_live_cols};
    Plain16Args pa{};
    pa.rs_inv = d_rowscale_inv;
    pa.cs_inv = d_cscale_inv;
    pa.Mrows = Mtiles * TM;
    pa.G = 1;
    pa.start[0] = 0;
    pa.start[1] = (int)Ntiles;
    pa.slab_light = nullptr;
    {
        lc::ScopedTimer timer_(lc::T_SERIES_SWEEP, s);
        if (terms == 1)
            hipLaunchKernelGGL((k_sweep_hi2<false>), dim3((unsigned)(F * Mtiles * Ntiles * 2)), dim3(H2_THREADS), H2_LDS_BYTES, s,
                               (const uint4*)d_pt, (const uint4*)d_yt, (int)(K / TK), F * Mtiles, sa, pa, bv, fv);
        else if (terms == 101)
            hipLaunchKernelGGL((k_sweep_f16x3<false, false, true, true, false, true>), dim3((unsigned)(F * Mtiles * Ntiles)),
                               dim3(512), LDS16_BYTES, s, (const uint4*)d_pt, (const uint4*)d_yt, (int)(K / TK), F * Mtiles,
                               sa, pa, bv, fv);
        else
            hipLaunchKernelGGL((k_sweep_f16x3<false, false, true, true>), dim3((unsigned)(F * Mtiles * Ntiles)), dim3(512),
                               LDS16_BYTES, s, (const uint4*)d_pt, (const uint4*)d_yt, (int)(K / TK), F * Mtiles, sa, pa, bv, fv);
    }
    if (int rc = lc::launched("k_sweep_f16x3<series moments>")) return rc;
    if (accumulate == 2) return LC_OK;                    // (see lc_series_sweep_finalize_folds)
    return lc_series_finalize_launch(d_part, d_ystat, d_yblk, M, h_n_val, F, (long long)V, d_coef, d_aidx, S, d_scores,
                                     accumulate, s, d_live_cols);
}

extern "C" int lc_series_sweep_finalize_folds(const float* d_part, const float* d_ystat, const float* d_yblk, int F, int M,
                                              const int32_t* h_n_val, int64_t V, const double* d_coef, const int32_t* d_aidx,
                                              int S, float* d_scores, int accumulate, const int32_t* d_live_cols,
                                              lc_stream_t stream) {
    LC_REQUIRE(d_part && d_ystat && d_yblk && d_scores && h_n_val && d_coef && d_aidx, LC_E_BADARG,
               "lc_series_sweep_finalize_folds: null pointer");
    LC_REQUIRE(F >= 1 && F <= MAX_FOLDS16 && M > 0 && M % LC_MB == 0 && V > 0 && V % 128 == 0 && S > 0, LC_E_SHAPE,
               "lc_series_sweep_finalize_folds: bad shape");
    for (int f = 0; f < F; ++f) LC_REQUIRE(h_n_val[f] > 1, LC_E_SHAPE, "lc_series_sweep_finalize_folds: need n_val > 1");
    return lc_series_finalize_launch(d_part, d_ystat, d_yblk, M, h_n_val, F, (long long)V, d_coef, d_aidx, S, d_scores,
                                     accumulate ? 1 : 0, lc::as_stream(stream), d_live_cols);
}
