/*
 * litcoder_hip.h -- C ABI of liblitcoder_hip.so, the MI355X (gfx950) implementation of
 * LITcoder's nested-CV ridge hot path (FIR delay stacking, Lanczos downsampling,
 * the per-alpha ridge solve / prediction contractions, per-voxel correlation scoring).
 *
 * Conventions
 *   - Every pointer named d_* is DEVICE memory owned by the caller (the Python host
 *     allocates it as torch tensors and passes data_ptr()); h_* is host memory.
 *     The library never allocates memory that outlives a call.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All work is
 *     enqueued asynchronously on it; nothing here synchronises unless stated.
 *   - Matrices are row-major; `ld*` is the row stride in ELEMENTS.
 *   - Return value: 0 = ok, <0 = error (LC_E_*); lc_last_error() gives the message
 *     of the calling thread's last failure.
 *   - Row-index lists are int32; an entry of -1 marks a padding row (contributes
 *     nothing).
 *
 * Each entry point cites the reference call site it replaces (paths under
 * /root/reference/encoding/).
 */
#ifndef LITCODER_HIP_H
#define LITCODER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_OK 0
#define LC_E_BADARG (-1)   /* null pointer, negative size, unsupported dtype ... */
#define LC_E_SHAPE (-2)    /* sizes that violate a documented multiple / limit */
#define LC_E_HIP (-3)      /* a HIP runtime call or kernel launch failed */
#define LC_E_NOTSPD (-4)   /* Cholesky met a non-positive pivot (alpha == 0 on a singular Gram) */
#define LC_E_ARCH (-5)     /* device is not gfx950 */

#define LC_F32 0
#define LC_F64 1
#define LC_I32 2           /* (collectives only) */

#define LC_SCORE_CORR 0    /* ridge_regression.py:122-125 */
#define LC_SCORE_R2 1      /* ridge_regression.py:126-130 */

/* Padding granules the batched routines require. */
#define LC_NB 64           /* Gram/Cholesky block: N (train rows, padded) % LC_NB == 0 */
#define LC_MB 32           /* score row-block: M (validation rows, padded) % LC_MB == 0 */

typedef void* lc_stream_t;

int lc_version(void);
const char* lc_last_error(void);
/* A HIP stream restricted to the compute units whose bits are set in mask (words x 32 bits, bit i = CU i of the
 * current device; hipExtStreamCreateWithCUMask).  Destroy with lc_stream_destroy. */
int lc_stream_create_cu_mask(const uint32_t* mask, int words, lc_stream_t* out);
int lc_stream_destroy(lc_stream_t stream);

/* 0 when device `dev` is a gfx950; LC_E_ARCH otherwise. */
int lc_check_device(int dev);

/* Optional per-kernel-class timing (bench.py's roofline leg): when enabled, entry points bracket
 * their launches with HIP events on `stream`.  lc_timing_read synchronises on the recorded events,
 * returns the summed milliseconds / call count of a slot and clears it.  Slot 0 is the fused
 * alpha-sweep MFMA kernel alone. */
int lc_timing_enable(int on);
/* ... for the slots whose bits are set only (bit i = slot i; 0 = off): every timed launch is bracketed by two event
 * records that keep it from overlapping its neighbours on the stream, so a measurement of ONE kernel class inside a timed
 * region (bench.py: slot 0 during the headline fits) should not time the ~900 other launches of a fit as well. */
int lc_timing_enable_slots(uint64_t mask);
int lc_timing_slots(void);
const char* lc_timing_name(int slot);
int lc_timing_read(int slot, double* total_ms, int* calls);

/* ---------------------------------------------------------------- preprocessing */

/* FIR.make_delayed (features/FIR_expander.py:24-43).  d_stim: (nt, ndim) f32 or f64,
 * d_out: (nt, ndim*nd) f64, column block k = stim shifted by h_delays[k] rows;
 * vacated rows zero, or wrapped when circpad.  Bit-exact copies. */
int lc_fir_delay(const void* d_stim, int dtype, int64_t nt, int64_t ndim, int64_t ld_in,
                 const int64_t* h_delays, int nd, int circpad,
                 double* d_out, int64_t ld_out, lc_stream_t stream);

/* lanczosinterp2D (downsample/interpdata.py:87-126) with lanczosfun (:45-63).
 * d_data (n_old, D) f32|f64; d_oldtime (n_old), d_newtime (n_new) f64;
 * cutoff = 1/mean(diff(newtime))*cutoff_mult is computed by the host (interpdata.py:107).
 * d_out (n_new, D) f64, or (n_new, 2*D) = [neg | pos] when rectify. */
int lc_lanczos_interp(const void* d_data, int dtype, int64_t n_old, int64_t D, int64_t ld_in,
                      const double* d_oldtime, const double* d_newtime, int64_t n_new,
                      double cutoff, double window, int rectify,
                      double* d_out, int64_t ld_out, lc_stream_t stream);

/* lanczosinterp2D for ALL stories of a run in one launch (encoding/trainer.py:125-157 + 174-201 call the downsampler
 * once per story).  Inputs and outputs of the stories are concatenated by rows; d_stories: n_stories records
 * {int64 old_off, n_old, new_off; double cutoff; int32 sorted (+ padding)} = 40 bytes each, where cutoff =
 * 1 / mean(diff(newtime_s)) * cutoff_mult (interpdata.py:107) and sorted = 1 when the story's sample times are
 * non-decreasing (only the samples within `window` lobes of an output time are then visited -- same weights, same
 * order of accumulation, same bits as lc_lanczos_interp); d_row_story (n_new_total int32): the story of each output row. */
int lc_lanczos_interp_stories(const void* d_data, int dtype, int64_t D, int64_t ld_in, const double* d_oldtime,
                              const double* d_newtime, int64_t n_new_total, const int32_t* d_row_story,
                              const void* d_stories, int n_stories, double window, int rectify, double* d_out,
                              int64_t ld_out, lc_stream_t stream);

/* The float32 design matrix of a story-structured fit in one launch (encoding/trainer.py:203-209 FIR.make_delayed per
 * story; :235-257 zs(features[start:end]) per story, np.vstack, np.nan_to_num; models/nested_cv.py:99 float32 cast):
 * d_feat = the stories' (downsampled) float64 features concatenated by rows; d_stories: n_stories records of five int64
 * {in_off, n_in, a, b, out_row0} -- trimmed rows [a, b) of the delayed story land in rows out_row0.. of d_x (ldx
 * floats per row, columns k*ndim + c for delay k).  Sums in numpy's order, no fused multiply-adds: the reference's bits. */
int lc_story_design_f32(const double* d_feat, int64_t ndim, int64_t ld_in, const void* d_stories, int n_stories,
                        const int64_t* h_delays, int nd, float* d_x, int64_t ldx, lc_stream_t stream);


/* sincinterp2D (downsample/interpdata.py:66-84) with sincfun (:29-42, array branch): same banded
 * weighted-row-sum kernel as Lanczos with the sinc weight, optional causal mask and per-output-row
 * renormalisation (weights / sum(weights) unless the sum is exactly 0). */
int lc_sinc_interp(const void* d_data, int dtype, int64_t n_old, int64_t D, int64_t ld_in,
                   const double* d_oldtime, const double* d_newtime, int64_t n_new,
                   double cutoff, double window, int causal, int renorm,
                   double* d_out, int64_t ld_out, lc_stream_t stream);

/* The per-TR reducers (downsample/downsampling.py:24-136,180-319: rect, average, sum, last and the legacy_*
 * chunk variants): out[s] = mean (0) | sum (1) | last (2) of rows d_idx[d_seg[s] .. d_seg[s+1]) of d_data,
 * zero for an empty segment.  The host turns time windows / TR labels / split points into (d_seg, d_idx). */
int lc_segment_reduce(const void* d_data, int dtype, int64_t D, int64_t ld_in, const int64_t* d_seg,
                      const int32_t* d_idx, int64_t n_seg, int how, double* d_out, int64_t ld_out,
                      lc_stream_t stream);

/* ---------------------------------------------------------------- the reference's SVD semantics (slow fp64 route)
 * ridge_utils.py:34-67 drops singular values <= singcutoff of the thin SVD of a training block; ridge_regression.py:
 * 56,117 shrink the kept ones by S / (S^2 + a^2), which is defined at a = 0.  In terms of K = X X' (eigenpairs
 * lambda = S^2, U) the operators are  R U_k diag(1 / (lambda_k + a^2)) U_k'  with R = K[va,tr] (hat matrix) or
 * R = [Xtr' ; K[te,tr]] (refit).  Taken only for penalty grids the Cholesky route cannot serve (alpha = 0, or a
 * singcutoff that is not negligible against the smallest penalty).
 * lc_batch_eigh_jacobi: F symmetric (n, n) fp64 systems d_a (n even: pad an odd system with a zero row and column;
 * destroyed) -> d_lam (F, n) eigenvalues, d_vt (F, n, n) eigenvectors AS ROWS, d_lmax (F, optional) the largest;
 * cyclic Jacobi, parallel ordering, at most max_sweeps sweeps to the relative off-diagonal tolerance tol (host
 * synchronisation once per sweep; *h_sweeps = sweeps done).
 * lc_batch_spectral_apply: d_h[h_slot ? h_slot[f A + a] : f A + a] (M, n) f32 = R_f V_kept diag(1/(lambda + a2[f A +
 * a])) V_kept', an eigenpair being kept when sqrt(lambda) > cutoff AND it is among the d_rank_cap[f] largest (the thin
 * SVD of an (n x p) block has min(n, p) values; NULL: no cap); d_kept (F, optional) = how many were. */
int64_t lc_batch_eigh_work_bytes(int F, int n);
int lc_batch_eigh_jacobi(double* d_a, int F, int n, double* d_vt, double* d_lam, double* d_lmax, void* d_work,
                         int64_t work_bytes, int max_sweeps, double tol, int32_t* h_sweeps, lc_stream_t stream);
int64_t lc_batch_spectral_work_bytes(int F, int A, int n, int M);
int lc_batch_spectral_apply(const double* d_lam, const double* d_vt, int F, int n, const double* d_r, int M,
                            const double* d_a2, int A, double cutoff, const int32_t* d_rank_cap, void* d_work,
                            int64_t work_bytes, float* d_h, const int32_t* h_slot, int32_t* d_kept, lc_stream_t stream);

/* ---------------------------------------------------------------- casts / gathers */

/* ---- host <-> device boundary of a fit (nested_cv.py:99-100: `torch.tensor(features / targets, dtype=float32)`;
 * :293-296: the float32 weights returned as a host array).  Targets and weights move as column panels of voxels.
 * lc_memcpy2d_async: `rows` rows of `width_bytes` with byte pitches, kind 0 = H2D, 1 = D2H, 2 = D2D, asynchronous on
 * `stream` when the host side is page-locked.  lc_fill2d_bytes: the same shape set to a byte value (zero padding).
 * lc_host_cast_f64_f32 / lc_host_copy_f32: HOST-side staging of a (rows, cols) block of the caller's pageable array
 * into page-locked memory, float64 -> float32 (round to nearest even, the cast the reference performs on the host) or
 * a plain copy; no device work, callable from any thread. */
int lc_memcpy2d_async(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes,
                      int64_t rows, int kind, lc_stream_t stream);
int lc_fill2d_bytes(void* d_ptr, int64_t pitch, int byte, int64_t width_bytes, int64_t rows, lc_stream_t stream);
int lc_host_cast_f64_f32(const double* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t cols);
int lc_host_copy_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t cols);

/* The whole upload of a fit's inputs on native threads (nested_cv.py:99-100 for features AND targets): a list of jobs,
 * each a column range [c0, c1) of a row block of a host matrix (float32 or float64, pageable) going to the same
 * columns of rows [dst_row0, dst_row0 + rows) of a float32 device matrix.  lc_upload_start returns at once; worker
 * threads stage row chunks into the caller's page-locked slots (n_slots buffers of slot_bytes; float64 is cast to
 * float32 there) and issue the copies on `stream` in job order.  lc_upload_wait blocks the calling host thread until
 * every copy of `job` has been issued and makes `consumer` wait for them on the device.
 * lc_upload_finish joins the threads and leaves the slots idle (the job sources, the slots and the destinations must
 * stay alive until it returns); lc_upload_free releases the handle once no thread waits on it any more. */
typedef struct lc_upload_job {
    const void* src;   /* host matrix block, row-major */
    int64_t ld_src;    /* elements between its rows */
    int dtype;         /* LC_F32 | LC_F64 */
    int64_t rows;      /* rows of the block */
    int64_t c0, c1;    /* columns copied */
    void* dst;         /* device float32 matrix (row 0, column 0) */
    int64_t ld_dst;    /* elements between its rows */
    int64_t dst_row0;  /* destination row of the block's first row */
    int transform;     /* LC_UPLOAD_CAST: the cast alone; LC_UPLOAD_ZSCORE: the block is ONE story and is z-scored on the way
                        * (encoding/utils.py:23-29 zs as trainer.py:235-257 applies it per story: population std, zero-std
                        * columns only de-meaned, in the block's own precision, numpy's summation order -- bit-identical to
                        * the reference's zs followed by its float32 cast) */
} lc_upload_job;
enum { LC_UPLOAD_CAST = 0, LC_UPLOAD_ZSCORE = 1 };
/* HOST code, no device: the z-scoring of one story block exactly as a LC_UPLOAD_ZSCORE job stages it -- float32
 * out[r, c] = fl32(zs(src)[r, c]) (utils.py:23-29 + nested_cv.py:99-100) -- for callers and tests without a GPU. */
int lc_host_zscore_story(const void* src, int dtype, int64_t ld_src, int64_t rows, int64_t cols, float* out,
                         int64_t ld_out);
typedef struct lc_upload lc_upload_t;
int lc_upload_start(const lc_upload_job* jobs, int n_jobs, void* const* pinned_slots, int n_slots, int64_t slot_bytes,
                    int n_threads, int device, lc_stream_t stream, lc_upload_t** out);
/* The same with device staging slots (n_slots device buffers of slot_bytes): a chunk then crosses the link as ONE
 * contiguous copy and is laid out into the panel's columns by a device-to-device 2-D copy -- a strided copy across PCIe
 * pays per row (measured: 30-38 GB/s for 40-48 KB rows against 57 GB/s contiguous). */
int lc_upload_start_staged(const lc_upload_job* jobs, int n_jobs, void* const* pinned_slots, void* const* device_slots,
                           int n_slots, int64_t slot_bytes, int n_threads, int device, lc_stream_t stream,
                           lc_upload_t** out);
int lc_upload_wait(lc_upload_t* upload, int job, lc_stream_t consumer);
int lc_upload_finish(lc_upload_t* upload);
int lc_upload_free(lc_upload_t* upload);

/* torch.tensor(x, dtype=float32) (models/nested_cv.py:99-100): f64 -> f32, (rows, cols)
 * into a (possibly wider, zero-padded by the caller) destination. */
int lc_cast_f64_f32(const double* d_in, int64_t ld_in, float* d_out, int64_t ld_out,
                    int64_t rows, int64_t cols, lc_stream_t stream);

/* out[r, j] = in[rows[r], cols ? cols[j] : j]   (nested_cv.py:200-201 row splits, plus the
 * alpha-sorted voxel order used by the grouped refit).  rows[r] == -1 or cols[j] == -1 -> 0.
 * d_live_cols (NULL: all; here and in lc_val_stats_folds / lc_col_scales_f16_flags / lc_split_cols_f16): device int32, read
 * by the kernel -- only the first *d_live_cols columns (rounded up to whole 256-column tiles) are touched: the screening
 * pass' refinement panel has a fixed capacity, how many undecided voxels it holds is known on the device only (round 6:
 * every pass over the panel costs what its voxels cost, not what its capacity would). */
int lc_gather_f32(const float* d_in, int64_t ld_in, const int32_t* d_rows, int64_t n_rows,
                  const int32_t* d_cols, int64_t n_cols, float* d_out, int64_t ld_out,
                  const int32_t* d_live_cols, lc_stream_t stream);

/* acc[:, cols[j]] += scale * w[:, j]  (np.mean(fold_weights) accumulated on device,
 * nested_cv.py:249,296); cols[j] == -1 skipped. */
int lc_scatter_axpy_f32(const float* d_w, int64_t ld_w, int64_t n_rows, const int32_t* d_cols,
                        int64_t n_cols, float scale, float* d_acc, int64_t ld_acc,
                        lc_stream_t stream);
/* d_dst[r, d_cols[j]] = d_src[r, j] for 4- or 8-byte elements (an overwrite; d_cols[j] < 0 skipped; the indices distinct):
 * the columns of the f32 side path back into the main path's score / weight / result matrices. */
int lc_scatter_cols(const void* d_src, int64_t ld_src, int64_t n_rows, int elem_bytes, const int32_t* d_cols,
                    int64_t n_cols, void* d_dst, int64_t ld_dst, lc_stream_t stream);
/* The same side path when it holds a handful of columns (ns <= 8; one outlier voxel is the usual case): the products of
 * ridge_regression.py:46-61, 104-120 for those columns alone, streamed instead of tiled --
 *     d_c[m, j] = sum_k d_a[m, k] * d_y[d_rows[k] * ldy + j]      (m < M, k < K; d_rows[k] < 0: 0; d_rows NULL: row k)
 * for the columns j < ns with d_sel[j] == want (d_sel NULL: all; the other columns of d_c are left alone), f32 inputs,
 * fp64 accumulation, f32 result.  K % 4 == 0, lda % 4 == 0, d_a 16-byte aligned. */
int lc_gemv_cols_f32(const float* d_a, int64_t lda, int64_t M, int64_t K, const float* d_y, int64_t ldy,
                     const int32_t* d_rows, const int32_t* d_sel, int ns, int32_t want, float* d_c, int64_t ldc,
                     lc_stream_t stream);

/* The same mean over folds (nested_cv.py:249,293-296) without a read-modify-write of the accumulator per fold: each
 * fold keeps its alpha-sorted weight matrix, lc_invert_perm notes where every voxel's column went
 * (d_pos[d_cols[j]] = base + j for d_cols[j] >= 0), and lc_combine_folds_f32 forms, for one voxel range,
 *     d_out[r, v] = sum_f scale[f] * w[f][r * ld_w[f] + pos[f][v]]      (folds in order; pos < 0 skipped)
 * with the same per-term expression as lc_scatter_axpy_f32 (bit-identical to accumulating fold by fold from zero).
 * w / ld_w / pos / scale are HOST arrays of n_folds entries (device pointers inside). */
int lc_invert_perm(const int32_t* d_cols, int64_t n_cols, int32_t base, int32_t* d_pos, lc_stream_t stream);
int lc_combine_folds_f32(const float* const* w, const int64_t* ld_w, const int32_t* const* pos, const float* scale,
                         int n_folds, int64_t n_rows, int64_t n_cols, float* d_out, int64_t ld_out,
                         lc_stream_t stream);

/* ---------------------------------------------------------------- column statistics */

/* DataNormalizer.fit (models/ridge_utils.py:94-125): per-column mean and UNBIASED std of
 * the rows listed in d_rows (NULL = rows 0..n-1).  d_mean/d_std: (n_cols) f32. */
int lc_col_mean_std_f32(const float* d_x, int64_t ld, const int32_t* d_rows, int64_t n_rows,
                        int64_t n_cols, float* d_mean, float* d_std, lc_stream_t stream);

/* DataNormalizer.transform (ridge_utils.py:127-166): x = (x - mean) / (std + eps), in place,
 * all n_rows rows. */
int lc_col_normalize_f32(float* d_x, int64_t ld, int64_t n_rows, int64_t n_cols,
                         const float* d_mean, const float* d_std, float eps, lc_stream_t stream);

/* The trainer's per-story z-score (utils.py:23-29 ``zs``; trainer.py:235-257): float64, population std,
 * zero-std columns only de-meaned; nan_to_num != 0 also applies np.nan_to_num to the result (features). */
int lc_zscore_story_f64(const double* d_x, int64_t ld_in, int64_t rows, int64_t cols, int nan_to_num,
                        double* d_out, int64_t ld_out, lc_stream_t stream);

/* scipy.stats.pearsonr per voxel (nested_cv.py:418-438): d_a, d_b (n, V) f32 -> d_r (V) f64
 * Pearson r, NaN (constant column) reported as NaN; the host maps NaN -> 0 like the
 * reference.  fp64 accumulation. */
int lc_pearson_cols(const float* d_a, int64_t lda, const float* d_b, int64_t ldb,
                    int64_t n, int64_t V, double* d_r, lc_stream_t stream);

/* The same r with operand a read THROUGH a row list and a column list: a[i][c] = d_y[d_rows[i] * ld_y + d_cols[c]]
 * (d_cols NULL: c; an entry < 0: a zero column) -- the test rows of the alpha-sorted voxels straight from the resident
 * targets, no gathered copy (nested_cv.py:152-155 on y_test[:, i]).  n <= 640 rows (held in registers). */
int lc_pearson_cols_gather(const float* d_y, int64_t ld_y, const int32_t* d_rows, const int32_t* d_cols,
                           const float* d_b, int64_t ldb, int64_t n, int64_t V, double* d_r, lc_stream_t stream);

/* The p-value scipy.stats.pearsonr attaches to r for n samples (nested_cv.py:434-436): two-sided, from the
 * Beta(n/2-1, n/2-1) null distribution, evaluated on the float32-rounded r like scipy does for float32
 * inputs; NaN r -> 1.  d_r, d_p: (V) f64. */
int lc_pearson_pvalues(const double* d_r, int64_t V, int64_t n, double* d_p, lc_stream_t stream);

/* ---------------------------------------------------------------- small dense fp64 */

/* K = X X' with fp64 accumulation; X (T, p) f32, K (T, T) f64.  Replaces the SVD of the
 * design matrix (ridge_utils.py:52): every fold's train Gram and validation cross-Gram
 * are row/column subsets of this one matrix. */
int lc_gram_f64(const float* d_x, int64_t ldx, int64_t T, int64_t p, double* d_k, int64_t ldk,
                lc_stream_t stream);
/* The same matrix through the fp64 MFMA (the deep-update kernel of lc_batch_chol_solve as one "N x T" product of the
 * widened design, lower-triangle tiles computed and mirrored): 3x the rate of the vector-ALU kernel, equal to it to
 * the rounding of the fp64 sums.  d_work: T * pad16(p) doubles. */
int lc_gram_f64_mfma(const float* d_x, int64_t ldx, int64_t T, int64_t p, double* d_work, double* d_k,
                     int64_t ldk, lc_stream_t stream);

/* S[0]^2 of a fold's design matrix (ridge_regression.py:39,97 `norm = S[0].item()`) = the largest eigenvalue of a Gram
 * block, by `steps` Lanczos iterations + bisection.  THREE entry points since round 6 (ten variants before), one per
 * way the systems are given:
 *   lc_lambda_max         F row lists (each N entries, -1 padded) into matrix f at d_k + f k_stride (k_stride = 0: all
 *                         lists index ONE matrix, the dual form's K; rows_per^2: the primal form's Gram block of fold f);
 *   lc_lambda_max_masked  up to 32 row sets of one matrix as membership bits, one pass over K per iteration;
 *   lc_lambda_max_dense   the leading n x n blocks of F matrices, streaming matvec.
 * d_work: F*(3*N + 2*steps + 8) f64.  d_lmax: (F) f64. */
int lc_lambda_max(const double* d_k, int64_t ldk, int64_t k_stride, const int32_t* d_rows, int F, int N, int steps,
                  double* d_work, double* d_lmax, lc_stream_t stream);

/* The same for up to 32 row sets of ONE Gram matrix in a single pass over K per iteration: bit f of
 * d_member[i] (T x uint32) says whether row i belongs to system f.  Every fold of a nested CV (outer
 * train sets and their inner train sets) is a principal submatrix of K, so one call serves the whole fit.
 * d_work: F*(3*T + 2*steps + 8) + 16*32*T f64 (the matvec -- K times the 32 systems' vectors on the fp64 MFMA -- is
 * split over up to 16 column ranges whose partial sums are added in fixed order).  d_lmax: (F) f64.
 * use_mfma = 0 takes the vector-ALU matvec of round 1 (a per-call choice: the library has no process-wide switches).
 * tol: a convergence stop (round 5): `steps` is the most a system runs; every 8 steps from the 24th on its top Ritz
 * value is looked at, and the system stops once that value has moved by <= tol (relative) over the last 8 steps (tol = 0:
 * never -- the default of every caller: a run short of the top eigenvector can sit on the SECOND eigenvalue for a while,
 * which such a stop takes for convergence; profiles/r05_lanczos_stop_misfire.txt). */
int lc_lambda_max_masked(const double* d_k, int64_t ldk, int T, const uint32_t* d_member, int F, int steps, double tol,
                         double* d_work, double* d_lmax, int use_mfma, lc_stream_t stream);
/* Systems that are the LEADING n x n blocks of their own matrices (the primal form's p x p Gram matrices: matrix f at
 * d_k + f k_stride, no row lists): a streaming matvec (16-byte loads, the vector through LDS), the same recurrence, the
 * same stop.  N = padded vector length (>= n, even); d_work: F*(3*N + 2*steps + 8) f64, 16-byte aligned like d_k. */
int lc_lambda_max_dense(const double* d_k, int64_t ldk, int64_t k_stride, int F, int N, int n, int steps, double tol,
                        double* d_work, double* d_lmax, lc_stream_t stream);

/* a2[f*A + a] = (alphas[a] * (normalpha ? sqrt(lmax[f]) : 1))^2
 * (ridge_regression.py:99-101,117: D = S/(S^2 + nalpha^2)). */
int lc_penalties(const double* d_lmax, int F, const double* d_alphas, int A, int normalpha,
                 double* d_a2, lc_stream_t stream);

/* Build B = F*A augmented systems, batch b = f*A + a:
 *   top    (N x N): K[tr_f, tr_f] + a2[b] I      (padding rows/cols -> identity)
 *   bottom (M x N): K[va_f, tr_f]                (d_rhs == NULL; padding -> 0)
 *                   or d_rhs[f] (M x N f64 panel, ld N) when d_rhs != NULL
 * d_aug: (B, N+M, N) f64.  d_tr: (F, N) int32, d_va: (F, M) int32 (ignored with d_rhs). */
int lc_batch_assemble(const double* d_k, int64_t ldk, const int32_t* d_tr, const int32_t* d_va,
                      const double* d_rhs, const double* d_a2, int F, int A, int N, int M,
                      double* d_aug, lc_stream_t stream);

/* The same for a SUBSET of that grid (voxel shards: the (fold, alpha) systems are dealt out over the ranks, each
 * solves its share and the f32 results are all-gathered): system j of the batch is grid system s = d_sys[j]
 * (int32, B entries), built from fold s / A and penalty d_a2[s].  d_aug: (B, N+M, N). */
int lc_batch_assemble_sel(const double* d_k, int64_t ldk, int64_t k_fold_stride, const int32_t* d_tr,
                          const int32_t* d_va, const double* d_rhs, const double* d_a2, const int32_t* d_sys,
                          int B, int A, int N, int M, double* d_aug, lc_stream_t stream);

/* Primal form of the same ridge systems, for p < n/2 (SURVEY 8d "for p<n use the primal form"; what the reference's
 * thin SVD of a tall Rstim, ridge_utils.py:52, is by construction):  pred = Pstim (Rstim'Rstim + a^2 I)^-1 Rstim'Rresp.
 *   lc_gather_transpose_f32: d_out (F * p_pad, N) f32, block f = the rows d_rows[f][0..N) (int32, -1 = zero column)
 *                            of X (ldx, p columns), transposed and zero-padded to p_pad rows: Rstim' of training set f;
 *   lc_gram_blocks_f64:      d_g (n_blocks, rows_per, rows_per) f64, block b = Xt_b Xt_b' with Xt_b rows
 *                            [b rows_per, (b+1) rows_per) of d_xt (ld ldx, `depth` columns): Rstim'Rstim of every set;
 *   lc_gather_rows_f64:      d_out (F, M, N) f64, row i of block f = row d_rows[f][i] of X (p columns, zero padded to
 *                            N); -1 -> zero row; -(2 + c) -> unit row e_c: the augmented rows Pstim (inner folds) /
 *                            [I ; X_test] (refit: weights operator above the test-row hat matrix);
 *   lc_batch_assemble_sel with k_fold_stride = rows_per^2 then takes the top block of fold f from ITS Gram matrix
 *   (0 = one shared matrix, the dual form), and lc_lambda_max's k_stride does the same for S[0]^2. */
int lc_gather_transpose_f32(const float* d_x, int64_t ldx, const int32_t* d_rows, int F, int N, int p,
                            int p_pad, float* d_out, lc_stream_t stream);
int lc_gram_blocks_f64(const float* d_xt, int64_t ldx, int n_blocks, int rows_per, int depth, double* d_g,
                       lc_stream_t stream);
int lc_gather_rows_f64(const float* d_x, int64_t ldx, const int32_t* d_rows, int F, int M, int p, int N,
                       double* d_out, lc_stream_t stream);

/* Primal form for a handful of features, p <= 16 (csrc/lc_primal.hip): every statistic the nested CV takes from a
 * prediction X w is a linear / quadratic form in w = (G + a^2 I)^-1 Rstim'y, so the V-wide work of an outer fold is one
 * pass over the targets (what ridge_regression.py:104-133 and nested_cv.py:150-155 compute through V-wide products).
 * Row sets: d_rows (n_sets, ldr) int32, the first d_nrows[s] entries of row s valid; d_shrow[s] = the row of Y whose
 * values are subtracted from the targets of set s (all sets of one outer fold share it).  p_pad = lc_primal_pad(p) in
 * {4, 8, 16}.
 *   lc_xty_f64          d_part ((n_sets, RS, p_pad + 2, V) f64): per chunk of `chunk` rows  X'(Y - shift), the sum of
 *                       the shifted targets and of their squares; slots beyond a set's rows are left untouched;
 *   lc_primal_set_stats d_xstat ((n_sets, p_pad + 2 p_pad^2) f64) = [column sums | X'X | centred scatter] per set;
 *   lc_primal_gsys      d_gsys ((n_sys, p_pad, p_pad) f64) = X'X[plus] - X'X[minus], d_sysdef (n_sys, 2) int32 =
 *                       (plus, minus or -1): the Gram matrix of a training set given as a block minus a sub-block;
 *   lc_primal_inverse   d_pinv ((n_sys * A, p_pad, p_pad) f64) = (G_sys + d_a2[sys * A + a] I)^-1 by Cholesky (padding
 *                       rows: identity); d_info (n_sys * A) nonzero = failed pivot;
 *   lc_primal_scores    d_scores (A, lds) f32 = sum over the F inner folds of the correlation score of every alpha
 *                       (fp32 sum in fold order; columns >= V zeroed); d_src (F, 3) int32 = (plus, minus or -1,
 *                       validation) set numbers; d_pinv: system f * A + a;
 *   lc_primal_refit     W[:, c] += scale * fl32(pinv[best[c]] Rstim'y) over set_train (d_pinv: the A systems of the
 *                       outer training set) and d_r[c] = Pearson r of the prediction with the targets over set_test. */
int lc_primal_pad(int p);
int lc_xty_f64(const float* d_x, int64_t ldx, int p, const float* d_y, int64_t ldy, int64_t V,
               const int32_t* d_rows, int ldr, const int32_t* d_nrows, const int32_t* d_shrow, int n_sets,
               int chunk, int RS, double* d_part, lc_stream_t stream);
int lc_primal_set_stats(const float* d_x, int64_t ldx, int p, const int32_t* d_rows, int ldr,
                        const int32_t* d_nrows, int n_sets, double* d_xstat, lc_stream_t stream);
int lc_primal_gsys(const double* d_xstat, const int32_t* d_sysdef, int n_sys, int p, double* d_gsys,
                   lc_stream_t stream);
int lc_primal_inverse(const double* d_gsys, const double* d_a2, int n_sys, int A, int p, double* d_pinv,
                      int32_t* d_info, lc_stream_t stream);
int lc_primal_scores(const double* d_part, int RS, int chunk, const int32_t* d_nrows, const int32_t* d_shrow,
                     const float* d_y, int64_t ldy, int64_t V, const int32_t* d_src, const double* d_xstat,
                     const double* d_pinv, int F, int A, int p, float* d_scores, int64_t lds,
                     lc_stream_t stream);
int lc_primal_refit(const double* d_part, int RS, int chunk, const int32_t* d_nrows, const int32_t* d_shrow,
                    const float* d_y, int64_t ldy, int64_t V, int set_train, int set_test,
                    const double* d_xstat, const double* d_pinv, const int32_t* d_best, int p, float scale,
                    float* d_w, int64_t ldw, double* d_r, lc_stream_t stream);

/* In place on every (N+M, N) system: Cholesky of the top block, then bottom <- bottom * inv(top)
 * i.e. the hat matrices  Xva Xtr' (Xtr Xtr' + a^2 I)^-1  (= Pstim Vh' diag(S/(S^2+a^2)) U',
 * ridge_regression.py:104-105,117-120).  Result written as f32 to d_h (B, M, N).
 * d_linv: workspace (B, N/LC_NB, LC_NB, LC_NB) f64.  d_info: (B) int32, nonzero = failed pivot.
 * d_slot: optional (B) int32, system b is written to slot d_slot[b] of d_h (NULL = b).
 * opt: per-call variants (lc_chol_options below; NULL = the defaults). */
struct lc_chol_options;
int lc_batch_chol_solve(double* d_aug, int B, int N, int M, double* d_linv, float* d_h, const int32_t* d_slot,
                        int32_t* d_info, const struct lc_chol_options* opt, lc_stream_t stream);

/* The explicit inverse of the same top blocks: d_aug (B, 2N, N) f64 with the N x N IDENTITY as bottom block (the
 * caller assembles it like any other right-hand side); d_p (B, N, N) f32 <- (top)^-1.  Same kernels; the rows of
 * I L^-T L^-1 are independent, row tile r is zero left of block column r after the first pass and only needed from block
 * column r on after the second (symmetry), so both passes skip the other tiles: N^3 flops in all instead of
 * N^3/3 + 2 N^3.  The refit applies  [Xtr' ; K[te,tr]] (K + a^2 I)^-1  (ridge_regression.py:56-61, nested_cv.py:151) as a
 * product with this inverse on the fp16x3 MFMA where its accuracy allows. */
int lc_batch_chol_inverse(double* d_aug, int B, int N, double* d_linv, float* d_p, const int32_t* d_slot,
                          int32_t* d_info, const struct lc_chol_options* opt, lc_stream_t stream);

/* The variants of lc_batch_chol_solve / lc_batch_chol_inverse as PER-CALL options (NULL = the defaults); the library
 * keeps no process-wide switches, so fits with different settings coexist in one process:
 *   outer_block  columns per outer block of the two-level blocking (a multiple of LC_NB; default 512)
 *   big_kernel   kernel of the deep updates: 2 = 4x4x4 fp64 MFMA (default: 128 x 128 tiles, 64 x 64 ones for small batches --
 *                the same bits), 3 = the 64 x 64 tiles always, 4 = the 128 x 128 ones always, 1 = vector ALU, 0 = 16x16x4 fp64 MFMA
 *   fused_steps  1 (default) = fused left-looking 64-column steps, 0 = the first version's panel + update launches
 *   left_deep    1 = the deep updates left-looking too (measured: no gain), 0 (default) = right-looking
 *   persistent   bit 0 (default on): the steps of the back substitution inside an outer block as ONE launch (a row tile's
 *                steps depend on no other workgroup); 0 = one launch per step */
typedef struct lc_chol_options {
    int outer_block, big_kernel, fused_steps, left_deep, persistent;
} lc_chol_options;
/* (round 6: the option pointer is a parameter of the two entry points above; the *_opt twins are gone) */

/* The same hat matrices for alphas whose penalty dwarfs the spectrum, as a polynomial in K[tr,tr]:
 *   K[va,tr] (K[tr,tr] + a^2 I)^-1  ~=  sum_{j<terms} c_sj K[va,tr] K[tr,tr]^j / scale_f^(j+1)
 * with a = alpha_s * sqrt(scale_f) (normalpha: scale_f = lambda_max of the fold) and c_s the coefficients of a
 * polynomial close to 1 / (x + alpha_s^2) on [0, 1] (d_coef: (S, terms) f64; Neumann/Taylor (-1)^j alpha^-2(j+1), or
 * the minimax ones of litcoder_core_amd/series.py).  The matrix powers are shared by the S alphas of a fold.
 * Output goes to slot f*A + d_aidx[s] of d_h (F*A, M, N) f32.  d_work: F*N*N + terms*F*M*N doubles. */
int lc_batch_series_hat(const double* d_k, int64_t ldk, const int32_t* d_tr, const int32_t* d_va,
                        int F, int N, int M, const double* d_scale, const double* d_coef,
                        const int32_t* d_aidx, int S, int A, int terms, double* d_work, float* d_h,
                        lc_stream_t stream);

/* The shared matrix powers of that series themselves, scaled:  P'_j = K[va,tr] K[tr,tr]^j / scale_f^(j+1),
 * j < terms <= 8, as f32 in d_p (F, terms, M, N).  With T_j = P'_j Y the prediction of every series alpha is
 * sum_j c_sj T_j (coefficients as in lc_batch_series_hat): ONE (terms*M x N x V) contraction serves all those alphas
 * (lc_series_scores).  d_scale: (F) f64 (lambda_max under normalpha).  d_work: F*N*N + terms*F*M*N doubles.
 * d_rowmap: optional (terms*M) int32, row of d_p (F, rows_p, N) that receives row i of term j (entry j*M + i);
 * NULL = j*M + i with rows_p = terms*M.  Rows of d_p that no entry names are left untouched (callers zero them). */
int lc_batch_series_terms(const double* d_k, int64_t ldk, const int32_t* d_tr, const int32_t* d_va,
                          int F, int N, int M, const double* d_scale, int terms,
                          double* d_work, float* d_p, const int32_t* d_rowmap, int rows_p, lc_stream_t stream);

/* Small data movers, so that a fit launches no framework kernels (everything on the caller's stream):
 *   lc_fill_bytes             hipMemsetAsync (zero-initialised buffers; 0xFF = the -1 padding of index lists);
 *   lc_gather_sub_f32_strided lc_gather_sub_f32 with explicit output strides (elements): the series chain keeps its
 *                             operand as (N, folds, M) so that all folds are column groups of one grouped GEMM;
 *   lc_series_place           P[f][rowmap[i]][n] = Q[n][f][i]: term j of that chain into the stacked, slab-padded
 *                             operand of the plain fp16x3 contraction (d_rowmap: M entries of term j, -1 = skip);
 *   lc_scale_cast_f64_f32     dst = (float)(src / divisor[0])  (refit rows over lambda_max for the polynomial route);
 *   lc_combine_terms_f32      out = c0 T0 + c1 T1 + ... (<= 4 terms, left to right in fp32, no contraction): the
 *                             polynomial form of  [Xtr' ; K_te](K + a^2 I)^-1  for one alpha from the shared powers. */
int lc_fill_bytes(void* d_ptr, int byte, int64_t nbytes, lc_stream_t stream);
int lc_gather_sub_f32_strided(const double* d_k, int64_t ldk, const int32_t* d_rows, const int32_t* d_cols,
                              int F, int R, int C, const double* d_scale, float* d_out, int64_t s_f,
                              int64_t s_r, int64_t s_c, lc_stream_t stream);
int lc_series_place(const float* d_q, int N, int F, int ldq, int M, const int32_t* d_rowmap, float* d_p,
                    int rows_p, lc_stream_t stream);
int lc_scale_cast_f64_f32(const double* d_src, const double* d_divisor, float* d_dst, int64_t n,
                          lc_stream_t stream);
int lc_combine_terms_f32(const float* const* h_terms, const float* h_coef, int terms, float* d_out, int64_t n,
                         lc_stream_t stream);
/* The same in float64: the p x p Gram matrices of the primal form's training sets as sums / differences of the
 * validation blocks' (an inner training set = the outer block minus its validation block, nested_cv.py:366-374). */
int lc_combine_terms_f64(const double* const* h_terms, const double* h_coef, int terms, double* d_out, int64_t n,
                         lc_stream_t stream);

/* out[f][i][j] = K[rows[f][i], cols[f][j]] in fp64 (index -1 -> 0): d_out (F, R, C) contiguous.  The test-row block
 * K[te, tr] of the refit's augmented rows (the hat matrix of the test rows, nested_cv.py:151,251). */
int lc_gather_sub_f64(const double* d_k, int64_t ldk, const int32_t* d_rows, const int32_t* d_cols, int F,
                      int R, int C, double* d_out, lc_stream_t stream);

/* out[f] (R, C) f32 = K[rows_f, cols_f] / scale[f] (d_scale NULL = 1; index -1 = zero row / column): the
 * f32 operands of the series terms when their chain  P'_j = P'_(j-1) (K[tr,tr] / scale)  runs on the f32 MFMA
 * (lc_gemm_grouped_f32) instead of in fp64 -- terms j >= 1 enter a prediction scaled by rho^j <= 1.6e-2 per
 * step, so fp32 products with fp32 accumulation leave the result at full fp32 accuracy. */
int lc_gather_sub_f32(const double* d_k, int64_t ldk, const int32_t* d_rows, const int32_t* d_cols,
                      int F, int R, int C, const double* d_scale, float* d_out, lc_stream_t stream);

/* rhs[f] (p x N) f64 <- X[tr_f]' for the refit systems (rows of X listed in d_tr, -1 -> 0). */
int lc_transpose_rows_f64(const float* d_x, int64_t ldx, const int32_t* d_tr, int N, int64_t p,
                          double* d_out, lc_stream_t stream);

/* ---------------------------------------------------------------- the V-wide contractions */

/* Column statistics of the validation targets of one inner fold, feeding the fused scorer:
 * d_ystat (3, V) f32 = [mean, unbiased std, unbiased var] of Y[va_rows] (z_score / .var,
 * ridge_regression.py:108,111); d_yblk (M/LC_MB, V) f32 = per-32-row-block sums of (y - mean);
 * d_yv (M, V) f32 = the validation rows gathered contiguously (zero padding rows) for the sweep epilogue. */
int lc_val_stats(const float* d_y, int64_t ldy, int64_t V, const int32_t* d_va, int M, int n_val,
                 float* d_ystat, float* d_yblk, float* d_yv, lc_stream_t stream);
/* The same for F (<= 64) inner folds in ONE launch: d_va (F, M), h_n_val (F, host), outputs (F, 3, V) / (F, M/32, V) /
 * (F, M, V).  The five validation blocks of an outer fold are independent; one launch instead of five. */
int lc_val_stats_folds(const float* d_y, int64_t ldy, int64_t V, const int32_t* d_va, int F, int M,
                       const int32_t* h_n_val, float* d_ystat, float* d_yblk, float* d_yv,
                       const int32_t* d_live_cols, lc_stream_t stream);

/* Fused alpha sweep of one inner fold (ridge_regression.py:115-133, K4+K5 of SURVEY 2.2):
 *   pred_a = H_a (M x N) . Y[tr_rows] (N x V)    for a = 0..A-1, never stored;
 *   score[a, v] = mean(z(Yva) z(pred_a))  or  signed sqrt|R2|,  NaN -> 0,
 *   d_scores[a, v] (+)= score     (accumulate != 0 adds: the inner-fold mean, nested_cv.py:391).
 * d_h (A*M, N) f32 hat matrices (row stride N); d_part workspace (A*M/LC_MB, 4, V) f32. */
int lc_alpha_sweep_scores(const float* d_h, int A, int M, int N,
                          const float* d_y, int64_t ldy, int64_t V,
                          const int32_t* d_tr, const float* d_yv, int n_val,
                          const float* d_ystat, const float* d_yblk, int mode,
                          float* d_part, float* d_scores, int accumulate, lc_stream_t stream);

/* ---- split-precision variant of the sweep (fp16 hi + lo operands, three fp16 MFMAs per product,
 * fp32 accumulate; fp32-level accuracy at ~5x the f32-MFMA rate).  Same contract and outputs as
 * lc_alpha_sweep_scores; operands are prepared once per fold by the three helpers below. ---- */

/* H (rows, K) f32 -> tiled fp16 hi/lo image (pad256(rows) * K * 2 halves) with an exact power-of-two
 * scale per row; d_rowscale_inv (pad256(rows)) receives 2^e undoing it.  K % 32 == 0. */
int lc_split_rows_f16(const float* d_h, int64_t ld, int64_t rows, int64_t K, void* d_tiled,
                      float* d_rowscale_inv, lc_stream_t stream);
/* The same for `groups` row blocks of `rows` rows each that follow one another in d_h (the inner folds' hat matrices /
 * series terms): every group is padded to whole 256-row tiles in the image and in d_rowscale_inv, one launch. */
int lc_split_rows_f16_groups(const float* d_h, int64_t ld, int groups, int64_t rows, int64_t K, void* d_tiled,
                             float* d_rowscale_inv, lc_stream_t stream);
/* The image of the MEAN of the folds' refit operators (round 6; replaces the per-fold weight products of ridge_torch,
 * ridge_regression.py:46-61, followed by np.mean(fold_weights), nested_cv.py:249,293-296, for voxels that chose the same
 * alpha tuple over the folds):  C[r][t] = scale * sum_f (maps[f][t] >= 0 ? mats[f][r * ld[f] + maps[f][t]] : 0), t < K, as
 * the tiled fp16 hi/lo image lc_split_rows_f16 would make of C (pad256(rows) * K * 2 halves; d_rowscale_inv as there).
 * h_mats / h_ld / h_maps: HOST arrays of n_folds (<= 16) entries -- device pointers to the (rows, ld) f32 operators and to
 * the (K,) int32 column maps (16-byte aligned).  K %% 16 == 0, K <= 8192. */
int lc_mean_operator_image_f16(const float* const* h_mats, const int64_t* h_ld, const int32_t* const* h_maps, int n_folds,
                               float scale, int64_t rows, int64_t K, void* d_tiled, float* d_rowscale_inv,
                               lc_stream_t stream);
/* The same for n_images alpha tuples in ONE launch.  d_table: DEVICE array of n_images x (n_folds + 1) int64 -- per image the
 * n_folds device addresses of the folds' (rows, ld[f]) operators, then the slot of its image: image i is written at d_tiled +
 * slot_i * pad256(rows) * K * 2 halves, its row scales at d_rowscale_inv + slot_i * pad256(rows).  h_ld / h_maps: as above
 * (shared by all images).  Bit for bit what n_images calls of lc_mean_operator_image_f16 write. */
int lc_mean_operator_images_f16(const int64_t* d_table, int n_images, const int64_t* h_ld, const int32_t* const* h_maps,
                                int n_folds, float scale, int64_t rows, int64_t K, void* d_tiled, float* d_rowscale_inv,
                                lc_stream_t stream);
/* The A image of lc_alpha_sweep_scores_f16x3: per group (inner fold) the A hat matrices H_a (M x K each, stacked alpha by
 * alpha in d_h: rows a M + i), with the 32-row blocks taken in the order (validation block, alpha) -- image block s =
 * rows [32 (s / A), 32 (s / A) + 32) of alpha s %% A -- so that every 256-row tile of the sweep holds all alphas of the
 * same few validation rows and its epilogue fetches those rows of the targets once, not once per alpha.  M %% 32 == 0. */
int lc_split_rows_f16_alphas(const float* d_h, int64_t ld, int groups, int A, int64_t M, int64_t K, void* d_tiled,
                             float* d_rowscale_inv, lc_stream_t stream);
/* The same of the FIRST A of the A_src hat matrices every group holds in d_h (group stride A_src * M rows): the screening
 * pass of the inner CV (one fp16 MFMA per product: scores good to ~2e-4 relative) takes the largest factorised alphas from the
 * shared series terms instead, where the 4-term series is accurate to << that (FitOptions.screen_series_tol). */
int lc_split_rows_f16_alphas_sel(const float* d_h, int64_t ld, int groups, int A_src, int A, int64_t M, int64_t K,
                                 void* d_tiled, float* d_rowscale_inv, lc_stream_t stream);

/* Per-voxel power-of-two scale from max|y| over rows 0..T-1: d_cscale[v] = 2^-e, d_cscale[V + v] = 2^e.
 * *d_flag (caller-zeroed) is OR-ed with 1 when some column is non-finite or has most of its entries more
 * than 2^9 below its maximum (outliers): the 22-bit split would then not be fp32-equivalent and callers
 * should use lc_alpha_sweep_scores. */
int lc_col_scales_f16(const float* d_y, int64_t ldy, int64_t T, int64_t V, float* d_cscale,
                      int32_t* d_flag, lc_stream_t stream);
/* ... with WHICH columns raised the flag (round 5): d_colflag (V) u8, 1 = the column's dynamic range is too wide for the
 * fp16 hi/lo split.  A handful of such voxels is recomputed on an exact-f32 side path and only they leave the f16x3
 * arithmetic (the reference treats every column alike in fp32, ridge_regression.py:104-125); needs d_flag. */
int lc_col_scales_f16_flags(const float* d_y, int64_t ldy, int64_t T, int64_t V, float* d_cscale, int32_t* d_flag,
                            uint8_t* d_colflag, const int32_t* d_live_cols, lc_stream_t stream);

/* The primal form's operand  B_f = B_all - B_val(f)  (Rstim'Rresp of an inner training set as the outer block's minus
 * the validation block's, nested_cv.py:366-374 -> ridge_regression.py:104-106) and its column scales in ONE pass:
 *   lc_combine_terms_colmax_f32  out = c0 T0 + c1 T1 + ... (<= 4 terms, lc_combine_terms_f32's arithmetic) over (rows,
 *                                cols) matrices of leading dimension ld (16-byte aligned, cols %% 4 == 0); d_colmax
 *                                (cols uint32, caller-zeroed, NULL = none) receives the bits of max |out| over the FINITE
 *                                entries of every column (several calls may accumulate into it);
 *   lc_col_scales_from_max       d_cscale[v] = 2^-e, d_cscale[V + v] = 2^e from those maxima: lc_col_scales_f16's scales. */
int lc_combine_terms_colmax_f32(const float* const* h_terms, const float* h_coef, int terms, float* d_out,
                                int64_t ld, int64_t rows, int64_t cols, uint32_t* d_colmax, lc_stream_t stream);
int lc_col_scales_from_max(const uint32_t* d_colmax, int64_t V, float* d_cscale, lc_stream_t stream);

/* Y[d_rows] (K rows incl. -1 padding, V columns) * cscale -> tiled fp16 hi/lo image
 * (pad256(V) * K * 2 halves).  K % 32 == 0. */
int lc_split_cols_f16(const float* d_y, int64_t ldy, int64_t V, const int32_t* d_rows, int K,
                      const float* d_cscale, void* d_tiled, const int32_t* d_live_cols, lc_stream_t stream);

/* The tiled image of the same K rows with the voxel columns permuted: column j of the output image is column
 * d_perm[j] of d_tiled (-1: a zero column), j < Vs (a multiple of 256).  The alpha-sorted operand of the refit
 * (ridge_regression.py:46-61 groups the voxels by alpha) from the image the inner CV already made of the outer training
 * rows, without a sorted fp32 copy of the targets.  Column scales are per column and travel with it. */
int lc_permute_cols_f16(const void* d_tiled, const int32_t* d_perm, int64_t Vs, int K, void* d_out,
                        lc_stream_t stream);

/* "B view" (both fp16x3 entry points): d_yt / d_bt may be the tiled image of MORE rows than the product
 * contracts -- the targets of a whole outer training set, split once -- of which the product skips one aligned gap
 * (the validation block of an inner fold).  b_rows = rows of the image (0: the image is exactly the K rows),
 * b_gap_begin / b_gap_rows = first row and length of the skipped block in image rows; all multiples of 16, and
 * K + b_gap_rows <= b_rows. */

/* Correlation scores (mode LC_SCORE_CORR, same formula and nan_to_num as lc_alpha_sweep_scores) of S series
 * alphas of ONE inner fold from d_t (terms*M, ldt) f32 = the stacked T_j = P'_j Y (lc_batch_series_terms +
 * lc_gemm_grouped_f16x3 / lc_gemm_grouped_f32): per voxel the fp64 moments of the T_j over the n_val
 * validation rows give mean, variance and covariance with y of every alpha's prediction as linear / quadratic
 * forms.  d_coef: (S, terms) f64 polynomial coefficients per alpha, d_aidx: (S) rows of d_scores (A, V) f32 that
 * receive (or accumulate) the scores.  d_yv, d_ystat as produced by lc_val_stats.
 * d_rowmap: optional (terms*M) int32 row of d_t holding row i of term j (as in lc_batch_series_terms). */
int lc_series_scores(const float* d_t, int64_t ldt, int terms, int M, int n_val, int64_t V,
                     const float* d_yv, const float* d_ystat, const double* d_coef,
                     const int32_t* d_aidx, int S, const int32_t* d_rowmap, float* d_scores, int accumulate,
                     lc_stream_t stream);

/* The two steps above in ONE contraction that never stores the terms (four series terms): d_pt = the tiled fp16 image
 * (lc_split_rows_f16) of the stacked terms in the layout  row = 256 t + 128 (j >> 1) + 32 (2 (j & 1) + (b & 1)) + i % 32
 * for term j, validation row i, b = i / 32, t = b / 2 -- every 256-row tile holds all four terms of two 32-row
 * validation blocks, terms 0, 1 in the wave row that issues three MFMAs per product, terms 2, 3 in the one that
 * issues hi * hi alone -- and the epilogue reduces the accumulators to the blocks' partial moments (d_part: (M / 32,
 * 18, V) f32 workspace), from which the scores of the S series alphas follow as in lc_series_scores (same formula:
 * ridge_regression.py:124-133 for pred = sum_j c_j T_j).  d_yt / d_cscale_inv / Ncols / b_*: the tiled target image as
 * in lc_gemm_grouped_f16x3;  d_yv, d_ystat, d_yblk: lc_val_stats of the fold;  d_scores: (A, V) f32, V %% 128 == 0. */


/* The entry points of the two contractions above, for F (1 <= F <= 64) inner folds of an outer fold in ONE launch each
 * (round 6: the single-fold forms lc_alpha_sweep_scores_f16x3 / lc_series_sweep_scores_f16x3 are gone -- F = 1 is that
 * call): the folds' tiled A images (and row
 * scales) are stacked, every fold padded to whole 256-row tiles; d_yv / d_ystat / d_yblk / d_part are (F, ...) stacks
 * (lc_val_stats_folds); all folds contract the same tiled target image d_yt, fold f skipping its own gap
 * (h_gap_begin[f], h_gap_rows[f]; b_rows = 0: no gap).  Scores of the folds are added in fp32, fold order
 * (nested_cv.py:373-380).  The folds are independent: one launch fills the chip where F small ones each end in a
 * partial round of workgroups (a rank of an 8-GPU job holds 10 000 voxels: 1.25 rounds per fold).
 * terms: 3 = every product as hi*hi + hi*lo + lo*hi (22-bit operands: fp32-level scores, what ridge_corr_torch's fp32
 * matmuls give, ridge_regression.py:120-133);  1 = the SCREENING pass of the inner CV (round 6): hi*hi
 * alone, 11-bit operands -- scores good to ~1e-5, from which the alpha of every voxel whose best two alphas are further
 * apart than that is already decided (nested_cv.py:408-411 only takes the argmax); the others are scored again with
 * terms = 3 (lc_undecided_cols picks them).  terms = 1 needs N (K) % 64 == 0 and correlation scores.
 * d_live_cols (NULL: all): device int32, read by the kernel -- only the column tiles below *d_live_cols do any work (the
 * refinement's column panel has a fixed capacity; how many undecided columns it holds is known on the device only). */
int lc_alpha_sweep_scores_f16x3_folds(const void* d_ht, const float* d_rowscale_inv, int F, int A, int M, int N,
                                      const void* d_yt, const float* d_cscale_inv, const float* d_yv,
                                      int64_t V, const int32_t* h_n_val, const float* d_ystat,
                                      const float* d_yblk, int mode, float* d_part, float* d_scores,
                                      int accumulate, int64_t b_rows, const int64_t* h_gap_begin,
                                      const int64_t* h_gap_rows, int terms, const int32_t* d_live_cols,
                                      lc_stream_t stream);
int lc_series_sweep_scores_f16x3_folds(const void* d_pt, const float* d_rowscale_inv, int F, int M,
                                       const int32_t* h_n_val, int64_t K, const void* d_yt,
                                       const float* d_cscale_inv, int64_t Ncols, const float* d_yv, int64_t V,
                                       const float* d_ystat, const float* d_yblk, const double* d_coef,
                                       const int32_t* d_aidx, int S, float* d_part, float* d_scores,
                                       int accumulate, int64_t b_rows, const int64_t* h_gap_begin,
                                       const int64_t* h_gap_rows, int terms, const int32_t* d_live_cols,
                                       lc_stream_t stream);

/* accumulate = 2 in the two calls above: the contraction alone (partials into d_part, no scores).  The second halves for F
 * such folds at once (d_part / d_ystat / d_yblk: the (F, ...) stacks the contractions wrote slice by slice) -- one pass over
 * all folds' partials instead of one small launch per fold (25 + 25 per fit at cfg2); folds added in order, as above. */
int lc_alpha_sweep_finalize_folds(const float* d_part, const float* d_ystat, const float* d_yblk, int F, int A, int M,
                                  const int32_t* h_n_val, int64_t V, int mode, float* d_scores, int accumulate,
                                  const int32_t* d_live_cols, lc_stream_t stream);
int lc_series_sweep_finalize_folds(const float* d_part, const float* d_ystat, const float* d_yblk, int F, int M,
                                   const int32_t* h_n_val, int64_t V, const double* d_coef, const int32_t* d_aidx, int S,
                                   float* d_scores, int accumulate, const int32_t* d_live_cols, lc_stream_t stream);

/* The voxels whose alpha the SCREENING pass of the inner CV (terms = 1 above) does not decide (round 6, DESIGN.md 4.2;
 * the reference takes the argmax of the fold-mean scores, nested_cv.py:408-411 -- a voxel whose best two alphas are
 * further apart than the screening error has that argmax already).  d_scores: (A, ld) f32 sums over the scored inner
 * folds; voxel v < V is undecided when its two largest sums are closer than tau_sum * kappa(v), kappa = rms / std of the
 * validation rows of d_ystat ((3, ld_stat) f32 of lc_val_stats: mean, std, variance; NULL: 1) -- a column on an offset
 * loses the operand bits the offset takes -- or when a sum is non-finite; a voxel whose sums are ALL exactly zero
 * (constant column) is decided (alphas[0] by the first-maximum rule in either arithmetic).
 * d_list (cap) int32 receives the undecided columns in ascending order, -1 behind them; d_count[0] = columns in the
 * list, d_count[1] = undecided columns found (> cap: the list does not hold them all -- the caller scores the whole
 * range again), d_count[2] = that condition as 0 / 1 (three int32).  Workspaces: d_flags (V) u8, d_block_count (ceil(V / 256)) int32.  Deterministic, no atomics. */
int lc_undecided_cols(const float* d_scores, int A, int64_t ld, int64_t V, float tau_sum, const float* d_ystat,
                      int64_t ld_stat, uint8_t* d_flags, int32_t* d_block_count, int32_t* d_list, int cap,
                      int32_t* d_count, lc_stream_t stream);

/* single_alpha (nested_cv.py:396-400: ONE alpha = argmax of the voxel MEAN of the scores) under the screening pass: the mean of
 * V screening scores is good to ~(screening error) x sqrt(sum kappa^2) / V when the voxels' errors are independent -- d_out[0] =
 * sum kappa, d_out[1] = sum kappa^2 over the V voxels (kappa as in lc_undecided_cols), from which the host decides whether the
 * two best alphas' means are far enough apart; if not the fit is repeated on three MFMAs. */
int lc_kappa_sums(const float* d_ystat, int64_t ld_stat, int64_t V, double* d_out, lc_stream_t stream);

/* ---------------------------------------------------------------- statistics tail */

/* Fisher's combination of k p-values per voxel, nested_cv.py:441-477 (`_combine_pvalues_across_folds`):
 * chi2.sf(-2 sum ln p, 2k) in closed form for the even degrees of freedom; all-ones rows give exactly 1.
 * d_p: (k, V) f64 row-major, NaN-free; d_out: (V) f64. */
int lc_fisher_combine(const double* d_p, int k, int64_t V, double* d_out, lc_stream_t stream);

/* Benjamini-Hochberg step-up (statsmodels `fdrcorrection(pvals, alpha, method="indep")`, call sites
 * nested_cv.py:158,263,282): d_reject (n) u8 and d_padj (n) f64 in input order.  d_work: at least
 * lc_bh_fdr_work_bytes(n) bytes (sorted copies + the scratch of the library's own LSD radix sort), caller-owned. */
int64_t lc_bh_fdr_work_bytes(int64_t n);
int lc_bh_fdr(const double* d_p, int64_t n, double alpha, uint8_t* d_reject, double* d_padj,
              void* d_work, int64_t work_bytes, lc_stream_t stream);

/* The rejection mask of the same procedure WITHOUT the adjusted p-values and without a sort (the per-fold masks of a
 * cross-validated fit, nested_cv.py:263-290: only they enter the majority vote): kmax = max{i : p_(i) <= (i/n) alpha} is the
 * largest fixed point of c -> #{p <= (c/n) alpha}, reached from c = n by a few counting passes; d_reject[i] =
 * p[i] <= (kmax / n) alpha.  Element for element the mask lc_bh_fdr returns.  The iteration is capped (48 passes:
 * p-values that hug the BH line from above make it fall one step at a time): *d_status = 0 and the mask is written, or
 * *d_status = 1 and it is not -- the caller then runs lc_bh_fdr on that vector. */
int lc_bh_reject(const double* d_p, int64_t n, double alpha, uint8_t* d_reject, int32_t* d_status, lc_stream_t stream);

/* The test rows of the refit without their predictions ever reaching HBM (nested_cv.py:151-155, 251-257: pred = Pstim wt,
 * then Pearson r per voxel; SURVEY K8 + K9): the grouped contraction of lc_gemm_grouped_f16x3 over Mrows = the test rows
 * (d_at: per alpha group the rows  X_te (K + a^2 I)^-1 ...  of the refit operator), whose epilogue reduces the fp32
 * predictions -- the very values lc_gemm_grouped_f16x3 would store -- per 128-row slab and column to count, means and
 * centred sums in fp64 against the test targets  d_y[d_y_rows[i] * ldy + d_y_cols[column]]  (NULL lists: row i / the
 * column itself; d_y_cols[j] < 0: a padding column, r = NaN); the slabs are merged by the pairwise update formulas:
 * d_r (Ncols) f64 = Pearson r as lc_pearson_cols gives it on the stored predictions (same two-pass arithmetic per slab;
 * agreement ~1e-15).  d_part: workspace, ceil(Mrows / 128) * 6 * Ncols doubles. */
int lc_gemm_grouped_f16x3_pearson(const void* d_at, const float* d_rowscale_inv, int64_t Mrows, const void* d_bt,
                                  const float* d_cscale_inv, int64_t Ncols, int64_t K,
                                  const int32_t* h_group_tiles, int G, const float* d_y, int64_t ldy,
                                  const int32_t* d_y_rows, const int32_t* d_y_cols, double* d_part, double* d_r,
                                  lc_stream_t stream);

/* (The diagnostic builds of that kernel -- in-kernel stamps, the 16x16x32 experiment -- are not part of this library since
 * round 5: tools/debug_kernels/, built into tools/bin/liblitcoder_debug.so.) */

/* Grouped GEMM of the refit (lc_gemm_grouped_f32's job) on the same fp16x3 scheme:
 * C[:, tile] = A_g(tile) . B[:, tile].  d_at: G tiled images made by lc_split_rows_f16 (one per group, each
 * pad256(Mrows) rows), d_rowscale_inv: (G * pad256(Mrows)); d_bt: tiled image of B (K x Ncols) made by
 * lc_split_cols_f16, d_cscale_inv: (Ncols).  Ncols % 256 == 0, K % 32 == 0; h_group_tiles: G+1 offsets in
 * 256-column tiles.  d_slab_light: optional (G * pad256(Mrows) / 128) bytes, one per 128-row slab of the tiled
 * image: nonzero = that slab's rows are computed from the fp16 hi parts alone (11-bit operands, one MFMA per
 * product instead of three) -- for rows that enter the caller's result scaled down by 2^-11 or more, such as
 * the higher terms of lc_batch_series_terms.  ldc may be smaller than Ncols (by less than one 256-column tile): only
 * the first ldc columns are then stored -- an output whose width is no multiple of the tile, such as the refit
 * operator  [Xtr' ; K[te,tr]] . (K + a^2 I)^-1  with the inverse from lc_batch_chol_inverse as the B operand (a symmetric
 * B's lc_split_rows_f16 image IS its column image).  128-row slabs without valid rows skip their MFMAs. */
int lc_gemm_grouped_f16x3(const void* d_at, const float* d_rowscale_inv, int64_t Mrows,
                          const void* d_bt, const float* d_cscale_inv, float* d_c, int64_t ldc,
                          int64_t Ncols, int64_t K, const int32_t* h_group_tiles, int G,
                          const uint8_t* d_slab_light,
                          int64_t b_rows, int64_t b_gap_begin, int64_t b_gap_rows, lc_stream_t stream);

/* best[v] = first argmax_a scores[a, v] / n_folds (nested_cv.py:391-408); also
 * d_rowsum[a] = sum_v scores[a, v] (f64) for the single-alpha path (:396-400, all-reduced
 * across ranks by the host).  Either output may be NULL. */
int lc_select_alpha(const float* d_scores, int A, int64_t V, int32_t* d_best, double* d_rowsum,
                    lc_stream_t stream);

/* single_alpha (nested_cv.py:396-403): best[v] = first argmax_a rowsum[a] for every v -- `rowsum` being the per-alpha
 * score sums after the all-reduce over the voxel shards, so the choice never visits the host. */
int lc_fill_argmax(const double* d_rowsum, int A, int32_t* d_best, int64_t V, lc_stream_t stream);
/* d_acc[i] += d_x[i]: the per-alpha score sums of the voxel ranges of ONE fold added up on the device when a
 * single_alpha fit works through its targets panel by panel (nested_cv.py:396-400 takes the mean over ALL voxels). */
int lc_accumulate_f64(const double* d_x, double* d_acc, int64_t n, lc_stream_t stream);

/* One fold's per-voxel results of this rank as ONE (4, ld) f64 block in natural voxel order, the unit of the
 * all-gather over voxel shards (SURVEY 8e "single gather"; nested_cv.py:152-158, 252-263 are its consumers):
 *   row 0: Pearson r (alpha-sorted d_r_sorted[j] belongs to voxel d_perm[j]; -1 = padding), NaN kept
 *   row 1: p-value, raw          row 2: chosen alpha index (d_best, natural order)
 *   row 3: [any(d_info_a != 0), any(d_info_b != 0), 0 ...]  -- Cholesky pivot flags (inner folds, refit)
 * Columns >= V are zero. */
int lc_fold_pack(const double* d_r_sorted, const double* d_p_sorted, const int32_t* d_perm, int64_t Vs,
                 const int32_t* d_best, int64_t V, const int32_t* d_info_a, int n_a,
                 const int32_t* d_info_b, int n_b, double* d_out, int64_t ld, lc_stream_t stream);

/* The same block filled panel by panel (voxel panels of one rank processed one after the other share the rank's
 * block): this panel's V voxels land in columns [col0, col0 + V); `clear` != 0 zeroes the whole block first (the first
 * panel of a fold); the flags of row 3 are OR-ed over the panels. */
int lc_fold_pack_at(const double* d_r_sorted, const double* d_p_sorted, const int32_t* d_perm, int64_t Vs,
                    const int32_t* d_best, int64_t V, const int32_t* d_info_a, int n_a,
                    const int32_t* d_info_b, int n_b, double* d_out, int64_t ld, int64_t col0, int clear,
                    lc_stream_t stream);

/* The gathered blocks of all ranks (world, 4, ld) -> V_total-long vectors: rank k's columns [0, lo[k+1]-lo[k])
 * land at [lo[k], lo[k+1]) (d_lo: world+1 int64 on the device; w_max = widest shard).  d_p_clean = p with
 * NaN-r voxels set to 1 (nested_cv.py:436), the input of the BH-FDR; d_bad (2 int32) = OR of the ranks' flags. */
int lc_fold_unpack(const double* d_src, int world, int64_t ld, const int64_t* d_lo, int64_t w_max,
                   double* d_r, double* d_p, int32_t* d_idx, double* d_p_clean, int32_t* d_bad,
                   lc_stream_t stream);

/* Counting sort of voxels by alpha index (torch.unique / nonzero grouping,
 * ridge_regression.py:46-50): d_perm = voxel ids grouped by alpha, stable, each group starting
 * at a multiple of `pad` columns (slots between groups are left untouched: pre-fill d_perm,
 * V + A*pad entries, with -1).  d_count (A) = group sizes.  A voxel whose index is outside 0 .. A-1 (-1) is in
 * no group (banded ridge with a search over band scales: the voxels another candidate's refit takes). */
int lc_group_by_alpha(const int32_t* d_best, int64_t V, int A, int pad, int32_t* d_perm,
                      int32_t* d_count, lc_stream_t stream);
/* The same for the alpha indices a0 .. a0+A-1 only (A <= 64 per call): the reference groups by ANY number of distinct
 * alphas (torch.unique, ridge_regression.py:46-50); a grid of more than 64 is grouped range by range, each range
 * into a perm of its own (V + A*pad entries, pre-filled with -1), d_count (A) = the sizes of ITS groups. */
int lc_group_by_alpha_range(const int32_t* d_best, int64_t V, int a0, int A, int pad, int32_t* d_perm,
                            int32_t* d_count, lc_stream_t stream);

/* Grouped plain GEMM, C[:, tile] = A_g(tile) . B[:, tile]   (f32 MFMA, 128-column tiles):
 *   ridge_torch's  wt[:, sel] = Vh' diag(D) U' Y[:, sel]  (ridge_regression.py:56-61) with
 *   A_g = X_tr'(K + a_g^2 I)^-1, and  y_pred = X_test @ wt  (nested_cv.py:151,251) with one group.
 * d_a: (G, Mrows, K) f32, group stride a_group_stride elements; d_b: (K, Ncols) f32;
 * d_c: (Mrows, Ncols) f32.  Ncols % 128 == 0, K % 32 == 0 (callers pad with zeros).
 * h_group_tiles: G+1 column-tile offsets (host) -- group g owns tiles [h[g], h[g+1]).
 * d_brows: optional row-index list for B (K entries, -1 = zero row). */
int lc_gemm_grouped_f32(const float* d_a, int64_t lda, int64_t a_group_stride,
                        const float* d_b, int64_t ldb, const int32_t* d_brows,
                        float* d_c, int64_t ldc, int64_t Mrows, int64_t Ncols, int64_t K,
                        const int32_t* h_group_tiles, int G, lc_stream_t stream);

/* ---------------------------------------------------------------- collectives over xGMI
 * Thin RCCL wrappers: what a voxel-sharded fit exchanges -- the V-independent f32 operators of the (fold, alpha) systems
 * dealt out over the ranks (all-gather; ridge_regression.py:104-125 is voxel-independent given them), the per-alpha score
 * sums of single_alpha (nested_cv.py:396-400) and the alpha histogram (all-reduce), the packed per-fold results
 * (all-gather) -- on device pointers and a HIP stream.  RCCL is resolved at the first call (the copy the process has
 * loaded, else librccl.so.1): a one-GPU process never loads it.
 *   lc_comm_unique_id        h_id (lc_comm_unique_id_bytes() bytes): a fresh id, made on ONE rank and handed to the others
 *                            by the host side (the Python layer broadcasts it once through torch.distributed);
 *   lc_comm_create           collective over `world` ranks: this rank's communicator on `device`;
 *   lc_allgather_f32 / _bytes  d_recv (world x count) = every rank's d_send, rank-major, ordered on `stream`;
 *   lc_allreduce             in place, op 0 = sum, 1 = max; dtype LC_F32 / LC_F64 / LC_I32;  lc_allreduce_sum_f32: the f32 sum. */
typedef struct lc_comm lc_comm_t;
int lc_comm_unique_id_bytes(void);
int lc_comm_unique_id(void* h_id, int bytes);
int lc_comm_create(const void* h_id, int bytes, int world, int rank, int device, lc_comm_t** out);
int lc_comm_destroy(lc_comm_t* comm);
int lc_allgather_f32(lc_comm_t* comm, const float* d_send, float* d_recv, int64_t count, lc_stream_t stream);
int lc_allgather_bytes(lc_comm_t* comm, const void* d_send, void* d_recv, int64_t bytes, lc_stream_t stream);
int lc_allreduce(lc_comm_t* comm, void* d_buf, int64_t count, int dtype, int op, lc_stream_t stream);
int lc_allreduce_sum_f32(lc_comm_t* comm, float* d_buf, int64_t count, lc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LITCODER_HIP_H */
