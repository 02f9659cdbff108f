// HBM-bound helpers around the fit: casts, row/column gathers, column statistics,
// per-voxel Pearson correlation, alpha selection and grouping.  All kernels put the voxel
// (column) axis on the lanes so every wave-level access is a contiguous row segment.
#include "lc_common.h"

namespace {

// ------------------------------------------------------------------ cast
__global__ void __launch_bounds__(256) k_cast_f64_f32(const double* __restrict__ in, long long ld_in,
                                                      float* __restrict__ out, long long ld_out, long long cols) {
    const long long r = blockIdx.x;
    const long long stride = (long long)gridDim.y * blockDim.x;
    const double* src = in + r * ld_in;
    float* dst = out + r * ld_out;
    for (long long j = (long long)blockIdx.y * blockDim.x + threadIdx.x; j < cols; j += stride) dst[j] = (float)src[j];
}

// ------------------------------------------------------------------ gather / scatter
// x = column chunk (fastest), y = row: the workgroups that are resident together read (or update) the same row, whose
// sectors -- each touched for one 4-byte element -- are then shared through L2 instead of being fetched per chunk
// (2.2x on the alpha-sorted copy of the targets; several rows per workgroup were slower again).
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ in, long long ld_in,
                                                const int* __restrict__ rows, const int* __restrict__ cols,
                                                long long n_cols, float* __restrict__ out, long long ld_out,
                                                const int* __restrict__ live) {
    const long long r = blockIdx.y;
    const long long src = rows ? rows[r] : r;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // live (the refinement's column panel): only the first *live columns (whole 256-column tiles) hold voxels
    if (live) n_cols = min(n_cols, ((long long)*live + 255) & ~255ll);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += stride) {
        float v = 0.f;
        const long long c = cols ? cols[j] : j;
        if (src >= 0 && c >= 0) v = in[src * ld_in + c];
        out[r * ld_out + j] = v;
    }
}

__global__ void __launch_bounds__(256) k_scatter_axpy(const float* __restrict__ w, long long ld_w,
                                                      const int* __restrict__ cols, long long n_cols, float scale,
                                                      float* __restrict__ acc, long long ld_acc) {
    const long long r = blockIdx.y;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += stride) {
        const long long c = cols ? cols[j] : j;
        if (c >= 0) acc[r * ld_acc + c] += scale * w[r * ld_w + j];
    }
}

// dst[r, cols[j]] = src[r, j]  (overwrite; cols[j] < 0: skipped) for 4- or 8-byte elements: a few columns recomputed on a
// side path -- the voxels whose dynamic range the fp16 split cannot carry, round 5 -- go back into the matrices of the
// main path.  Column indices must be distinct.
template <typename E>
__global__ void __launch_bounds__(256) k_scatter_cols(const E* __restrict__ src, long long ld_src, const int* __restrict__ cols,
                                                      long long n_cols, E* __restrict__ dst, long long ld_dst) {
    const long long r = blockIdx.y;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += stride) {
        const long long c = cols[j];
        if (c >= 0) dst[r * ld_dst + c] = src[r * ld_src + j];
    }
}

// A few columns of a V-wide product, streamed (round 5: the f32 side panel when it holds a handful of columns -- one outlier
// voxel among 80 000 is the usual case).  C[m][j] = sum_k A[m][k] Y[rows[k]][j] for the ns <= GV_NS columns j whose
// sel[j] == want (sel == NULL: all), fp64 accumulation of the f32 products, written as f32.  The 128-column MFMA tiles of
// lc_alpha_sweep_scores / lc_gemm_grouped_f32 spend 127 / 128 of their work on padding there and hold 15-30 CUs for
// ~0.3 ms per launch beside the fit's fp16 sweeps; this reads A once at HBM speed (a hat-matrix stack of 15 MB: ~10 us).
// Block: 8 rows of A (two per wave), the Y columns staged through LDS in chunks of GV_KC rows.
constexpr int GV_NS = 8, GV_KC = 1024;
__global__ void __launch_bounds__(256) k_gemv_cols(const float* __restrict__ A, long long lda, long long M, long long K,
                                                   const float* __restrict__ Y, long long ldy, const int* __restrict__ rows,
                                                   const int* __restrict__ sel, int ns, int want, float* __restrict__ C,
                                                   long long ldc) {
    __shared__ float ys[GV_NS][GV_KC];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned active = 0;
    for (int j = 0; j < ns; ++j)
        if (!sel || sel[j] == want) active |= 1u << j;
    if (!active) return;                                 // (block-uniform)
    const long long m0 = (long long)blockIdx.x * 8 + 2 * w;
    double acc[2][GV_NS];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < GV_NS; ++j) acc[r][j] = 0.0;
    for (long long k0 = 0; k0 < K; k0 += GV_KC) {
        const int kc = (int)(K - k0 < GV_KC ? K - k0 : GV_KC);
        __syncthreads();                                 // the previous chunk has been consumed
        for (int e = t; e < kc * ns; e += 256) {
            const int k = e / ns, j = e - k * ns;
            const long long row = rows ? rows[k0 + k] : k0 + k;
            ys[j][k] = (row >= 0 && ((active >> j) & 1u)) ? Y[row * ldy + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long m = m0 + r;
            if (m >= M) break;
            const float* a = A + m * lda + k0;
            for (int k = 4 * lane; k < kc; k += 256) {
                const float4 av = *reinterpret_cast<const float4*>(a + k);          // (K % 4 == 0)
#pragma unroll
                for (int j = 0; j < GV_NS; ++j) {
                    if (!((active >> j) & 1u)) continue;
                    const float4 yv = *reinterpret_cast<const float4*>(&ys[j][k]);
                    acc[r][j] += (double)av.x * (double)yv.x + (double)av.y * (double)yv.y + (double)av.z * (double)yv.z +
                                 (double)av.w * (double)yv.w;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < GV_NS; ++j) {
            if (!((active >> j) & 1u)) continue;         // (wave-uniform)
            double v = acc[r][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && m0 + r < M) C[(m0 + r) * ldc + j] = (float)v;
        }
}

// Mean of the folds' weights without a read-modify-write per fold: every fold leaves its alpha-SORTED weight matrix where
// the refit contraction wrote it, k_invert_perm notes where each voxel's column went, and ONE pass per voxel range forms
//     out[r, v] = sum_f scale_f * w_f[r, pos_f[v]]          (folds in order, the same expression per term as
// k_scatter_axpy's  acc += scale * w,  starting from 0: bit-identical to the accumulate it replaces)
// -- per fold one 4-byte gather per element (the sorted neighbours of a group are neighbours in natural order too, so
// the sectors are shared) instead of a gather, a read and a write of the accumulator.
constexpr int CF_MAX = 8, CF_ROWS = 4;
struct CombineArgs {
    const float* w[CF_MAX];
    const int* pos[CF_MAX];
    long long ld[CF_MAX];
    float scale[CF_MAX];
    int n;
};

__global__ void __launch_bounds__(256) k_invert_perm(const int* __restrict__ cols, long long n_cols, int base,
                                                     int* __restrict__ pos) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols) return;
    const int c = cols[j];
    if (c >= 0) pos[c] = base + (int)j;
}

__global__ void __launch_bounds__(256) k_combine_folds(const CombineArgs a, long long n_rows, long long n_cols,
                                                       float* __restrict__ out, long long ld_out, int accumulate) {
    const long long r0 = (long long)blockIdx.y * CF_ROWS;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n_cols; v += stride) {
        int p[CF_MAX];
#pragma unroll
        for (int f = 0; f < CF_MAX; ++f) p[f] = f < a.n ? a.pos[f][v] : -1;
#pragma unroll
        for (int rr = 0; rr < CF_ROWS; ++rr) {
            const long long r = r0 + rr;
            if (r >= n_rows) break;
            float acc = accumulate ? out[r * ld_out + v] : 0.f;
#pragma unroll
            for (int f = 0; f < CF_MAX; ++f)
                if (f < a.n && p[f] >= 0) acc += a.scale[f] * a.w[f][r * a.ld[f] + p[f]];
            out[r * ld_out + v] = acc;
        }
    }
}

// ------------------------------------------------------------------ column moments
// Block = 64 columns (x) by RG row groups (y).  Two passes over the listed rows: mean, then
// centred second moment, both in fp64; the second pass re-reads a panel that is still in L2.
constexpr int CM_RG = 8;

template <int RG>
__device__ inline double block_colsum(double v, double (*sm)[64]) {
    sm[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < RG; ++g) t += sm[g][threadIdx.x];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(64 * CM_RG) k_col_mean_std(const float* __restrict__ x, long long ld,
                                                             const int* __restrict__ rows, long long n_rows,
                                                             long long n_cols, float* __restrict__ mean_out,
                                                             float* __restrict__ std_out) {
    __shared__ double sm[CM_RG][64];
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = c < n_cols;
    double s = 0.0;
    if (live)
        for (long long i = threadIdx.y; i < n_rows; i += CM_RG) {
            const long long r = rows ? rows[i] : i;
            if (r >= 0) s += (double)x[r * ld + c];
        }
    const double mean = block_colsum<CM_RG>(s, sm) / (double)n_rows;
    double q = 0.0;
    if (live)
        for (long long i = threadIdx.y; i < n_rows; i += CM_RG) {
            const long long r = rows ? rows[i] : i;
            if (r >= 0) { const double d = (double)x[r * ld + c] - mean; q += d * d; }
        }
    const double m2 = block_colsum<CM_RG>(q, sm);
    if (live && threadIdx.y == 0) {
        mean_out[c] = (float)mean;
        std_out[c] = (float)sqrt(m2 / (double)(n_rows - 1));   // n_rows == 1 -> NaN, like torch.std
    }
}

__global__ void __launch_bounds__(256) k_col_normalize(float* __restrict__ x, long long ld, long long n_cols,
                                                       const float* __restrict__ mean, const float* __restrict__ sd,
                                                       float eps) {
    const long long r = blockIdx.x;
    const long long stride = (long long)gridDim.y * blockDim.x;
    for (long long j = (long long)blockIdx.y * blockDim.x + threadIdx.x; j < n_cols; j += stride)
        x[r * ld + j] = (x[r * ld + j] - mean[j]) / (sd[j] + eps);
}

// Per-story z-scoring of the trainer (utils.py:23-29 ``zs``): float64, POPULATION std, a column whose std is
// exactly 0 is only de-meaned; optional np.nan_to_num on the result (trainer.py:235,250 applies it to X).
// One thread per column walks the rows in order -- numpy reduces axis 0 of a C-ordered matrix row by row, one running
// sum per column -- without fused multiply-adds: mean, std and the normalised values are the reference's bit for bit
// (round 4; rounds 1-3 summed in row groups and were a few ulps off).  Lanes across columns: coalesced.
__global__ void __launch_bounds__(256) k_zscore_story(const double* __restrict__ x, long long ld_in, long long rows,
                                                      long long cols, int nan_to_num, double* __restrict__ out,
                                                      long long ld_out) {
#pragma clang fp contract(off)
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const double* col = x + c;
    double s = 0.0;
    for (long long i = 0; i < rows; ++i) s = s + col[i * ld_in];
    const double mean = s / (double)rows;
    double q = 0.0;
    for (long long i = 0; i < rows; ++i) {
        const double d = col[i * ld_in] - mean;
        q = q + d * d;
    }
    const double sd = sqrt(q / (double)rows);
    for (long long i = 0; i < rows; ++i) {
        double v = col[i * ld_in] - mean;
        if (sd != 0.0) v = v / sd;
        if (nan_to_num) {
            if (v != v) v = 0.0;
            else if (v > 1.7976931348623157e308) v = 1.7976931348623157e308;
            else if (v < -1.7976931348623157e308) v = -1.7976931348623157e308;
        }
        out[i * ld_out + c] = v;
    }
}

// Validation-target statistics for the fused scorer.  ystat = [mean | std | var] (unbiased),
// yblk[b, v] = sum over the b-th 32-row block of fl32(y - mean): exactly the centred values
// the GEMM epilogue multiplies with;  yv = the M validation rows gathered (zero padding rows) in the row-quad
// interleaved layout of lc_epilogue.h, so that the sweep epilogue reads them with 16-byte coalesced loads.
// blockIdx.y = inner fold: fold f reads the row list va + f M and writes slice f of (F, 3, V) / (F, M/32, V) / (F, M, V)
struct FoldRows {
    int n_val[64];
};

// ------------------------------------------------------------------ Pearson r per column
constexpr int PR_RG = 16;

// ONE pass over both operands (round 6; two before: the second pass missed L2 and the kernel ran at 0.24 of the HBM roof):
// every thread sums  d = x - x0  about the column's FIRST-ROW values (a sample of the same column: the centred sums that
// follow lose a digit or two of sixteen, and a constant column gives exact zeros -> 0 / 0 = NaN as before) with
// PR_UNROLL independent loads of each operand in flight; fp64 throughout, fixed reduction order.
constexpr int PR_UNROLL = 8;
__global__ void __launch_bounds__(64 * PR_RG) k_pearson_cols(const float* __restrict__ a, long long lda,
                                                             const float* __restrict__ b, long long ldb,
                                                             long long n, long long V, double* __restrict__ r_out) {
    __shared__ double sm[PR_RG][64];
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);      // row group = wave: scalar row offsets
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = c < V;
    const float a0 = live ? a[c] : 0.f, b0 = live ? b[c] : 0.f;
    double sa = 0.0, sb = 0.0, qa = 0.0, qb = 0.0, qab = 0.0;
    if (live)
        for (long long i0 = ty; i0 < n; i0 += (long long)PR_RG * PR_UNROLL) {
            float va[PR_UNROLL], vb[PR_UNROLL];
#pragma unroll
            for (int u = 0; u < PR_UNROLL; ++u) {
                const long long i = i0 + (long long)u * PR_RG;
                va[u] = i < n ? a[i * lda + c] : a0;                 // (past the last row: the shift itself, d = 0)
                vb[u] = i < n ? b[i * ldb + c] : b0;
            }
#pragma unroll
            for (int u = 0; u < PR_UNROLL; ++u) {
                const double da = (double)va[u] - (double)a0, db = (double)vb[u] - (double)b0;
                sa += da;
                sb += db;
                qa += da * da;
                qb += db * db;
                qab += da * db;
            }
        }
    sa = block_colsum<PR_RG>(sa, sm);
    sb = block_colsum<PR_RG>(sb, sm);
    qa = block_colsum<PR_RG>(qa, sm);
    qb = block_colsum<PR_RG>(qb, sm);
    qab = block_colsum<PR_RG>(qab, sm);
    if (live && ty == 0) {
        const double inv = 1.0 / (double)n;
        qa -= sa * sa * inv;
        qb -= sb * sb * inv;
        qab -= sa * sb * inv;
        if (qa < 0.0) qa = 0.0;                      // (rounding of the shifted form; the two-pass form cannot go below 0)
        if (qb < 0.0) qb = 0.0;
        double r = qab / (sqrt(qa) * sqrt(qb));       // 0/0 -> NaN for a constant column
        if (r > 1.0) r = 1.0;
        if (r < -1.0) r = -1.0;
        r_out[c] = r;
    }
}

// The same sums with the column's rows held in REGISTERS (n <= PQ_RG * RPT rows: one pass over the operands instead of
// two -- the second pass of the kernel above misses L2: 64 columns x 600 rows x 2 operands per workgroup, 1250
// workgroups), and with operand a optionally read THROUGH a row list and a column list (the test rows of the
// alpha-sorted voxels straight from the resident targets: no gathered copy).  Block = 32 columns x 32 row groups: a
// thread keeps at most 20 rows of each operand (40 rows per thread at 16 row groups spilled 128 VGPRs to scratch and ran
// no faster than the two-pass kernel).  fp64 sums, two-pass formula on the registers, fixed reduction order.
constexpr int PQ_C = 32, PQ_RG = 32;

__device__ inline double block_colsum32(double v, double (*sm)[PQ_C]) {
    sm[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll 4
    for (int g = 0; g < PQ_RG; ++g) t += sm[g][threadIdx.x];
    __syncthreads();
    return t;
}

template <int RPT>
__global__ void __launch_bounds__(PQ_C * PQ_RG) k_pearson_cols_regs(const float* __restrict__ a, long long lda,
                                                                    const int* __restrict__ a_rows, const int* __restrict__ a_cols,
                                                                    const float* __restrict__ b, long long ldb, long long n,
                                                                    long long V, double* __restrict__ r_out) {
    __shared__ double sm[PQ_RG][PQ_C];
    const int ty = threadIdx.y;
    const long long c = (long long)blockIdx.x * PQ_C + threadIdx.x;
    const bool live = c < V;
    const long long ca = live ? (a_cols ? (long long)a_cols[c] : c) : -1;
    float va[RPT], vb[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const long long i = ty + (long long)k * PQ_RG;
        va[k] = 0.f;
        vb[k] = 0.f;
        if (live && i < n) {
            const long long ra = a_rows ? (long long)a_rows[i] : i;
            if (ca >= 0 && ra >= 0) va[k] = a[ra * lda + ca];
            vb[k] = b[i * ldb + c];
        }
    }
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (ty + (long long)k * PQ_RG < n) { sa += (double)va[k]; sb += (double)vb[k]; }
    const double ma = block_colsum32(sa, sm) / (double)n;
    const double mb = block_colsum32(sb, sm) / (double)n;
    double qa = 0.0, qb = 0.0, qab = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (ty + (long long)k * PQ_RG < n) {
            const double da = (double)va[k] - ma;
            const double db = (double)vb[k] - mb;
            qa += da * da;
            qb += db * db;
            qab += da * db;
        }
    qa = block_colsum32(qa, sm);
    qb = block_colsum32(qb, sm);
    qab = block_colsum32(qab, sm);
    if (live && ty == 0) {
        double r = qab / (sqrt(qa) * sqrt(qb));
        if (r > 1.0) r = 1.0;
        if (r < -1.0) r = -1.0;
        r_out[c] = r;
    }
}

template <int RPT>
void launch_pearson_regs(const float* a, long long lda, const int* rows, const int* cols, const float* b, long long ldb,
                         long long n, long long V, double* r, hipStream_t s) {
    hipLaunchKernelGGL(k_pearson_cols_regs<RPT>, dim3((unsigned)lc::ceil_div<long long>(V, PQ_C)), dim3(PQ_C, PQ_RG), 0, s, a,
                       lda, rows, cols, b, ldb, n, V, r);
}

// ------------------------------------------------------------------ Pearson p-values
// Two-sided p-value of r under H0 as scipy.stats.pearsonr forms it (r ~ Beta(n/2-1, n/2-1) on [-1, 1]):
// p = 2 * (1 - I_x(ab, ab)), x = (|r| + 1) / 2 evaluated in fp32 like scipy does for float32 inputs, the
// regularised incomplete beta by Lentz's continued fraction in fp64.  NaN r -> p = 1.
__device__ inline double betainc_sym(double ab, double x) {        // I_x(ab, ab)
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const bool flip = x > 0.5;
    const double xx = flip ? 1.0 - x : x;
    const double lbeta = 2.0 * lgamma(ab) - lgamma(2.0 * ab);
    const double front = exp(ab * log(xx) + ab * log1p(-xx) - lbeta) / ab;
    const double tiny = 1e-300;
    double c = 1.0, d = 1.0 - 2.0 * ab * xx / (ab + 1.0);
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 400; ++m) {
        const double m2 = 2.0 * m;
        double num = m * (ab - m) * xx / ((ab + m2 - 1.0) * (ab + m2));
        d = 1.0 + num * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + num / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        num = -(ab + m) * (2.0 * ab + m) * xx / ((ab + m2) * (ab + m2 + 1.0));
        d = 1.0 + num * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + num / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double delta = d * c;
        h *= delta;
        if (fabs(delta - 1.0) < 1e-16) break;
    }
    const double val = front * h;
    return flip ? 1.0 - val : val;
}

__global__ void __launch_bounds__(256) k_pearson_pvalues(const double* __restrict__ r, long long V, long long n,
                                                         double* __restrict__ p) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float r32 = (float)r[v];
    double out = 1.0;
    if (r32 == r32 && n > 2) {
        const float a = fminf(fabsf(r32), 1.f);
        const double x = (double)((a + 1.f) / 2.f);          // fp32, as scipy >= 1.14 does for fp32 statistics
        out = fmin(2.0 * betainc_sym((double)n / 2.0 - 1.0, 1.0 - x), 1.0);
    }
    p[v] = out;
}

// ------------------------------------------------------------------ undecided voxels of the screening pass (round 6)
// scores: (A, ld) f32, the SUM over the scored inner folds of the screening pass (one MFMA per product).  Voxel v is
// UNDECIDED when the two largest of its A sums lie closer than  tau_sum * kappa(v)  -- the screening scores are good to
// ~1e-5 of a fold-mean score for a centred column; a column riding on an offset loses the bits the offset takes:
// kappa = rms / std of its validation rows (>= 1) -- unless all A sums are exactly zero (a constant / non-finite
// column: every alpha scores 0 in either arithmetic, the first-maximum rule takes alphas[0]).  Two kernels, no atomics:
// flags + per-block counts, then every block places its own columns behind the blocks before it -- the list is in
// ascending column order, run to run.
constexpr int UD_THREADS = 256;

__global__ void __launch_bounds__(UD_THREADS) k_undecided_flags(const float* __restrict__ scores, int A, long long ld,
                                                               long long V, float tau_sum, const float* __restrict__ ystat,
                                                               long long ld_stat, unsigned char* __restrict__ flags,
                                                               int* __restrict__ block_count) {
    __shared__ int wsum[UD_THREADS / 64];
    const long long v = (long long)blockIdx.x * UD_THREADS + threadIdx.x;
    bool und = false;
    if (v < V) {
        float top = -__builtin_huge_valf(), second = -__builtin_huge_valf();
        bool all_zero = true, finite = true;
        for (int a = 0; a < A; ++a) {
            const float sc = scores[(long long)a * ld + v];
            all_zero = all_zero && sc == 0.f;
            finite = finite && fabsf(sc) < 3.0e38f;
            if (sc > top) { second = top; top = sc; }
            else if (sc > second) second = sc;
        }
        float kappa = 1.f;
        if (ystat != nullptr) {
            const float m = ystat[v], sd = ystat[ld_stat + v], var = ystat[2 * ld_stat + v];
            const float k2 = sd > 0.f ? sqrtf(m * m + var) / sd : 1.f;
            kappa = (k2 >= 1.f && k2 < 3.0e38f) ? k2 : 1.f;
        }
        if (all_zero) und = false;
        else if (!finite) und = true;
        else if (A < 2) und = false;
        else und = !(top - second >= tau_sum * kappa);           // (a NaN gap counts as undecided)
        flags[v] = und ? 1 : 0;
    }
    const unsigned long long m = __ballot(und);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < UD_THREADS / 64; ++w) t += wsum[w];
        block_count[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(UD_THREADS) k_undecided_place(const unsigned char* __restrict__ flags, long long V,
                                                               const int* __restrict__ block_count, int n_blocks,
                                                               int* __restrict__ list, int cap, int* __restrict__ count) {
    __shared__ int red[2][UD_THREADS];
    __shared__ int wbase[UD_THREADS / 64 + 1];
    int before = 0, total = 0;
    for (int b = threadIdx.x; b < n_blocks; b += UD_THREADS) {
        const int c = block_count[b];
        total += c;
        if (b < (int)blockIdx.x) before += c;
    }
    red[0][threadIdx.x] = before;
    red[1][threadIdx.x] = total;
    __syncthreads();
    for (int w = UD_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    before = red[0][0];
    total = red[1][0];
    const long long v = (long long)blockIdx.x * UD_THREADS + threadIdx.x;
    const bool und = v < V && flags[v] != 0;
    const unsigned long long m = __ballot(und);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wbase[wave + 1] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        wbase[0] = 0;
        for (int w = 1; w <= UD_THREADS / 64; ++w) wbase[w] += wbase[w - 1];
    }
    __syncthreads();
    if (und) {
        const int pos = before + wbase[wave] + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < cap) list[pos] = (int)v;
    }
    // the tail of the list: no column
    for (long long j = (long long)total + (long long)blockIdx.x * UD_THREADS + threadIdx.x; j < cap;
         j += (long long)gridDim.x * UD_THREADS)
        list[j] = -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        count[0] = total < cap ? total : cap;       // columns in the list (what the refinement's launches cover)
        count[1] = total;                           // undecided columns found: > cap = the list does not hold them all
        count[2] = total > cap ? 1 : 0;             // ... as a flag (voxel shards: MAX all-reduced in place)
    }
}

// ------------------------------------------------------------------ alpha selection
__global__ void __launch_bounds__(256) k_argmax_alpha(const float* __restrict__ scores, int A, long long V,
                                                      int* __restrict__ best) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float top = scores[v];
    int arg = 0;
    for (int a = 1; a < A; ++a) {
        const float s = scores[(long long)a * V + v];
        if (s > top) { top = s; arg = a; }      // strict: the first maximum wins (torch.argmax)
    }
    best[v] = arg;
}

// One block per alpha; fixed-order tree reduction -> deterministic.
__global__ void __launch_bounds__(1024) k_rowsum(const float* __restrict__ scores, long long V,
                                                 double* __restrict__ out) {
    __shared__ double sm[1024];
    const float* row = scores + (long long)blockIdx.x * V;
    double s = 0.0;
    for (long long v = threadIdx.x; v < V; v += 1024) s += (double)row[v];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

// sum of kappa and of kappa^2 over the voxels, kappa = rms / std of the validation rows of d_ystat (as in k_undecided_flags):
// what the error of a voxel-MEAN of screening scores scales with (single_alpha: nested_cv.py:396-400).  One block, fixed order.
__global__ void __launch_bounds__(1024) k_kappa_sums(const float* __restrict__ ystat, long long ld, long long V,
                                                     double* __restrict__ out) {
    __shared__ double sm[2][1024];
    double k1 = 0.0, k2 = 0.0;
    for (long long v = threadIdx.x; v < V; v += 1024) {
        const float m = ystat[v], sd = ystat[ld + v], var = ystat[2 * ld + v];
        const float k = sd > 0.f ? sqrtf(m * m + var) / sd : 1.f;
        const double kap = (k >= 1.f && k < 3.0e38f) ? (double)k : 1.0;
        k1 += kap;
        k2 += kap * kap;
    }
    sm[0][threadIdx.x] = k1;
    sm[1][threadIdx.x] = k2;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + w];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sm[0][0]; out[1] = sm[1][0]; }
}

// Stable counting sort of voxel ids by alpha index, one block.  Thread t owns the contiguous
// voxel segment [t*L, (t+1)*L); hist[a][t] is scanned alpha-major so that equal alphas keep
// voxel order.
constexpr int GB_THREADS = 512;
constexpr int GB_MAX_A = 64;

__global__ void __launch_bounds__(GB_THREADS) k_group_by_alpha(const int* __restrict__ best, long long V, int a0, int A, int pad,
                                                               int* __restrict__ perm, int* __restrict__ count, int staged) {
    extern __shared__ int hist[];            // A * GB_THREADS, then (staged) V bytes of alpha indices
    __shared__ int carry;
    const int t = threadIdx.x;
    const long long L = (V + GB_THREADS - 1) / GB_THREADS;
    const long long lo = (long long)t * L, hi = min(V, lo + L);
    // every thread walks its own contiguous segment twice: straight from global memory that is one dependent ~1 us
    // load per voxel and thread (2 x 157 at V = 80 000: the whole kernel).  With `staged` the indices (< 64: one byte)
    // are first copied into LDS with coalesced loads and the segments are walked there.  Indices are taken relative
    // to a0 (a grid of more than 64 alphas is grouped in ranges of 64, lc_group_by_alpha_range); 255 = in no group.
    unsigned char* stage = reinterpret_cast<unsigned char*>(hist + A * GB_THREADS);
    if (staged) {
        for (long long v = t; v < V; v += GB_THREADS) {
            const unsigned rel = (unsigned)(best[v] - a0);
            stage[v] = rel < (unsigned)A ? (unsigned char)rel : (unsigned char)255;
        }
        __syncthreads();
    }
    for (int a = 0; a < A; ++a) hist[a * GB_THREADS + t] = 0;
    // an index outside 0 .. A-1 (-1: the voxel belongs to another candidate's refit, banded.py) is in no group
    if (staged) {
        for (long long v = lo; v < hi; ++v)
            if ((int)stage[v] < A) hist[stage[v] * GB_THREADS + t] += 1;
    } else {
        for (long long v = lo; v < hi; ++v)
            if ((unsigned)(best[v] - a0) < (unsigned)A) hist[(best[v] - a0) * GB_THREADS + t] += 1;
    }
    if (t == 0) carry = 0;
    __syncthreads();
    // exclusive scan over the A*GB_THREADS table in chunks of GB_THREADS (Hillis-Steele per chunk)
    __shared__ int buf[2][GB_THREADS];
    for (int a = 0; a < A; ++a) {
        const int mine = hist[a * GB_THREADS + t];
        int cur = 0;
        buf[0][t] = mine;
        __syncthreads();
        for (int off = 1; off < GB_THREADS; off <<= 1) {
            int v = buf[cur][t];
            if (t >= off) v += buf[cur][t - off];
            buf[cur ^ 1][t] = v;
            cur ^= 1;
            __syncthreads();
        }
        const int incl = buf[cur][t];
        const int base = carry;
        hist[a * GB_THREADS + t] = base + incl - mine;
        __syncthreads();
        if (t == GB_THREADS - 1) {
            carry = ((base + incl + pad - 1) / pad) * pad;      // next group starts on a pad boundary
            count[a] = incl;
        }
        __syncthreads();
    }
    for (long long v = lo; v < hi; ++v) {
        const int a = staged ? (int)stage[v] : best[v] - a0;
        if ((unsigned)a < (unsigned)A) perm[hist[a * GB_THREADS + t]++] = (int)v;
    }
}

}  // namespace

extern "C" int lc_cast_f64_f32(const double* d_in, int64_t ld_in, float* d_out, int64_t ld_out, int64_t rows,
                               int64_t cols, lc_stream_t stream) {
    LC_REQUIRE(d_in && d_out, LC_E_BADARG, "lc_cast_f64_f32: null pointer");
    LC_REQUIRE(rows >= 0 && cols >= 0 && ld_in >= cols && ld_out >= cols, LC_E_SHAPE, "lc_cast_f64_f32: bad shape");
    if (rows == 0 || cols == 0) return LC_OK;
    dim3 grid((unsigned)rows, (unsigned)lc::imin(lc::ceil_div<long long>(cols, 1024), 1024));
    lc::ScopedTimer timer_(lc::T_CAST, lc::as_stream(stream));
    hipLaunchKernelGGL(k_cast_f64_f32, grid, dim3(256), 0, lc::as_stream(stream), d_in, (long long)ld_in, d_out,
                       (long long)ld_out, (long long)cols);
    return lc::launched("k_cast_f64_f32");
}

extern "C" int lc_gather_f32(const float* d_in, int64_t ld_in, const int32_t* d_rows, int64_t n_rows,
                             const int32_t* d_cols, int64_t n_cols, float* d_out, int64_t ld_out,
                             const int32_t* d_live_cols, lc_stream_t stream) {
    LC_REQUIRE(d_in && d_out, LC_E_BADARG, "lc_gather_f32: null pointer");
    LC_REQUIRE(n_rows >= 0 && n_cols >= 0 && ld_out >= n_cols, LC_E_SHAPE, "lc_gather_f32: bad shape");
    if (n_rows == 0 || n_cols == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_GATHER, lc::as_stream(stream));
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {          // grid.y carries the row
        const int64_t nr = lc::imin(65535, n_rows - r0);
        dim3 grid((unsigned)lc::imin(lc::ceil_div<long long>(n_cols, 1024), 1024), (unsigned)nr);
        hipLaunchKernelGGL(k_gather, grid, dim3(256), 0, lc::as_stream(stream), d_rows ? d_in : d_in + r0 * ld_in,
                           (long long)ld_in, d_rows ? d_rows + r0 : nullptr, d_cols, (long long)n_cols, d_out + r0 * ld_out,
                           (long long)ld_out, d_live_cols);
    }
    return lc::launched("k_gather");
}

extern "C" int lc_scatter_axpy_f32(const float* d_w, int64_t ld_w, int64_t n_rows, const int32_t* d_cols,
                                   int64_t n_cols, float scale, float* d_acc, int64_t ld_acc, lc_stream_t stream) {
    LC_REQUIRE(d_w && d_acc, LC_E_BADARG, "lc_scatter_axpy_f32: null pointer");
    LC_REQUIRE(n_rows >= 0 && n_cols >= 0, LC_E_SHAPE, "lc_scatter_axpy_f32: bad shape");
    if (n_rows == 0 || n_cols == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_SCATTER, lc::as_stream(stream));
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {          // grid.y carries the row
        const int64_t nr = lc::imin(65535, n_rows - r0);
        dim3 grid((unsigned)lc::imin(lc::ceil_div<long long>(n_cols, 1024), 1024), (unsigned)nr);
        hipLaunchKernelGGL(k_scatter_axpy, grid, dim3(256), 0, lc::as_stream(stream), d_w + r0 * ld_w, (long long)ld_w, d_cols,
                           (long long)n_cols, scale, d_acc + r0 * ld_acc, (long long)ld_acc);
    }
    return lc::launched("k_scatter_axpy");
}

extern "C" int lc_scatter_cols(const void* d_src, int64_t ld_src, int64_t n_rows, int elem_bytes, const int32_t* d_cols,
                               int64_t n_cols, void* d_dst, int64_t ld_dst, lc_stream_t stream) {
    LC_REQUIRE(d_src && d_cols && d_dst, LC_E_BADARG, "lc_scatter_cols: null pointer");
    LC_REQUIRE(n_rows >= 0 && n_cols >= 0 && ld_src >= n_cols && (elem_bytes == 4 || elem_bytes == 8), LC_E_SHAPE,
               "lc_scatter_cols: bad shape (elements of 4 or 8 bytes)");
    if (n_rows == 0 || n_cols == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_SCATTER, lc::as_stream(stream));
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {          // grid.y carries the row
        const int64_t nr = lc::imin(65535, n_rows - r0);
        dim3 grid((unsigned)lc::imin(lc::ceil_div<long long>(n_cols, 256), 1024), (unsigned)nr);
        if (elem_bytes == 4)
            hipLaunchKernelGGL(k_scatter_cols<float>, grid, dim3(256), 0, lc::as_stream(stream),
                               static_cast<const float*>(d_src) + r0 * ld_src, (long long)ld_src, d_cols, (long long)n_cols,
                               static_cast<float*>(d_dst) + r0 * ld_dst, (long long)ld_dst);
        else
            hipLaunchKernelGGL(k_scatter_cols<double>, grid, dim3(256), 0, lc::as_stream(stream),
                               static_cast<const double*>(d_src) + r0 * ld_src, (long long)ld_src, d_cols, (long long)n_cols,
                               static_cast<double*>(d_dst) + r0 * ld_dst, (long long)ld_dst);
    }
    return lc::launched("k_scatter_cols");
}

extern "C" int lc_gemv_cols_f32(const float* d_a, int64_t lda, int64_t M, int64_t K, const float* d_y, int64_t ldy,
                                const int32_t* d_rows, const int32_t* d_sel, int ns, int32_t want, float* d_c, int64_t ldc,
                                lc_stream_t stream) {
    LC_REQUIRE(d_a && d_y && d_c, LC_E_BADARG, "lc_gemv_cols_f32: null pointer");
    LC_REQUIRE(M >= 0 && K >= 0 && ns >= 1 && ns <= GV_NS && K % 4 == 0 && lda >= K && lda % 4 == 0 && ldc >= ns &&
                   (reinterpret_cast<uintptr_t>(d_a) & 15) == 0,
               LC_E_SHAPE, "lc_gemv_cols_f32: need 1 <= ns <= %d, K %% 4 == 0, lda >= K, lda %% 4 == 0, A 16-byte aligned", GV_NS);
    if (M == 0) return LC_OK;                        // (K == 0: an empty sum -- the kernel writes the zeros)
    hipLaunchKernelGGL(k_gemv_cols, dim3((unsigned)lc::ceil_div<long long>(M, 8)), dim3(256), 0, lc::as_stream(stream), d_a,
                       (long long)lda, (long long)M, (long long)K, d_y, (long long)ldy, d_rows, d_sel, ns, (int)want, d_c,
                       (long long)ldc);
    return lc::launched("k_gemv_cols");
}

extern "C" int lc_invert_perm(const int32_t* d_cols, int64_t n_cols, int32_t base, int32_t* d_pos, lc_stream_t stream) {
    LC_REQUIRE(d_cols && d_pos, LC_E_BADARG, "lc_invert_perm: null pointer");
    LC_REQUIRE(n_cols >= 0 && n_cols < (1ll << 31) && base >= 0, LC_E_SHAPE, "lc_invert_perm: bad shape");
    if (n_cols == 0) return LC_OK;
    hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)lc::ceil_div<long long>(n_cols, 256)), dim3(256), 0,
                       lc::as_stream(stream), d_cols, (long long)n_cols, (int)base, d_pos);
    return lc::launched("k_invert_perm");
}

extern "C" int lc_combine_folds_f32(const float* const* w, const int64_t* ld_w, const int32_t* const* pos,
                                    const float* scale, int n_folds, int64_t n_rows, int64_t n_cols, float* d_out,
                                    int64_t ld_out, lc_stream_t stream) {
    LC_REQUIRE(w && ld_w && pos && scale && d_out, LC_E_BADARG, "lc_combine_folds_f32: null pointer");
    LC_REQUIRE(n_folds > 0 && n_rows >= 0 && n_cols >= 0 && ld_out >= n_cols, LC_E_SHAPE, "lc_combine_folds_f32: bad shape");
    if (n_rows == 0 || n_cols == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_SCATTER, lc::as_stream(stream));
    const dim3 grid((unsigned)lc::imin(lc::ceil_div<long long>(n_cols, 256), 4096),
                    (unsigned)lc::ceil_div<long long>(n_rows, CF_ROWS));
    LC_REQUIRE(grid.y <= 65535, LC_E_SHAPE, "lc_combine_folds_f32: too many rows");
    for (int f0 = 0; f0 < n_folds; f0 += CF_MAX) {               // folds in order, CF_MAX per pass
        CombineArgs a{};
        a.n = (int)lc::imin(CF_MAX, n_folds - f0);
        for (int f = 0; f < a.n; ++f) {
            LC_REQUIRE(w[f0 + f] && pos[f0 + f], LC_E_BADARG, "lc_combine_folds_f32: null fold pointer");
            a.w[f] = w[f0 + f];
            a.pos[f] = pos[f0 + f];
            a.ld[f] = ld_w[f0 + f];
            a.scale[f] = scale[f0 + f];
        }
        hipLaunchKernelGGL(k_combine_folds, grid, dim3(256), 0, lc::as_stream(stream), a, (long long)n_rows, (long long)n_cols,
                           d_out, (long long)ld_out, f0 > 0 ? 1 : 0);
    }
    return lc::launched("k_combine_folds");
}

extern "C" int lc_col_mean_std_f32(const float* d_x, int64_t ld, const int32_t* d_rows, int64_t n_rows,
                                   int64_t n_cols, float* d_mean, float* d_std, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_mean && d_std, LC_E_BADARG, "lc_col_mean_std_f32: null pointer");
    LC_REQUIRE(n_rows > 0 && n_cols >= 0, LC_E_SHAPE, "lc_col_mean_std_f32: need at least one row");
    if (n_cols == 0) return LC_OK;
    hipLaunchKernelGGL(k_col_mean_std, dim3((unsigned)lc::ceil_div<long long>(n_cols, 64)), dim3(64, CM_RG), 0,
                       lc::as_stream(stream), d_x, ld, d_rows, n_rows, n_cols, d_mean, d_std);
    return lc::launched("k_col_mean_std");
}

extern "C" int lc_col_normalize_f32(float* d_x, int64_t ld, int64_t n_rows, int64_t n_cols, const float* d_mean,
                                    const float* d_std, float eps, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_mean && d_std, LC_E_BADARG, "lc_col_normalize_f32: null pointer");
    if (n_rows <= 0 || n_cols <= 0) return LC_OK;
    dim3 grid((unsigned)n_rows, (unsigned)lc::imin(lc::ceil_div<long long>(n_cols, 1024), 1024));
    hipLaunchKernelGGL(k_col_normalize, grid, dim3(256), 0, lc::as_stream(stream), d_x, ld, n_cols, d_mean, d_std,
                       eps);
    return lc::launched("k_col_normalize");
}

namespace {
// The same statistics with the validation rows held in registers: one pass over Y instead of three.  A wave owns
// whole 32-row blocks (b = ty, ty + CM_RG, ...; at most NBLK of them), so the values it loaded are the ones whose
// block sums and row-quad copies it writes.  The fp64 sums behind mean and variance run over a wave's own rows first, then
// over the waves of the block.
template <int NBLK>
__global__ void __launch_bounds__(64 * CM_RG) k_val_stats_regs(const float* __restrict__ y, long long ldy, long long V,
                                                               const int* __restrict__ va, int M, FoldRows fr,
                                                               float* __restrict__ ystat, float* __restrict__ yblk,
                                                               float* __restrict__ yv, const int* __restrict__ live_cols) {
    if (live_cols && (long long)blockIdx.x * 64 >= (((long long)*live_cols + 255) & ~255ll)) return;   // (block-uniform: no voxel here)
    __shared__ double sm[CM_RG][64];
    const int fold = blockIdx.y, n_val = fr.n_val[fold];
    va += (long long)fold * M;
    ystat += (long long)fold * 3 * V;
    yblk += (long long)fold * (M / LC_MB) * V;
    yv += (long long)fold * M * V;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = c < V;
    const int nblocks = M / LC_MB;
    float cache[NBLK][LC_MB];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
        const int b = ty + k * CM_RG;
#pragma unroll
        for (int r = 0; r < LC_MB; ++r) {
            const int i = b * LC_MB + r;
            cache[k][r] = (live && b < nblocks && i < n_val) ? y[(long long)va[i] * ldy + c] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NBLK; ++k)
#pragma unroll
        for (int r = 0; r < LC_MB; ++r) s += (double)cache[k][r];          // padding entries are exact zeros
    const double mean = block_colsum<CM_RG>(s, sm) / (double)n_val;
    const float meanf = (float)mean;
    double q = 0.0;
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
        const int b = ty + k * CM_RG;
#pragma unroll
        for (int r = 0; r < LC_MB; ++r)
            if (b * LC_MB + r < n_val && b < nblocks) {
                const double d = (double)cache[k][r] - mean;
                q += d * d;
            }
    }
    const double m2 = block_colsum<CM_RG>(q, sm);
    if (live && ty == 0) {
        const double var = m2 / (double)(n_val - 1);
        ystat[c] = meanf;
        ystat[V + c] = (float)sqrt(var);
        ystat[2 * V + c] = (float)var;
    }
    if (live)
#pragma unroll
        for (int k = 0; k < NBLK; ++k) {
            const int b = ty + k * CM_RG;
            if (b < nblocks) {
                float t = 0.f;
#pragma unroll
                for (int r4 = 0; r4 < LC_MB; r4 += 4) {
                    float4 quad;                             // row-quad interleaved layout, see lc_epilogue.h
                    float* qv = reinterpret_cast<float*>(&quad);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        qv[j] = cache[k][r4 + j];
                        if (b * LC_MB + r4 + j < n_val) t += qv[j] - meanf;
                    }
                    reinterpret_cast<float4*>(yv)[(long long)((b * LC_MB + r4) >> 2) * V + c] = quad;
                }
                yblk[(long long)b * V + c] = t;
            }
        }
}

// More than 512 validation rows (the LeBel-style pipeline: 1 844 per inner fold): the same statistics in two passes over
// the rows -- pass 1 the mean, pass 2 the variance about it, the block sums and the row-quad copies -- a wave taking every
// CM_RG-th 32-row block with two register sets, so that the 32 loads of its next block are in flight while it works on
// this one.  (Rounds 1-3 walked the rows three times with one load in flight per thread: 0.8 TB/s of the 2.95 GB x 2 per
// fit there; a first round-4 version loaded 64 rows per trip and waited for them: 2.1 TB/s.)
template <int NBLK>
__global__ void __launch_bounds__(64 * CM_RG) k_val_stats_chunks(const float* __restrict__ y, long long ldy, long long V,
                                                                 const int* __restrict__ va, int M, FoldRows fr,
                                                                 float* __restrict__ ystat, float* __restrict__ yblk,
                                                                 float* __restrict__ yv, const int* __restrict__ live_cols) {
    static_assert(NBLK == 1, "one 32-row block per wave and trip, two register sets in flight");
    if (live_cols && (long long)blockIdx.x * 64 >= (((long long)*live_cols + 255) & ~255ll)) return;   // (block-uniform: no voxel here)
    __shared__ double sm[CM_RG][64];
    const int fold = blockIdx.y, n_val = fr.n_val[fold];
    va += (long long)fold * M;
    ystat += (long long)fold * 3 * V;
    yblk += (long long)fold * (M / LC_MB) * V;
    yv += (long long)fold * M * V;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = c < V;
    const int nblocks = M / LC_MB;
    // a wave walks the blocks ty, ty + CM_RG, ...; the loads of its NEXT block are in flight while it works on this one
    auto load = [&](float (&buf)[LC_MB], int b) {
#pragma unroll
        for (int r = 0; r < LC_MB; ++r) {
            const int i = b * LC_MB + r;
            buf[r] = (live && b < nblocks && i < n_val) ? y[(long long)va[i] * ldy + c] : 0.f;
        }
    };
    float A[LC_MB], B[LC_MB];
    double s = 0.0;
    auto sum = [&](const float (&buf)[LC_MB]) {
#pragma unroll
        for (int r = 0; r < LC_MB; ++r) s += (double)buf[r];              // padding entries are exact zeros
    };
    load(A, ty);
    for (int b = ty; b < nblocks; b += 2 * CM_RG) {
        load(B, b + CM_RG);
        sum(A);
        load(A, b + 2 * CM_RG);
        sum(B);
    }
    const double mean = block_colsum<CM_RG>(s, sm) / (double)n_val;
    const float meanf = (float)mean;
    double q = 0.0;
    auto finish = [&](const float (&buf)[LC_MB], int b) {
        if (b >= nblocks) return;
        float t = 0.f;
#pragma unroll
        for (int r4 = 0; r4 < LC_MB; r4 += 4) {
            float4 quad;                                     // row-quad interleaved layout, see lc_epilogue.h
            float* qv = reinterpret_cast<float*>(&quad);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qv[j] = buf[r4 + j];
                if (b * LC_MB + r4 + j < n_val) {
                    const double d = (double)qv[j] - mean;
                    q += d * d;
                    t += qv[j] - meanf;
                }
            }
            if (live) reinterpret_cast<float4*>(yv)[(long long)((b * LC_MB + r4) >> 2) * V + c] = quad;
        }
        if (live) yblk[(long long)b * V + c] = t;
    };
    load(A, ty);
    for (int b = ty; b < nblocks; b += 2 * CM_RG) {
        load(B, b + CM_RG);
        finish(A, b);
        load(A, b + 2 * CM_RG);
        finish(B, b + CM_RG);
    }
    const double m2 = block_colsum<CM_RG>(q, sm);
    if (live && ty == 0) {
        const double var = m2 / (double)(n_val - 1);
        ystat[c] = meanf;
        ystat[V + c] = (float)sqrt(var);
        ystat[2 * V + c] = (float)var;
    }
}
}  // namespace

extern "C" int lc_val_stats_folds(const float* d_y, int64_t ldy, int64_t V, const int32_t* d_va, int F, int M,
                                  const int32_t* h_n_val, float* d_ystat, float* d_yblk, float* d_yv,
                                  const int32_t* d_live_cols, lc_stream_t stream) {
    LC_REQUIRE(d_y && d_va && h_n_val && d_ystat && d_yblk && d_yv, LC_E_BADARG, "lc_val_stats: null pointer");
    LC_REQUIRE(F >= 1 && F <= 64 && M > 0 && M % LC_MB == 0, LC_E_SHAPE, "lc_val_stats: need 1 <= F <= 64, M %% %d == 0",
               LC_MB);
    FoldRows fr{};
    for (int f = 0; f < F; ++f) {
        LC_REQUIRE(h_n_val[f] > 0 && h_n_val[f] <= M, LC_E_SHAPE, "lc_val_stats: need 0 < n_val <= M");
        fr.n_val[f] = h_n_val[f];
    }
    if (V <= 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_VAL_STATS, lc::as_stream(stream));
    const dim3 grid((unsigned)lc::ceil_div<long long>(V, 64), (unsigned)F), block(64, CM_RG);
    if (M / LC_MB <= 2 * CM_RG)                              // up to 512 validation rows: held in registers, one pass
        hipLaunchKernelGGL(k_val_stats_regs<2>, grid, block, 0, lc::as_stream(stream), d_y, (long long)ldy, (long long)V, d_va,
                           M, fr, d_ystat, d_yblk, d_yv, d_live_cols);
    else                                                     // any number: two passes of register-sized chunks
        hipLaunchKernelGGL(k_val_stats_chunks<1>, grid, block, 0, lc::as_stream(stream), d_y, (long long)ldy, (long long)V,
                           d_va, M, fr, d_ystat, d_yblk, d_yv, d_live_cols);
    return lc::launched("k_val_stats");
}

extern "C" int lc_val_stats(const float* d_y, int64_t ldy, int64_t V, const int32_t* d_va, int M, int n_val,
                            float* d_ystat, float* d_yblk, float* d_yv, lc_stream_t stream) {
    const int32_t n = n_val;
    return lc_val_stats_folds(d_y, ldy, V, d_va, 1, M, &n, d_ystat, d_yblk, d_yv, nullptr, stream);
}

extern "C" int lc_pearson_cols(const float* d_a, int64_t lda, const float* d_b, int64_t ldb, int64_t n, int64_t V,
                               double* d_r, lc_stream_t stream) {
    LC_REQUIRE(d_a && d_b && d_r, LC_E_BADARG, "lc_pearson_cols: null pointer");
    LC_REQUIRE(n > 0 && V >= 0 && lda >= V && ldb >= V, LC_E_SHAPE, "lc_pearson_cols: bad shape");
    if (V == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_PEARSON, lc::as_stream(stream));
    if (n <= PQ_RG * 8)
        launch_pearson_regs<8>(d_a, lda, nullptr, nullptr, d_b, ldb, n, V, d_r, lc::as_stream(stream));
    else if (n <= PQ_RG * 20)
        launch_pearson_regs<20>(d_a, lda, nullptr, nullptr, d_b, ldb, n, V, d_r, lc::as_stream(stream));
    else
        hipLaunchKernelGGL(k_pearson_cols, dim3((unsigned)lc::ceil_div<long long>(V, 64)), dim3(64, PR_RG), 0,
                           lc::as_stream(stream), d_a, lda, d_b, ldb, n, V, d_r);
    return lc::launched("k_pearson_cols");
}

extern "C" int lc_pearson_cols_gather(const float* d_y, int64_t ld_y, const int32_t* d_rows, const int32_t* d_cols,
                                      const float* d_b, int64_t ldb, int64_t n, int64_t V, double* d_r,
                                      lc_stream_t stream) {
    LC_REQUIRE(d_y && d_b && d_r && d_rows, LC_E_BADARG, "lc_pearson_cols_gather: null pointer");
    LC_REQUIRE(n > 0 && n <= PQ_RG * 20 && V >= 0 && ldb >= V, LC_E_SHAPE,
               "lc_pearson_cols_gather: 1 <= n <= %d rows", PQ_RG * 20);
    if (V == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_PEARSON, lc::as_stream(stream));
    if (n <= PQ_RG * 8)
        launch_pearson_regs<8>(d_y, ld_y, d_rows, d_cols, d_b, ldb, n, V, d_r, lc::as_stream(stream));
    else
        launch_pearson_regs<20>(d_y, ld_y, d_rows, d_cols, d_b, ldb, n, V, d_r, lc::as_stream(stream));
    return lc::launched("k_pearson_cols_regs");
}

extern "C" int lc_undecided_cols(const float* d_scores, int A, int64_t ld, int64_t V, float tau_sum, const float* d_ystat,
                                 int64_t ld_stat, uint8_t* d_flags, int32_t* d_block_count, int32_t* d_list, int cap,
                                 int32_t* d_count, lc_stream_t stream) {
    LC_REQUIRE(d_scores && d_flags && d_block_count && d_list && d_count, LC_E_BADARG, "lc_undecided_cols: null pointer");
    LC_REQUIRE(A > 0 && V > 0 && V < (1ll << 31) && ld >= V && cap > 0 && tau_sum >= 0.f && (!d_ystat || ld_stat >= V),
               LC_E_SHAPE, "lc_undecided_cols: need A > 0, 0 < V <= ld, cap > 0, tau_sum >= 0");
    hipStream_t s = lc::as_stream(stream);
    const int nb = (int)lc::ceil_div<long long>(V, UD_THREADS);
    hipLaunchKernelGGL(k_undecided_flags, dim3((unsigned)nb), dim3(UD_THREADS), 0, s, d_scores, A, (long long)ld, (long long)V,
                       tau_sum, d_ystat, (long long)ld_stat, d_flags, d_block_count);
    if (int rc = lc::launched("k_undecided_flags")) return rc;
    hipLaunchKernelGGL(k_undecided_place, dim3((unsigned)nb), dim3(UD_THREADS), 0, s, d_flags, (long long)V, d_block_count, nb,
                       d_list, cap, d_count);
    return lc::launched("k_undecided_place");
}

extern "C" int lc_kappa_sums(const float* d_ystat, int64_t ld_stat, int64_t V, double* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_ystat && d_out, LC_E_BADARG, "lc_kappa_sums: null pointer");
    LC_REQUIRE(V > 0 && ld_stat >= V, LC_E_SHAPE, "lc_kappa_sums: need 0 < V <= ld_stat");
    hipLaunchKernelGGL(k_kappa_sums, dim3(1), dim3(1024), 0, lc::as_stream(stream), d_ystat, (long long)ld_stat, (long long)V,
                       d_out);
    return lc::launched("k_kappa_sums");
}

extern "C" int lc_select_alpha(const float* d_scores, int A, int64_t V, int32_t* d_best, double* d_rowsum,
                               lc_stream_t stream) {
    LC_REQUIRE(d_scores && A > 0 && V >= 0, LC_E_BADARG, "lc_select_alpha: bad argument");
    if (V == 0) return LC_OK;
    hipStream_t s = lc::as_stream(stream);
    if (d_best) {
        hipLaunchKernelGGL(k_argmax_alpha, dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0, s, d_scores,
                           A, V, d_best);
        if (int rc = lc::launched("k_argmax_alpha")) return rc;
    }
    if (d_rowsum) {
        hipLaunchKernelGGL(k_rowsum, dim3((unsigned)A), dim3(1024), 0, s, d_scores, V, d_rowsum);
        if (int rc = lc::launched("k_rowsum")) return rc;
    }
    return LC_OK;
}

extern "C" int lc_group_by_alpha_range(const int32_t* d_best, int64_t V, int a0, int A, int pad, int32_t* d_perm,
                                       int32_t* d_count, lc_stream_t stream) {
    LC_REQUIRE(d_best && d_perm && d_count, LC_E_BADARG, "lc_group_by_alpha: null pointer");
    LC_REQUIRE(a0 >= 0 && A > 0 && A <= GB_MAX_A && pad >= 1, LC_E_SHAPE,
               "lc_group_by_alpha: a0 >= 0, A in 1..%d per call (larger grids: one call per range of alphas), pad >= 1", GB_MAX_A);
    LC_REQUIRE(V >= 0 && V < (1ll << 31), LC_E_SHAPE, "lc_group_by_alpha: V out of range");
    size_t lds = (size_t)A * GB_THREADS * sizeof(int);
    const int staged = lds + (size_t)V + 16 <= 150 * 1024 ? 1 : 0;      // the indices as bytes behind the histogram table
    if (staged) lds += ((size_t)V + 15) / 16 * 16;
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_group_by_alpha), 150 * 1024)) return rc;
    hipLaunchKernelGGL(k_group_by_alpha, dim3(1), dim3(GB_THREADS), lds, lc::as_stream(stream), d_best, V, a0, A, pad,
                       d_perm, d_count, staged);
    return lc::launched("k_group_by_alpha");
}

extern "C" int lc_group_by_alpha(const int32_t* d_best, int64_t V, int A, int pad, int32_t* d_perm,
                                 int32_t* d_count, lc_stream_t stream) {
    return lc_group_by_alpha_range(d_best, V, 0, A, pad, d_perm, d_count, stream);
}

extern "C" int lc_pearson_pvalues(const double* d_r, int64_t V, int64_t n, double* d_p, lc_stream_t stream) {
    LC_REQUIRE(d_r && d_p, LC_E_BADARG, "lc_pearson_pvalues: null pointer");
    LC_REQUIRE(V >= 0 && n >= 0, LC_E_SHAPE, "lc_pearson_pvalues: bad shape");
    if (V == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_PEARSON, lc::as_stream(stream));
    hipLaunchKernelGGL(k_pearson_pvalues, dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0,
                       lc::as_stream(stream), d_r, (long long)V, (long long)n, d_p);
    return lc::launched("k_pearson_pvalues");
}

extern "C" int lc_zscore_story_f64(const double* d_x, int64_t ld_in, int64_t rows, int64_t cols, int nan_to_num,
                                   double* d_out, int64_t ld_out, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_out, LC_E_BADARG, "lc_zscore_story_f64: null pointer");
    LC_REQUIRE(rows > 0 && cols >= 0 && ld_in >= cols && ld_out >= cols, LC_E_SHAPE, "lc_zscore_story_f64: bad shape");
    if (cols == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_COLSTATS, lc::as_stream(stream));
    hipLaunchKernelGGL(k_zscore_story, dim3((unsigned)lc::ceil_div<long long>(cols, 256)), dim3(256), 0,
                       lc::as_stream(stream), d_x, (long long)ld_in, (long long)rows, (long long)cols, nan_to_num, d_out,
                       (long long)ld_out);
    return lc::launched("k_zscore_story");
}
