// Primal form for a handful of features (p <= 16): the whole nested CV from block products X'Y  (gfx950).
//
// With G = Rstim'Rstim (p x p) the ridge weights of one voxel are  w = (G + a^2 I)^-1 Rstim'y  and the prediction on a
// row set is X w (what the reference's thin SVD of a tall Rstim gives, ridge_utils.py:52, ridge_regression.py:104-120).
// For p this small every statistic the reference takes from a prediction -- its mean and unbiased std (z_score,
// ridge_utils.py:6-15), its co-moment with the targets (ridge_regression.py:124-133, nested_cv.py:152-155) -- is a
// p-dimensional linear or quadratic form in w:
//     sum_i pred_i = sx.w ,   sum_i (pred_i - mean)^2 = w'Sc w ,   sum_i (pred_i - mean)(y_i - ybar) = w.(X_c'y_c)
// (sx, Sc: column sums and centred scatter matrix of the rows' features), and  Rstim'y  of a training set is the sum of
// the block products of its row blocks.  So one HBM-bound pass over the targets per outer fold (k_xty: X'Y and the
// targets' own first two moments per row set, fp64 accumulation of exact fp32 products) replaces every V-wide GEMM of
// the fold, and two per-voxel kernels do the rest in fp64: k_primal_scores (all inner folds x all alphas) and
// k_primal_refit (weights at the chosen alpha, accumulated into W; Pearson r of the test rows).  No sorting by alpha,
// no grouped contraction, no fp16 split.
//
// Targets enter SHIFTED by one row of the fold (d = y - y[shrow]): a constant voxel then has exactly zero moments, as
// the reference's z_score / pearsonr of a constant vector do, and the centred sums lose nothing to a large mean.
#include "lc_common.h"

namespace {

constexpr int XT_RG = 4;          // row groups (waves) per block
constexpr int XT_UR = 8;          // rows in flight per thread

// ---- block products: part[s][y][k][c] = sum over rows chunk y of set s of X[row][k] * (Y[row][c] - Y[shrow[s]][c]),
//      k = PT: sum of the shifted targets, k = PT + 1: sum of their squares.  Slots y >= ceil(nrows[s] / chunk) are
//      not written (and not read by the consumers).  A thread owns CPT adjacent columns (one 8- or 16-byte load per
//      row: the pass is HBM-bound and needs the bytes in flight), a block 64 * CPT columns x XT_RG row groups.
//      Y: ldy a multiple of 4, columns V .. ldy-1 readable (the resident targets are zero-padded to the tile width).
template <int PT, int CPT>
__global__ void __launch_bounds__(64 * XT_RG) k_xty(const float* __restrict__ X, long long ldx, int p,
                                                    const float* __restrict__ Y, long long ldy, long long V,
                                                    const int* __restrict__ rows, int ldr,
                                                    const int* __restrict__ nrows, const int* __restrict__ shrow,
                                                    int chunk, int RS, double* __restrict__ part) {
    typedef float vec_t __attribute__((ext_vector_type(CPT)));
    __shared__ double red[XT_RG - 1][PT + 2][CPT][64];
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.z, yb = blockIdx.y;
    const int n = nrows[s];
    const int r0 = yb * chunk;
    if (r0 >= n) return;
    const int r1 = min(n, r0 + chunk);
    const long long c = ((long long)blockIdx.x * 64 + lane) * CPT;
    const bool live = c < V;
    const long long cc = live ? c : 0;
    const int* rl = rows + (long long)s * ldr;
    const vec_t shv = *reinterpret_cast<const vec_t*>(Y + (long long)shrow[s] * ldy + cc);
    double sh[CPT], acc[PT][CPT], sd[CPT], sdd[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        sh[j] = (double)shv[j];
        sd[j] = 0.0;
        sdd[j] = 0.0;
#pragma unroll
        for (int k = 0; k < PT; ++k) acc[k][j] = 0.0;
    }
    for (int i0 = r0 + g; i0 < r1; i0 += XT_RG * XT_UR) {
        vec_t yv[XT_UR];
        int ri[XT_UR];
#pragma unroll
        for (int u = 0; u < XT_UR; ++u) {
            ri[u] = rl[min(i0 + XT_RG * u, r1 - 1)];                 // wave-uniform: scalar loads
            yv[u] = *reinterpret_cast<const vec_t*>(Y + (long long)ri[u] * ldy + cc);
        }
#pragma unroll
        for (int u = 0; u < XT_UR; ++u) {
            if (i0 + XT_RG * u >= r1) break;
            const float* xr = X + (long long)ri[u] * ldx;
            double d[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                d[j] = (double)yv[u][j] - sh[j];
                sd[j] += d[j];
                sdd[j] += d[j] * d[j];
            }
#pragma unroll
            for (int k = 0; k < PT; ++k)
                if (k < p) {
                    const double xk = (double)xr[k];
#pragma unroll
                    for (int j = 0; j < CPT; ++j) acc[k][j] += xk * d[j];
                }
        }
    }
    if (g > 0) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
#pragma unroll
            for (int k = 0; k < PT; ++k) red[g - 1][k][j][lane] = acc[k][j];
            red[g - 1][PT][j][lane] = sd[j];
            red[g - 1][PT + 1][j][lane] = sdd[j];
        }
    }
    __syncthreads();
    if (g != 0 || !live) return;
    for (int q = 0; q < XT_RG - 1; ++q) {                            // fixed order: deterministic
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
#pragma unroll
            for (int k = 0; k < PT; ++k) acc[k][j] += red[q][k][j][lane];
            sd[j] += red[q][PT][j][lane];
            sdd[j] += red[q][PT + 1][j][lane];
        }
    }
    double* dst = part + ((long long)s * RS + yb) * (PT + 2) * V + c;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        if (c + j >= V) break;
#pragma unroll
        for (int k = 0; k < PT; ++k) dst[(long long)k * V + j] = acc[k][j];
        dst[(long long)PT * V + j] = sd[j];
        dst[(long long)(PT + 1) * V + j] = sdd[j];
    }
}

// ---- per row set: xstat[s] = [sx (PT) | X'X (PT x PT) | Sc (PT x PT)]: column sums, raw second moments and centred
//      scatter matrix  Sc = X'X - sx sx'/n  of the features over the set's rows, fp64 (products of fp32 are exact)
template <int PT>
__global__ void __launch_bounds__(1024) k_set_stats(const float* __restrict__ X, long long ldx, int p,
                                                    const int* __restrict__ rows, int ldr,
                                                    const int* __restrict__ nrows, double* __restrict__ xstat) {
    constexpr int NQ = PT + 2 * PT * PT, SL = 1024 / (PT * PT), TR = 512;     // TR rows staged per round trip
    __shared__ float tile[TR][PT + 1];
    __shared__ double red[SL][PT + PT * PT];
    __shared__ double tot[PT + PT * PT];
    const int s = blockIdx.x, n = nrows[s];
    const int* rl = rows + (long long)s * ldr;
    const int pair = threadIdx.x % (PT * PT), sl = threadIdx.x / (PT * PT);
    const int k = pair / PT, l = pair % PT;
    double q = 0.0, sx = 0.0;
    for (int i0 = 0; i0 < n; i0 += TR) {
        const int nr = min(TR, n - i0);
        for (int e = threadIdx.x; e < nr * PT; e += 1024) {
            const int r = e / PT, c = e % PT;
            tile[r][c] = c < p ? X[(long long)rl[i0 + r] * ldx + c] : 0.f;
        }
        __syncthreads();
        for (int r = sl; r < nr; r += SL) {
            const double a = (double)tile[r][k];
            q += a * (double)tile[r][l];
            if (l == 0) sx += a;
        }
        __syncthreads();
    }
    red[sl][PT + pair] = q;
    if (l == 0) red[sl][k] = sx;
    __syncthreads();
    if (threadIdx.x < PT + PT * PT) {
        double t = 0.0;
        for (int j = 0; j < SL; ++j) t += red[j][threadIdx.x];
        tot[threadIdx.x] = t;
        xstat[(long long)s * NQ + threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < PT * PT)
        xstat[(long long)s * NQ + PT + PT * PT + threadIdx.x] = tot[PT + pair] - tot[k] * tot[l] / (double)n;
}

// ---- per ridge system b = sys * A + a:  (G_sys + a2[b] I)^-1  with  G_sys = X'X[plus] - X'X[minus]  (a training set
//      as its outer block minus its validation block; minus = -1: the set itself).  Rows / columns >= p: identity.
//      gsys (optional): the G_sys themselves, (n_sys, PT, PT), for S[0]^2 = lambda_max.
template <int PT>
__global__ void __launch_bounds__(64) k_primal_gsys(const double* __restrict__ xstat, const int* __restrict__ sysdef,
                                                    int p, double* __restrict__ gsys) {
    constexpr int NQ = PT + 2 * PT * PT;
    const int sy = blockIdx.x, plus = sysdef[2 * sy], minus = sysdef[2 * sy + 1];
    for (int e = threadIdx.x; e < PT * PT; e += 64) {
        const int k = e / PT, l = e % PT;
        double v = xstat[(long long)plus * NQ + PT + e];
        if (minus >= 0) v -= xstat[(long long)minus * NQ + PT + e];
        if (k >= p || l >= p) v = 0.0;
        gsys[(long long)sy * PT * PT + e] = v;
    }
}

template <int PT>
__global__ void __launch_bounds__(64) k_primal_inverse(const double* __restrict__ gsys, const double* __restrict__ a2,
                                                       int A, int p, double* __restrict__ pinv,
                                                       int* __restrict__ info) {
    __shared__ double L[PT][PT + 1];
    __shared__ int bad;
    const int b = blockIdx.x, sy = b / A, t = threadIdx.x;
    if (t == 0) bad = 0;
    for (int e = t; e < PT * PT; e += 64) {
        const int k = e / PT, l = e % PT;
        double v = gsys[(long long)sy * PT * PT + e];
        if (k == l) v = k < p ? v + a2[b] : 1.0;
        L[k][l] = v;
    }
    __syncthreads();
    for (int j = 0; j < PT; ++j) {                                   // right-looking Cholesky, lower triangle
        if (t == 0) {
            double d = L[j][j];
            if (!(d > 0.0)) { bad = 1; d = 1.0; }
            L[j][j] = sqrt(d);
        }
        __syncthreads();
        if (t > j && t < PT) L[t][j] /= L[j][j];
        __syncthreads();
        for (int e = t; e < PT * PT; e += 64) {
            const int i = e / PT, l = e % PT;
            if (l > j && i >= l) L[i][l] -= L[i][j] * L[l][j];
        }
        __syncthreads();
    }
    if (t < PT) {                                                    // column t of the inverse: L z = e_t, L' x = z
        double z[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            double v = i == t ? 1.0 : 0.0;
#pragma unroll
            for (int l = 0; l < PT; ++l)
                if (l < i) v -= L[i][l] * z[l];
            z[i] = v / L[i][i];
        }
#pragma unroll
        for (int i = PT - 1; i >= 0; --i) {
            double v = z[i];
#pragma unroll
            for (int l = 0; l < PT; ++l)
                if (l > i) v -= L[l][i] * z[l];
            z[i] = v / L[i][i];
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) pinv[((long long)b * PT + i) * PT + t] = z[i];
    }
    if (t == 0) info[b] = bad;
}

// Sum of a set's partial block products for one quantity and one column.
__device__ inline double part_sum(const double* __restrict__ part, int RS, int nq, long long V, int s, int nchunks,
                                  int k, long long c) {
    const double* src = part + ((long long)s * RS * nq + k) * V + c;
    double t = 0.0;
    for (int y = 0; y < nchunks; ++y) t += src[(long long)y * nq * V];
    return t;
}

// ---- validation scores of every alpha, summed over the inner folds of one outer fold (correlation scoring,
//      ridge_regression.py:124-133: mean over the validation rows of z(y) z(pred), unbiased std + 1e-8 in both,
//      NaN -> 0; nested_cv.py:373-380 adds the folds' scores in fp32, fold order).
//      src: (F, 3) set numbers  [plus, minus (-1: none), validation]  -- training set = plus \ minus;
//      pinv: system f * A + a;  all sets of the fold share one shift row.
template <int PT>
__global__ void __launch_bounds__(256) k_primal_scores(const double* __restrict__ part, int RS, int chunk,
                                                       const int* __restrict__ nrows, const int* __restrict__ shrow,
                                                       const float* __restrict__ Y, long long ldy, long long V,
                                                       const int* __restrict__ src, const double* __restrict__ xstat,
                                                       const double* __restrict__ pinv, int F, int A, int p,
                                                       float* __restrict__ scores, long long lds) {
    constexpr int NQ = PT + 2 * PT * PT, NP = PT + 2;
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= V) {
        if (c < lds)
            for (int a = 0; a < A; ++a) scores[(long long)a * lds + c] = 0.f;
        return;
    }
    for (int f = 0; f < F; ++f) {
        const int plus = src[3 * f], minus = src[3 * f + 1], va = src[3 * f + 2];
        const int cp = (nrows[plus] + chunk - 1) / chunk, cv = (nrows[va] + chunk - 1) / chunk;
        const int cm = minus >= 0 ? (nrows[minus] + chunk - 1) / chunk : 0;
        const double sh = (double)Y[(long long)shrow[va] * ldy + c];
        const double n = (double)nrows[va];
        const double* xs_p = xstat + (long long)plus * NQ;
        const double* xs_m = xstat + (long long)(minus >= 0 ? minus : 0) * NQ;
        const double* xs_v = xstat + (long long)va * NQ;
        const double sd = part_sum(part, RS, NP, V, va, cv, PT, c);
        const double sdd = part_sum(part, RS, NP, V, va, cv, PT + 1, c);
        double b[PT], cy[PT];
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            b[k] = 0.0;
            cy[k] = 0.0;
            if (k < p) {
                double t = part_sum(part, RS, NP, V, plus, cp, k, c), sx = xs_p[k];
                if (minus >= 0) {
                    t -= part_sum(part, RS, NP, V, minus, cm, k, c);
                    sx -= xs_m[k];
                }
                b[k] = t + sh * sx;                                  // Rstim'y of the training rows (unshifted y)
                cy[k] = part_sum(part, RS, NP, V, va, cv, k, c) - xs_v[k] / n * sd;
            }
        }
        const double m2y = sdd - sd * sd / n;
        const double sy = sqrt(fmax(m2y, 0.0) / (n - 1.0));
        for (int a = 0; a < A; ++a) {
            const double* P = pinv + (long long)(f * A + a) * PT * PT;
            double w[PT];
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                double t = 0.0;
#pragma unroll
                for (int l = 0; l < PT; ++l) t += P[k * PT + l] * b[l];
                w[k] = t;
            }
            double m2 = 0.0, cov = 0.0;
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                double t = 0.0;
#pragma unroll
                for (int l = 0; l < PT; ++l) t += xs_v[PT + PT * PT + k * PT + l] * w[l];    // centred scatter, validation rows
                m2 += w[k] * t;
                cov += w[k] * cy[k];
            }
            const double sp = sqrt(fmax(m2, 0.0) / (n - 1.0));
            float score = (float)(cov / (n * (sy + 1e-8) * (sp + 1e-8)));
            if (score != score) score = 0.f;
            else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
            else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
            float* dst = scores + (long long)a * lds + c;
            *dst = f > 0 ? *dst + score : score;
        }
    }
}

// ---- refit at the chosen alpha and the test rows' Pearson r (ridge_regression.py:9-63, nested_cv.py:150-155):
//      w = pinv[best[c]] Rstim'y (outer training set `tr`),  W[:, c] += scale * fl32(w),
//      r = w.(X_c'y_c) / sqrt(w'Sc w . sum (y - ybar)^2) over the rows of set `te`, clipped to [-1, 1]; 0/0 -> NaN.
template <int PT>
__global__ void __launch_bounds__(256) k_primal_refit(const double* __restrict__ part, int RS, int chunk,
                                                      const int* __restrict__ nrows, const int* __restrict__ shrow,
                                                      const float* __restrict__ Y, long long ldy, long long V, int tr,
                                                      int te, const double* __restrict__ xstat,
                                                      const double* __restrict__ pinv, const int* __restrict__ best,
                                                      int p, float scale, float* __restrict__ W, long long ldw,
                                                      double* __restrict__ r_out) {
    constexpr int NQ = PT + 2 * PT * PT, NP = PT + 2;
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= V) return;
    const int ct = (nrows[tr] + chunk - 1) / chunk, ce = (nrows[te] + chunk - 1) / chunk;
    const double sh = (double)Y[(long long)shrow[tr] * ldy + c];
    const double n = (double)nrows[te];
    const double* xs_t = xstat + (long long)tr * NQ;
    const double* xs_e = xstat + (long long)te * NQ;
    const double sd = part_sum(part, RS, NP, V, te, ce, PT, c);
    const double sdd = part_sum(part, RS, NP, V, te, ce, PT + 1, c);
    double b[PT], cy[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        b[k] = 0.0;
        cy[k] = 0.0;
        if (k < p) {
            b[k] = part_sum(part, RS, NP, V, tr, ct, k, c) + sh * xs_t[k];
            cy[k] = part_sum(part, RS, NP, V, te, ce, k, c) - xs_e[k] / n * sd;
        }
    }
    const double* P = pinv + (long long)best[c] * PT * PT;
    double w[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < PT; ++l) t += P[k * PT + l] * b[l];
        w[k] = t;
    }
    double m2 = 0.0, cov = 0.0;
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < PT; ++l) t += xs_e[PT + PT * PT + k * PT + l] * w[l];
        m2 += w[k] * t;
        cov += w[k] * cy[k];
        if (k < p) {
            float* dst = W + (long long)k * ldw + c;
            *dst += scale * (float)w[k];
        }
    }
    const double m2y = sdd - sd * sd / n;
    double r = cov / (sqrt(fmax(m2, 0.0)) * sqrt(fmax(m2y, 0.0)));
    if (r > 1.0) r = 1.0;
    if (r < -1.0) r = -1.0;
    r_out[c] = r;
}

}  // namespace

#define LC_PT_DISPATCH(p_, CALL)                   \
    do {                                           \
        if ((p_) <= 4) { CALL(4); }                \
        else if ((p_) <= 8) { CALL(8); }           \
        else { CALL(16); }                         \
    } while (0)

extern "C" int lc_primal_pad(int p) { return p <= 4 ? 4 : (p <= 8 ? 8 : (p <= 16 ? 16 : -1)); }

extern "C" int lc_xty_f64(const float* d_x, int64_t ldx, int p, const float* d_y, int64_t ldy, int64_t V,
                          const int32_t* d_rows, int ldr, const int32_t* d_nrows, const int32_t* d_shrow, int n_sets,
                          int chunk, int RS, double* d_part, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_y && d_rows && d_nrows && d_shrow && d_part, LC_E_BADARG, "lc_xty_f64: null pointer");
    LC_REQUIRE(p > 0 && p <= 16 && ldx >= p && V >= 0 && ldy >= V && n_sets > 0 && n_sets <= 65535 && ldr > 0 &&
                   chunk > 0 && RS > 0 && RS <= 65535 && (long long)RS * chunk >= ldr && ldy % 4 == 0 &&
                   ldy >= (V + 3) / 4 * 4 && reinterpret_cast<uintptr_t>(d_y) % 16 == 0,
               LC_E_SHAPE, "lc_xty_f64: need 1 <= p <= 16, ldx >= p, RS * chunk >= ldr, and 16-byte aligned target rows "
               "padded to a multiple of 4 columns (ldy %% 4 == 0, ldy >= V rounded up)");
    if (V == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_GROUPED_GEMM, lc::as_stream(stream));
    // columns per thread: 4 (16-byte loads) while the accumulators fit a high occupancy, 2 for p_pad = 16
#define CALL(PT_)                                                                                                     \
    {                                                                                                                 \
        constexpr int CPT_ = PT_ <= 8 ? 4 : 2;                                                                        \
        const dim3 grid((unsigned)lc::ceil_div<long long>(V, 64 * CPT_), (unsigned)RS, (unsigned)n_sets);             \
        hipLaunchKernelGGL((k_xty<PT_, CPT_>), grid, dim3(64 * XT_RG), 0, lc::as_stream(stream), d_x, (long long)ldx, \
                           p, d_y, (long long)ldy, (long long)V, d_rows, ldr, d_nrows, d_shrow, chunk, RS, d_part);   \
    }
    LC_PT_DISPATCH(p, CALL);
#undef CALL
    return lc::launched("k_xty");
}

extern "C" int lc_primal_set_stats(const float* d_x, int64_t ldx, int p, const int32_t* d_rows, int ldr,
                                   const int32_t* d_nrows, int n_sets, double* d_xstat, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_rows && d_nrows && d_xstat, LC_E_BADARG, "lc_primal_set_stats: null pointer");
    LC_REQUIRE(p > 0 && p <= 16 && ldx >= p && n_sets > 0 && ldr > 0, LC_E_SHAPE, "lc_primal_set_stats: bad shape");
#define CALL(PT_)                                                                                                  \
    hipLaunchKernelGGL((k_set_stats<PT_>), dim3((unsigned)n_sets), dim3(1024), 0, lc::as_stream(stream), d_x,      \
                       (long long)ldx, p, d_rows, ldr, d_nrows, d_xstat)
    LC_PT_DISPATCH(p, CALL);
#undef CALL
    return lc::launched("k_set_stats");
}

extern "C" int lc_primal_gsys(const double* d_xstat, const int32_t* d_sysdef, int n_sys, int p, double* d_gsys,
                              lc_stream_t stream) {
    LC_REQUIRE(d_xstat && d_sysdef && d_gsys, LC_E_BADARG, "lc_primal_gsys: null pointer");
    LC_REQUIRE(p > 0 && p <= 16 && n_sys > 0, LC_E_SHAPE, "lc_primal_gsys: bad shape");
#define CALL(PT_)                                                                                                  \
    hipLaunchKernelGGL((k_primal_gsys<PT_>), dim3((unsigned)n_sys), dim3(64), 0, lc::as_stream(stream), d_xstat,   \
                       d_sysdef, p, d_gsys)
    LC_PT_DISPATCH(p, CALL);
#undef CALL
    return lc::launched("k_primal_gsys");
}

extern "C" int lc_primal_inverse(const double* d_gsys, const double* d_a2, int n_sys, int A, int p, double* d_pinv,
                                 int32_t* d_info, lc_stream_t stream) {
    LC_REQUIRE(d_gsys && d_a2 && d_pinv && d_info, LC_E_BADARG, "lc_primal_inverse: null pointer");
    LC_REQUIRE(p > 0 && p <= 16 && n_sys > 0 && A > 0, LC_E_SHAPE, "lc_primal_inverse: bad shape");
    lc::ScopedTimer timer_(lc::T_CHOL_SOLVE, lc::as_stream(stream));
#define CALL(PT_)                                                                                                  \
    hipLaunchKernelGGL((k_primal_inverse<PT_>), dim3((unsigned)(n_sys * A)), dim3(64), 0, lc::as_stream(stream),   \
                       d_gsys, d_a2, A, p, d_pinv, d_info)
    LC_PT_DISPATCH(p, CALL);
#undef CALL
    return lc::launched("k_primal_inverse");
}

extern "C" int lc_primal_scores(const double* d_part, int RS, int chunk, const int32_t* d_nrows, const int32_t* d_shrow,
                                const float* d_y, int64_t ldy, int64_t V, const int32_t* d_src, const double* d_xstat,
                                const double* d_pinv, int F, int A, int p, float* d_scores, int64_t lds,
                                lc_stream_t stream) {
    LC_REQUIRE(d_part && d_nrows && d_shrow && d_y && d_src && d_xstat && d_pinv && d_scores, LC_E_BADARG,
               "lc_primal_scores: null pointer");
    LC_REQUIRE(p > 0 && p <= 16 && F > 0 && A > 0 && RS > 0 && chunk > 0 && V >= 0 && ldy >= V && lds >= V, LC_E_SHAPE,
               "lc_primal_scores: bad shape");
    if (lds == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_SWEEP_FINALIZE, lc::as_stream(stream));
#define CALL(PT_)                                                                                                    \
    hipLaunchKernelGGL((k_primal_scores<PT_>), dim3((unsigned)lc::ceil_div<long long>(lds, 256)), dim3(256), 0,       \
                       lc::as_stream(stream), d_part, RS, chunk, d_nrows, d_shrow, d_y, (long long)ldy, (long long)V, \
                       d_src, d_xstat, d_pinv, F, A, p, d_scores, (long long)lds)
    LC_PT_DISPATCH(p, CALL);
#undef CALL
    return lc::launched("k_primal_scores");
}

extern "C" int lc_primal_refit(const double* d_part, int RS, int chunk, const int32_t* d_nrows, const int32_t* d_shrow,
                               const float* d_y, int64_t ldy, int64_t V, int set_train, int set_test,
                               const double* d_xstat, const double* d_pinv, const int32_t* d_best, int p, float scale,
                               float* d_w, int64_t ldw, double* d_r, lc_stream_t stream) {
    LC_REQUIRE(d_part && d_nrows && d_shrow && d_y && d_xstat && d_pinv && d_best && d_w && d_r, LC_E_BADARG,
               "lc_primal_refit: null pointer");
    LC_REQUIRE(p > 0 && p <= 16 && RS > 0 && chunk > 0 && V >= 0 && ldy >= V && ldw >= V && set_train >= 0 &&
                   set_test >= 0, LC_E_SHAPE, "lc_primal_refit: bad shape");
    if (V == 0) return LC_OK;
    lc::ScopedTimer timer_(lc::T_PEARSON, lc::as_stream(stream));
#define CALL(PT_)                                                                                                   \
    hipLaunchKernelGGL((k_primal_refit<PT_>), dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0,         \
                       lc::as_stream(stream), d_part, RS, chunk, d_nrows, d_shrow, d_y, (long long)ldy, (long long)V, \
                       set_train, set_test, d_xstat, d_pinv, d_best, p, scale, d_w, (long long)ldw, d_r)
    LC_PT_DISPATCH(p, CALL);
#undef CALL
    return lc::launched("k_primal_refit");
}
