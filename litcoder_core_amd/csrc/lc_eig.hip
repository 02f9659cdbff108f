// The reference's SVD semantics where the Cholesky route cannot follow it (gfx950, fp64): alpha = 0 and a `singcutoff`
// that really drops directions.
//
// ridge_utils.py:34-67 (`svd_wrapper`) takes the thin SVD  Rstim = U S Vh  and DROPS singular values <= singcutoff;
// ridge_regression.py:56,117 then shrink the kept ones by  S / (S^2 + a^2)  -- defined for a = 0 (pseudo-inverse).
// With  K = Rstim Rstim' = U S^2 U'  the same operators are
//       Pstim Vh' diag(S/(S^2+a^2)) U'  =  K[va,tr] U_k diag(1 / (lambda_k + a^2)) U_k'          (hat matrix)
//       Vh' diag(S/(S^2+a^2)) U'        =  Rstim'   U_k diag(1 / (lambda_k + a^2)) U_k'          (weights operator)
// over the kept eigenpairs (lambda_k = S_k^2 > singcutoff^2, at most min(n, p) of them: the thin SVD has no more).
// So this route needs the symmetric eigendecomposition of K[tr,tr] itself:
//   * lc_batch_eigh_jacobi -- cyclic Jacobi with the round-robin parallel ordering: per step n/2 disjoint rotations,
//     A <- J' A J done row-wise (a workgroup owns the row pair (p, q): it first applies the step's COLUMN rotations to
//     its two rows -- element pairs inside a row, through LDS -- then mixes the two rows), V' <- J' V' the same way.
//     Quadratically convergent, every access coalesced, no pivoting or deflation logic; ~10 sweeps of n - 1 steps.
//     Slow next to the Cholesky route (memory-bound: 4 n^2 doubles move per step) -- it is taken only for penalty
//     grids that route cannot serve (nested_cv.check_penalties).
//   * lc_batch_spectral_apply -- H[f, a] = (R_f V_f) diag(keep / (lambda + a^2)) V_f' as two fp64 tile products per
//     (fold, alpha), written as f32 like the Cholesky route's operators.
#include "lc_common.h"

#include <vector>

namespace {

constexpr int JT = 256;

// pair k (0 <= k < n/2) of step s of the round-robin tournament on n players (n even): player n-1 stays, the others
// rotate.  Every pair of players meets exactly once in the n-1 steps of a sweep.
__device__ inline void jac_pair(int n, int s, int k, int& p, int& q) {
    const int m = n - 1;
    int a, b;
    if (k == 0) { a = m; b = s % m; }
    else { a = (s + k) % m; b = (s - k + m) % m; }
    p = a < b ? a : b;
    q = a < b ? b : a;
}

// Rotation of every pair of the step from the current diagonal / off-diagonal entries (Rutishauser's formulas), and per
// COLUMN j the triple (partner, c, sigma) with  (A J)[i][j] = c A[i][j] + sigma A[i][partner].
struct ColRot { int partner; double c, sg; };

__global__ void __launch_bounds__(JT) k_jac_params(const double* __restrict__ A, int n, int n_real, int step, double tol,
                                                   const double* __restrict__ norm, double* __restrict__ rot,
                                                   ColRot* __restrict__ col, int* __restrict__ flag) {
    const int k = blockIdx.x * JT + threadIdx.x, b = blockIdx.y;
    if (k >= n / 2) return;
    int p, q;
    jac_pair(n, step, k, p, q);
    const double* a = A + (long long)b * n * n;
    double c = 1.0, s = 0.0;
    if (q < n_real) {                                    // (the padding player of an odd-sized problem never rotates)
        const double app = a[(long long)p * n + p], aqq = a[(long long)q * n + q], apq = a[(long long)p * n + q];
        // relative to the pair's own diagonal, but never below tol x 1e-2 x the system's largest diagonal entry: inside
        // the numerically-zero part of the spectrum of a rank-deficient block nothing is worth rotating (and would
        // never settle) -- what is dropped or kept is decided at >= 1e-6 of lambda_max
        const double lim = tol * fmax(sqrt(fabs(app) * fabs(aqq)), 1.0e-2 * norm[b]);
        if (fabs(apq) > lim && apq != 0.0) {
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            s = t * c;
            atomicOr(flag + b, 1);
        }
    }
    double* r = rot + ((long long)b * (n / 2) + k) * 2;
    r[0] = c; r[1] = s;
    ColRot* cr = col + (long long)b * n;
    cr[p] = ColRot{q, c, -s};
    cr[q] = ColRot{p, c, s};
}

// One workgroup per row pair (p, q) of the step: rows p, q of A (column rotations inside each row, then the row
// rotation) and rows p, q of V' (row rotation only).
__global__ void __launch_bounds__(JT) k_jac_apply(double* __restrict__ A, double* __restrict__ Vt, int n, int step,
                                                  const double* __restrict__ rot, const ColRot* __restrict__ col) {
    extern __shared__ double sh[];                       // [2][n]
    const int k = blockIdx.x, b = blockIdx.y;
    int p, q;
    jac_pair(n, step, k, p, q);
    double* a = A + (long long)b * n * n;
    double* v = Vt + (long long)b * n * n;
    const double* r = rot + ((long long)b * (n / 2) + k) * 2;
    const double c = r[0], s = r[1];
    const ColRot* cr = col + (long long)b * n;
    double* rp = sh;
    double* rq = sh + n;
    for (int j = threadIdx.x; j < n; j += JT) {
        rp[j] = a[(long long)p * n + j];
        rq[j] = a[(long long)q * n + j];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += JT) {
        const ColRot cj = cr[j];
        const double xp = cj.c * rp[j] + cj.sg * rp[cj.partner];     // (A J)[p][j]
        const double xq = cj.c * rq[j] + cj.sg * rq[cj.partner];     // (A J)[q][j]
        a[(long long)p * n + j] = c * xp - s * xq;                   // (J' A J)[p][j]
        a[(long long)q * n + j] = s * xp + c * xq;
        const double vp = v[(long long)p * n + j], vq = v[(long long)q * n + j];
        v[(long long)p * n + j] = c * vp - s * vq;
        v[(long long)q * n + j] = s * vp + c * vq;
    }
}

__global__ void __launch_bounds__(JT) k_eye_batch(double* __restrict__ Vt, int n) {
    const int i = blockIdx.x, b = blockIdx.y;
    double* row = Vt + ((long long)b * n + i) * n;
    for (int j = threadIdx.x; j < n; j += JT) row[j] = j == i ? 1.0 : 0.0;
}

// largest |diagonal entry| per system: the scale of the convergence test
__global__ void __launch_bounds__(JT) k_jac_norm(const double* __restrict__ A, int n, double* __restrict__ norm) {
    __shared__ double red[JT];
    const int b = blockIdx.x;
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += JT) m = fmax(m, fabs(A[((long long)b * n + i) * n + i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = JT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) norm[b] = red[0];
}

// eigenvalues off the diagonal; lam_max per system
__global__ void __launch_bounds__(JT) k_eig_diag(const double* __restrict__ A, int n, int n_real, double* __restrict__ lam,
                                                 double* __restrict__ lmax) {
    __shared__ double red[JT];
    const int b = blockIdx.x;
    double m = -1.0e300;
    for (int i = threadIdx.x; i < n; i += JT) {
        const double x = i < n_real ? A[((long long)b * n + i) * n + i] : 0.0;
        lam[(long long)b * n + i] = x;
        if (i < n_real) m = fmax(m, x);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = JT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && lmax) lmax[b] = red[0];
}

// keep[f][j] = 1 when eigenpair j of system f survives: lambda > cutoff^2 AND among the rank_cap largest (the thin SVD
// of an (n x p) block has min(n, p) singular values; the rest of K's spectrum is rounding noise around zero).  One
// workgroup per system: rank of lambda_j = number of eigenvalues larger than it (ties by index).
__global__ void __launch_bounds__(JT) k_eig_keep(const double* __restrict__ lam, int n, int n_real, double cutoff2,
                                                 const int* __restrict__ rank_cap, unsigned char* __restrict__ keep,
                                                 int* __restrict__ n_kept) {
    __shared__ int cnt;
    const int f = blockIdx.x;
    const double* l = lam + (long long)f * n;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int cap = rank_cap ? rank_cap[f] : n_real;
    for (int j = threadIdx.x; j < n; j += JT) {
        int k = 0;
        if (j < n_real && l[j] > cutoff2) {
            int bigger = 0;
            for (int i = 0; i < n_real; ++i) bigger += (l[i] > l[j]) || (l[i] == l[j] && i < j);
            k = bigger < cap;
        }
        keep[(long long)f * n + j] = (unsigned char)k;
        if (k) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && n_kept) n_kept[f] = cnt;
}

// C (rows x cols) = A (rows x depth) B, fp64, 32 x 32 tiles through LDS.  BT: B is stored [cols][depth] (C = A B'),
// else [depth][cols].  SCALE: column k of A is multiplied by d[k] on the way in.  OUT32: C is stored as float.
template <bool BT, bool SCALE, bool OUT32>
__global__ void __launch_bounds__(256) k_mm64s(const double* __restrict__ A, long long lda, const double* __restrict__ B,
                                               long long ldb, void* __restrict__ Cv, long long ldc, int rows, int cols,
                                               int depth, const double* __restrict__ d) {
    __shared__ double sa[32][33], sb[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8 threads, 4 rows each
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < depth; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = ty * 4 + i;
            const int ar = r0 + rr, ak = k0 + tx;
            double x = (ar < rows && ak < depth) ? A[(long long)ar * lda + ak] : 0.0;
            if (SCALE && ak < depth) x *= d[ak];
            sa[rr][tx] = x;
            if (BT) {                                               // sb[k][c] = B[c0 + c][k0 + k]
                const int bc = c0 + rr, bk = k0 + tx;
                sb[tx][rr] = (bc < cols && bk < depth) ? B[(long long)bc * ldb + bk] : 0.0;
            } else {
                const int bk = k0 + rr, bc = c0 + tx;
                sb[rr][tx] = (bk < depth && bc < cols) ? B[(long long)bk * ldb + bc] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const double bv = sb[k][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fma(sa[ty * 4 + i][k], bv, acc[i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty * 4 + i, c = c0 + tx;
        if (r < rows && c < cols) {
            if (OUT32) static_cast<float*>(Cv)[(long long)r * ldc + c] = (float)acc[i];
            else static_cast<double*>(Cv)[(long long)r * ldc + c] = acc[i];
        }
    }
}

// d[f, a][j] = keep[f][j] / (lambda[f][j] + a2[f * A + a])
__global__ void __launch_bounds__(JT) k_eig_shrink(const double* __restrict__ lam, const unsigned char* __restrict__ keep,
                                                   const double* __restrict__ a2, int n, int A, double* __restrict__ d) {
    const int fa = blockIdx.x, f = fa / A;
    const double pen = a2[fa];
    for (int j = threadIdx.x; j < n; j += JT)
        d[(long long)fa * n + j] = keep[(long long)f * n + j] ? 1.0 / (lam[(long long)f * n + j] + pen) : 0.0;
}

}  // namespace

extern "C" int64_t lc_batch_eigh_work_bytes(int F, int n) {
    if (F <= 0 || n <= 0) return -1;
    const int64_t np = n + (n & 1);
    return (int64_t)F * (np / 2) * 2 * 8 + (int64_t)F * np * (int64_t)sizeof(ColRot) + (int64_t)F * 8 + (int64_t)F * 4 + 256;
}

extern "C" int lc_batch_eigh_jacobi(double* d_a, int F, int n, double* d_vt, double* d_lam, double* d_lmax, void* d_work,
                                    int64_t work_bytes, int max_sweeps, double tol, int32_t* h_sweeps, lc_stream_t stream) {
    LC_REQUIRE(d_a && d_vt && d_lam && d_work, LC_E_BADARG, "lc_batch_eigh_jacobi: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && n > 0 && n % 2 == 0 && n <= 8192 && max_sweeps > 0 && tol > 0.0, LC_E_SHAPE,
               "lc_batch_eigh_jacobi: need an even n <= 8192 (pad an odd system with a zero row and column)");
    LC_REQUIRE(work_bytes >= lc_batch_eigh_work_bytes(F, n), LC_E_SHAPE, "lc_batch_eigh_jacobi: workspace too small");
    hipStream_t s = lc::as_stream(stream);
    double* rot = static_cast<double*>(d_work);
    ColRot* col = reinterpret_cast<ColRot*>(rot + (long long)F * (n / 2) * 2);
    double* norm = reinterpret_cast<double*>(col + (long long)F * n);
    int* flag = reinterpret_cast<int*>(norm + F);
    const int lds = 2 * n * (int)sizeof(double);
    if (lds > 48 * 1024)
        if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_jac_apply), lds)) return rc;
    hipLaunchKernelGGL(k_eye_batch, dim3(n, F), dim3(JT), 0, s, d_vt, n);
    std::vector<int> host_flag(F);
    int sweeps = 0;
    for (; sweeps < max_sweeps; ++sweeps) {
        LC_HIP(hipMemsetAsync(flag, 0, sizeof(int) * F, s));
        hipLaunchKernelGGL(k_jac_norm, dim3(F), dim3(JT), 0, s, d_a, n, norm);
        for (int step = 0; step < n - 1; ++step) {
            hipLaunchKernelGGL(k_jac_params, dim3((unsigned)lc::ceil_div(n / 2, JT), (unsigned)F), dim3(JT), 0, s, d_a, n, n,
                               step, tol, norm, rot, col, flag);
            hipLaunchKernelGGL(k_jac_apply, dim3((unsigned)(n / 2), (unsigned)F), dim3(JT), lds, s, d_a, d_vt, n, step, rot,
                               col);
        }
        if (int rc = lc::launched("jacobi sweep")) return rc;
        // one look per sweep: did any pair of any system still rotate?  (the slow route: a host round trip per sweep is
        // nothing next to the sweep's n - 1 steps)
        LC_HIP(hipMemcpyAsync(host_flag.data(), flag, sizeof(int) * F, hipMemcpyDeviceToHost, s));
        LC_HIP(hipStreamSynchronize(s));
        bool any = false;
        for (int f = 0; f < F; ++f) any |= host_flag[f] != 0;
        if (!any) { ++sweeps; break; }
    }
    if (h_sweeps) *h_sweeps = sweeps;
    hipLaunchKernelGGL(k_eig_diag, dim3(F), dim3(JT), 0, s, d_a, n, n, d_lam, d_lmax);
    return lc::launched("k_eig_diag");
}

extern "C" int64_t lc_batch_spectral_work_bytes(int F, int A, int n, int M) {
    if (F <= 0 || A <= 0 || n <= 0 || M <= 0) return -1;
    return (int64_t)F * n + (int64_t)F * 4 + ((int64_t)F * M * n + (int64_t)F * A * n) * 8 + 512;
}

extern "C" int lc_batch_spectral_apply(const double* d_lam, const double* d_vt, int F, int n, const double* d_r, int M,
                                       const double* d_a2, int A, double cutoff, const int32_t* d_rank_cap, void* d_work,
                                       int64_t work_bytes, float* d_h, const int32_t* h_slot, int32_t* d_kept,
                                       lc_stream_t stream) {
    LC_REQUIRE(d_lam && d_vt && d_r && d_a2 && d_work && d_h, LC_E_BADARG, "lc_batch_spectral_apply: null pointer");
    LC_REQUIRE(F > 0 && n > 0 && M > 0 && A > 0 && cutoff >= 0.0, LC_E_SHAPE, "lc_batch_spectral_apply: bad shape");
    LC_REQUIRE(work_bytes >= lc_batch_spectral_work_bytes(F, A, n, M), LC_E_SHAPE, "lc_batch_spectral_apply: workspace too small");
    hipStream_t s = lc::as_stream(stream);
    unsigned char* keep = static_cast<unsigned char*>(d_work);
    char* base = static_cast<char*>(d_work) + (((int64_t)F * n + 255) / 256) * 256;
    double* Z = reinterpret_cast<double*>(base);                          // (F, M, n) = R V
    double* D = Z + (long long)F * M * n;                                 // (F A, n)
    hipLaunchKernelGGL(k_eig_keep, dim3(F), dim3(JT), 0, s, d_lam, n, n, cutoff * cutoff, d_rank_cap, keep, d_kept);
    hipLaunchKernelGGL(k_eig_shrink, dim3((unsigned)(F * A)), dim3(JT), 0, s, d_lam, keep, d_a2, n, A, D);
    const dim3 grid((unsigned)lc::ceil_div(n, 32), (unsigned)lc::ceil_div(M, 32));
    for (int f = 0; f < F; ++f) {
        const double* Vt = d_vt + (long long)f * n * n;
        // Z_f = R_f V_f :  Z[m][j] = sum_i R[m][i] V'[j][i]   (B stored [cols][depth])
        hipLaunchKernelGGL((k_mm64s<true, false, false>), grid, dim3(256), 0, s, d_r + (long long)f * M * n, (long long)n, Vt,
                           (long long)n, (void*)(Z + (long long)f * M * n), (long long)n, M, n, n, (const double*)nullptr);
        for (int a = 0; a < A; ++a) {
            const long long slot = h_slot ? h_slot[f * A + a] : (long long)f * A + a;
            // H = (Z diag(d)) V'
            hipLaunchKernelGGL((k_mm64s<false, true, true>), grid, dim3(256), 0, s, Z + (long long)f * M * n, (long long)n, Vt,
                               (long long)n, (void*)(d_h + slot * M * n), (long long)n, M, n, n,
                               D + ((long long)f * A + a) * n);
        }
    }
    return lc::launched("lc_batch_spectral_apply");
}
