// The V-independent dense algebra of the fit, in fp64:
//   Gram matrix K = X X', largest eigenvalue of fold sub-Grams (Lanczos + bisection), the assembly of the
//   augmented ridge systems and the polynomial-series operands.  The batched Cholesky solve that turns the
//   systems into hat matrices  H_a = Xva Xtr' (Xtr Xtr' + a^2 I)^-1  lives in lc_chol.hip.  Together they replace
//   the 30 thin SVDs of the reference (ridge_utils.py:52): with U S Vh = svd(Xtr),
//   Pstim Vh' diag(S/(S^2+a^2)) U' == H_a exactly.
//
// The GEMM-shaped pieces of THIS file go through one 64x64x64 register-tiled fp64 product.
#include "lc_common.h"

namespace {

constexpr int NB = LC_NB;   // 64

// ------------------------------------------------------------------ 64x64 tile product
// acc[4][4] (rows ty*4.., cols tx*4..) += sum_k Aop[m][k] * Bop[k][n] over a 64-deep K.
// The A operand is always given as global [m][k] (row stride lda); the B operand either as
// global [n][k] (BT = true: "N x T" product) or as [k][n] (BT = false).  Rows beyond
// a_rows / b_rows are treated as zero.  256 threads.
constexpr int TS_LD = 64 + 2;   // LDS row stride (doubles) of the k-major staged tiles

__device__ inline void stage_t(double* s, const double* __restrict__ g, long long ld, int rows_valid, int k0) {
    // global [idx][k] -> s[k][idx];  thread: idx = tid % 64, k = 8*(tid/64) .. +7
    const int idx = threadIdx.x & 63, kb = (threadIdx.x >> 6) * 8;
    if (idx < rows_valid) {
        const double* src = g + (long long)idx * ld + k0 + kb;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[(kb + e) * TS_LD + idx] = src[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[(kb + e) * TS_LD + idx] = 0.0;
    }
}

__device__ inline void stage_n(double* s, const double* __restrict__ g, long long ld, int k0) {
    // global [k][n] -> s[k][n];  thread: n = tid % 64, k = 8*(tid/64) .. +7
    const int n = threadIdx.x & 63, kb = (threadIdx.x >> 6) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[(kb + e) * TS_LD + n] = g[(long long)(k0 + kb + e) * ld + n];
}

template <bool BT>
__device__ inline void tile_product(double (&acc)[4][4], const double* __restrict__ A, long long lda, int a_rows,
                                    const double* __restrict__ B, long long ldb, int b_rows, double* sA, double* sB) {
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    for (int k0 = 0; k0 < NB; k0 += 32) {
        __syncthreads();
        stage_t(sA, A, lda, a_rows, k0);
        if (BT) stage_t(sB, B, ldb, b_rows, k0);
        else stage_n(sB, B, ldb, k0);
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[k * TS_LD + ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[k * TS_LD + tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------ Gram matrix
// K[i][j] = sum_k X[i][k] X[j][k], fp32 inputs, fp64 products and sums.  Lower-triangle tiles are
// computed and mirrored.
__global__ void __launch_bounds__(256) k_gram(const float* __restrict__ X, long long ldx, int T, int p,
                                              double* __restrict__ Kmat, long long ldk) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    __shared__ double sA[16 * TS_LD], sB[16 * TS_LD];
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;    // loader: row 0..63, k offset 0,4,8,12
    const int ra = bi * 64 + lr, rb = bj * 64 + lr;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < p; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + lk + e;
            sA[(lk + e) * TS_LD + lr] = (ra < T && k < p) ? (double)X[(long long)ra * ldx + k] : 0.0;
            sB[(lk + e) * TS_LD + lr] = (rb < T && k < p) ? (double)X[(long long)rb * ldx + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[k * TS_LD + ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[k * TS_LD + tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = bi * 64 + ty * 4 + i, c = bj * 64 + tx * 4 + j;
            if (r < T && c < T) {
                Kmat[(long long)r * ldk + c] = acc[i][j];
                if (bi != bj) Kmat[(long long)c * ldk + r] = acc[i][j];
            }
        }
}

// ------------------------------------------------------------------ lambda_max (Lanczos)
// Per fold f the work area holds: v[N], vprev[N], w[N], a[steps], b[steps], meta[8].
__device__ inline double* lz_base(double* work, int f, int N, int steps) {
    return work + (long long)f * (3ll * N + 2ll * steps + 8);
}

__global__ void __launch_bounds__(256) k_lz_init(const int* __restrict__ rows, int N, int steps, double* work,
                                                 int n_dense = 0) {
    __shared__ double red[256];
    const int f = blockIdx.x;
    double* base = lz_base(work, f, N, steps);
    double* v = base;
    double* vp = base + N;
    const int* rw = rows ? rows + (long long)f * N : nullptr;      // NULL: the leading n_dense entries are the system
    double ss = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        // fixed pseudo-random start vector (integer hash -> (-1, 1)), zero on padding rows
        unsigned h = (unsigned)i * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const bool live = rw ? rw[i] >= 0 : i < n_dense;
        const double x = live ? ((double)(h & 0xFFFFFF) / 8388608.0 - 1.0) + 1.5 : 0.0;
        v[i] = x;
        vp[i] = 0.0;
        ss += x * x;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const double inv = red[0] > 0.0 ? 1.0 / sqrt(red[0]) : 0.0;
    for (int i = threadIdx.x; i < N; i += 256) v[i] *= inv;
    if (threadIdx.x < 8) base[3ll * N + 2ll * steps + threadIdx.x] = 0.0;   // meta: [0]=steps done, [1]=stopped, [2]=prev beta
}

// w = K[rows, rows] v : one wave per row.
__global__ void __launch_bounds__(256) k_lz_symv(const double* __restrict__ Kmat, long long ldk, long long k_stride,
                                                 const int* __restrict__ rows, int N, int steps, double* work) {
    const int f = blockIdx.y;
    Kmat += (long long)f * k_stride;                  // 0: every system indexes the same matrix
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    double* base = lz_base(work, f, N, steps);
    if (base[3ll * N + 2ll * steps + 1] != 0.0) return;   // the system has stopped (invariant subspace / converged)
    const double* v = base;
    double* w = base + 2ll * N;
    const int* rw = rows + (long long)f * N;
    const int r = rw[i];
    double s = 0.0;
    if (r >= 0) {
        const double* krow = Kmat + (long long)r * ldk;
        for (int j = lane; j < N; j += 64) {
            const int c = rw[j];
            if (c >= 0) s += krow[c] * v[j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) w[i] = s;
}

// w = K_f[0:n, 0:n] v for systems that ARE the leading blocks of their own matrices (the primal form's p x p Gram
// matrices: no row lists).  The gather version above moves one 8-byte K element, one index and one vector element per
// lane and trip with one row per wave: 2.2 TB/s on six 3072 x 3072 systems -- and the Lanczos run sits at the head of a
// LeBel-shaped fit's critical path (13 of its 150 ms, round 4).  Here a block takes 16 rows of one system: the vector
// chunk goes through LDS once per block, a wave streams FOUR rows at a time with 16-byte loads (eight in flight per
// lane), so the iteration is bound by streaming the matrices.  Sums: per row a fixed lane-strided order, butterfly at the
// end (deterministic; not the gather version's order -- the two differ in the last bits).
constexpr int LZD_ROWS = 16, LZD_CHUNK = 4096;
__global__ void __launch_bounds__(256) k_lz_symv_dense(const double* __restrict__ Kmat, long long ldk, long long k_stride,
                                                       int N, int n, int steps, double* work) {
    __shared__ double vs[LZD_CHUNK];
    const int f = blockIdx.y;
    double* base = lz_base(work, f, N, steps);
    if (base[3ll * N + 2ll * steps + 1] != 0.0) return;   // stopped
    const double* v = base;
    double* w = base + 2ll * N;
    Kmat += (long long)f * k_stride;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i0 = blockIdx.x * LZD_ROWS + wv * 4;
    const double* kr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kr[q] = Kmat + (long long)min(i0 + q, n - 1) * ldk;    // rows past the end: clamped, unused
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int n2 = (n + 1) & ~1;                       // (n odd: the vector's entry n is a zero; K's column n is in the row's padding)
    for (int c0 = 0; c0 < n2; c0 += LZD_CHUNK) {
        const int cw = min(LZD_CHUNK, n2 - c0);
        __syncthreads();
        for (int j = threadIdx.x * 2; j < cw; j += 512) *reinterpret_cast<double2*>(vs + j) = *reinterpret_cast<const double2*>(v + c0 + j);
        __syncthreads();
        int j = lane * 2;
        for (; j + 128 < cw; j += 256) {               // two trips per pass: eight 16-byte loads in flight
            double2 k0[4], k1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                k0[q] = *reinterpret_cast<const double2*>(kr[q] + c0 + j);
                k1[q] = *reinterpret_cast<const double2*>(kr[q] + c0 + j + 128);
            }
            const double2 v0 = *reinterpret_cast<const double2*>(vs + j), v1 = *reinterpret_cast<const double2*>(vs + j + 128);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] += k0[q].x * v0.x;
                acc[q] += k0[q].y * v0.y;
                acc[q] += k1[q].x * v1.x;
                acc[q] += k1[q].y * v1.y;
            }
        }
        for (; j < cw; j += 128) {
            const double2 v0 = *reinterpret_cast<const double2*>(vs + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 k0 = *reinterpret_cast<const double2*>(kr[q] + c0 + j);
                acc[q] += k0.x * v0.x;
                acc[q] += k0.y * v0.y;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double t = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0 && i0 + q < N) w[i0 + q] = i0 + q < n ? t : 0.0;     // (the padding entries of the vectors stay zero)
    }
}

// Largest eigenvalue of the k x k tridiagonal (al, be) by 64-way multisection on the Sturm count, one wave: count(x) =
// number of eigenvalues < x; [lo, hi] shrinks around the smallest x with count(x) == k, 65-fold per round.
__device__ inline double lz_theta(const double* al, const double* be, int k, int rounds, int lane) {
    double lo = al[0], hi = al[0];
    for (int i = 0; i < k; ++i) {
        const double bl = i > 0 ? fabs(be[i - 1]) : 0.0, br = i + 1 < k ? fabs(be[i]) : 0.0;
        lo = fmin(lo, al[i] - bl - br);
        hi = fmax(hi, al[i] + bl + br);
    }
    for (int round = 0; round < rounds && hi > lo; ++round) {
        const double w = (hi - lo) / 65.0;
        const double x = lo + w * (lane + 1);
        int cnt = 0;
        double q = al[0] - x;
        if (q < 0.0) ++cnt;
        for (int i = 1; i < k; ++i) {
            const double den = q != 0.0 ? q : 1e-300;
            q = al[i] - x - be[i - 1] * be[i - 1] / den;
            if (q < 0.0) ++cnt;
        }
        const unsigned long long full = __ballot(cnt >= k);      // lanes whose x is above every eigenvalue
        const int first = full ? __ffsll((long long)full) - 1 : 64;
        const double nlo = first == 0 ? lo : lo + w * first;     // x of lane first-1
        const double nhi = first == 64 ? hi : lo + w * (first + 1);
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

// Sum over the 1024 threads of a block, every thread gets it: butterfly inside the waves, the 16 wave sums through LDS
// in fixed order (two barriers; the tree through LDS it replaces had eleven -- this kernel is 128 dependent launches of a
// Lanczos run and nothing but latency).
__device__ inline double block_sum_1024(double x, double* red) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    __syncthreads();
    return t;
}

// One Lanczos recurrence step per fold: a = v.w ; w -= a v + b_prev vprev ; b = |w| ; rotate.
// tol > 0: every LZ_CHECK steps from LZ_CHECK_FROM on the top Ritz value is computed (one wave, ~10 us) and the run of
// this system STOPS once it has moved by <= tol (relative) over the last LZ_CHECK steps; the later launches of the system
// return at once.  The top Ritz value rises monotonically and, past its first few digits, geometrically -- by 1e-3 per 16
// steps on the bench designs (profiles/r04_lanczos_convergence.txt): a move of <= 1e-6 over 8 steps leaves ~3e-8.
constexpr int LZ_CHECK = 8, LZ_CHECK_FROM = 24, LZ_CHECK_MAX = 256;
__global__ void __launch_bounds__(1024) k_lz_step(int N, int steps, int step, double* work,
                                                  const double* __restrict__ part = nullptr, int nsplit = 0,
                                                  const unsigned* __restrict__ member = nullptr, double tol = 0.0) {
    __shared__ double red[1024];
    __shared__ double s_al[LZ_CHECK_MAX], s_be[LZ_CHECK_MAX];
    __shared__ int s_conv;
    const int f = blockIdx.x;
    double* base = lz_base(work, f, N, steps);
    double* v = base;
    double* vp = base + N;
    double* w = base + 2ll * N;
    double* al = base + 3ll * N;
    double* be = al + steps;
    double* meta = be + steps;
    if (meta[1] != 0.0) return;                       // already stopped (invariant subspace found)
    if (part)                                         // masked multi-system matvec: column-split partial sums, fixed order
        for (int i = threadIdx.x; i < N; i += 1024) {
            double sum = 0.0;
            for (int sp = 0; sp < nsplit; ++sp) sum += part[((long long)sp * 32 + f) * N + i];
            w[i] = (member[i] >> f) & 1u ? sum : 0.0;
        }
    double d = 0.0;
    for (int i = threadIdx.x; i < N; i += 1024) d += v[i] * w[i];
    const double a = block_sum_1024(d, red);
    const double bprev = meta[2];
    double nn = 0.0;
    for (int i = threadIdx.x; i < N; i += 1024) {
        const double x = w[i] - a * v[i] - bprev * vp[i];
        w[i] = x;
        nn += x * x;
    }
    const double b = sqrt(block_sum_1024(nn, red));
    const bool stop = !(b > 1e-13 * (fabs(a) + bprev));
    const double inv = stop ? 0.0 : 1.0 / b;
    for (int i = threadIdx.x; i < N; i += 1024) {
        vp[i] = v[i];
        v[i] = w[i] * inv;
    }
    const int kk = step + 1;
    bool conv = false;
    if (tol > 0.0 && !stop && kk >= LZ_CHECK_FROM && kk % LZ_CHECK == 0 && kk <= LZ_CHECK_MAX) {     // (block-uniform)
        for (int i = threadIdx.x; i < kk; i += 1024) {
            s_al[i] = i == step ? a : al[i];
            s_be[i] = i == step ? b : be[i];
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const double theta = lz_theta(s_al, s_be, kk, 8, threadIdx.x);
            if (threadIdx.x == 0) {
                const double prev = meta[3];
                meta[3] = theta;
                s_conv = (prev > 0.0 && theta - prev <= tol * theta) ? 1 : 0;
            }
        }
        __syncthreads();
        conv = s_conv != 0;
    }
    if (threadIdx.x == 0) {
        al[step] = a;
        be[step] = b;
        meta[0] = (double)(step + 1);
        meta[2] = b;
        if (stop) meta[1] = 1.0;
        else if (conv) meta[1] = 2.0;
    }
}

// Largest eigenvalue of the k x k tridiagonal (a, b): 64-way multisection on the Sturm count, one
// wave per fold.  count(x) = number of eigenvalues < x; we shrink [lo, hi] around the smallest x
// with count(x) == k.
__global__ void __launch_bounds__(64) k_lz_eig(int N, int steps, double* work, double* __restrict__ lmax) {
    const int f = blockIdx.x, lane = threadIdx.x;
    double* base = lz_base(work, f, N, steps);
    const double* al = base + 3ll * N;
    const double* be = al + steps;
    const int k = (int)(be + steps)[0];
    if (k <= 0) { if (lane == 0) lmax[f] = 0.0; return; }
    const double hi = lz_theta(al, be, k, 14, lane);
    if (lane == 0) lmax[f] = hi;
}

// ---- all systems of a fit at once: K[I_f, I_f] for up to 32 row sets I_f of ONE Gram matrix -----------------
// Every fold's matrix is a principal submatrix of K, so with the Lanczos vectors kept at full length T (zero
// outside I_f) one pass over K serves all systems:  w_f = mask_f .* (K v_f).  member[i] bit f = (i in I_f).
// The per-fold gather version above reads K once per system and iteration (5 inner + 1 outer train set per
// outer fold: ~4x the bytes of K per outer fold); this one reads K once per iteration for the whole fit.
__global__ void __launch_bounds__(256) k_lz_init_masked(const unsigned* __restrict__ member, int T, int steps,
                                                        double* work) {
    __shared__ double red[256];
    const int f = blockIdx.x;
    double* base = lz_base(work, f, T, steps);
    double* v = base;
    double* vp = base + T;
    double ss = 0.0;
    for (int i = threadIdx.x; i < T; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const double x = (member[i] >> f) & 1u ? ((double)(h & 0xFFFFFF) / 8388608.0 - 1.0) + 1.5 : 0.0;
        v[i] = x;
        vp[i] = 0.0;
        ss += x * x;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const double inv = red[0] > 0.0 ? 1.0 / sqrt(red[0]) : 0.0;
    for (int i = threadIdx.x; i < T; i += 256) v[i] *= inv;
    if (threadIdx.x < 8) base[3ll * T + 2ll * steps + threadIdx.x] = 0.0;
}

constexpr int LZM_ROWS = 8;      // rows of K per block
constexpr int LZM_JT = 128;      // columns of K per LDS tile
constexpr int LZM_SPLIT = 4;     // column splits of the matvec (work buffer: + LZM_SPLIT * 32 * T doubles)

// block: 8 rows of K x all F <= 32 systems.  thread (f = tid & 31, g = tid >> 5): partial sums over the g-th
// 16-column slice of every tile for the 8 rows; K and V tiles go through LDS (V transposed to [j][f]), the next
// tile is fetched into registers while the current one is consumed; fixed-order reduction over g at the end.
__global__ void __launch_bounds__(256) k_lz_symv_multi(const double* __restrict__ Kmat, long long ldk,
                                                       const unsigned* __restrict__ member, int T, int F, int steps,
                                                       double* work, double* __restrict__ part, int jspan) {
    __shared__ double Ks[LZM_ROWS][LZM_JT];
    __shared__ double Vs[LZM_JT][33];
    const int tid = threadIdx.x, f = tid & 31, g = tid >> 5;
    const int i0 = blockIdx.x * LZM_ROWS;
    if (__syncthreads_count(tid < F && lz_base(work, tid, T, steps)[3ll * T + 2ll * steps + 1] == 0.0) == 0) return;  // all stopped
    double kreg[4], vreg[16];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + 256 * k, r = e >> 7, j = j0 + (e & 127);
            kreg[k] = (i0 + r < T && j < T) ? Kmat[(long long)(i0 + r) * ldk + j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = tid + 256 * k, ff = e >> 7, j = j0 + (e & 127);
            vreg[k] = (ff < F && j < T) ? lz_base(work, ff, T, steps)[j] : 0.0;
        }
    };
    double acc[LZM_ROWS];
#pragma unroll
    for (int r = 0; r < LZM_ROWS; ++r) acc[r] = 0.0;
    // blockIdx.y = column split: this block sums over columns [jb, je) only; the splits are added in k_lz_step (each block
    // is a chain of load latencies, one per 128-column tile: 4 splits = a quarter of the chain and 4x the blocks in flight)
    const int jb = blockIdx.y * jspan, je = min(T, jb + jspan);
    fetch(jb);
    for (int j0 = jb; j0 < je; j0 += LZM_JT) {
        __syncthreads();                                   // previous tile consumed
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int e = tid + 256 * k; Ks[e >> 7][e & 127] = kreg[k]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int e = tid + 256 * k; Vs[e & 127][e >> 7] = vreg[k]; }
        __syncthreads();
        if (j0 + LZM_JT < je) fetch(j0 + LZM_JT);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const double vj = Vs[g * 16 + jj][f];
#pragma unroll
            for (int r = 0; r < LZM_ROWS; ++r) acc[r] += Ks[r][g * 16 + jj] * vj;
        }
    }
    __syncthreads();
    double* red = &Vs[0][0];                               // [g][r][f]: 8 * 8 * 32 doubles
#pragma unroll
    for (int r = 0; r < LZM_ROWS; ++r) red[(g * LZM_ROWS + r) * 32 + f] = acc[r];
    __syncthreads();
    const int r = g, i = i0 + r;                           // thread (r, f) finishes one output
    if (i < T && f < F) {
        double sum = 0.0;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) sum += red[(gg * LZM_ROWS + r) * 32 + f];
        part[((long long)blockIdx.y * 32 + f) * T + i] = sum;
    }
}

// The same product on the fp64 MFMA (v_mfma_f64_16x16x4): as a vector-ALU kernel the matvec above is bound by its LDS
// reads -- one broadcast read of K per FMA, 2.6 GB of LDS traffic per iteration, 48 us -- while K itself is 72 MB.
// Here a block holds 64 rows (a wave 16) and a 32-column tile of K plus the 32 systems' vector slices in LDS, a lane
// reads ONE K element and two V elements per pair of MFMAs (16 rows x 4 k x 32 systems), and 16 column splits put
// ~750 blocks in flight, so that the iteration is bound by streaming K.  Output: the same column-split partial sums.
constexpr int LZQ_ROWS = 64, LZQ_JT = 32, LZQ_SPLIT = 8, LZQ_LD = LZQ_JT + 2;
typedef double lzq_f64x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_lz_symv_mfma(const double* __restrict__ Kmat, long long ldk, int T, int F,
                                                      int steps, double* work, double* __restrict__ part, int jspan) {
    __shared__ double Ks[LZQ_ROWS][LZQ_LD];            // [row][k]
    __shared__ double Vs[32][LZQ_LD];                  // [system][k]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int i0 = blockIdx.x * LZQ_ROWS;
    const int jb = blockIdx.y * jspan, je = min(T, jb + jspan);
    if (jb >= je) return;
    if (__syncthreads_count(tid < F && lz_base(work, tid, T, steps)[3ll * T + 2ll * steps + 1] == 0.0) == 0) return;  // all stopped
    double kreg[8], vreg[4];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q, r = e >> 5, j = j0 + (e & 31);
            kreg[q] = (i0 + r < T && j < je) ? Kmat[(long long)(i0 + r) * ldk + j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, n = e >> 5, j = j0 + (e & 31);
            vreg[q] = (n < F && j < je) ? lz_base(work, n, T, steps)[j] : 0.0;
        }
    };
    lzq_f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    fetch(jb);
    for (int j0 = jb; j0 < je; j0 += LZQ_JT) {
        __syncthreads();                               // previous tile consumed
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = tid + 256 * q; Ks[e >> 5][e & 31] = kreg[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = tid + 256 * q; Vs[e >> 5][e & 31] = vreg[q]; }
        __syncthreads();
        if (j0 + LZQ_JT < je) fetch(j0 + LZQ_JT);
#pragma unroll
        for (int k4 = 0; k4 < LZQ_JT / 4; ++k4) {
            const double a = Ks[w * 16 + li][4 * k4 + lq];
            const double b0 = Vs[li][4 * k4 + lq], b1 = Vs[16 + li][4 * k4 + lq];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
        }
    }
    // accumulator r of a lane: row lq + 4 r of the wave's 16, system li (acc0) / 16 + li (acc1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + w * 16 + lq + 4 * r;
        if (i < T) {
            if (li < F) part[((long long)blockIdx.y * 32 + li) * T + i] = acc0[r];
            if (16 + li < F) part[((long long)blockIdx.y * 32 + 16 + li) * T + i] = acc1[r];
        }
    }
}

__global__ void k_penalties(const double* __restrict__ lmax, int F, const double* __restrict__ alphas, int A,
                            int normalpha, double* __restrict__ a2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * A) return;
    const int f = i / A, a = i - f * A;
    const double s0 = normalpha ? sqrt(lmax[f]) : 1.0;
    const double na = alphas[a] * s0;
    a2[i] = na * na;
}

// ------------------------------------------------------------------ batch assembly
__global__ void __launch_bounds__(256) k_assemble(const double* __restrict__ Kmat, long long ldk,
                                                  const int* __restrict__ tr, const int* __restrict__ va,
                                                  const double* __restrict__ rhs, const double* __restrict__ a2,
                                                  const int* __restrict__ sys, long long k_fold_stride, int A,
                                                  int N, int M, double* __restrict__ aug) {
    const int i = blockIdx.x;             // row of the (N+M) x N system
    const int b = blockIdx.y;             // system of the batch being built ...
    const int sg = sys ? sys[b] : b;      // ... which is system sg = f * A + a of the full (fold, alpha) grid
    const int f = sg / A;
    Kmat += (long long)f * k_fold_stride; // 0: one matrix for all folds (dual form); else fold f's own (primal form)
    const int* trf = tr + (long long)f * N;
    double* dst = aug + ((long long)b * (N + M) + i) * N;
    if (i < N) {
        const int r = trf[i];
        const double diag = r >= 0 ? a2[sg] : 1.0;
        for (int j = threadIdx.x; j < N; j += 256) {
            const int c = trf[j];
            double v = (r >= 0 && c >= 0) ? Kmat[(long long)r * ldk + c] : 0.0;
            if (j == i) v = (r >= 0 ? v : 0.0) + diag;
            dst[j] = v;
        }
    } else if (rhs) {
        const double* src = rhs + ((long long)f * M + (i - N)) * N;
        for (int j = threadIdx.x; j < N; j += 256) dst[j] = src[j];
    } else {
        const int r = va[(long long)f * M + (i - N)];
        for (int j = threadIdx.x; j < N; j += 256) {
            const int c = trf[j];
            dst[j] = (r >= 0 && c >= 0) ? Kmat[(long long)r * ldk + c] : 0.0;
        }
    }
}

// rhs[c][j] = X[tr[j]][c]  (p x N panel for the refit systems), 32x32 LDS transpose.
__global__ void __launch_bounds__(256) k_transpose_rows(const float* __restrict__ X, long long ldx,
                                                        const int* __restrict__ tr, int N, long long p,
                                                        double* __restrict__ out) {
    __shared__ float t[32][33];
    const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int jj = ty; jj < 32; jj += 8) {
        const int j = j0 + jj;
        const int r = j < N ? tr[j] : -1;
        const long long c = c0 + tx;
        t[jj][tx] = (r >= 0 && c < p) ? X[(long long)r * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
        const long long c = c0 + cc;
        const int j = j0 + tx;
        if (c < p && j < N) out[c * N + j] = (double)t[tx][cc];
    }
}

// ------------------------------------------------------------------ polynomial-series hat matrices
// For a^2 >> lambda_max(K):  G (K + a^2 I)^-1 = sum_j (-1)^j G K^j / a^(2j+2), truncated after J+1 terms
// (relative error (lambda_max/a^2)^(J+1)).  The powers P_j = G K^j are shared by every such alpha of a
// fold, so a handful of fp64 GEMMs replaces one Cholesky factorisation + two triangular solves per alpha.

// out[f][i][j] = K[rows[f][i], cols[f][j]]   (-1 -> 0)
__global__ void __launch_bounds__(256) k_gather_sub(const double* __restrict__ Kmat, long long ldk,
                                                    const int* __restrict__ rows, const int* __restrict__ cols, int R,
                                                    int C, double* __restrict__ out) {
    const int i = blockIdx.x, f = blockIdx.y;
    const int r = rows[(long long)f * R + i];
    const int* cf = cols + (long long)f * C;
    double* dst = out + ((long long)f * R + i) * C;
    for (int j = threadIdx.x; j < C; j += 256) {
        const int c = cf[j];
        dst[j] = (r >= 0 && c >= 0) ? Kmat[(long long)r * ldk + c] : 0.0;
    }
}

// out[f][i][j] = K[rows_f[i]][cols_f[j]] / scale[f] as f32 (zeros where an index is -1)
__global__ void __launch_bounds__(256) k_gather_sub_f32(const double* __restrict__ Kmat, long long ldk,
                                                        const int* __restrict__ rows, const int* __restrict__ cols, int R,
                                                        int C, const double* __restrict__ scale, float* __restrict__ out) {
    const int i = blockIdx.x, f = blockIdx.y;
    const int r = rows[(long long)f * R + i];
    const int* cf = cols + (long long)f * C;
    const double w = scale ? 1.0 / scale[f] : 1.0;
    float* dst = out + ((long long)f * R + i) * C;
    for (int j = threadIdx.x; j < C; j += 256) {
        const int c = cf[j];
        dst[j] = (r >= 0 && c >= 0) ? (float)(Kmat[(long long)r * ldk + c] * w) : 0.f;
    }
}

// C[f] (M x N) = A[f] (M x Kd) . B[f] (Kd x N), row-major, 64 x 64 output tiles, Kd % 64 == 0, N % 64 == 0.
__global__ void __launch_bounds__(256) k_gemm_f64_nn(const double* __restrict__ A, const double* __restrict__ B,
                                                     double* __restrict__ C, int M, int N, int Kd) {
    __shared__ double sA[32 * TS_LD], sB[32 * TS_LD];
    const int f = blockIdx.z;
    const int c0 = blockIdx.x * NB, r0 = blockIdx.y * NB;
    const double* a = A + (long long)f * M * Kd + (long long)r0 * Kd;
    const double* b = B + (long long)f * Kd * N + c0;
    const int rows = min(NB, M - r0);
    double acc[4][4] = {};
    for (int kb = 0; kb < Kd; kb += NB) tile_product<false>(acc, a + kb, Kd, rows, b + (long long)kb * N, N, NB, sA, sB);
    double* c = C + (long long)f * M * N + (long long)r0 * N + c0;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ty * 4 + i < rows)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[(long long)(ty * 4 + i) * N + tx * 4 + j] = acc[i][j];
}

// H[f*A + aidx[s]] = sum_j coef[s][j] P_j[f] / scale[f]^(j+1)   (Horner in P_j / scale^j)
__global__ void __launch_bounds__(256) k_series_hat(const double* __restrict__ P, long long p_stride, int terms,
                                                    const double* __restrict__ scale, const double* __restrict__ coef,
                                                    const int* __restrict__ aidx, int A, int M, int N,
                                                    float* __restrict__ h) {
    const int i = blockIdx.x, s = blockIdx.y, f = blockIdx.z;
    const double inv = 1.0 / scale[f];
    const double* c = coef + (long long)s * terms;
    const double* src = P + ((long long)f * M + i) * N;
    float* dst = h + (((long long)f * A + aidx[s]) * M + i) * N;
    for (int j = threadIdx.x; j < N; j += 256) {
        double acc = 0.0;
        for (int t = terms - 1; t >= 0; --t) acc = c[t] * src[(long long)t * p_stride + j] + inv * acc;
        dst[j] = (float)(inv * acc);
    }
}

// P'_j[f] = P_j[f] / scale[f]^(j+1) as f32, laid out (F, terms, M, N): the shared matrix powers themselves, for
// callers that contract them with the targets once and combine the per-alpha predictions afterwards.
__global__ void __launch_bounds__(256) k_series_terms(const double* __restrict__ P, long long p_stride, int terms,
                                                      const double* __restrict__ scale, int M, int N,
                                                      float* __restrict__ out, const int* __restrict__ rowmap,
                                                      int rows_p) {
    const int i = blockIdx.x, t = blockIdx.y, f = blockIdx.z;
    const double inv = 1.0 / scale[f];
    double w = inv;
    for (int k = 0; k < t; ++k) w *= inv;
    const double* src = P + (long long)t * p_stride + ((long long)f * M + i) * N;
    const int row = rowmap ? rowmap[t * M + i] : t * M + i;
    float* dst = out + ((long long)f * rows_p + row) * N;
    for (int j = threadIdx.x; j < N; j += 256) dst[j] = (float)(src[j] * w);
}


// ---- primal form (p << n): Gram matrices of the FEATURE axis, one per training set --------------------------
// G_s = X_s' X_s for row set s.  The sets are gathered and transposed once into a stack Xt (n_sets * p_pad rows,
// set s = rows [s p_pad, (s+1) p_pad), column j = row rows[s][j] of X, -1 -> zero column); every set's Gram matrix
// is then the diagonal block of Xt Xt', computed by the tile kernel below with one grid plane per set.
__global__ void __launch_bounds__(256) k_gather_transpose_f32(const float* __restrict__ X, long long ldx,
                                                              const int* __restrict__ rows, int N, int p, int p_pad,
                                                              float* __restrict__ out) {
    __shared__ float t[32][33];
    const int f = blockIdx.z;
    const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int* rw = rows + (long long)f * N;
    for (int jj = ty; jj < 32; jj += 8) {
        const int j = j0 + jj;
        const int r = j < N ? rw[j] : -1;
        const int c = c0 + tx;
        t[jj][tx] = (r >= 0 && c < p) ? X[(long long)r * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
        const int c = c0 + cc, j = j0 + tx;
        if (c < p_pad && j < N) out[((long long)f * p_pad + c) * N + j] = t[tx][cc];
    }
}

__global__ void __launch_bounds__(256) k_gram_blocks(const float* __restrict__ X, long long ldx, int rows_per, int depth,
                                                     double* __restrict__ G) {
    const int bi = blockIdx.y, bj = blockIdx.x, b = blockIdx.z;
    if (bj > bi) return;
    X += (long long)b * rows_per * ldx;
    G += (long long)b * rows_per * rows_per;
    __shared__ double sA[16 * TS_LD], sB[16 * TS_LD];
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
    const int ra = bi * 64 + lr, rb = bj * 64 + lr;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < depth; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + lk + e;
            sA[(lk + e) * TS_LD + lr] = (ra < rows_per && k < depth) ? (double)X[(long long)ra * ldx + k] : 0.0;
            sB[(lk + e) * TS_LD + lr] = (rb < rows_per && k < depth) ? (double)X[(long long)rb * ldx + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[k * TS_LD + ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = sB[k * TS_LD + tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], bb[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = bi * 64 + ty * 4 + i, c = bj * 64 + tx * 4 + j;
            if (r < rows_per && c < rows_per) {
                G[(long long)r * rows_per + c] = acc[i][j];
                if (bi != bj) G[(long long)c * rows_per + r] = acc[i][j];
            }
        }
}

// out[f][i][:] (N doubles) = row rows[f][i] of X (p columns, then zeros);  -1 -> zero row;  -(2 + c) -> unit row e_c
__global__ void __launch_bounds__(256) k_gather_rows_f64(const float* __restrict__ X, long long ldx,
                                                         const int* __restrict__ rows, int M, int p, int N,
                                                         double* __restrict__ out) {
    const int i = blockIdx.x, f = blockIdx.y;
    const int r = rows[(long long)f * M + i];
    double* dst = out + ((long long)f * M + i) * N;
    for (int c = threadIdx.x; c < N; c += 256) {
        double v = 0.0;
        if (r >= 0) v = c < p ? (double)X[(long long)r * ldx + c] : 0.0;
        else if (r <= -2) v = (c == -(r + 2)) ? 1.0 : 0.0;
        dst[c] = v;
    }
}

// ---- small data movers that keep the fit free of framework kernels ------------------------------------------
// out[i * s_r + f * s_f + j * s_c] = (float)(K[rows[f][i], cols[f][j]] / scale[f])  (generalised k_gather_sub_f32: the
// series chain wants its operand as (N, folds, M) so that all folds are column groups of one grouped GEMM)
__global__ void __launch_bounds__(256) k_gather_sub_f32_strided(const double* __restrict__ Kmat, long long ldk,
                                                                const int* __restrict__ rows, const int* __restrict__ cols,
                                                                int R, int C, const double* __restrict__ scale,
                                                                float* __restrict__ out, long long s_f, long long s_r,
                                                                long long s_c) {
    const int i = blockIdx.x, f = blockIdx.y;
    const int r = rows[(long long)f * R + i];
    const int* cf = cols + (long long)f * C;
    const double w = scale ? 1.0 / scale[f] : 1.0;
    float* dst = out + (long long)f * s_f + (long long)i * s_r;
    for (int j = threadIdx.x; j < C; j += 256) {
        const int c = cf[j];
        dst[(long long)j * s_c] = (r >= 0 && c >= 0) ? (float)(Kmat[(long long)r * ldk + c] * w) : 0.f;
    }
}

// P[f][rowmap[i]][n] = Q[n][f][i]   (Q: (N, F, ldq) f32 -- term j of the series chain, transposed; P: (F, rows_p, N)):
// 32 x 32 LDS transpose per (fold, tile)
__global__ void __launch_bounds__(256) k_series_place(const float* __restrict__ Q, int N, int F, int ldq, int M,
                                                      const int* __restrict__ rowmap, float* __restrict__ P, int rows_p) {
    __shared__ float t[32][33];
    const int f = blockIdx.z, n0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int nn = ty; nn < 32; nn += 8) {
        const int n = n0 + nn, i = i0 + tx;
        t[nn][tx] = (n < N && i < M) ? Q[((long long)n * F + f) * ldq + i] : 0.f;
    }
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {
        const int i = i0 + ii, n = n0 + tx;
        if (i < M && n < N) {
            const int row = rowmap[i];
            if (row >= 0) P[((long long)f * rows_p + row) * N + n] = t[tx][ii];
        }
    }
}

__global__ void __launch_bounds__(256) k_scale_cast(const double* __restrict__ src, const double* __restrict__ divisor,
                                                    float* __restrict__ dst, long long n) {
    const double w = 1.0 / divisor[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = (float)(src[i] * w);
}

__global__ void __launch_bounds__(256) k_combine_terms(const float* __restrict__ t0, const float* __restrict__ t1,
                                                       const float* __restrict__ t2, const float* __restrict__ t3,
                                                       float c0, float c1, float c2, float c3, int terms,
                                                       float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // the order and arithmetic of  M = c0 T0; M += c1 T1; ...  (fl32 product, fl32 sum: no fma contraction)
        float m = __fmul_rn(t0[i], c0);
        if (terms > 1) m = __fadd_rn(m, __fmul_rn(t1[i], c1));
        if (terms > 2) m = __fadd_rn(m, __fmul_rn(t2[i], c2));
        if (terms > 3) m = __fadd_rn(m, __fmul_rn(t3[i], c3));
        out[i] = m;
    }
}

__global__ void __launch_bounds__(256) k_combine_terms_f64(const double* __restrict__ t0, const double* __restrict__ t1,
                                                           const double* __restrict__ t2, const double* __restrict__ t3,
                                                           double c0, double c1, double c2, double c3, int terms,
                                                           double* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double m = __dmul_rn(t0[i], c0);
        if (terms > 1) m = __dadd_rn(m, __dmul_rn(t1[i], c1));
        if (terms > 2) m = __dadd_rn(m, __dmul_rn(t2[i], c2));
        if (terms > 3) m = __dadd_rn(m, __dmul_rn(t3[i], c3));
        out[i] = m;
    }
}
}  // namespace

extern "C" int lc_combine_terms_f64(const double* const* h_terms, const double* h_coef, int terms, double* d_out,
                                    int64_t n, lc_stream_t stream) {
    LC_REQUIRE(h_terms && h_coef && d_out && terms >= 1 && terms <= 4 && n >= 0, LC_E_BADARG,
               "lc_combine_terms_f64: need 1..4 terms");
    for (int j = 0; j < terms; ++j) LC_REQUIRE(h_terms[j], LC_E_BADARG, "lc_combine_terms_f64: null term");
    if (n == 0) return LC_OK;
    const double* t[4] = {h_terms[0], terms > 1 ? h_terms[1] : nullptr, terms > 2 ? h_terms[2] : nullptr,
                          terms > 3 ? h_terms[3] : nullptr};
    double c[4] = {h_coef[0], terms > 1 ? h_coef[1] : 0.0, terms > 2 ? h_coef[2] : 0.0, terms > 3 ? h_coef[3] : 0.0};
    hipLaunchKernelGGL(k_combine_terms_f64, dim3((unsigned)lc::imin(lc::ceil_div<long long>(n, 256), 65535)), dim3(256), 0,
                       lc::as_stream(stream), t[0], t[1], t[2], t[3], c[0], c[1], c[2], c[3], terms, d_out, (long long)n);
    return lc::launched("k_combine_terms_f64");
}

extern "C" int lc_gram_f64(const float* d_x, int64_t ldx, int64_t T, int64_t p, double* d_k, int64_t ldk,
                           lc_stream_t stream) {
    LC_REQUIRE(d_x && d_k, LC_E_BADARG, "lc_gram_f64: null pointer");
    LC_REQUIRE(T > 0 && p > 0 && ldx >= p && ldk >= T && T < (1 << 30), LC_E_SHAPE, "lc_gram_f64: bad shape");
    const unsigned nt = (unsigned)lc::ceil_div<long long>(T, 64);
    lc::ScopedTimer timer_(lc::T_GRAM, lc::as_stream(stream));
    hipLaunchKernelGGL(k_gram, dim3(nt, nt), dim3(256), 0, lc::as_stream(stream), d_x, (long long)ldx, (int)T, (int)p,
                       d_k, (long long)ldk);
    return lc::launched("k_gram");
}


extern "C" int lc_gather_transpose_f32(const float* d_x, int64_t ldx, const int32_t* d_rows, int F, int N, int p,
                                       int p_pad, float* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_rows && d_out, LC_E_BADARG, "lc_gather_transpose_f32: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && N > 0 && p > 0 && p_pad >= p && ldx >= p, LC_E_SHAPE, "lc_gather_transpose_f32: bad shape");
    dim3 grid((unsigned)lc::ceil_div(N, 32), (unsigned)lc::ceil_div(p_pad, 32), (unsigned)F);
    hipLaunchKernelGGL(k_gather_transpose_f32, grid, dim3(256), 0, lc::as_stream(stream), d_x, (long long)ldx, d_rows, N, p,
                       p_pad, d_out);
    return lc::launched("k_gather_transpose_f32");
}

extern "C" int lc_gram_blocks_f64(const float* d_xt, int64_t ldx, int n_blocks, int rows_per, int depth, double* d_g,
                                  lc_stream_t stream) {
    LC_REQUIRE(d_xt && d_g, LC_E_BADARG, "lc_gram_blocks_f64: null pointer");
    LC_REQUIRE(n_blocks > 0 && n_blocks <= 65535 && rows_per > 0 && depth > 0 && ldx >= depth, LC_E_SHAPE,
               "lc_gram_blocks_f64: bad shape");
    const unsigned nt = (unsigned)lc::ceil_div(rows_per, 64);
    lc::ScopedTimer timer_(lc::T_GRAM, lc::as_stream(stream));
    hipLaunchKernelGGL(k_gram_blocks, dim3(nt, nt, (unsigned)n_blocks), dim3(256), 0, lc::as_stream(stream), d_xt,
                       (long long)ldx, rows_per, depth, d_g);
    return lc::launched("k_gram_blocks");
}

extern "C" int lc_gather_rows_f64(const float* d_x, int64_t ldx, const int32_t* d_rows, int F, int M, int p, int N,
                                  double* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_rows && d_out, LC_E_BADARG, "lc_gather_rows_f64: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && M > 0 && p > 0 && N >= p && ldx >= p, LC_E_SHAPE, "lc_gather_rows_f64: bad shape");
    hipLaunchKernelGGL(k_gather_rows_f64, dim3((unsigned)M, (unsigned)F), dim3(256), 0, lc::as_stream(stream), d_x,
                       (long long)ldx, d_rows, M, p, N, d_out);
    return lc::launched("k_gather_rows_f64");
}

extern "C" int lc_lambda_max(const double* d_k, int64_t ldk, int64_t k_stride, const int32_t* d_rows, int F,
                             int N, int steps, double* d_work, double* d_lmax, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_rows && d_work && d_lmax, LC_E_BADARG, "lc_lambda_max: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && N > 0 && steps > 0 && k_stride >= 0, LC_E_SHAPE, "lc_lambda_max: bad shape");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LAMBDA_MAX, s);
    hipLaunchKernelGGL(k_lz_init, dim3(F), dim3(256), 0, s, d_rows, N, steps, d_work);
    if (int rc = lc::launched("k_lz_init")) return rc;
    for (int it = 0; it < steps; ++it) {
        hipLaunchKernelGGL(k_lz_symv, dim3((unsigned)lc::ceil_div(N, 4), (unsigned)F), dim3(256), 0, s, d_k,
                           (long long)ldk, (long long)k_stride, d_rows, N, steps, d_work);
        hipLaunchKernelGGL(k_lz_step, dim3(F), dim3(1024), 0, s, N, steps, it, d_work);
    }
    if (int rc = lc::launched("k_lz_step")) return rc;
    hipLaunchKernelGGL(k_lz_eig, dim3(F), dim3(64), 0, s, N, steps, d_work, d_lmax);
    return lc::launched("k_lz_eig");
}

extern "C" int lc_lambda_max_dense(const double* d_k, int64_t ldk, int64_t k_stride, int F, int N, int n, int steps,
                                   double tol, double* d_work, double* d_lmax, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_work && d_lmax, LC_E_BADARG, "lc_lambda_max_dense: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && n > 0 && N >= n && N % 2 == 0 && steps > 0 && k_stride >= 0 && ldk >= n && ldk % 2 == 0 &&
                   k_stride % 2 == 0 && tol >= 0.0, LC_E_SHAPE,
               "lc_lambda_max_dense: need N >= n, even N / ldk / k_stride, tol >= 0");
    LC_REQUIRE((reinterpret_cast<uintptr_t>(d_k) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0, LC_E_BADARG,
               "lc_lambda_max_dense: matrices and work area must be 16-byte aligned");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LAMBDA_MAX, s);
    hipLaunchKernelGGL(k_lz_init, dim3(F), dim3(256), 0, s, (const int*)nullptr, N, steps, d_work, n);
    if (int rc = lc::launched("k_lz_init")) return rc;
    for (int it = 0; it < steps; ++it) {
        hipLaunchKernelGGL(k_lz_symv_dense, dim3((unsigned)lc::ceil_div(N, LZD_ROWS), (unsigned)F), dim3(256), 0, s, d_k,
                           (long long)ldk, (long long)k_stride, N, n, steps, d_work);
        hipLaunchKernelGGL(k_lz_step, dim3(F), dim3(1024), 0, s, N, steps, it, d_work, (const double*)nullptr, 0,
                           (const unsigned*)nullptr, tol);
    }
    if (int rc = lc::launched("k_lz_step")) return rc;
    hipLaunchKernelGGL(k_lz_eig, dim3(F), dim3(64), 0, s, N, steps, d_work, d_lmax);
    return lc::launched("k_lz_eig");
}

// use_mfma: the masked multi-system matvec on the fp64 MFMA (k_lz_symv_mfma, the default) or on the vector ALU
// (k_lz_symv_multi) -- a per-call choice, no process-wide switch

extern "C" int lc_lambda_max_masked(const double* d_k, int64_t ldk, int T, const uint32_t* d_member, int F, int steps,
                                    double tol, double* d_work, double* d_lmax, int use_mfma, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_member && d_work && d_lmax, LC_E_BADARG, "lc_lambda_max_masked: null pointer");
    LC_REQUIRE(F > 0 && F <= 32 && T > 0 && steps > 0 && ldk >= T && tol >= 0.0, LC_E_SHAPE,
               "lc_lambda_max_masked: need 1 <= F <= 32 systems, T > 0, steps > 0, tol >= 0");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LAMBDA_MAX, s);
    hipLaunchKernelGGL(k_lz_init_masked, dim3(F), dim3(256), 0, s, d_member, T, steps, d_work);
    if (int rc = lc::launched("k_lz_init_masked")) return rc;
    double* part = d_work + (long long)F * (3ll * T + 2ll * steps + 8);      // (LZQ_SPLIT, 32, T) partial matvecs
    const bool mfma = use_mfma != 0;
    const int jspan = mfma ? lc::ceil_div(lc::ceil_div(T, LZQ_SPLIT), LZQ_JT) * LZQ_JT
                           : lc::ceil_div(lc::ceil_div(T, LZM_SPLIT), LZM_JT) * LZM_JT;
    const int nsplit = lc::ceil_div(T, jspan);
    for (int it = 0; it < steps; ++it) {
        if (mfma)
            hipLaunchKernelGGL(k_lz_symv_mfma, dim3((unsigned)lc::ceil_div(T, LZQ_ROWS), (unsigned)nsplit), dim3(256), 0, s, d_k,
                               (long long)ldk, T, F, steps, d_work, part, jspan);
        else
            hipLaunchKernelGGL(k_lz_symv_multi, dim3((unsigned)lc::ceil_div(T, LZM_ROWS), (unsigned)nsplit), dim3(256), 0, s, d_k,
                               (long long)ldk, d_member, T, F, steps, d_work, part, jspan);
        hipLaunchKernelGGL(k_lz_step, dim3(F), dim3(1024), 0, s, T, steps, it, d_work, (const double*)part, nsplit,
                           (const unsigned*)d_member, tol);
    }
    if (int rc = lc::launched("k_lz_step")) return rc;
    hipLaunchKernelGGL(k_lz_eig, dim3(F), dim3(64), 0, s, T, steps, d_work, d_lmax);
    return lc::launched("k_lz_eig");
}

extern "C" int lc_penalties(const double* d_lmax, int F, const double* d_alphas, int A, int normalpha, double* d_a2,
                            lc_stream_t stream) {
    LC_REQUIRE(d_alphas && d_a2 && (d_lmax || !normalpha), LC_E_BADARG, "lc_penalties: null pointer");
    LC_REQUIRE(F > 0 && A > 0, LC_E_SHAPE, "lc_penalties: bad shape");
    hipLaunchKernelGGL(k_penalties, dim3((unsigned)lc::ceil_div(F * A, 256)), dim3(256), 0, lc::as_stream(stream),
                       d_lmax, F, d_alphas, A, normalpha, d_a2);
    return lc::launched("k_penalties");
}

extern "C" int lc_batch_assemble(const double* d_k, int64_t ldk, const int32_t* d_tr, const int32_t* d_va,
                                 const double* d_rhs, const double* d_a2, int F, int A, int N, int M, double* d_aug,
                                 lc_stream_t stream) {
    LC_REQUIRE(d_k && d_tr && d_a2 && d_aug && (d_va || d_rhs), LC_E_BADARG, "lc_batch_assemble: null pointer");
    LC_REQUIRE(F > 0 && A > 0 && N > 0 && N % LC_NB == 0 && M > 0 && M % LC_MB == 0 && F * A <= 65535, LC_E_SHAPE,
               "lc_batch_assemble: need N %% %d == 0, M %% %d == 0, F*A <= 65535", LC_NB, LC_MB);
    lc::ScopedTimer timer_(lc::T_ASSEMBLE, lc::as_stream(stream));
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)(N + M), (unsigned)(F * A)), dim3(256), 0, lc::as_stream(stream), d_k,
                       (long long)ldk, d_tr, d_va, d_rhs, d_a2, (const int*)nullptr, 0ll, A, N, M, d_aug);
    return lc::launched("k_assemble");
}

extern "C" int lc_batch_assemble_sel(const double* d_k, int64_t ldk, int64_t k_fold_stride, const int32_t* d_tr,
                                     const int32_t* d_va, const double* d_rhs, const double* d_a2, const int32_t* d_sys,
                                     int B, int A, int N, int M, double* d_aug, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_tr && d_a2 && d_sys && d_aug && (d_va || d_rhs), LC_E_BADARG, "lc_batch_assemble_sel: null pointer");
    LC_REQUIRE(B > 0 && B <= 65535 && A > 0 && N > 0 && N % LC_NB == 0 && M > 0 && M % LC_MB == 0, LC_E_SHAPE,
               "lc_batch_assemble_sel: need N %% %d == 0, M %% %d == 0, B <= 65535", LC_NB, LC_MB);
    lc::ScopedTimer timer_(lc::T_ASSEMBLE, lc::as_stream(stream));
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)(N + M), (unsigned)B), dim3(256), 0, lc::as_stream(stream), d_k,
                       (long long)ldk, d_tr, d_va, d_rhs, d_a2, d_sys, (long long)k_fold_stride, A, N, M, d_aug);
    return lc::launched("k_assemble");
}

extern "C" int lc_transpose_rows_f64(const float* d_x, int64_t ldx, const int32_t* d_tr, int N, int64_t p,
                                     double* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_tr && d_out, LC_E_BADARG, "lc_transpose_rows_f64: null pointer");
    LC_REQUIRE(N > 0 && p > 0 && ldx >= p, LC_E_SHAPE, "lc_transpose_rows_f64: bad shape");
    dim3 grid((unsigned)lc::ceil_div(N, 32), (unsigned)lc::ceil_div<long long>(p, 32));
    hipLaunchKernelGGL(k_transpose_rows, grid, dim3(256), 0, lc::as_stream(stream), d_x, (long long)ldx, d_tr, N,
                       (long long)p, d_out);
    return lc::launched("k_transpose_rows");
}


extern "C" int lc_fill_bytes(void* d_ptr, int byte, int64_t nbytes, lc_stream_t stream) {
    LC_REQUIRE(d_ptr || nbytes == 0, LC_E_BADARG, "lc_fill_bytes: null pointer");
    LC_REQUIRE(nbytes >= 0 && byte >= 0 && byte <= 255, LC_E_SHAPE, "lc_fill_bytes: bad argument");
    if (nbytes) LC_HIP(hipMemsetAsync(d_ptr, byte, (size_t)nbytes, lc::as_stream(stream)));
    return LC_OK;
}

extern "C" int lc_gather_sub_f32_strided(const double* d_k, int64_t ldk, const int32_t* d_rows, const int32_t* d_cols,
                                         int F, int R, int C, const double* d_scale, float* d_out, int64_t s_f,
                                         int64_t s_r, int64_t s_c, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_rows && d_cols && d_out, LC_E_BADARG, "lc_gather_sub_f32_strided: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && R > 0 && C > 0, LC_E_SHAPE, "lc_gather_sub_f32_strided: bad shape");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_SERIES, s);
    hipLaunchKernelGGL(k_gather_sub_f32_strided, dim3(R, F), dim3(256), 0, s, d_k, (long long)ldk, d_rows, d_cols, R, C,
                       d_scale, d_out, (long long)s_f, (long long)s_r, (long long)s_c);
    return lc::launched("k_gather_sub_f32_strided");
}

extern "C" int lc_series_place(const float* d_q, int N, int F, int ldq, int M, const int32_t* d_rowmap, float* d_p,
                               int rows_p, lc_stream_t stream) {
    LC_REQUIRE(d_q && d_rowmap && d_p, LC_E_BADARG, "lc_series_place: null pointer");
    LC_REQUIRE(N > 0 && F > 0 && F <= 65535 && M > 0 && ldq >= M && rows_p > 0, LC_E_SHAPE, "lc_series_place: bad shape");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_SERIES, s);
    hipLaunchKernelGGL(k_series_place, dim3((unsigned)lc::ceil_div(N, 32), (unsigned)lc::ceil_div(M, 32), (unsigned)F),
                       dim3(256), 0, s, d_q, N, F, ldq, M, d_rowmap, d_p, rows_p);
    return lc::launched("k_series_place");
}

extern "C" int lc_scale_cast_f64_f32(const double* d_src, const double* d_divisor, float* d_dst, int64_t n,
                                     lc_stream_t stream) {
    LC_REQUIRE(d_src && d_divisor && d_dst && n >= 0, LC_E_BADARG, "lc_scale_cast_f64_f32: bad argument");
    if (n == 0) return LC_OK;
    hipLaunchKernelGGL(k_scale_cast, dim3((unsigned)lc::imin(lc::ceil_div<long long>(n, 256), 65535)), dim3(256), 0,
                       lc::as_stream(stream), d_src, d_divisor, d_dst, (long long)n);
    return lc::launched("k_scale_cast");
}

extern "C" int lc_combine_terms_f32(const float* const* h_terms, const float* h_coef, int terms, float* d_out, int64_t n,
                                    lc_stream_t stream) {
    LC_REQUIRE(h_terms && h_coef && d_out && terms >= 1 && terms <= 4 && n >= 0, LC_E_BADARG,
               "lc_combine_terms_f32: need 1..4 terms");
    for (int j = 0; j < terms; ++j) LC_REQUIRE(h_terms[j], LC_E_BADARG, "lc_combine_terms_f32: null term");
    if (n == 0) return LC_OK;
    const float* t[4] = {h_terms[0], terms > 1 ? h_terms[1] : nullptr, terms > 2 ? h_terms[2] : nullptr,
                         terms > 3 ? h_terms[3] : nullptr};
    float c[4] = {h_coef[0], terms > 1 ? h_coef[1] : 0.f, terms > 2 ? h_coef[2] : 0.f, terms > 3 ? h_coef[3] : 0.f};
    hipLaunchKernelGGL(k_combine_terms, dim3((unsigned)lc::imin(lc::ceil_div<long long>(n, 256), 65535)), dim3(256), 0,
                       lc::as_stream(stream), t[0], t[1], t[2], t[3], c[0], c[1], c[2], c[3], terms, d_out, (long long)n);
    return lc::launched("k_combine_terms");
}

extern "C" int lc_gather_sub_f64(const double* d_k, int64_t ldk, const int32_t* d_rows, const int32_t* d_cols, int F,
                                 int R, int C, double* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_rows && d_cols && d_out, LC_E_BADARG, "lc_gather_sub_f64: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && R > 0 && C > 0, LC_E_SHAPE, "lc_gather_sub_f64: bad shape");
    hipLaunchKernelGGL(k_gather_sub, dim3(R, F), dim3(256), 0, lc::as_stream(stream), d_k, (long long)ldk, d_rows, d_cols, R, C,
                       d_out);
    return lc::launched("k_gather_sub");
}

extern "C" int lc_gather_sub_f32(const double* d_k, int64_t ldk, const int32_t* d_rows, const int32_t* d_cols, int F,
                                 int R, int C, const double* d_scale, float* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_rows && d_cols && d_out, LC_E_BADARG, "lc_gather_sub_f32: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && R > 0 && C > 0, LC_E_SHAPE, "lc_gather_sub_f32: bad shape");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_SERIES, s);
    hipLaunchKernelGGL(k_gather_sub_f32, dim3(R, F), dim3(256), 0, s, d_k, (long long)ldk, d_rows, d_cols, R, C, d_scale,
                       d_out);
    return lc::launched("k_gather_sub_f32");
}

extern "C" int lc_batch_series_terms(const double* d_k, int64_t ldk, const int32_t* d_tr, const int32_t* d_va, int F,
                                     int N, int M, const double* d_scale, int terms, double* d_work, float* d_p,
                                     const int32_t* d_rowmap, int rows_p, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_tr && d_va && d_scale && d_work && d_p, LC_E_BADARG, "lc_batch_series_terms: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && N > 0 && N % LC_NB == 0 && M > 0 && M % LC_MB == 0 && terms >= 1 && terms <= 8,
               LC_E_SHAPE, "lc_batch_series_terms: need N %% %d == 0, M %% %d == 0, 1 <= terms <= 8", LC_NB, LC_MB);
    LC_REQUIRE(rows_p >= terms * M, LC_E_SHAPE, "lc_batch_series_terms: rows_p < terms * M");
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_SERIES, s);
    double* Kf = d_work;                                   // (F, N, N), then P_0 .. P_{terms-1}, each (F, M, N)
    double* P = d_work + (long long)F * N * N;
    const long long p_stride = (long long)F * M * N;
    hipLaunchKernelGGL(k_gather_sub, dim3(N, F), dim3(256), 0, s, d_k, (long long)ldk, d_tr, d_tr, N, N, Kf);
    hipLaunchKernelGGL(k_gather_sub, dim3(M, F), dim3(256), 0, s, d_k, (long long)ldk, d_va, d_tr, M, N, P);
    for (int t = 1; t < terms; ++t)
        hipLaunchKernelGGL(k_gemm_f64_nn, dim3(N / NB, lc::ceil_div(M, NB), F), dim3(256), 0, s, P + (t - 1) * p_stride, Kf,
                           P + t * p_stride, M, N, N);
    hipLaunchKernelGGL(k_series_terms, dim3(M, terms, F), dim3(256), 0, s, P, p_stride, terms, d_scale, M, N, d_p,
                       d_rowmap, rows_p);
    return lc::launched("lc_batch_series_terms");
}

extern "C" int lc_batch_series_hat(const double* d_k, int64_t ldk, const int32_t* d_tr, const int32_t* d_va, int F, int N,
                                   int M, const double* d_scale, const double* d_coef, const int32_t* d_aidx, int S,
                                   int A, int terms, double* d_work, float* d_h, lc_stream_t stream) {
    LC_REQUIRE(d_k && d_tr && d_va && d_scale && d_coef && d_aidx && d_work && d_h, LC_E_BADARG,
               "lc_batch_series_hat: null pointer");
    LC_REQUIRE(F > 0 && F <= 65535 && S > 0 && S <= 65535 && N > 0 && N % LC_NB == 0 && M > 0 && M % LC_MB == 0 &&
                   terms >= 1 && terms <= 16,
               LC_E_SHAPE, "lc_batch_series_hat: need N %% %d == 0, M %% %d == 0, 1 <= terms <= 16", LC_NB, LC_MB);
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_SERIES, s);
    // work: Kf (F, N, N) then P_0 .. P_{terms-1}, each (F, M, N)
    double* Kf = d_work;
    double* P = d_work + (long long)F * N * N;
    const long long p_stride = (long long)F * M * N;
    hipLaunchKernelGGL(k_gather_sub, dim3(N, F), dim3(256), 0, s, d_k, (long long)ldk, d_tr, d_tr, N, N, Kf);
    hipLaunchKernelGGL(k_gather_sub, dim3(M, F), dim3(256), 0, s, d_k, (long long)ldk, d_va, d_tr, M, N, P);
    for (int t = 1; t < terms; ++t)
        hipLaunchKernelGGL(k_gemm_f64_nn, dim3(N / NB, lc::ceil_div(M, NB), F), dim3(256), 0, s, P + (t - 1) * p_stride, Kf,
                           P + t * p_stride, M, N, N);
    hipLaunchKernelGGL(k_series_hat, dim3(M, S, F), dim3(256), 0, s, P, p_stride, terms, d_scale, d_coef, d_aidx, A, M,
                       N, d_h);
    return lc::launched("lc_batch_series_hat");
}
