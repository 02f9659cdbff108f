// Batched Cholesky solve of the augmented ridge systems, fp64 (gfx950).
//
// One system is an (N + M) x N row-major block: rows 0..N-1 hold A = K[tr,tr] + a^2 I (lower triangle used), rows
// N..N+M-1 hold G = K[va,tr] (inner folds) or the stacked [X'; K[te,tr]] rows (refit).  The solve leaves
// H = G A^-1 in the bottom rows:  A = L L',  Z = G L^-T (the augmented rows ride along with the panel and trailing
// updates of the factorisation),  H = Z L^-1 (block back substitution).
//
// Two-level right-looking blocking.  Columns are factored in NB = 64 wide steps (diagonal tile in LDS by one
// workgroup per system, its inverse Linv kept, panel = P Linv'), but the rank-64 updates of a step only touch the
// remaining columns of the current OUTER block (512 columns, lc_chol_outer_block); everything to the right is
// updated once per outer block with a product of that depth.  At NB = 64 depth the trailing update re-reads and
// re-writes the whole trailing matrix 30 times for N = 1920 -- 10 GB per batch of 20 systems, the largest single
// cost of the first version -- at depth 512 an eighth of that.  The back substitution is blocked the same way.
//
// Every GEMM-shaped piece is a tile product on LDS-staged operands: the short step kernels on
// v_mfma_f64_16x16x4_f64 (k_mm64<2>), the deep updates on v_mfma_f64_4x4x4_4b_f64 (k_mm64q, below), with a
// v_fma_f64 version (k_mm64v) kept for comparison.  What the three pipes deliver on this part was measured first
// (tools/mfma_f64_rate.hip); the datasheet's "matrix fp64 = vector fp64 = 78.6 TF" holds only for the 4x4x4 form.
#include "lc_common.h"
#include <type_traits>

namespace {

constexpr int NB = LC_NB;   // 64

// Diagonal block k: L_kk = chol(A_kk) in LDS, four columns per barrier pair, then Linv = inv(L_kk) by forward
// substitution with four lanes per column (details at the two loops).  One workgroup per system.
constexpr int PD_LD = NB + 1;

// 1 / sqrt(d) for d > 0: hardware estimate + two Newton steps (full fp64 accuracy); sqrt(d) = d * rsqrt(d).
// The software sqrt and divide of the pivots were the longest dependent chain of the diagonal kernel.
__device__ inline double rsqrt_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = fma(r, fma(-h * r, r, 0.5), r);
    r = fma(r, fma(-h * r, r, 0.5), r);
    return r;
}

// The body, for the workgroup of system b (256 threads): L = NB x PD_LD doubles and rdiag = NB + 768 doubles of LDS.
// PRELOADED: the caller has put the tile into L already (a barrier follows here either way).
// (MODE is for tools/potrf_bench.hip only: 0 = everything, 1 = no Linv, 2 = load and store alone, 3 = Linv alone)
template <bool PRELOADED, int MODE = 0>
__device__ inline void potrf_diag_tile_v(double* __restrict__ aug, int N, int M, int k, int b, double* __restrict__ linv,
                                         int* __restrict__ info, double* L, double* rdiag) {
    const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
    double* a = aug + (long long)b * (N + M) * N + (long long)k * NB * N + k * NB;
    if (!PRELOADED)
        for (int e = t; e < NB * NB; e += 256) L[(e >> 6) * PD_LD + (e & 63)] = a[(long long)(e >> 6) * N + (e & 63)];
    __syncthreads();
    // Four columns per step (16 steps, two barriers each): every thread factors the 4 x 4 pivot block redundantly
    // in registers, transforms the panel rows it needs on the fly (l = a G^-T, six FMAs) and applies the rank-4
    // update; the scaled panel is written after the second barrier.  Latency, not flops, is what this kernel
    // costs (one workgroup per system), and the barrier count is its latency.
    for (int j = 0; j < (MODE >= 2 ? 0 : NB); j += 4) {
        __syncthreads();                              // columns j..j+3 are final up to their scaling
        double g00 = L[j * PD_LD + j];
        const double a10 = L[(j + 1) * PD_LD + j], a11 = L[(j + 1) * PD_LD + j + 1];
        const double a20 = L[(j + 2) * PD_LD + j], a21 = L[(j + 2) * PD_LD + j + 1], a22 = L[(j + 2) * PD_LD + j + 2];
        const double a30 = L[(j + 3) * PD_LD + j], a31 = L[(j + 3) * PD_LD + j + 1], a32 = L[(j + 3) * PD_LD + j + 2],
                     a33 = L[(j + 3) * PD_LD + j + 3];
        int bad = 0;
        if (!(g00 > 0.0)) bad = 1;
        const double r0 = rsqrt_nr(g00);
        g00 *= r0;
        const double g10 = a10 * r0, g20 = a20 * r0, g30 = a30 * r0;
        double g11 = a11 - g10 * g10;
        if (!bad && !(g11 > 0.0)) bad = 2;
        const double r1 = rsqrt_nr(g11);
        g11 *= r1;
        const double g21 = (a21 - g20 * g10) * r1, g31 = (a31 - g30 * g10) * r1;
        double g22 = a22 - g20 * g20 - g21 * g21;
        if (!bad && !(g22 > 0.0)) bad = 3;
        const double r2 = rsqrt_nr(g22);
        g22 *= r2;
        const double g32 = (a32 - g30 * g20 - g31 * g21) * r2;
        double g33 = a33 - g30 * g30 - g31 * g31 - g32 * g32;
        if (!bad && !(g33 > 0.0)) bad = 4;
        const double r3 = rsqrt_nr(g33);
        g33 *= r3;
        if (t == 0 && bad && info[b] == 0) info[b] = k * NB + j + bad;
        // l = a G^-T for a panel row a = (x0..x3)
#define LC_ROW_TRANSFORM(x0, x1, x2, x3, l0, l1, l2, l3)                 \
        const double l0 = (x0) * r0;                                      \
        const double l1 = ((x1) - l0 * g10) * r1;                         \
        const double l2 = ((x2) - l0 * g20 - l1 * g21) * r2;              \
        const double l3 = ((x3) - l0 * g30 - l1 * g31 - l2 * g32) * r3;
        // the panel rows are transformed ONCE (one thread per row, in place), then the rank-4 update reads them: the
        // first version transformed the rows it needed on the fly inside the update -- 16 x the arithmetic of this
        // phase for one barrier's worth of latency that the pivot block above hides anyway
        if (j + 4 + t < NB) {
            double* ai = L + (j + 4 + t) * PD_LD + j;
            LC_ROW_TRANSFORM(ai[0], ai[1], ai[2], ai[3], l0, l1, l2, l3)
            ai[0] = l0; ai[1] = l1; ai[2] = l2; ai[3] = l3;
        }
        __syncthreads();                              // the scaled panel is complete
        for (int i = j + 4 + ti; i < NB; i += 16) {
            const double* ai = L + i * PD_LD + j;
            const double li0 = ai[0], li1 = ai[1], li2 = ai[2], li3 = ai[3];
            for (int c = j + 4 + tj; c <= i; c += 16) {
                const double* ac = L + c * PD_LD + j;
                L[i * PD_LD + c] -= li0 * ac[0] + li1 * ac[1] + li2 * ac[2] + li3 * ac[3];
            }
        }
#undef LC_ROW_TRANSFORM
        if (t == 255) {                               // the pivot block itself
            L[j * PD_LD + j] = g00;
            L[(j + 1) * PD_LD + j] = g10; L[(j + 1) * PD_LD + j + 1] = g11;
            L[(j + 2) * PD_LD + j] = g20; L[(j + 2) * PD_LD + j + 1] = g21; L[(j + 2) * PD_LD + j + 2] = g22;
            L[(j + 3) * PD_LD + j] = g30; L[(j + 3) * PD_LD + j + 1] = g31; L[(j + 3) * PD_LD + j + 2] = g32;
            L[(j + 3) * PD_LD + j + 3] = g33;
            rdiag[j] = r0; rdiag[j + 1] = r1; rdiag[j + 2] = r2; rdiag[j + 3] = r3;
        }
    }
    __syncthreads();
    for (int e = t; e < NB * NB; e += 256) {
        const int i = e >> 6, j = e & 63;
        a[(long long)i * N + j] = j <= i ? L[i * PD_LD + j] : 0.0;
    }
    // Linv = inv(L), blocked 4 x 4 in 16 x 16 blocks, in place over L in LDS (round 3; one forward substitution over all
    // 64 rows -- 64 dependent steps of ~0.26 us -- was 17 of the kernel's 41 us):
    //   (a) the four diagonal blocks at once: column c by forward substitution INSIDE its block (16 dependent steps), four
    //       lanes per column -- lane part r keeps the entries x_q with q = r (mod 4), two shuffles combine their shares;
    //   (b) block row bi = 1, 2, 3:  X[bi][j] = -X[bi][bi] (sum_{kb = j .. bi-1} L[bi][kb] X[kb][j]),  one element per
    //       thread and block, the inner sums through a 3 x 256 scratch (rdiag + NB: the callers provide NB + 768 doubles).
    // The rows of L above block row bi already hold X; L itself went to global memory just above.
    if (MODE == 0 || MODE == 3) {
        if (MODE == 3 && t < NB) rdiag[t] = 1.0 / L[t * PD_LD + t];
        __syncthreads();                               // the store loop above has read L; rdiag is complete
        double* T = rdiag + NB;
        double* lo = linv + ((long long)b * (N / NB) + k) * NB * NB;
        {
            const int c = t >> 2, r = t & 3, d0 = (c >> 4) << 4;
            double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) {
                const int i = d0 + ii;
                double sp = 0.0;
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (4 * m < ii) {                        // static bound; the lane's own entry 4m + r may still be >= ii
                        if (4 * m + r < ii) sp = fma(L[i * PD_LD + d0 + 4 * m + r], x[m], sp);
                    }
                sp += __shfl_xor(sp, 1);
                sp += __shfl_xor(sp, 2);
                const double xi = ((i == c ? 1.0 : 0.0) - sp) * rdiag[i];
                if (r == (ii & 3)) x[ii >> 2] = xi;
            }
            __syncthreads();                           // every lane has read its diagonal block
#pragma unroll
            for (int m = 0; m < 4; ++m) L[(d0 + 4 * m + r) * PD_LD + c] = x[m];     // zeros above the diagonal included
        }
        __syncthreads();
        const int ra = t >> 4, cc = t & 15;
#pragma unroll
        for (int bi = 1; bi < 4; ++bi) {
            double tv[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < bi) {
                    double sacc = 0.0;
#pragma unroll
                    for (int kb = 0; kb < 3; ++kb)
                        if (kb >= j && kb < bi) {
#pragma unroll
                            for (int q = 0; q < 16; ++q)
                                sacc = fma(L[(16 * bi + ra) * PD_LD + 16 * kb + q], L[(16 * kb + q) * PD_LD + 16 * j + cc], sacc);
                        }
                    tv[j] = sacc;
                }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < bi) T[j * 256 + ra * 16 + cc] = tv[j];
            __syncthreads();                           // T complete; every read of L[bi][*] (the original blocks) is done
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < bi) {
                    double sacc = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) sacc = fma(L[(16 * bi + ra) * PD_LD + 16 * bi + q], T[j * 256 + q * 16 + cc], sacc);
                    tv[j] = -sacc;
                }
            __syncthreads();                           // T has been read
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < bi) L[(16 * bi + ra) * PD_LD + 16 * j + cc] = tv[j];
            __syncthreads();
        }
        for (int e = t; e < NB * NB; e += 256) {
            const int i = e >> 6, j = e & 63;
            lo[e] = j <= i ? L[i * PD_LD + j] : 0.0;
        }
    }
}

template <bool PRELOADED>
__device__ inline void potrf_diag_tile(double* __restrict__ aug, int N, int M, int k, int b, double* __restrict__ linv,
                                       int* __restrict__ info, double* L, double* rdiag) {
    potrf_diag_tile_v<PRELOADED, 0>(aug, N, M, k, b, linv, info, L, rdiag);
}

__global__ void __launch_bounds__(256) k_potrf_diag(double* __restrict__ aug, int N, int M, int k,
                                                    double* __restrict__ linv, int* __restrict__ info) {
    __shared__ double L[NB * PD_LD];
    __shared__ double rdiag[NB + 768];                 // 1 / L[i][i], then the scratch of the blocked inverse
    potrf_diag_tile<false>(aug, N, M, k, blockIdx.x, linv, info, L, rdiag);
}

// ------------------------------------------------------------------ fp64 MFMA tile product
// C (rows x cols) = or -= A B for every system of a batch.  A is global [m][k] (row stride lda); B is [n][k]
// (BT: "N x T" product, C[m][n] = sum_k A[m][k] B[n][k]) or [k][n].  A workgroup (4 waves, 2 x 2) owns a
// TS x TS tile of C, TS = 32 WB, each wave WB x WB MFMA blocks of 16 x 16.  Operands go through LDS in depth-16
// chunks stored [row][k] with a row stride of 18 doubles: the 32 lanes ds_read_b64 serves per cycle (16 rows x 2
// k) then fall on 32 distinct 8-byte bank pairs.  The next chunk is fetched into registers while the current
// one is multiplied.  C may alias A when cols <= TS and depth covers all of A's columns (panel, back_diag): a
// workgroup has consumed all its A rows before it stores.
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int MM_KC = 16;
constexpr int MM_LD = MM_KC + 2;

struct MMArgs {
    double* c;
    const double* a;
    const double* b;
    long long c_sys, a_sys, b_sys;   // strides between systems
    long long lda, ldb, ldc;
    int rows, cols, depth;           // depth % 16 == 0
    int row0, col0;                  // matrix coordinates of C[0][0], for the triangle test
    int tri;                         // skip tiles that lie entirely above the diagonal
    int subtract;                    // C -= A B, else C = A B
};

template <int WB, bool BT>
__global__ void __launch_bounds__(256, 2) k_mm64(const MMArgs g) {
    constexpr int TS = 32 * WB;
    __shared__ double sA[TS * MM_LD], sB[TS * MM_LD];
    const int r0 = blockIdx.y * TS, c0 = blockIdx.x * TS;
    if (g.tri && g.row0 + r0 + TS - 1 < g.col0 + c0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1, li = lane & 15, lq = lane >> 4;
    const double* A = g.a + (long long)blockIdx.z * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)blockIdx.z * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(TS, g.rows - r0), b_n = min(TS, g.cols - c0);

    f64x2 ra[WB], rb[WB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < WB; ++q) {
            const int e = t + 256 * q;
            const int row = e >> 3, kp = e & 7;
            ra[q] = row < a_rows ? *reinterpret_cast<const f64x2*>(A + (long long)row * g.lda + k0 + 2 * kp)
                                 : f64x2{0.0, 0.0};
            if (BT) {
                rb[q] = row < b_n ? *reinterpret_cast<const f64x2*>(B + (long long)row * g.ldb + k0 + 2 * kp)
                                  : f64x2{0.0, 0.0};
            } else {
                const int kr = e / (TS / 2), cp = e % (TS / 2);
                rb[q] = 2 * cp < b_n ? *reinterpret_cast<const f64x2*>(B + (long long)(k0 + kr) * g.ldb + 2 * cp)
                                     : f64x2{0.0, 0.0};
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < WB; ++q) {
            const int e = t + 256 * q;
            const int row = e >> 3, kp = e & 7;
            *reinterpret_cast<f64x2*>(sA + row * MM_LD + 2 * kp) = ra[q];
            if (BT) {
                *reinterpret_cast<f64x2*>(sB + row * MM_LD + 2 * kp) = rb[q];
            } else {
                const int kr = e / (TS / 2), cp = e % (TS / 2);
                sB[(2 * cp) * MM_LD + kr] = rb[q].x;
                sB[(2 * cp + 1) * MM_LD + kr] = rb[q].y;
            }
        }
    };

    f64x4 acc[WB][WB];
#pragma unroll
    for (int i = 0; i < WB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    fetch(0);
    stash();
    __syncthreads();
    const double* pa = sA + (wm * 16 * WB + li) * MM_LD + lq;
    const double* pb = sB + (wn * 16 * WB + li) * MM_LD + lq;
    for (int k0 = 0; k0 < g.depth; k0 += MM_KC) {
        const bool more = k0 + MM_KC < g.depth;
        if (more) fetch(k0 + MM_KC);
#pragma unroll
        for (int k4 = 0; k4 < MM_KC / 4; ++k4) {
            double a[WB], b[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) a[i] = pa[i * 16 * MM_LD + 4 * k4];
#pragma unroll
            for (int j = 0; j < WB; ++j) b[j] = pb[j * 16 * MM_LD + 4 * k4];
#pragma unroll
            for (int i = 0; i < WB; ++i)
#pragma unroll
                for (int j = 0; j < WB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            __syncthreads();
            stash();
            __syncthreads();
        }
    }

    // C -= acc as batches of loads, then the stores of the batch: written as "*dst = *dst - acc" per element the
    // compiler must assume that a store aliases the next load and serialises WB * WB * 4 memory round trips
    double* C = g.c + (long long)blockIdx.z * g.c_sys + (long long)r0 * g.ldc + c0;
#pragma unroll
    for (int i = 0; i < WB; ++i) {
        double old[4][WB];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wm * 16 * WB + i * 16 + lq + 4 * r;
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                const int col = wn * 16 * WB + j * 16 + li;
                old[r][j] = (g.subtract && row < a_rows && col < b_n) ? C[(long long)row * g.ldc + col] : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wm * 16 * WB + i * 16 + lq + 4 * r;
            if (row < a_rows)
#pragma unroll
                for (int j = 0; j < WB; ++j) {
                    const int col = wn * 16 * WB + j * 16 + li;
                    if (col < b_n) C[(long long)row * g.ldc + col] = g.subtract ? old[r][j] - acc[i][j][r] : acc[i][j][r];
                }
        }
    }
}

template <int WB, bool BT>
void launch_mm(const MMArgs& g, int B, hipStream_t s) {
    constexpr int TS = 32 * WB;
    if (g.rows <= 0 || g.cols <= 0) return;
    hipLaunchKernelGGL((k_mm64<WB, BT>), dim3((unsigned)lc::ceil_div(g.cols, TS), (unsigned)lc::ceil_div(g.rows, TS), (unsigned)B),
                       dim3(256), 0, s, g);
}

// The same product on the vector ALU (kept for comparison, lc_debug_chol_big_kernel(1); it served the deep updates
// until the 4x4x4 MFMA below): 128 x 128 tile, 8 x 8 accumulators per thread, two workgroups per CU.  v_fma_f64
// out-runs the 16x16x4 fp64 MFMA once a SIMD holds two waves (tools/mfma_f64_rate.hip: 57 vs 45 TFLOP/s), and an
// 8 x 8 register tile needs only 8 ds_read_b128 per 64 FMAs.  Operands sit k-major in LDS ([k][128 + 2]); a thread
// owns rows 32 i + 2 ty + {0, 1} and columns 32 j + 2 tx + {0, 1}, so the 16 lanes one LDS cycle serves read 256
// contiguous bytes (B) or two broadcast addresses (A).
constexpr int MV_TS = 128, MV_KC = 16, MV_LD = MV_TS + 2;

template <bool BT>
__global__ void __launch_bounds__(256, 2) k_mm64v(const MMArgs g) {
    __shared__ double sA[MV_KC * MV_LD], sB[MV_KC * MV_LD];
    const int r0 = blockIdx.y * MV_TS, c0 = blockIdx.x * MV_TS;
    if (g.tri && g.row0 + r0 + MV_TS - 1 < g.col0 + c0) return;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const double* A = g.a + (long long)blockIdx.z * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)blockIdx.z * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(MV_TS, g.rows - r0), b_n = min(MV_TS, g.cols - c0);

    f64x2 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + 256 * q;
            const int row = e >> 3, kp = e & 7;
            ra[q] = row < a_rows ? *reinterpret_cast<const f64x2*>(A + (long long)row * g.lda + k0 + 2 * kp)
                                 : f64x2{0.0, 0.0};
            if (BT) {
                rb[q] = row < b_n ? *reinterpret_cast<const f64x2*>(B + (long long)row * g.ldb + k0 + 2 * kp)
                                  : f64x2{0.0, 0.0};
            } else {
                const int kr = e >> 6, cp = e & 63;
                rb[q] = 2 * cp < b_n ? *reinterpret_cast<const f64x2*>(B + (long long)(k0 + kr) * g.ldb + 2 * cp)
                                     : f64x2{0.0, 0.0};
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + 256 * q;
            const int row = e >> 3, kp = e & 7;
            sA[(2 * kp) * MV_LD + row] = ra[q].x;
            sA[(2 * kp + 1) * MV_LD + row] = ra[q].y;
            if (BT) {
                sB[(2 * kp) * MV_LD + row] = rb[q].x;
                sB[(2 * kp + 1) * MV_LD + row] = rb[q].y;
            } else {
                const int kr = e >> 6, cp = e & 63;
                *reinterpret_cast<f64x2*>(sB + kr * MV_LD + 2 * cp) = rb[q];
            }
        }
    };

    double acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;

    fetch(0);
    stash();
    __syncthreads();
    const double* pa = sA + 2 * ty;
    const double* pb = sB + 2 * tx;
    for (int k0 = 0; k0 < g.depth; k0 += MV_KC) {
        const bool more = k0 + MV_KC < g.depth;
        if (more) fetch(k0 + MV_KC);
#pragma unroll 4
        for (int k = 0; k < MV_KC; ++k) {
            f64x2 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f64x2*>(pa + k * MV_LD + 32 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f64x2*>(pb + k * MV_LD + 32 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * i][2 * j] = fma(a[i].x, b[j].x, acc[2 * i][2 * j]);
                    acc[2 * i][2 * j + 1] = fma(a[i].x, b[j].y, acc[2 * i][2 * j + 1]);
                    acc[2 * i + 1][2 * j] = fma(a[i].y, b[j].x, acc[2 * i + 1][2 * j]);
                    acc[2 * i + 1][2 * j + 1] = fma(a[i].y, b[j].y, acc[2 * i + 1][2 * j + 1]);
                }
        }
        if (more) {
            __syncthreads();
            stash();
            __syncthreads();
        }
    }

    // loads of a batch first, then its stores (see k_mm64: element-wise read-modify-write serialises the round trips)
    double* C = g.c + (long long)blockIdx.z * g.c_sys + (long long)r0 * g.ldc + c0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f64x2 old[4][4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * h + ii;
            const int row = 32 * (i >> 1) + 2 * ty + (i & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 32 * j + 2 * tx;
                old[ii][j] = (g.subtract && row < a_rows && col < b_n)
                                 ? *reinterpret_cast<const f64x2*>(C + (long long)row * g.ldc + col) : f64x2{0.0, 0.0};
            }
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * h + ii;
            const int row = 32 * (i >> 1) + 2 * ty + (i & 1);
            if (row < a_rows)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 32 * j + 2 * tx;
                    if (col < b_n) {
                        f64x2 v = {acc[i][2 * j], acc[i][2 * j + 1]};
                        if (g.subtract) v = old[ii][j] - v;
                        *reinterpret_cast<f64x2*>(C + (long long)row * g.ldc + col) = v;
                    }
                }
        }
    }
}

// The deep updates on the matrix pipe after all -- with v_mfma_f64_4x4x4_4b_f64.  Measured on this part
// (tools/mfma_f64_rate.hip): the 16x16x4 fp64 MFMA issues every ~130 cycles from one wave (34 TFLOP/s, 45-50 with two
// waves per SIMD), the 4x4x4 four-block form every 16-17 cycles (65-70 TFLOP/s from a single wave per SIMD, the
// datasheet rate).  Its four blocks are independent 4x4x4 products with lanes  A: 16 k + 4 blk + i,  B: 16 k + 4 blk + j,
// D: 16 i + 4 blk + j  (tools/mfma_f64_4x4_layout.hip; CBSZ / ABID have no effect); with the A block replicated
// over blk -- an LDS broadcast read -- one instruction is a 4 x 16 x 4 product.  Tile 128 x 128 per workgroup, wave
// w owns rows 32 w .. 32 w + 31 and all 128 columns (8 row groups x 8 column groups = 64 accumulators in AGPRs).
// Software pipeline: LDS double buffer (one barrier per depth-16 chunk), B fragments of the next depth-4 step in a
// second register set, every A fragment reloaded in place right after its eight MFMAs, next chunk in registers.
// The MFMAs are inline asm with a tied accumulator: left to the register allocator a quarter of them came out as
// D != C with write-after-read chains between neighbours and the loop ran at 40 cycles per MFMA instead of 16.
constexpr int MQ_TS = 128, MQ_KC = 16, MQ_LD = MQ_KC + 2;
constexpr int MQ_BUF = MQ_TS * MQ_LD;          // doubles per operand buffer
constexpr int MQ_LDS_BYTES = 2 * 2 * MQ_BUF * 8;

template <bool BT>
__global__ void __launch_bounds__(256, 2) k_mm64q(const MMArgs g) {
    extern __shared__ double smem[];            // [2 buffers][A, B][MQ_BUF]
    const int r0 = blockIdx.y * MQ_TS, c0 = blockIdx.x * MQ_TS;
    if (g.tri && g.row0 + r0 + MQ_TS - 1 < g.col0 + c0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const double* A = g.a + (long long)blockIdx.z * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)blockIdx.z * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(MQ_TS, g.rows - r0), b_n = min(MQ_TS, g.cols - c0);

    // loader: 4 x 16 B per operand and thread (rows (t >> 3) + 32 q, k pair t & 7); rows past the edge are clamped
    // to the last valid one -- their products are never stored
    const int lrow = t >> 3, lkp = t & 7;
    f64x2 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = lrow + 32 * q;
            ra[q] = *reinterpret_cast<const f64x2*>(A + (long long)min(row, a_rows - 1) * g.lda + k0 + 2 * lkp);
            if (BT) {
                rb[q] = *reinterpret_cast<const f64x2*>(B + (long long)min(row, b_n - 1) * g.ldb + k0 + 2 * lkp);
            } else {
                const int e = t + 256 * q, kr = e >> 6, cp = e & 63;
                rb[q] = *reinterpret_cast<const f64x2*>(B + (long long)(k0 + kr) * g.ldb + min(2 * cp, b_n - 2));
            }
        }
    };
    auto stash = [&](int buf) {
        double* sA = smem + buf * 2 * MQ_BUF;
        double* sB = sA + MQ_BUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int so = (lrow + 32 * q) * MQ_LD + 2 * lkp;
            *reinterpret_cast<f64x2*>(sA + so) = ra[q];
            if (BT) {
                *reinterpret_cast<f64x2*>(sB + so) = rb[q];
            } else {
                const int e = t + 256 * q, kr = e >> 6, cp = e & 63;
                sB[(2 * cp) * MQ_LD + kr] = rb[q].x;
                sB[(2 * cp + 1) * MQ_LD + kr] = rb[q].y;
            }
        }
    };

    // v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks; lanes  A: 16 k + 4 blk + i,  B: 16 k + 4 blk + j,
    // D: 16 i + 4 blk + j.  With the A block replicated over blk (an LDS broadcast) it is a 4 x 16 x 4 product.
    // Wave w owns rows 32 w .. 32 w + 31 of the tile and all 128 columns: 8 row groups x 8 column groups.
    double acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;

    const int oa = (w * 32 + (lane & 3)) * MQ_LD + lq, ob = MQ_BUF + li * MQ_LD + lq;
    double fa[8], fb[2][8];
    // one depth-4 step: the B fragments of the next step go to the other register set up front, every A fragment is
    // reloaded in place as soon as its eight MFMAs are issued
    auto step = [&](int cur, bool next, int nbuf, int nk4) {
        const double* base = smem + nbuf * 2 * MQ_BUF + 4 * nk4;
        if (next) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[cur ^ 1][j] = base[ob + j * 16 * MQ_LD];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[i]), "v"(fb[cur][j]));
            if (next) fa[i] = base[oa + i * 4 * MQ_LD];
        }
    };

    const int nk = g.depth / MQ_KC;
    fetch(0);
    stash(0);
    if (nk > 1) fetch(MQ_KC);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = smem[oa + i * 4 * MQ_LD];
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[0][j] = smem[ob + j * 16 * MQ_LD];
    auto chunk = [&](int k, auto last) {
        constexpr bool LAST = decltype(last)::value;
        const int buf = k & 1;
        step(0, true, buf, 1);
        step(1, true, buf, 2);
        if (!LAST) stash(buf ^ 1);          // chunk k+1 (in registers since the previous iteration)
        step(0, true, buf, 3);
        if (!LAST && k + 2 < nk) fetch((k + 2) * MQ_KC);
        __syncthreads();                  // chunk k+1 visible; everyone has read all of chunk k
        step(1, !LAST, buf ^ 1, 0);
    };
    for (int k = 0; k + 1 < nk; ++k) chunk(k, std::false_type{});
    chunk(nk - 1, std::true_type{});

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the MFMAs above are opaque to the hazard recogniser
    double* C = g.c + (long long)blockIdx.z * g.c_sys + (long long)r0 * g.ldc + c0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double old[4][8];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int row = w * 32 + 4 * (4 * h + ii) + lq;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                old[ii][j] = (g.subtract && row < a_rows && 16 * j + li < b_n) ? C[(long long)row * g.ldc + 16 * j + li] : 0.0;
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int row = w * 32 + 4 * (4 * h + ii) + lq;
            if (row < a_rows)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (16 * j + li < b_n)
                        C[(long long)row * g.ldc + 16 * j + li] = g.subtract ? old[ii][j] - acc[4 * h + ii][j] : acc[4 * h + ii][j];
        }
    }
}


// ------------------------------------------------------------------ fused 64-column steps (left-looking inside an outer block)
// The first version ran three launches per 64-column step -- diagonal tile, panel P = A Linv', and a rank-64 update of
// the rest of the outer block in which every 64 x 64 tile was its own workgroup doing one depth-64 product and a
// read-modify-write of C: ~170 workgroups per system and step, ~10 us each for 0.5 MFLOP.  Here a step is the diagonal
// tile plus ONE launch with one workgroup per 64-row tile i > k, which
//   (1) applies the one contribution its tile of column k still lacks,  C = A[i,k] - L[i,k-1] L[k,k-1]'  (see (3)),
//   (2) forms  L[i,k] = C Linv_kk'  (C goes through LDS as the A operand) and stores it, and
//   (3) PRE-UPDATES its tile of the next column with everything that is already final:
//           A[i,k+1] -= sum_{j = K0 .. k-1} L[i,j] L[k+1,j]'      (one product of depth 64 (k - K0), no RMW per 64),
//       the j = k term being the one step (1) of the next launch adds -- except for the DIAGONAL tiles of the outer
//       block, which every row tile keeps current step by step ((2'), round 5), so that the next diagonal tile is final
//       one depth-64 product after L[k+1,k] and is factored in the same launch (4).
// Same flops, one read-modify-write per tile and block column instead of one per 64 columns of depth, 38 instead of
// ~170 workgroups per system and step.  The back substitution is restructured the same way (k_bstep: one launch per
// step,  H[i,k] = (Z[i,k] - sum_{j > k} H[i,j] L[j,k]) Linv_kk ).
constexpr int ST_LDC = NB + 2;     // row stride of the C tile in LDS

// acc += A B' (BT: B stored [n][k]) or A B (B stored [k][n]) for one 64 x 64 tile, depth % 16 == 0, operands staged
// through LDS in depth-16 chunks like k_mm64.  The products run on v_mfma_f64_4x4x4_4b_f64 (see k_mm64q: the only
// fp64 MFMA form that issues at the datasheet rate on this part): wave w owns rows 16 w .. 16 w + 15 and all 64
// columns, acc[i][j] = rows 4 i + (lane >> 4) of its strip x columns 16 j + (lane & 15); one depth-4 step is 4 A
// fragments (LDS broadcast reads), 4 B fragments and 16 MFMAs with tied AGPR accumulators.  A_LDS: A is a 64 x 64 tile
// already in LDS (row stride ST_LDC) and depth == 64.
template <bool BT, bool A_LDS>
__device__ __forceinline__ void tile_mac(double (&acc)[4][4], const double* A, long long lda, int a_rows,
                                         const double* B, long long ldb, int depth, double* sA, double* sB) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    // Operand chunks are fetched TWO chunks ahead into two register sets (round 3): one chunk of MFMAs is ~0.5 us, a
    // global load 1-2 us, so with the next chunk alone in flight every chunk waited for its operands (matrix pipe busy
    // 0.2-0.3 in the step kernels).  Same MFMA order, same sums: the same bits.
    f64x2 ra[2][2], rb[2][2];
    auto fetch = [&](int k0, f64x2 (&xa)[2], f64x2 (&xb)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = t + 256 * q;
            const int row = e >> 3, kp = e & 7;
            if (!A_LDS)
                xa[q] = row < a_rows ? *reinterpret_cast<const f64x2*>(A + (long long)row * lda + k0 + 2 * kp) : f64x2{0.0, 0.0};
            if (BT) {
                xb[q] = *reinterpret_cast<const f64x2*>(B + (long long)row * ldb + k0 + 2 * kp);
            } else {
                const int kr = e >> 5, cp = e & 31;
                xb[q] = *reinterpret_cast<const f64x2*>(B + (long long)(k0 + kr) * ldb + 2 * cp);
            }
        }
    };
    auto stash = [&](const f64x2 (&xa)[2], const f64x2 (&xb)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = t + 256 * q;
            const int row = e >> 3, kp = e & 7;
            if (!A_LDS) *reinterpret_cast<f64x2*>(sA + row * MM_LD + 2 * kp) = xa[q];
            if (BT) {
                *reinterpret_cast<f64x2*>(sB + row * MM_LD + 2 * kp) = xb[q];
            } else {
                const int kr = e >> 5, cp = e & 31;
                sB[(2 * cp) * MM_LD + kr] = xb[q].x;
                sB[(2 * cp + 1) * MM_LD + kr] = xb[q].y;
            }
        }
    };
    const double* pb = sB + li * MM_LD + lq;
    auto mma = [&](int k0) {
        const double* pa = A_LDS ? A + (w * 16 + (lane & 3)) * ST_LDC + k0 + lq : sA + (w * 16 + (lane & 3)) * MM_LD + lq;
        constexpr int LDA = A_LDS ? ST_LDC : MM_LD;
#pragma unroll
        for (int k4 = 0; k4 < MM_KC / 4; ++k4) {
            double fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = pa[i * 4 * LDA + 4 * k4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = pb[j * 16 * MM_LD + 4 * k4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]));
        }
    };
    const int nk = depth / MM_KC;
    fetch(0, ra[0], rb[0]);
    if (nk > 1) fetch(MM_KC, ra[1], rb[1]);
    __syncthreads();                                   // the previous user of sA / sB is done
    stash(ra[0], rb[0]);
    __syncthreads();
    int k = 0;
    for (; k + 2 <= nk; k += 2) {                      // chunk k is in LDS, chunk k+1 in register set 1, set 0 is free
        if (k + 2 < nk) fetch((k + 2) * MM_KC, ra[0], rb[0]);
        mma(k * MM_KC);
        __syncthreads();
        stash(ra[1], rb[1]);
        __syncthreads();
        if (k + 3 < nk) fetch((k + 3) * MM_KC, ra[1], rb[1]);
        mma((k + 1) * MM_KC);
        if (k + 2 < nk) {
            __syncthreads();
            stash(ra[0], rb[0]);
            __syncthreads();
        }
    }
    if (k < nk) mma(k * MM_KC);                        // odd chunk count: the last one was stashed by the pair before it
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the MFMAs above are opaque to the hazard recogniser
}

// element (row, col) of the 64 x 64 tile that accumulator acc[i][j] of this lane holds
#define LC_TILE_ROW(i) (w * 16 + (i) * 4 + lq)
#define LC_TILE_COL(j) ((j) * 16 + li)
#define LC_FOR_TILE(i, j) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)

__global__ void __launch_bounds__(256, 3) k_lstep(double* __restrict__ aug, int N, int M, int k, int K0, int K1,
                                                  double* __restrict__ linv, int* __restrict__ info) {
    __shared__ double sA[NB * MM_LD], sB[NB * MM_LD], sC[NB * ST_LDC];
    const int R = N + M, nb = N / NB;
    const int i_t = k + 1 + blockIdx.x, b = blockIdx.y;
    const int r0 = i_t * NB, a_rows = min(NB, R - r0);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    double* Ab = aug + (long long)b * R * N;
    double* Arow = Ab + (long long)r0 * N;
    const double* Lk = linv + ((long long)b * nb + k) * NB * NB;
    double acc[4][4];
    // (1) C = A[i,k] - L[i,k-1] L[k,k-1]'  -> LDS
    LC_FOR_TILE(i, j) acc[i][j] = 0.0;
    if (k > K0) tile_mac<true, false>(acc, Arow + (k - 1) * NB, N, a_rows, Ab + (long long)k * NB * N + (k - 1) * NB, N, NB, sA, sB);
    {
        const double* src = Arow + k * NB;
        double old[4][4];
        LC_FOR_TILE(i, j) old[i][j] = LC_TILE_ROW(i) < a_rows ? src[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] : 0.0;
        LC_FOR_TILE(i, j) sC[LC_TILE_ROW(i) * ST_LDC + LC_TILE_COL(j)] = old[i][j] - acc[i][j];
    }
    // (2) L[i,k] = C Linv_kk'   (tile_mac's first barrier makes sC visible)
    LC_FOR_TILE(i, j) acc[i][j] = 0.0;
    tile_mac<true, true>(acc, sC, 0, a_rows, Lk, NB, NB, sA, sB);
    {
        double* dst = Arow + k * NB;
        LC_FOR_TILE(i, j) if (LC_TILE_ROW(i) < a_rows) dst[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] = acc[i][j];
    }
    if (k + 1 >= K1) return;
    const bool own = i_t == k + 1;                     // row tile k+1: the next diagonal tile is its own
    // (2') round 5: every row tile INSIDE the outer block keeps its own diagonal tile current, step by step:
    //          A[i,i] -= L[i,k] L[i,k]'           (both operands are the tile just formed)
    // so that the next diagonal tile is final after ONE product of depth 64 -- before, the workgroup of row tile k+1 first
    // ran the pre-update (3) of its column-(k+1) tile, which IS the diagonal tile, at depth 64 (k - K0): ~14 us on average
    // on the path every step waits for (the chain of a rank's 3 systems is 30 such steps).  Same flops; the diagonal tiles
    // are summed step by step instead of once per outer block.
    double old[4][4];
    if (i_t < K1) {
        __syncthreads();                               // everyone has read sC as the A operand of (2)
        LC_FOR_TILE(i, j) sC[LC_TILE_ROW(i) * ST_LDC + LC_TILE_COL(j)] = acc[i][j];
        LC_FOR_TILE(i, j) acc[i][j] = 0.0;
        __syncthreads();                               // tile_mac fetches B (= sC here) before its first barrier
        tile_mac<true, true>(acc, sC, 0, a_rows, sC, ST_LDC, NB, sA, sB);
        double* dd = Arow + i_t * NB;
        LC_FOR_TILE(i, j) old[i][j] = dd[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] - acc[i][j];
        if (!own) LC_FOR_TILE(i, j) dd[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] = old[i][j];
    }
    if (own) {
        // (4) the next diagonal tile is complete: factor it right away (L, Linv: potrf_diag_tile on the values at hand, in
        // the LDS of sC / sA) -- one launch per step instead of two, beside the other workgroups' tiles
        __syncthreads();                               // sC was an operand of the product above
        LC_FOR_TILE(i, j) sC[LC_TILE_ROW(i) * PD_LD + LC_TILE_COL(j)] = old[i][j];
        potrf_diag_tile<true>(aug, N, M, k + 1, b, linv, info, sC, sA);
        return;
    }
    // (3) pre-update of the tile in column k+1 with everything that is final
    if (k == K0) return;                               // nothing final to apply yet
    LC_FOR_TILE(i, j) acc[i][j] = 0.0;
    tile_mac<true, false>(acc, Arow + K0 * NB, N, a_rows, Ab + (long long)(k + 1) * NB * N + K0 * NB, N, (k - K0) * NB, sA, sB);
    double* dst = Arow + (k + 1) * NB;
    LC_FOR_TILE(i, j) old[i][j] = LC_TILE_ROW(i) < a_rows ? dst[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] : 0.0;
    LC_FOR_TILE(i, j) old[i][j] -= acc[i][j];
    LC_FOR_TILE(i, j) if (LC_TILE_ROW(i) < a_rows) dst[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] = old[i][j];
}

// one 64-row tile of the bottom block, one step of the back substitution (see k_lstep's header)
__device__ __forceinline__ void bstep_tile(double* __restrict__ aug, int N, int M, int k, int K1, const double* __restrict__ linv,
                                  int tile, int b, double* sA, double* sB, double* sC) {
    const int R = N + M, nb = N / NB;
    const int r0 = tile * NB, a_rows = min(NB, M - r0);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    double* Ab = aug + (long long)b * R * N;
    double* Zrow = Ab + (long long)(N + r0) * N;
    const double* Lk = linv + ((long long)b * nb + k) * NB * NB;
    double acc[4][4];
    LC_FOR_TILE(i, j) acc[i][j] = 0.0;
    // C = Z[i,k] - sum_{j = k+1 .. K1-1} H[i,j] L[j,k]  -> LDS
    const int depth = (K1 - 1 - k) * NB;
    if (depth > 0)
        tile_mac<false, false>(acc, Zrow + (k + 1) * NB, N, a_rows, Ab + (long long)(k + 1) * NB * N + k * NB, N, depth, sA, sB);
    {
        const double* src = Zrow + k * NB;
        double old[4][4];
        LC_FOR_TILE(i, j) old[i][j] = LC_TILE_ROW(i) < a_rows ? src[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] : 0.0;
        LC_FOR_TILE(i, j) sC[LC_TILE_ROW(i) * ST_LDC + LC_TILE_COL(j)] = old[i][j] - acc[i][j];
    }
    // H[i,k] = C Linv_kk
    LC_FOR_TILE(i, j) acc[i][j] = 0.0;
    tile_mac<false, true>(acc, sC, 0, a_rows, Lk, NB, NB, sA, sB);
    double* dst = Zrow + k * NB;
    LC_FOR_TILE(i, j) if (LC_TILE_ROW(i) < a_rows) dst[(long long)LC_TILE_ROW(i) * N + LC_TILE_COL(j)] = acc[i][j];
}

__global__ void __launch_bounds__(256, 3) k_bstep(double* __restrict__ aug, int N, int M, int k, int K1,
                                                  const double* __restrict__ linv) {
    __shared__ double sA[NB * MM_LD], sB[NB * MM_LD], sC[NB * ST_LDC];
    bstep_tile(aug, N, M, k, K1, linv, blockIdx.x, blockIdx.y, sA, sB, sC);
}

// All the steps of an outer block in ONE launch (round 3): a row tile's steps only read what the SAME workgroup wrote
// in its earlier steps (its own rows of H) besides L and Linv, which are final -- no dependency between workgroups, so
// the chain of K1 - K0 dependent launches (each waiting its turn beside the sweeps' workgroups in a busy fit) becomes a
// loop.  Between steps: every wave's stores have reached L2 (vmcnt), then ONE wave invalidates the CU's L1 -- the
// line of block column k+1 was read (old value) before this workgroup overwrote it -- and a barrier releases the rest.
// inverse: row tile r works on block columns k >= r only (the block-upper triangle).
__global__ void __launch_bounds__(256, 3) k_bsteps(double* __restrict__ aug, int N, int M, int K0, int K1,
                                                   const double* __restrict__ linv, int inverse) {
    __shared__ double sA[NB * MM_LD], sB[NB * MM_LD], sC[NB * ST_LDC];
    const int tile = blockIdx.x;
    for (int k = K1 - 1; k >= K0; --k) {
        if (inverse && tile > k) break;
        bstep_tile(aug, N, M, k, K1, linv, tile, blockIdx.y, sA, sB, sC);
        if (k > K0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x < 64) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
}
// The deep updates of SMALL batches (round 5): C (rows x cols) -= A B' (BT) or A B per system in 64 x 64 tiles on tile_mac
// -- the same MFMA, the same order of the sums over the depth as k_mm64q: the same bits -- for batches whose 128 x 128 tiles
// do not fill the chip: a rank of 8 holds 3 systems of a fold's 20, its first deep update is ~300 tiles of 150 us each on
// 256 CUs; in 64 x 64 tiles it is 1200 of ~30 us, three to a CU.
template <bool BT>
__global__ void __launch_bounds__(256, 3) k_mm64s(const MMArgs g) {
    __shared__ double sA[NB * MM_LD], sB[NB * MM_LD];
    const int r0 = blockIdx.y * NB, c0 = blockIdx.x * NB;
    if (g.tri && g.row0 + r0 + NB - 1 < g.col0 + c0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const double* A = g.a + (long long)blockIdx.z * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)blockIdx.z * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(NB, g.rows - r0);
    double acc[4][4];
    LC_FOR_TILE(i, j) acc[i][j] = 0.0;
    tile_mac<BT, false>(acc, A, g.lda, a_rows, B, g.ldb, g.depth, sA, sB);
    double* C = g.c + (long long)blockIdx.z * g.c_sys + (long long)r0 * g.ldc + c0;
    double old[4][4];
    LC_FOR_TILE(i, j) old[i][j] = (g.subtract && LC_TILE_ROW(i) < a_rows) ? C[(long long)LC_TILE_ROW(i) * g.ldc + LC_TILE_COL(j)] : 0.0;
    LC_FOR_TILE(i, j) if (LC_TILE_ROW(i) < a_rows)
        C[(long long)LC_TILE_ROW(i) * g.ldc + LC_TILE_COL(j)] = g.subtract ? old[i][j] - acc[i][j] : acc[i][j];
}
#undef LC_TILE_ROW
#undef LC_TILE_COL
#undef LC_FOR_TILE

// deep updates: big_kernel 2 = 4x4x4 MFMA (k_mm64q, the default; k_mm64s -- the same bits -- while the batch has fewer than
// MM_SMALL_BELOW 128 x 128 tiles: measured -9 .. -12 % of a whole solve from 1 to 20 systems of 1920, -5 % at 40, equal at
// 80: tools/chol_small_tiles.py), 3 = k_mm64s always, 4 = k_mm64q always, 1 = vector ALU (k_mm64v), 0 = 16x16x4 MFMA (k_mm64<4>)
constexpr long long MM_SMALL_BELOW = 4096;
template <bool BT>
void launch_big(const MMArgs& g, int B, hipStream_t s, int big_kernel, long long small_tiles_below = 0) {
    if (g.rows <= 0 || g.cols <= 0) return;
    const dim3 grid((unsigned)lc::ceil_div(g.cols, 128), (unsigned)lc::ceil_div(g.rows, 128), (unsigned)B);
    // small batches: 64 x 64 tiles (k_mm64s: same bits) while the 128 x 128 ones would not fill the chip twice over
    const long long tiles128 = (long long)grid.x * grid.y * grid.z / (g.tri ? 2 : 1);
    if ((big_kernel == 2 && tiles128 < small_tiles_below && g.cols % NB == 0) || big_kernel == 3) {
        hipLaunchKernelGGL((k_mm64s<BT>), dim3((unsigned)lc::ceil_div(g.cols, NB), (unsigned)lc::ceil_div(g.rows, NB), (unsigned)B),
                           dim3(256), 0, s, g);
        return;
    }
    if (big_kernel == 2 || big_kernel == 4) {        // LDS attribute: set (and checked) by lc_batch_chol_solve
        hipLaunchKernelGGL((k_mm64q<BT>), grid, dim3(256), MQ_LDS_BYTES, s, g);
    } else if (big_kernel == 1) {
        hipLaunchKernelGGL((k_mm64v<BT>), grid, dim3(256), 0, s, g);
    } else {
        launch_mm<4, BT>(g, B, s);
    }
}

__global__ void __launch_bounds__(256) k_extract_h(const double* __restrict__ aug, int N, int M, float* __restrict__ h,
                                                   const int* __restrict__ slot) {
    const int i = blockIdx.x, b = blockIdx.y;
    const double* src = aug + ((long long)b * (N + M) + N + i) * N;
    float* dst = h + ((long long)(slot ? slot[b] : b) * M + i) * N;
    for (int j = threadIdx.x; j < N; j += 256) dst[j] = (float)src[j];
}

// inverse mode: the bottom block holds the inverse's block-upper triangle (row tile <= column tile); mirror the rest
__global__ void __launch_bounds__(256) k_extract_sym(const double* __restrict__ aug, int N, float* __restrict__ h,
                                                     const int* __restrict__ slot) {
    const int i = blockIdx.x, b = blockIdx.y;
    const double* bot = aug + ((long long)b * 2 * N + N) * N;
    float* dst = h + ((long long)(slot ? slot[b] : b) * N + i) * N;
    const int bi = i / NB;
    for (int j = threadIdx.x; j < N; j += 256)
        dst[j] = (float)(j / NB >= bi ? bot[(long long)i * N + j] : bot[(long long)j * N + i]);
}

}  // namespace

// The variants of the blocking are PER-CALL options (lc_chol_options, NULL = the defaults): columns per outer block
// (multiple of NB, default 512); the deep-update kernel; fused left-looking 64-column steps (k_lstep / k_bstep, default)
// or the first version's three launches per step; the deep updates left-looking too (one product of the full depth per
// block column: measured, no gain -- the deep-update kernel is not bound by its C read-modify-write -- and less parallel
// for small batches).  No process-wide switches: two fits with different settings coexist in one process.
static lc_chol_options chol_defaults() { return lc_chol_options{512, 2, 1, 0, 1}; }

// inverse: the bottom block is the N x N identity and only the block-upper triangle of  I (top)^-1  is formed -- the
// rows of the product are independent, row tile r of Z = I L^-T is zero left of block column r, and row tile r of the
// result is only needed from block column r on (the rest by symmetry): in every step the bottom row tiles beyond the
// current block column are skipped, N^3 / 3 flops per pass instead of N^3.
static int chol_solve_impl(double* d_aug, int B, int N, int M, double* d_linv, float* d_h, const int32_t* d_slot,
                           int32_t* d_info, lc_stream_t stream, bool inverse, const lc_chol_options* opt);

extern "C" int lc_batch_chol_solve(double* d_aug, int B, int N, int M, double* d_linv, float* d_h,
                                   const int32_t* d_slot, int32_t* d_info, const lc_chol_options* opt,
                                   lc_stream_t stream) {
    return chol_solve_impl(d_aug, B, N, M, d_linv, d_h, d_slot, d_info, stream, false, opt);
}

extern "C" int lc_batch_chol_inverse(double* d_aug, int B, int N, double* d_linv, float* d_p, const int32_t* d_slot,
                                     int32_t* d_info, const lc_chol_options* opt, lc_stream_t stream) {
    return chol_solve_impl(d_aug, B, N, N, d_linv, d_p, d_slot, d_info, stream, true, opt);
}

static int chol_solve_impl(double* d_aug, int B, int N, int M, double* d_linv, float* d_h, const int32_t* d_slot,
                           int32_t* d_info, lc_stream_t stream, bool inverse, const lc_chol_options* opt) {
    const lc_chol_options o = opt ? *opt : chol_defaults();
    LC_REQUIRE(o.outer_block > 0 && o.outer_block % NB == 0 && o.big_kernel >= 0 && o.big_kernel <= 4, LC_E_SHAPE,
               "lc_batch_chol_solve: options: outer_block must be a multiple of %d, big_kernel in 0..4", NB);
    LC_REQUIRE(d_aug && d_linv && d_h && d_info, LC_E_BADARG, "lc_batch_chol_solve: null pointer");
    LC_REQUIRE(B > 0 && B <= 65535 && N > 0 && N % LC_NB == 0 && M > 0 && M % LC_MB == 0, LC_E_SHAPE,
               "lc_batch_chol_solve: need N %% %d == 0, M %% %d == 0, B <= 65535", LC_NB, LC_MB);
    hipStream_t s = lc::as_stream(stream);
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_mm64q<true>), MQ_LDS_BYTES)) return rc;
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_mm64q<false>), MQ_LDS_BYTES)) return rc;
    lc::ScopedTimer timer_(lc::T_CHOL_SOLVE, s);
    LC_HIP(hipMemsetAsync(d_info, 0, sizeof(int32_t) * B, s));
    const int nb = N / NB;
    const int R = N + M;
    const int ob = o.outer_block / NB;                       // NB-steps per outer block
    const long long sys = (long long)R * N;
    const bool fused = o.fused_steps || inverse;             // the triangular limits exist for the default kernels only
    const bool left_deep = o.left_deep && !inverse;
    MMArgs g{};
    g.c_sys = g.a_sys = g.b_sys = sys;
    g.lda = g.ldb = g.ldc = N;
    for (int K0 = 0; K0 < nb; K0 += ob) {
        const int K1 = min(K0 + ob, nb);
        if (left_deep && K0 > 0) {
            // block column [K0, K1) takes the contributions of ALL earlier block columns in one product of depth 64 K0:
            // every C tile is read and written once per factorisation instead of once per outer block to its left
            const int c0 = K0 * NB;
            g.a = d_aug + (long long)c0 * N;                 // L[rows >= c0][0 .. c0)
            g.b = d_aug + (long long)c0 * N;                 // L[c0 .. c1 rows][0 .. c0)   ([n][k])
            g.a_sys = g.b_sys = g.c_sys = sys;
            g.ldb = N;
            g.c = d_aug + (long long)c0 * N + c0;
            g.rows = R - c0; g.cols = (K1 - K0) * NB; g.depth = c0;
            g.row0 = g.col0 = c0; g.tri = 1; g.subtract = 1;
            launch_big<true>(g, B, s, o.big_kernel, MM_SMALL_BELOW);
        }
        for (int k = K0; k < K1; ++k) {
            // fused steps: only the first diagonal tile of an outer block (completed by the deep update) needs a launch
            // of its own -- the others are factored by the previous step's workgroup of that row tile (k_lstep (4))
            if (!fused || k == K0)
                hipLaunchKernelGGL(k_potrf_diag, dim3(B), dim3(256), 0, s, d_aug, N, M, k, d_linv, d_info);
            const int below = (k + 1) * NB;                  // first row under the diagonal tile
            if (fused) {
                // inverse: bottom row tiles 0 .. k only (tile r is zero left of block column r)
                const int tiles = inverse ? (nb - k - 1) + min(M / NB, k + 1) : lc::ceil_div(R - below, NB);
                if (tiles > 0)
                    hipLaunchKernelGGL(k_lstep, dim3((unsigned)tiles, (unsigned)B), dim3(256), 0, s, d_aug, N, M, k, K0, K1,
                                       d_linv, d_info);
                continue;
            }
            // panel:  P <- P Linv_kk'   (rows below the diagonal tile, including the M augmented rows; in place)
            g.a = g.c = d_aug + (long long)below * N + k * NB;
            g.a_sys = g.c_sys = sys;
            g.b = d_linv + (long long)k * NB * NB;
            g.b_sys = (long long)nb * NB * NB;
            g.ldb = NB;
            g.rows = R - below; g.cols = NB; g.depth = NB;
            g.row0 = below; g.col0 = k * NB; g.tri = 0; g.subtract = 0;
            launch_mm<2, true>(g, B, s);
            // rank-64 update of the rest of the outer block
            g.b = g.a; g.b_sys = sys; g.ldb = N;
            g.c = d_aug + (long long)below * N + below;
            g.cols = K1 * NB - below;
            g.col0 = below; g.tri = 1; g.subtract = 1;
            launch_mm<2, true>(g, B, s);
        }
        if (left_deep) continue;
        // everything right of the outer block, once, at the block's full depth
        const int c1 = K1 * NB;
        g.a = g.b = d_aug + (long long)c1 * N + K0 * NB;
        g.a_sys = g.b_sys = g.c_sys = sys;
        g.ldb = N;
        g.c = d_aug + (long long)c1 * N + c1;
        g.rows = inverse ? (N - c1) + min(M, c1) : R - c1;   // inverse: bottom rows beyond c1 are still zero in this block
        g.cols = N - c1; g.depth = (K1 - K0) * NB;
        g.row0 = g.col0 = c1; g.tri = 1; g.subtract = 1;
        launch_big<true>(g, B, s, o.big_kernel, MM_SMALL_BELOW);
    }
    if (int rc = lc::launched("cholesky sweep")) return rc;
    // H L = Z from the last block column to the first:  H_k = Z_k Linv_kk,  Z_j -= H_k L_kj (j < k)
    const int last_full = ((nb - 1) / ob) * ob;
    for (int K0 = last_full; K0 >= 0; K0 -= ob) {
        const int K1 = min(K0 + ob, nb);
        if (left_deep && K1 < nb) {
            // Z[:, K0 .. K1) -= H[:, K1 .. nb) L[K1 .. nb rows, K0 .. K1 cols]: all finished block columns to the right at once
            g.a = d_aug + (long long)N * N + K1 * NB;
            g.b = d_aug + (long long)K1 * NB * N + K0 * NB;
            g.c = d_aug + (long long)N * N + K0 * NB;
            g.a_sys = g.b_sys = g.c_sys = sys;
            g.ldb = N;
            g.rows = M; g.cols = (K1 - K0) * NB; g.depth = (nb - K1) * NB;
            g.row0 = g.col0 = 0; g.tri = 0; g.subtract = 1;
            launch_big<false>(g, B, s, o.big_kernel, MM_SMALL_BELOW);
        }
        if (fused && (o.persistent & 1)) {
            // every step of the outer block in one launch (k_bsteps); inverse: row tiles 0 .. K1-1 at most
            const int rt = inverse ? min(M / NB, K1) : lc::ceil_div(M, NB);
            hipLaunchKernelGGL(k_bsteps, dim3((unsigned)rt, (unsigned)B), dim3(256), 0, s, d_aug, N, M, K0, K1, d_linv,
                               inverse ? 1 : 0);
        }
        for (int k = K1 - 1; k >= K0 && !(fused && (o.persistent & 1)); --k) {
            if (fused) {
                // inverse: only row tiles 0 .. k of block column k (the block-upper triangle)
                const int rt = inverse ? min(M / NB, k + 1) : lc::ceil_div(M, NB);
                hipLaunchKernelGGL(k_bstep, dim3((unsigned)rt, (unsigned)B), dim3(256), 0, s, d_aug, N, M, k, K1, d_linv);
                continue;
            }
            g.a = g.c = d_aug + (long long)N * N + k * NB;
            g.a_sys = g.c_sys = sys;
            g.b = d_linv + (long long)k * NB * NB;
            g.b_sys = (long long)nb * NB * NB;
            g.ldb = NB;
            g.rows = M; g.cols = NB; g.depth = NB;
            g.row0 = g.col0 = 0; g.tri = 0; g.subtract = 0;
            launch_mm<2, false>(g, B, s);
            g.b = d_aug + (long long)k * NB * N + K0 * NB;   // L[k rows][columns of the outer block left of k]
            g.b_sys = sys; g.ldb = N;
            g.c = d_aug + (long long)N * N + K0 * NB;
            g.cols = (k - K0) * NB;
            g.subtract = 1;
            launch_mm<2, false>(g, B, s);
        }
        if (left_deep) continue;
        g.a = d_aug + (long long)N * N + K0 * NB;
        g.b = d_aug + (long long)K0 * NB * N;
        g.c = d_aug + (long long)N * N;
        g.a_sys = g.b_sys = g.c_sys = sys;
        g.ldb = N;
        g.rows = inverse ? min(M, K0 * NB) : M;              // inverse: block columns < K0 need row tiles < K0 only
        g.cols = K0 * NB; g.depth = (K1 - K0) * NB;
        g.tri = 0; g.subtract = 1;
        launch_big<false>(g, B, s, o.big_kernel, MM_SMALL_BELOW);
    }
    if (int rc = lc::launched("back substitution")) return rc;
    if (inverse) hipLaunchKernelGGL(k_extract_sym, dim3(N, B), dim3(256), 0, s, d_aug, N, d_h, d_slot);
    else hipLaunchKernelGGL(k_extract_h, dim3(M, B), dim3(256), 0, s, d_aug, N, M, d_h, d_slot);
    return lc::launched("k_extract_h");
}

// ------------------------------------------------------------------ Gram matrix on the fp64 MFMA
// K = X X' (fp32 inputs, fp64 products and sums) as one "N x T" product of the deep-update kernel: the vector-ALU
// kernel of lc_gram_f64 is bound by its LDS reads (one 8-byte read per two FMAs: 20 TFLOP/s), k_mm64q runs the same
// flops at the matrix pipe's rate.  X is widened to fp64 once (zero-padded to 16 columns), the lower-triangle tiles
// are computed, the rest mirrored.  d_work: T * pad16(p) doubles.
namespace {
__global__ void __launch_bounds__(256) k_widen_f32_f64(const float* __restrict__ x, long long ldx, int p, int p16,
                                                       double* __restrict__ out) {
    const long long r = blockIdx.x;
    for (int c = threadIdx.x; c < p16; c += 256) out[r * p16 + c] = c < p ? (double)x[r * ldx + c] : 0.0;
}
__global__ void __launch_bounds__(256) k_mirror_upper(double* __restrict__ K, long long ldk, int T) {
    const int i = blockIdx.x;                              // row i takes K[i][j] = K[j][i] for the tiles right of its own
    for (int j = (i / MQ_TS + 1) * MQ_TS + threadIdx.x; j < T; j += 256) K[(long long)i * ldk + j] = K[(long long)j * ldk + i];
}
}  // namespace

extern "C" int lc_gram_f64_mfma(const float* d_x, int64_t ldx, int64_t T, int64_t p, double* d_work, double* d_k,
                                int64_t ldk, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_work && d_k, LC_E_BADARG, "lc_gram_f64_mfma: null pointer");
    LC_REQUIRE(T > 0 && p > 0 && ldx >= p && ldk >= T && T < (1 << 30) && p < (1 << 30), LC_E_SHAPE,
               "lc_gram_f64_mfma: bad shape");
    hipStream_t s = lc::as_stream(stream);
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_mm64q<true>), MQ_LDS_BYTES)) return rc;
    const int p16 = (int)((p + 15) / 16 * 16);
    lc::ScopedTimer timer_(lc::T_GRAM, s);
    hipLaunchKernelGGL(k_widen_f32_f64, dim3((unsigned)T), dim3(256), 0, s, d_x, (long long)ldx, (int)p, p16, d_work);
    MMArgs g{};
    g.a = g.b = d_work;
    g.c = d_k;
    g.lda = g.ldb = p16;
    g.ldc = ldk;
    g.rows = g.cols = (int)T;
    g.depth = p16;
    g.row0 = g.col0 = 0;
    g.tri = 1;
    g.subtract = 0;
    const dim3 grid((unsigned)lc::ceil_div<long long>(T, MQ_TS), (unsigned)lc::ceil_div<long long>(T, MQ_TS), 1);
    hipLaunchKernelGGL((k_mm64q<true>), grid, dim3(256), MQ_LDS_BYTES, s, g);
    hipLaunchKernelGGL(k_mirror_upper, dim3((unsigned)T), dim3(256), 0, s, d_k, (long long)ldk, (int)T);
    return lc::launched("lc_gram_f64_mfma");
}
