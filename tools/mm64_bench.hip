// Diagnostic: the fp64 tile-product kernels of lc_chol.hip on the deep-update shape of the inner-fold Cholesky
// (20 systems, C 2144 x 1664 lower trapezoid, depth 256), timed in isolation.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mm64_bench.hip -o tools/bin/mm64_bench && tools/bin/mm64_bench
#include "../litcoder_core_amd/csrc/lc_core.hip"
#include "../litcoder_core_amd/csrc/lc_chol.hip"
#include <type_traits>
#include <vector>


// ---- experimental: software-pipelined MFMA tile (fragment double buffer, LDS double buffer, one barrier per chunk)
namespace {
__device__ unsigned long long g_stamp[4];
constexpr int MP_TS = 128, MP_KC = 16, MP_LD = MP_KC + 2;
constexpr int MP_BUF = MP_TS * MP_LD;          // doubles per operand buffer

template <bool BT, int FAKE>
__global__ void __launch_bounds__(256, 2) k_mm64p(const MMArgs g) {
    extern __shared__ double smem[];            // [2 buffers][A, B][MP_BUF]
    const int r0 = blockIdx.y * MP_TS, c0 = blockIdx.x * MP_TS;
    if (g.tri && g.row0 + r0 + MP_TS - 1 < g.col0 + c0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1, li = lane & 15, lq = lane >> 4;
    const double* A = g.a + (long long)blockIdx.z * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)blockIdx.z * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(MP_TS, g.rows - r0), b_n = min(MP_TS, g.cols - c0);

    // loader: 4 x 16 B per operand and thread; out-of-range rows are clamped (their products are never stored)
    const double* pa_g[4];
    const double* pb_g[4];
    int so_a[4], so_b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = t + 256 * q;
        const int row = e >> 3, kp = e & 7;
        pa_g[q] = A + (long long)min(row, a_rows - 1) * g.lda + 2 * kp;
        so_a[q] = row * MP_LD + 2 * kp;
        if (BT) {
            pb_g[q] = B + (long long)min(row, b_n - 1) * g.ldb + 2 * kp;
            so_b[q] = row * MP_LD + 2 * kp;
        } else {
            const int kr = e >> 6, cp = e & 63;
            pb_g[q] = B + (long long)kr * g.ldb + min(2 * cp, b_n - 2);
            so_b[q] = (2 * cp) * MP_LD + kr;
        }
    }
    f64x2 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const f64x2*>(pa_g[q] + k0);
            rb[q] = *reinterpret_cast<const f64x2*>(pb_g[q] + (BT ? (long long)k0 : (long long)k0 * g.ldb));
        }
    };
    auto stash = [&](int buf) {
        double* sA = smem + buf * 2 * MP_BUF;
        double* sB = sA + MP_BUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f64x2*>(sA + so_a[q]) = ra[q];
            if (BT) *reinterpret_cast<f64x2*>(sB + so_b[q]) = rb[q];
            else { sB[so_b[q]] = rb[q].x; sB[so_b[q] + MP_LD] = rb[q].y; }
        }
    };

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    const int oa = (wm * 64 + li) * MP_LD + lq, ob = MP_BUF + (wn * 64 + li) * MP_LD + lq;
    double fa[2][4], fb[2][4];
    auto frags = [&](int buf, int k4, int slot) {
        const double* base = smem + buf * 2 * MP_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[slot][i] = base[oa + i * 16 * MP_LD + 4 * k4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[slot][j] = base[ob + j * 16 * MP_LD + 4 * k4];
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
    };

    const int nk = g.depth / MP_KC;
    fetch(0);
    stash(0);
    if (nk > 1) fetch(MP_KC);
    __syncthreads();
    frags(0, 0, 0);
    for (int k = 0; k < nk; ++k) {
        const int buf = k & 1;
        frags(buf, 1, 1);
        mfmas(0);
        frags(buf, 2, 0);
        mfmas(1);
        if (FAKE < 1 && k + 1 < nk) stash(buf ^ 1);     // chunk k+1 (in registers since the previous iteration)
        frags(buf, 3, 1);
        mfmas(0);
        if (FAKE < 1 && k + 2 < nk) fetch((k + 2) * MP_KC);
        if (FAKE < 2) __syncthreads();                                // chunk k+1 visible; everyone has read all of chunk k
        if (k + 1 < nk) frags(buf ^ 1, 0, 0);
        mfmas(1);
    }

    double* C = g.c + (long long)blockIdx.z * g.c_sys + (long long)r0 * g.ldc + c0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wm * 64 + i * 16 + lq + 4 * r;
            if (row < a_rows)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wn * 64 + j * 16 + li;
                    if (col < b_n) {
                        double* dst = C + (long long)row * g.ldc + col;
                        *dst = g.subtract ? *dst - acc[i][j][r] : acc[i][j][r];
                    }
                }
        }
}

template <bool BT, int FAKE>
__global__ void __launch_bounds__(256, 2) k_mm64q(const MMArgs g) {
    extern __shared__ double smem[];            // [2 buffers][A, B][MP_BUF]
    unsigned long long tb;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
    const int r0 = blockIdx.y * MP_TS, c0 = blockIdx.x * MP_TS;
    if (g.tri && g.row0 + r0 + MP_TS - 1 < g.col0 + c0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const double* A = g.a + (long long)blockIdx.z * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)blockIdx.z * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(MP_TS, g.rows - r0), b_n = min(MP_TS, g.cols - c0);

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
        double* sA = smem + buf * 2 * MP_BUF;
        double* sB = sA + MP_BUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int so = (lrow + 32 * q) * MP_LD + 2 * lkp;
            *reinterpret_cast<f64x2*>(sA + so) = ra[q];
            if (BT) {
                *reinterpret_cast<f64x2*>(sB + so) = rb[q];
            } else {
                const int e = t + 256 * q, kr = e >> 6, cp = e & 63;
                sB[(2 * cp) * MP_LD + kr] = rb[q].x;
                sB[(2 * cp + 1) * MP_LD + kr] = rb[q].y;
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

    const int oa = (w * 32 + (lane & 3)) * MP_LD + lq, ob = MP_BUF + li * MP_LD + lq;
    double fa[8], fb[2][8];
    // one depth-4 step: the B fragments of the next step go to the other register set up front, every A fragment is
    // reloaded in place as soon as its eight MFMAs are issued
    auto step = [&](int cur, bool next, int nbuf, int nk4) {
        const double* base = smem + nbuf * 2 * MP_BUF + 4 * nk4;
        if (next) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[cur ^ 1][j] = base[ob + j * 16 * MP_LD];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[i]), "v"(fb[cur][j]));
            if (next) fa[i] = base[oa + i * 4 * MP_LD];
        }
    };

    const int nk = g.depth / MP_KC;
    fetch(0);
    stash(0);
    if (nk > 1) fetch(MP_KC);
    __syncthreads();
    unsigned tv[4] = {0, 0, 0, 0};
    if (FAKE == -1) {                                   // touch the C tile: its HBM reads happen under the main loop
        const char* Ct = reinterpret_cast<const char*>(g.c + (long long)blockIdx.z * g.c_sys + (long long)r0 * g.ldc + c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = t + 256 * q;
            const unsigned off = (unsigned)min(id >> 3, a_rows - 1) * (unsigned)(g.ldc * 8) + (unsigned)min((id & 7) * 128, (b_n - 16) * 8);
            tv[q] = *reinterpret_cast<const unsigned*>(Ct + off);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = smem[oa + i * 4 * MP_LD];
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[0][j] = smem[ob + j * 16 * MP_LD];
    auto chunk = [&](int k, auto last) {
        constexpr bool LAST = decltype(last)::value;
        const int buf = k & 1;
        step(0, FAKE < 3, buf, 1);
        step(1, FAKE < 3, buf, 2);
        if (FAKE < 1 && !LAST) stash(buf ^ 1);          // chunk k+1 (in registers since the previous iteration)
        step(0, FAKE < 3, buf, 3);
        if (FAKE < 1 && !LAST && k + 2 < nk) fetch((k + 2) * MP_KC);
        if (FAKE < 2) __syncthreads();                  // chunk k+1 visible; everyone has read all of chunk k
        step(1, FAKE < 3 && !LAST, buf ^ 1, 0);
    };
    unsigned long long t0, t1, q0, q1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int k = 0; k + 1 < nk; ++k) chunk(k, std::false_type{});
    chunk(nk - 1, std::true_type{});
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q1)::"memory");
    if (t == 0 && blockIdx.x == 0 && blockIdx.y == 8 && blockIdx.z == 3) { g_stamp[0] = t1 - t0; g_stamp[1] = q1 - q0; }

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
    unsigned long long te;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te)::"memory");
    if ((tv[0] ^ tv[1] ^ tv[2] ^ tv[3]) == 0x9e3779b9u) g_stamp[0] = 1;   // keeps the touches alive
    if (t == 0 && blockIdx.x == 0 && blockIdx.y == 8 && blockIdx.z == 3) { g_stamp[2] = t0 - tb; g_stamp[3] = te - t1; }
}

template <bool BT, int FAKE>
__global__ void __launch_bounds__(256, 2) k_mm64s(const MMArgs g, int gx, int gy, int per_sys, int total, int skew_cycles) {
    extern __shared__ double smem[];            // [2 buffers][A, B][MP_BUF]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    // persistent workgroups, two per CU; the second half starts late so that the two workgroups of a CU do not
    // reach their C read-modify-write at the same moment
    if ((int)blockIdx.x >= (int)gridDim.x / 2 && skew_cycles > 0) {
        unsigned long long s0, s1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
        do {
            asm volatile("s_sleep 32\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
        } while ((long long)(s1 - s0) < (long long)skew_cycles);
    }
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
    const int bz = id / per_sys;
    int rem = id - bz * per_sys, bx = 0;
    if (g.tri) { while (rem >= gy - bx) { rem -= gy - bx; ++bx; } }
    else { bx = rem / gy; rem -= bx * gy; }
    const int by = (g.tri ? bx : 0) + rem;
    __syncthreads();                            // the previous tile's fragments are read by all waves
    const int r0 = by * MP_TS, c0 = bx * MP_TS;
    const double* A = g.a + (long long)bz * g.a_sys + (long long)r0 * g.lda;
    const double* B = g.b + (long long)bz * g.b_sys + (BT ? (long long)c0 * g.ldb : (long long)c0);
    const int a_rows = min(MP_TS, g.rows - r0), b_n = min(MP_TS, g.cols - c0);

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
        double* sA = smem + buf * 2 * MP_BUF;
        double* sB = sA + MP_BUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int so = (lrow + 32 * q) * MP_LD + 2 * lkp;
            *reinterpret_cast<f64x2*>(sA + so) = ra[q];
            if (BT) {
                *reinterpret_cast<f64x2*>(sB + so) = rb[q];
            } else {
                const int e = t + 256 * q, kr = e >> 6, cp = e & 63;
                sB[(2 * cp) * MP_LD + kr] = rb[q].x;
                sB[(2 * cp + 1) * MP_LD + kr] = rb[q].y;
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

    const int oa = (w * 32 + (lane & 3)) * MP_LD + lq, ob = MP_BUF + li * MP_LD + lq;
    double fa[8], fb[2][8];
    // one depth-4 step: the B fragments of the next step go to the other register set up front, every A fragment is
    // reloaded in place as soon as its eight MFMAs are issued
    auto step = [&](int cur, bool next, int nbuf, int nk4) {
        const double* base = smem + nbuf * 2 * MP_BUF + 4 * nk4;
        if (next) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[cur ^ 1][j] = base[ob + j * 16 * MP_LD];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[i]), "v"(fb[cur][j]));
            if (next) fa[i] = base[oa + i * 4 * MP_LD];
        }
    };

    const int nk = g.depth / MP_KC;
    fetch(0);
    stash(0);
    if (nk > 1) fetch(MP_KC);
    __syncthreads();
    unsigned tv[4] = {0, 0, 0, 0};
    if (FAKE == -1) {                                   // touch the C tile: its HBM reads happen under the main loop
        const char* Ct = reinterpret_cast<const char*>(g.c + (long long)bz * g.c_sys + (long long)r0 * g.ldc + c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = t + 256 * q;
            const unsigned off = (unsigned)min(id >> 3, a_rows - 1) * (unsigned)(g.ldc * 8) + (unsigned)min((id & 7) * 128, (b_n - 16) * 8);
            tv[q] = *reinterpret_cast<const unsigned*>(Ct + off);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = smem[oa + i * 4 * MP_LD];
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[0][j] = smem[ob + j * 16 * MP_LD];
    auto chunk = [&](int k, auto last) {
        constexpr bool LAST = decltype(last)::value;
        const int buf = k & 1;
        step(0, FAKE < 3, buf, 1);
        step(1, FAKE < 3, buf, 2);
        if (FAKE < 1 && !LAST) stash(buf ^ 1);          // chunk k+1 (in registers since the previous iteration)
        step(0, FAKE < 3, buf, 3);
        if (FAKE < 1 && !LAST && k + 2 < nk) fetch((k + 2) * MP_KC);
        if (FAKE < 2) __syncthreads();                  // chunk k+1 visible; everyone has read all of chunk k
        step(1, FAKE < 3 && !LAST, buf ^ 1, 0);
    };
    for (int k = 0; k + 1 < nk; ++k) chunk(k, std::false_type{});
    chunk(nk - 1, std::true_type{});

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the MFMAs above are opaque to the hazard recogniser
    double* C = g.c + (long long)bz * g.c_sys + (long long)r0 * g.ldc + c0;
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
    if ((tv[0] ^ tv[1] ^ tv[2] ^ tv[3]) == 0x9e3779b9u) g_stamp[0] = 1;   // keeps the touches alive
    }
}

template <bool BT, int FAKE>
void launch_p(const MMArgs& g, int B, hipStream_t s) {
    static bool done = false;
    const int lds = 2 * 2 * MP_BUF * 8;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mm64p<BT, FAKE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((k_mm64p<BT, FAKE>), dim3((unsigned)((g.cols + 127) / 128), (unsigned)((g.rows + 127) / 128), (unsigned)B),
                       dim3(256), lds, s, g);
}
template <bool BT, int FAKE>
void launch_q(const MMArgs& g, int B, hipStream_t s) {
    static bool done = false;
    const int lds = 2 * 2 * MP_BUF * 8;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mm64q<BT, FAKE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    hipLaunchKernelGGL((k_mm64q<BT, FAKE>), dim3((unsigned)((g.cols + 127) / 128), (unsigned)((g.rows + 127) / 128), (unsigned)B),
                       dim3(256), lds, s, g);
}
template <bool BT, int FAKE>
void launch_s(const MMArgs& g, int B, int skew, hipStream_t s) {
    static bool done = false;
    const int lds = 2 * 2 * MP_BUF * 8;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mm64s<BT, FAKE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    const int gx = (g.cols + 127) / 128, gy = (g.rows + 127) / 128;
    int per = 0;
    for (int x = 0; x < gx; ++x) per += g.tri ? gy - x : gy;
    hipLaunchKernelGGL((k_mm64s<BT, FAKE>), dim3(512), dim3(256), lds, s, g, gx, gy, per, per * B, skew);
}
}  // namespace

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int B = 20, N = 1920, M = 480, R = N + M;
    const int c1 = 256, depth = 256;
    const size_t n = (size_t)B * R * N;
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    double *d, *d0;
    (void)hipMalloc(&d, n * 8); (void)hipMalloc(&d0, n * 8);
    (void)hipMemcpy(d0, h.data(), n * 8, hipMemcpyHostToDevice);
    MMArgs g{};
    g.a = g.b = d + (long long)c1 * N;
    g.c = d + (long long)c1 * N + c1;
    g.a_sys = g.b_sys = g.c_sys = (long long)R * N;
    g.lda = g.ldb = g.ldc = N;
    g.rows = R - c1; g.cols = N - c1; g.depth = depth; g.row0 = g.col0 = c1; g.tri = 1; g.subtract = 1;
    long long tiles = 0;
    for (int x = 0; x < (g.cols + 127) / 128; ++x) for (int y = 0; y < (g.rows + 127) / 128; ++y) if (y >= x) ++tiles;
    const double flop = 2.0 * 128 * 128 * depth * tiles * B;
    auto report = [&](const char* name, double ms) { printf("%-46s %.3f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9); };
    (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    report("k_mm64v (v_fma_f64, 8x8 per thread)", time_ms([&] { g_big_kernel = 1; launch_big<true>(g, B, 0); }, 5));
    (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    report("k_mm64<4> (MFMA, 64x64 per wave)", time_ms([&] { g_big_kernel = 0; launch_big<true>(g, B, 0); }, 5));
    (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    report("k_mm64p (MFMA, pipelined fragments, LDS x2)", time_ms([&] { launch_p<true, 0>(g, B, 0); }, 5));
    report("  same, no global loads / LDS writes in the loop", time_ms([&] { launch_p<true, 1>(g, B, 0); }, 5));
    report("  same, and no barriers", time_ms([&] { launch_p<true, 2>(g, B, 0); }, 5));
    {
        std::vector<double> r1(n), r3(n);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        g_big_kernel = 1; launch_big<true>(g, B, 0);
        (void)hipMemcpy(r1.data(), d, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        launch_p<true, 0>(g, B, 0);
        (void)hipMemcpy(r3.data(), d, n * 8, hipMemcpyDeviceToHost);
        double md = 0;
        for (size_t i = 0; i < n; ++i) md = fmax(md, fabs(r1[i] - r3[i]));
        printf("max |VALU - pipelined MFMA| = %.3e\n", md);
    }
    (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    report("k_mm64q (4x4x4 MFMA as 4x16x4, 32x128 per wave)", time_ms([&] { launch_q<true, 0>(g, B, 0); }, 5));
    report("  same + C tile touched at the start", time_ms([&] { launch_q<true, -1>(g, B, 0); }, 5));
    {
        MMArgs g2 = g; g2.subtract = 0;
        report("  same, C = A B (no read of C)", time_ms([&] { launch_q<true, 0>(g2, B, 0); }, 5));
        unsigned long long st[4];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof st);
        printf("      loop %llu cycles, after it %llu cycles\n", st[0], st[3]);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    }
    report("  same, no global loads / LDS writes in the loop", time_ms([&] { launch_q<true, 1>(g, B, 0); }, 5));
    report("  same, and no barriers", time_ms([&] { launch_q<true, 2>(g, B, 0); }, 5));
    report("  same, and no fragment reloads (MFMAs only)", time_ms([&] { launch_q<true, 3>(g, B, 0); }, 5));
    for (int v = 0; v < 2; ++v) {
        if (v == 0) launch_q<true, 0>(g, B, 0); else launch_q<true, 3>(g, B, 0);
        unsigned long long st[4];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof st);
        printf("  main loop of one workgroup (%s): %llu cycles = %.1f per MFMA and wave, shader clock %.2f GHz, %.1f us\n",
               v ? "MFMAs only" : "full", st[0], (double)st[0] / 4096.0, (double)st[0] / ((double)st[1] * 10.0), st[1] * 0.01);
        printf("      before the loop %llu cycles, after it (C read-modify-write, stores retired) %llu cycles\n", st[2], st[3]);
    }
    {
        std::vector<double> r1(n), r3(n);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        g_big_kernel = 1; launch_big<true>(g, B, 0);
        (void)hipMemcpy(r1.data(), d, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        launch_q<true, 0>(g, B, 0);
        (void)hipMemcpy(r3.data(), d, n * 8, hipMemcpyDeviceToHost);
        double md = 0;
        for (size_t i = 0; i < n; ++i) md = fmax(md, fabs(r1[i] - r3[i]));
        printf("max |VALU - 4x4x4 MFMA| = %.3e\n", md);
    }
    for (int skew : {0, 20000, 40000, 70000, 100000}) {
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        char name[96];
        snprintf(name, sizeof name, "k_mm64s persistent, touch, skew %d cycles", skew);
        report(name, time_ms([&] { launch_s<true, -1>(g, B, skew, 0); }, 5));
    }
    {
        std::vector<double> r1(n), r3(n);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        g_big_kernel = 1; launch_big<true>(g, B, 0);
        (void)hipMemcpy(r1.data(), d, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
        launch_s<true, -1>(g, B, 40000, 0);
        (void)hipMemcpy(r3.data(), d, n * 8, hipMemcpyDeviceToHost);
        double md = 0;
        for (size_t i = 0; i < n; ++i) md = fmax(md, fabs(r1[i] - r3[i]));
        printf("  max |VALU - k_mm64s| = %.3e\n", md);
    }
    {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm64q<true, 0>, 256, 2 * 2 * MP_BUF * 8);
        printf("occupancy: k_mm64q %d", nb);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm64p<true, 0>, 256, 2 * 2 * MP_BUF * 8);
        printf(", k_mm64p %d", nb);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm64v<true>, 256, 0);
        printf(", k_mm64v %d", nb);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm64<4, true>, 256, 0);
        printf(", k_mm64<4> %d workgroups per CU\n", nb);
    }
    // results of the two must agree
    std::vector<double> r1(n), r2(n);
    (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    g_big_kernel = 1; launch_big<true>(g, B, 0);
    (void)hipMemcpy(r1.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(d, d0, n * 8, hipMemcpyDeviceToDevice);
    g_big_kernel = 0; launch_big<true>(g, B, 0);
    (void)hipMemcpy(r2.data(), d, n * 8, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < n; ++i) md = fmax(md, fabs(r1[i] - r2[i]));
    printf("max |VALU - MFMA| = %.3e\n", md);
    return 0;
}
