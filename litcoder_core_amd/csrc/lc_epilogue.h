// Score epilogue shared by the two alpha-sweep kernels (lc_gemm.hip: f32 MFMA, lc_gemm16.hip: fp16x3 MFMA).
//
// Both kernels finish with 32x32 MFMA accumulators (v_mfma_f32_32x32x*): lane (li = lane & 31, lh = lane >> 5)
// holds, for column li, the 16 rows  8q + 4lh + j  (q = r >> 2, j = r & 3) of a 32-row block.  The epilogue
// turns one such block x 32 columns of predictions into the block's partial moments
//     S1 = sum p,   M2 = sum (p - mean_block)^2,   S3 = sum (p - mean_block) (y - mean_y)
// (p = prediction for corr, fl32(y - prediction) for R2; ridge_regression.py:124-133) which k_score_finalize
// merges in fp64.  In R2 mode S3 is not needed; its slot and the fourth one carry S1 / M2 of the raw targets
// y of the block, formed by the very same instruction sequence as those of the residual: the reference's
// ``1 - resvar / Presp.var()`` is EXACTLY 0 whenever the prediction is too small to change y in fp32 (large
// alphas), ties that its first-maximum argmax then resolves, and two identical computations on identical
// inputs reproduce such ties bit for bit where two different variance algorithms would not.
// Nothing overlaps the epilogue (one block per CU), so it is written on packed fp32 pairs
// (v_pk_mul/add/fma_f32) with the row masks only on blocks that contain padding rows.
//
// The gathered validation targets come "row-quad interleaved" from lc_val_stats:
//     yv[((i >> 2) * V + c) * 4 + (i & 3)] = y[va[i], c]        (zeros for padding rows i >= n_val)
// so the four consecutive rows a lane owns are ONE 16-byte load and a half-wave reads 512 contiguous bytes.
#pragma once
#include <hip/hip_runtime.h>

namespace lc {

typedef float ep_f32x2 __attribute__((ext_vector_type(2)));
typedef float ep_f32x4 __attribute__((ext_vector_type(4)));
typedef float ep_f32x16 __attribute__((ext_vector_type(16)));

__device__ inline long long yv_index(int i, long long c, long long V) {
    return ((long long)(i >> 2) * V + c) * 4 + (i & 3);
}

// the four row quads of one 32-row block (first row i0, a multiple of 32) for column c
struct EpiTargets {
    ep_f32x4 y[4];
};

__device__ inline void epi_load_targets(const float* __restrict__ yv, long long V, int i0, int lh, long long c,
                                        EpiTargets& t) {
    const ep_f32x4* base = reinterpret_cast<const ep_f32x4*>(yv) + ((long long)((i0 >> 2) + lh) * V + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) t.y[q] = base[(long long)(2 * q) * V];
}

// One 32-row block x this lane's column.  SCALED: prediction = acc * rs[row] * cs (power-of-two scales of the
// fp16x3 kernel);  CORR: statistics of the prediction, else of the residual;  MASK: rows >= n_val are padding.
// Writes S1 / M2 / S3 of the block (R2: S1 / M2 of the residual, S1 / M2 of y) to part[0], part[V], part[2V]
// (, part[3V]) from the lanes with lh == 0 and `store`.
template <bool SCALED, bool CORR, bool MASK>
__device__ inline void epi_block(const ep_f32x16& acc, const EpiTargets& t, const ep_f32x4 (&rs)[4], float cs,
                                 float ymean, int i0, int n_val, int lh, float* __restrict__ part, long long V,
                                 bool store) {
    ep_f32x2 p[8], yc[8];                         // R2: yc holds the raw targets
    ep_f32x2 s1v = {0.f, 0.f}, s1yv = {0.f, 0.f};
    const ep_f32x2 ym2 = {ymean, ymean};
    const ep_f32x2 cs2 = {cs, cs};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * q + h;
            ep_f32x2 a = {acc[4 * q + 2 * h], acc[4 * q + 2 * h + 1]};
            const ep_f32x2 y = {t.y[q][2 * h], t.y[q][2 * h + 1]};
            if (SCALED) {
                const ep_f32x2 r2 = {rs[q][2 * h], rs[q][2 * h + 1]};
                a = a * r2 * cs2;
            }
            yc[k] = CORR ? y - ym2 : y;
            p[k] = CORR ? a : y - a;
            if (MASK) {
                const int row = i0 + 8 * q + 4 * lh + 2 * h;
                if (row >= n_val) { p[k].x = 0.f; yc[k].x = 0.f; }
                if (row + 1 >= n_val) { p[k].y = 0.f; yc[k].y = 0.f; }
            }
            s1v += p[k];
            if (!CORR) s1yv += yc[k];
        }
    float s1 = s1v.x + s1v.y;
    s1 += __shfl_xor(s1, 32);
    const int nb = MASK ? min(32, n_val - i0) : 32;
    const float mean_b = nb > 0 ? s1 / (float)nb : 0.f;
    const ep_f32x2 mb2 = {mean_b, mean_b};
    ep_f32x2 m2v = {0.f, 0.f}, s3v = {0.f, 0.f};
    if (!CORR) {
        float s1y = s1yv.x + s1yv.y;
        s1y += __shfl_xor(s1y, 32);
        const float mean_y = nb > 0 ? s1y / (float)nb : 0.f;
        const ep_f32x2 my2 = {mean_y, mean_y};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ep_f32x2 d = p[k] - mb2, dy = yc[k] - my2;
            if (MASK) {
                const int row = i0 + 8 * (k >> 1) + 4 * lh + 2 * (k & 1);
                if (row >= n_val) { d.x = 0.f; dy.x = 0.f; }
                if (row + 1 >= n_val) { d.y = 0.f; dy.y = 0.f; }
            }
            m2v += d * d;
            s3v += dy * dy;
        }
        float m2 = m2v.x + m2v.y, m2y = s3v.x + s3v.y;
        m2 += __shfl_xor(m2, 32);
        m2y += __shfl_xor(m2y, 32);
        if (lh == 0 && store) {
            part[0] = s1;
            part[V] = m2;
            part[2 * V] = s1y;
            part[3 * V] = m2y;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ep_f32x2 d = p[k] - mb2;
        if (MASK) {
            const int row = i0 + 8 * (k >> 1) + 4 * lh + 2 * (k & 1);
            if (row >= n_val) d.x = 0.f;
            if (row + 1 >= n_val) d.y = 0.f;
        }
        m2v += d * d;
        s3v += d * yc[k];
    }
    float m2 = m2v.x + m2v.y, s3 = s3v.x + s3v.y;
    m2 += __shfl_xor(m2, 32);
    s3 += __shfl_xor(s3, 32);
    if (lh == 0 && store) {
        part[0] = s1;
        part[V] = m2;
        part[2 * V] = s3;
    }
}

// Series terms: one 32-row block x this lane's column of the FOUR shared terms T_j = P'_j Y (true values, scales
// undone) -> the block's partial moments, from which lc_series_sweep's finalisation forms the score of every alpha on
// the polynomial series (prediction = sum_j c_j T_j):
//     part[j]              = sum T_j                                   j = 0..3
//     part[4 + j]          = sum (T_j - mean_j,block) (y - mean_y)
//     part[8 + idx(j, l)]  = sum (T_j - mean_j,block) (T_l - mean_l,block)   j <= l, row-major upper triangle (10)
// written from the lanes with lh == 0 and `store`.  MASK: rows >= n_val are padding.
constexpr int EPI_SERIES_PARTS = 18;

template <bool MASK>
__device__ inline void epi_series_block(const float (&T)[4][16], const EpiTargets& t, float ymean, int i0, int n_val,
                                        int lh, float* __restrict__ part, long long V, bool store) {
    const int nb = MASK ? min(32, n_val - i0) : 32;
    float s1[4], mean[4];
    bool live[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) live[r] = !MASK || (i0 + 8 * (r >> 2) + 4 * lh + (r & 3)) < n_val;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += live[r] ? T[j][r] : 0.f;
        a += __shfl_xor(a, 32);
        s1[j] = a;
        mean[j] = nb > 0 ? a / (float)nb : 0.f;
    }
    float cy[4] = {0.f, 0.f, 0.f, 0.f}, sc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) sc[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float yc = live[r] ? t.y[r >> 2][r & 3] - ymean : 0.f;
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = live[r] ? T[j][r] - mean[j] : 0.f;
        int k = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cy[j] += d[j] * yc;
#pragma unroll
            for (int l = j; l < 4; ++l) sc[k++] += d[j] * d[l];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) cy[j] += __shfl_xor(cy[j], 32);
#pragma unroll
    for (int k = 0; k < 10; ++k) sc[k] += __shfl_xor(sc[k], 32);
    if (lh == 0 && store) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[(long long)j * V] = s1[j];
            part[(long long)(4 + j) * V] = cy[j];
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) part[(long long)(8 + k) * V] = sc[k];
    }
}

// dispatch on the two wave-uniform run-time switches (score mode, padding rows in the block)
template <bool SCALED>
__device__ inline void epi_block_dispatch(bool corr, const ep_f32x16& acc, const EpiTargets& t,
                                          const ep_f32x4 (&rs)[4], float cs, float ymean, int i0, int n_val, int lh,
                                          float* __restrict__ part, long long V, bool store) {
    const bool full = i0 + 32 <= n_val;
    if (corr) {
        if (full) epi_block<SCALED, true, false>(acc, t, rs, cs, ymean, i0, n_val, lh, part, V, store);
        else epi_block<SCALED, true, true>(acc, t, rs, cs, ymean, i0, n_val, lh, part, V, store);
    } else {
        if (full) epi_block<SCALED, false, false>(acc, t, rs, cs, ymean, i0, n_val, lh, part, V, store);
        else epi_block<SCALED, false, true>(acc, t, rs, cs, ymean, i0, n_val, lh, part, V, store);
    }
}

}  // namespace lc
