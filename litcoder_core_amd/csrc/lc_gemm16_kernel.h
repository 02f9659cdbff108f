// The fp16 x 3 MFMA contraction kernel of the V-wide sweeps and its operand-preparation kernels (gfx950): shared by
// csrc/lc_gemm16.hip (the product's entry points) and tools/debug_kernels/lc_debug_gemm16.hip (the stamped diagnostic
// build and the 16x16x32 experiment kernel, which are NOT part of the product library since round 5).  Everything
// lives in an anonymous namespace: each translation unit that includes this header gets its own copies.
#pragma once
#include <type_traits>

#include "lc_common.h"
#include "lc_epilogue.h"

// Experiment bits of the sweep kernel (tools/debug_kernels builds only: tools/sweep_delivery_probe.py); 0 in the product
#ifndef LC_SWEEP_EXPERIMENTS
#define LC_SWEEP_EXPERIMENTS 0
#endif
#ifndef LC_SWEEP_EXP_BITS
#define LC_SWEEP_EXP_BITS (-1)                     // >= 0: the bits as a compile-time constant (no run-time branches in the loop)
#endif

namespace {

constexpr bool EXPK = LC_SWEEP_EXPERIMENTS != 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int TM = 256, TN = 256, TK = 16;
constexpr int KG = TK / 8;                        // 8-k groups per K-tile
constexpr int CHUNK16 = 2 * KG * 256;             // 16-byte units per (tile, K-tile) chunk = 16 KB
constexpr int STAGE16 = 2 * CHUNK16;              // A chunk + B chunk = 32 KB
constexpr int NSTAGE = 4;                         // LDS ring
constexpr int LDS16_BYTES = NSTAGE * STAGE16 * 16 + TM * 4;   // 128 KB ring + the tile's 256 row scales

__device__ inline int xcd_tile_id16(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------ operand preparation
// One wave per row of H: exact power-of-two scale to [0.5, 1), split, scatter into the tiled layout.
// Groups (the inner folds of an outer fold): group g = source rows [g rows, (g + 1) rows), padded to rows_pad (whole
// 256-row tiles) in the image and in rs_inv, so that the groups' images are stacked tile-aligned.
// il_A > 0 (the hat matrices of the fused sweep: il_A alphas x M = rows / il_A validation rows each, stacked alpha by
// alpha in h): the image takes the 32-row blocks in the order (validation block, alpha) -- image block s = source block
// (s / il_A) of alpha (s % il_A) -- so that a 256-row tile of the sweep holds ALL alphas of a few validation blocks and
// its epilogue needs those few blocks of the validation targets, not eight different ones (round 4: the fused launch
// fetched the targets once per alpha, 0.61 GB of its 2.75 GB at cfg2).
// src_rows: the rows a group occupies in h (>= rows; the screening pass' image of the fused sweep takes the FIRST il_A alphas
// of every group and leaves the others out).
__global__ void __launch_bounds__(256) k_split_rows_f16(const float* __restrict__ h, long long ld, int rows, int K,
                                                        uint4* __restrict__ out, float* __restrict__ rs_inv,
                                                        int rows_pad, int groups, int il_A, int src_rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);          // row of the stacked, padded image
    const int lane = threadIdx.x & 63;
    if (r >= rows_pad * groups) return;
    const int KT = K / TK;
    const int g = r / rows_pad, rg = r - g * rows_pad;
    const bool live = rg < rows;
    int rsrc = rg;
    if (il_A > 0 && live) {
        const int s = rg >> 5, ib = s / il_A, a = s - ib * il_A;
        rsrc = a * (rows / il_A) + ib * 32 + (rg & 31);
    }
    const float* src = h + ((long long)g * src_rows + rsrc) * ld;
    float mx = 0.f;
    if (live)
        for (int k = lane * 4; k < K; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(src + k);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) frexpf(mx, &e);         // mx = f * 2^e, f in [0.5, 1)
    e = max(-120, min(120, e));
    const float s = ldexpf(1.f, -e);
    if (lane == 0) rs_inv[r] = ldexpf(1.f, e);
    const long long tile_base = (long long)(r >> 8) * KT * CHUNK16;
    const int rr = r & 255;
    for (int c = lane; c < K / 8; c += 64) {
        h8 hi, lo;
        if (live) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + c * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(src + c * 8 + 4);
            const float x[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hi[j] = (_Float16)x[j];
                lo[j] = (_Float16)(x[j] - (float)hi[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; }
        }
        const long long o = tile_base + (long long)(c / KG) * CHUNK16 + (c % KG) * 256 + rr;
        out[o] = *reinterpret_cast<uint4*>(&hi);
        out[o + KG * 256] = *reinterpret_cast<uint4*>(&lo);
    }
}

// The image of the MEAN of several folds' refit operators (round 6): row r of the result is
//     C[r][t] = scale * sum_f ( map_f[t] >= 0 ? M_f[r][map_f[t]] : 0 ),   t = 0 .. K-1   (fp32 sums, folds in order)
// -- M_f (rows x ld_f) the operator of fold f at the alpha a group of voxels chose there (its columns are the fold's
// training rows), map_f[t] the column of M_f that belongs to row t of the targets (-1: row t is not a training row of fold
// f).  nested_cv.py:293-296 returns the mean of the folds' weight matrices and nothing else of them, and
// mean_f (M_f Y[tr_f]) = (mean_f M_f scattered to all rows) Y: one contraction of depth T per group of voxels with the
// same alpha in every fold instead of one of depth n_train per fold.  One wave per row: the sums stay in registers between
// the row maximum and the split (MO_UNITS units of 8 columns per lane: K <= 512 MO_UNITS), image layout of k_split_rows_f16.
constexpr int MO_MAX_FOLDS = 16, MO_UNITS = 16;
struct MeanOpArgs {
    const float* m[MO_MAX_FOLDS];
    long long ld[MO_MAX_FOLDS];
    const int* map[MO_MAX_FOLDS];
    int nf;
    float scale;
};
// BATCH: the images of several alpha tuples in one launch (one launch per tuple was ~45 us of host time each: 1.8 ms at the
// tail of a cfg2 fit, with the chip idle) -- blockIdx.y = entry of ``table``: n_folds device pointers (the folds'
// operators for this tuple; ld / map / scale from ``a``) and the slot of the tuple's image in ``out`` / ``rs_inv``.
template <bool BATCH>
__global__ void __launch_bounds__(256) k_mean_operator_image(const MeanOpArgs a, const long long* __restrict__ table, int rows,
                                                             int K, uint4* __restrict__ out, float* __restrict__ rs_inv,
                                                             int rows_pad) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows_pad) return;
    const long long* ent = BATCH ? table + (long long)blockIdx.y * (a.nf + 1) : nullptr;
    if (BATCH) {
        const long long slot = ent[a.nf];
        out += slot * ((long long)rows_pad * K / 4);          // (rows_pad x K x 2 halves per image, 8 halves per uint4)
        rs_inv += slot * rows_pad;
    }
    const int KT = K / TK, units = K / 8;
    const bool live = r < rows;
    float x[MO_UNITS][8];
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < MO_UNITS; ++u) {
        const int c = lane + 64 * u;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[u][j] = 0.f;
        if (!live || c >= units) continue;
        for (int f = 0; f < a.nf; ++f) {
            const int4 m0 = *reinterpret_cast<const int4*>(a.map[f] + c * 8);
            const int4 m1 = *reinterpret_cast<const int4*>(a.map[f] + c * 8 + 4);
            const float* src = (BATCH ? reinterpret_cast<const float*>(ent[f]) : a.m[f]) + (long long)r * a.ld[f];
            const int mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            if (m0.x >= 0 && m1.w == m0.x + 7 && (m0.x & 3) == 0 && (a.ld[f] & 3) == 0) {       // an aligned run of 8 columns
                const float4 v0 = *reinterpret_cast<const float4*>(src + m0.x);
                const float4 v1 = *reinterpret_cast<const float4*>(src + m0.x + 4);
                x[u][0] = __fadd_rn(x[u][0], v0.x); x[u][1] = __fadd_rn(x[u][1], v0.y);
                x[u][2] = __fadd_rn(x[u][2], v0.z); x[u][3] = __fadd_rn(x[u][3], v0.w);
                x[u][4] = __fadd_rn(x[u][4], v1.x); x[u][5] = __fadd_rn(x[u][5], v1.y);
                x[u][6] = __fadd_rn(x[u][6], v1.z); x[u][7] = __fadd_rn(x[u][7], v1.w);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (mm[j] >= 0) x[u][j] = __fadd_rn(x[u][j], src[mm[j]]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[u][j] = __fmul_rn(x[u][j], a.scale);
            mx = fmaxf(mx, fabsf(x[u][j]));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) frexpf(mx, &e);
    e = max(-120, min(120, e));
    const float s = ldexpf(1.f, -e);
    if (lane == 0) rs_inv[r] = ldexpf(1.f, e);
    const long long tile_base = (long long)(r >> 8) * KT * CHUNK16;
    const int rr = r & 255;
#pragma unroll
    for (int u = 0; u < MO_UNITS; ++u) {
        const int c = lane + 64 * u;
        if (c >= units) continue;
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[u][j] * s;
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
        }
        const long long o = tile_base + (long long)(c / KG) * CHUNK16 + (c % KG) * 256 + rr;
        out[o] = *reinterpret_cast<uint4*>(&hi);
        out[o + KG * 256] = *reinterpret_cast<uint4*>(&lo);
    }
}

// Per-column power-of-two scale from max |y| over all T rows: cs[v] = 2^-e, cs[V + v] = 2^e.
// *flag is OR-ed with 1 when most of a FINITE column's entries lie more than 2^9 below its maximum (outliers: the
// 22-bit hi+lo split, whose precision is absolute w.r.t. the column maximum, would then resolve the typical entries
// worse than fp32 does): the host keeps the f32 path in that case.  A column holding a NaN or an Inf does NOT raise
// the flag (round 4; it did before, and one masked-out voxel in 80 000 put the whole fit on the 5x slower f32 path):
// every V-wide kernel keeps a voxel's arithmetic inside its own column, so such a voxel ends where the reference's own
// fp32 arithmetic ends it -- every score NaN -> 0 (ridge_regression.py:133), alpha = alphas[0], non-finite weights,
// r = NaN -> (0, 1) (nested_cv.py:434-436) -- and its neighbours never see it (tests/test_gpu_parity.py).
constexpr int CS_RG = 16, CS_UNROLL = 8;   // row groups per block, rows in flight per thread
__global__ void __launch_bounds__(64 * CS_RG) k_col_scales(const float* __restrict__ y, long long ldy, int T, long long V,
                                                           float* __restrict__ cs, int* __restrict__ flag,
                                                           unsigned char* __restrict__ colflag = nullptr,
                                                           const int* __restrict__ live = nullptr) {
    // live (the refinement's column panel): only the first *live columns, in whole 256-column tiles, hold voxels
    if (live && (long long)blockIdx.x * 64 >= (((long long)*live + 255) & ~255ll)) return;
    __shared__ float sm[CS_RG][64];
    __shared__ int cnt[CS_RG][64];
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
    float mx = 0.f;
    if (c < V) {
        const float* col = y + c;
        // CS_UNROLL independent loads per trip: a column walk with one load in flight ran at 0.6-1.2 TB/s (round 4)
        for (int i0 = ty; i0 < T; i0 += CS_RG * CS_UNROLL) {
            float v[CS_UNROLL];
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u) {
                const int i = i0 + u * CS_RG;
                v[u] = i < T ? col[(long long)i * ldy] : 0.f;
            }
            // the scale comes from the FINITE entries: a voxel with one Inf / NaN sample still has outer folds whose
            // training rows are clean (the reference then chooses a real alpha there), and those must be split well
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u)
                if (fabsf(v[u]) < 3.0e38f) mx = fmaxf(mx, fabsf(v[u]));
        }
    }
    sm[threadIdx.y][threadIdx.x] = mx;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < CS_RG; ++g) mx = fmaxf(mx, sm[g][threadIdx.x]);
    const float small = mx * (1.f / 512.f);
    int n_small = 0;
    if (flag != nullptr) {                 // (block-uniform; without a flag pointer the scales alone: one pass over y)
        if (c < V) {
            const float* col = y + c;
            for (int i0 = ty; i0 < T; i0 += CS_RG * CS_UNROLL) {
                float v[CS_UNROLL];
#pragma unroll
                for (int u = 0; u < CS_UNROLL; ++u) {
                    const int i = i0 + u * CS_RG;
                    v[u] = i < T ? col[(long long)i * ldy] : __builtin_huge_valf();
                }
#pragma unroll
                for (int u = 0; u < CS_UNROLL; ++u) n_small += fabsf(v[u]) < small;
            }
        }
        cnt[threadIdx.y][threadIdx.x] = n_small;
        __syncthreads();
    }
    if (threadIdx.y == 0 && c < V) {
        if (flag != nullptr) {
#pragma unroll
            for (int g = 1; g < CS_RG; ++g) n_small += cnt[g][threadIdx.x];
        }
        int e = 0;
        if (mx > 0.f) frexpf(mx, &e);
        e = max(-120, min(120, e));
        cs[c] = ldexpf(1.f, -e);
        cs[V + c] = ldexpf(1.f, e);
        const bool wide = flag != nullptr && mx > 0.f && 2 * n_small > T;
        if (wide) atomicOr(flag, 1);
        if (colflag != nullptr) colflag[c] = wide ? 1 : 0;       // WHICH columns (round 5: only they leave the fp16 path)
    }
}

// out = c0 T0 + c1 T1 + ... (k_combine_terms' arithmetic: fl32 products and sums, left to right, no contraction) over a
// (rows, ld) matrix, 16 bytes per lane, and -- in the same pass -- the columns' maxima of |out| over the FINITE entries,
// as float bits in colmax (caller-zeroed; atomicMax on the bits of a non-negative float orders like the float): the
// primal form's  B_f = B_all - B_val(f)  needs its column scales before its fp16 image can be written, and a separate
// k_col_scales pass over it was a fifth of the HBM traffic of an inner fold at the LeBel shape.
constexpr int CC_ROWS = 64;                // rows per block: 4 waves x 16 rows, 256 columns
__global__ void __launch_bounds__(256) k_combine_colmax(const float* __restrict__ t0, const float* __restrict__ t1,
                                                        const float* __restrict__ t2, const float* __restrict__ t3,
                                                        float c0, float c1, float c2, float c3, int terms,
                                                        float* __restrict__ out, long long ld, int rows, long long cols4,
                                                        unsigned* __restrict__ colmax) {
    __shared__ float4 sm[4][64];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long c4 = (long long)blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * CC_ROWS, r1 = min(rows, r0 + CC_ROWS);
    float4 mx = {0.f, 0.f, 0.f, 0.f};
    if (c4 < cols4) {
#pragma unroll 4
        for (int r = r0 + w; r < r1; r += 4) {
            const long long o = (long long)r * ld + c4 * 4;
            float4 m = *reinterpret_cast<const float4*>(t0 + o);
            m.x = __fmul_rn(m.x, c0); m.y = __fmul_rn(m.y, c0); m.z = __fmul_rn(m.z, c0); m.w = __fmul_rn(m.w, c0);
            if (terms > 1) {
                const float4 t = *reinterpret_cast<const float4*>(t1 + o);
                m.x = __fadd_rn(m.x, __fmul_rn(t.x, c1)); m.y = __fadd_rn(m.y, __fmul_rn(t.y, c1));
                m.z = __fadd_rn(m.z, __fmul_rn(t.z, c1)); m.w = __fadd_rn(m.w, __fmul_rn(t.w, c1));
            }
            if (terms > 2) {
                const float4 t = *reinterpret_cast<const float4*>(t2 + o);
                m.x = __fadd_rn(m.x, __fmul_rn(t.x, c2)); m.y = __fadd_rn(m.y, __fmul_rn(t.y, c2));
                m.z = __fadd_rn(m.z, __fmul_rn(t.z, c2)); m.w = __fadd_rn(m.w, __fmul_rn(t.w, c2));
            }
            if (terms > 3) {
                const float4 t = *reinterpret_cast<const float4*>(t3 + o);
                m.x = __fadd_rn(m.x, __fmul_rn(t.x, c3)); m.y = __fadd_rn(m.y, __fmul_rn(t.y, c3));
                m.z = __fadd_rn(m.z, __fmul_rn(t.z, c3)); m.w = __fadd_rn(m.w, __fmul_rn(t.w, c3));
            }
            *reinterpret_cast<float4*>(out + o) = m;
            if (fabsf(m.x) < 3.0e38f) mx.x = fmaxf(mx.x, fabsf(m.x));
            if (fabsf(m.y) < 3.0e38f) mx.y = fmaxf(mx.y, fabsf(m.y));
            if (fabsf(m.z) < 3.0e38f) mx.z = fmaxf(mx.z, fabsf(m.z));
            if (fabsf(m.w) < 3.0e38f) mx.w = fmaxf(mx.w, fabsf(m.w));
        }
    }
    if (colmax == nullptr) return;                         // (kernel-uniform)
    sm[w][lane] = mx;
    __syncthreads();
    if (w == 0 && c4 < cols4) {
#pragma unroll
        for (int g = 1; g < 4; ++g) {
            const float4 o = sm[g][lane];
            mx.x = fmaxf(mx.x, o.x); mx.y = fmaxf(mx.y, o.y); mx.z = fmaxf(mx.z, o.z); mx.w = fmaxf(mx.w, o.w);
        }
        unsigned* dst = colmax + c4 * 4;
        if (mx.x > 0.f) atomicMax(dst + 0, __float_as_uint(mx.x));
        if (mx.y > 0.f) atomicMax(dst + 1, __float_as_uint(mx.y));
        if (mx.z > 0.f) atomicMax(dst + 2, __float_as_uint(mx.z));
        if (mx.w > 0.f) atomicMax(dst + 3, __float_as_uint(mx.w));
    }
}

// k_col_scales' scales from such maxima: cs[v] = 2^-e, cs[V + v] = 2^e.
__global__ void __launch_bounds__(256) k_scales_from_max(const unsigned* __restrict__ colmax, long long V,
                                                         float* __restrict__ cs) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= V) return;
    const float mx = __uint_as_float(colmax[c]);
    int e = 0;
    if (mx > 0.f) frexpf(mx, &e);
    e = max(-120, min(120, e));
    cs[c] = ldexpf(1.f, -e);
    cs[V + c] = ldexpf(1.f, e);
}

// Tiled fp16 hi/lo image of Y[rows] (K = padded row count, -1 rows -> 0): thread = (column, 8-row group).
__global__ void __launch_bounds__(256) k_split_cols_f16(const float* __restrict__ y, long long ldy, long long V,
                                                        const int* __restrict__ rows, int K, const float* __restrict__ cs,
                                                        uint4* __restrict__ out, const int* __restrict__ live = nullptr) {
    const int nt = blockIdx.x, g = blockIdx.y;              // g = K-tile * KG + k-group
    if (live && (long long)nt * 256 >= (long long)*live) return;     // (the refinement's panel: no voxel in this column tile)
    const int col = threadIdx.x;
    const long long c = (long long)nt * 256 + col;
    const int KT = K / TK;
    h8 hi, lo;
    if (c < V) {
        const float s = cs[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = rows[g * 8 + j];
            const float x = r >= 0 ? y[(long long)r * ldy + c] * s : 0.f;
            hi[j] = (_Float16)x;
            lo[j] = (_Float16)(x - (float)hi[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; }
    }
    const long long o = ((long long)nt * KT + g / KG) * CHUNK16 + (g % KG) * 256 + col;
    out[o] = *reinterpret_cast<uint4*>(&hi);
    out[o + KG * 256] = *reinterpret_cast<uint4*>(&lo);
}

// The tiled image of the SAME rows with the columns permuted (and -1 entries of the permutation as zero columns): the
// refit contracts the outer training rows of the alpha-SORTED voxels, and the inner CV has already split exactly those
// rows in natural voxel order (one image per outer fold) -- so the sorted operand is a gather of 16-byte units from that
// image (column scales travel with their columns) instead of a 4-byte gather of the fp32 targets, a sorted fp32 copy in
// HBM and a second split pass over it.  Thread = output column, block = R unit rows ((K-tile, plane, k-group)) of a tile.
__global__ void __launch_bounds__(256) k_permute_cols_f16(const uint4* __restrict__ in, const int* __restrict__ perm,
                                                          long long Vs, int rows16, uint4* __restrict__ out) {
    const long long cp = (long long)blockIdx.x * 256 + threadIdx.x;     // output column
    const int c = cp < Vs ? perm[cp] : -1;
    const uint4* src = in + ((long long)(c >> 8) * rows16) * 256 + (c & 255);
    uint4* dst = out + ((long long)blockIdx.x * rows16) * 256 + threadIdx.x;
    const int g0 = blockIdx.y * 8;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c >= 0 && g0 + j < rows16) ? src[(long long)(g0 + j) * 256] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (g0 + j < rows16) dst[(long long)(g0 + j) * 256] = v[j];
}

// ------------------------------------------------------------------ the fused sweep on fp16 x 3
struct Score16Args {
    const float* yv;       // gathered validation targets (M, V), zero padding rows
    const float* ymean;
    const float* rs_inv;   // per H row: 2^e undoing the row pre-scale
    const float* cs_inv;   // per voxel: 2^e undoing the column pre-scale
    float* part;
    long long V;           // padded voxel count of part / scores (multiple of 128)
    int M, n_val, mode, Mrows;
    int A;                 // score mode: alphas in the image, whose 32-row blocks are ordered (validation block, alpha)
    const int* live_cols;  // optional (score / series-moments modes): column tiles from *live_cols on do nothing
};

// Which K-tiles of the tiled B image a launch contracts: the image may hold MORE rows than the product uses (the
// targets of a whole outer training set, split once), and the product skips one aligned gap of it (the validation
// block of the inner fold):  K-tile kt of the product is tile kt + (kt >= cut ? skip : 0) of the image.
struct BView {
    int kt_total;          // K-tiles per column tile of the image
    int cut, skip;
};

// Score and series-moments modes: several inner folds in ONE launch, stacked along the M-tiles -- fold f owns tiles
// [f mt_per_fold, (f + 1) mt_per_fold) of the A image and of the row scales, slice f of the targets / statistics /
// partials, and its own gap of the shared B image.  (The folds of an outer fold are independent; one launch fills the
// chip where five small ones each end in a partial round of workgroups.)
constexpr int MAX_FOLDS16 = 64;
struct FoldViews {
    int mt_per_fold;
    long long part_stride;             // floats per fold in sa.part
    int n_val[MAX_FOLDS16];
    int cut[MAX_FOLDS16], skip[MAX_FOLDS16];
};

// plain (store) mode: C[:, tile] = A_g(tile) . B[:, tile] with one A matrix per column group
constexpr int MAX_GROUPS16 = 64;
struct Plain16Args {
    float* c;              // (Mrows, ldc) f32 output
    long long ldc;
    const float* rs_inv;   // (G * Mtiles * 256)
    const float* cs_inv;   // (Ncols)
    int Mrows;             // real rows per group
    int G;
    int start[MAX_GROUPS16 + 1];   // first 256-column tile of each group; start[G] = number of column tiles
    const unsigned char* slab_light;   // optional, per 128-row slab (G * Mtiles * 2): nonzero = hi*hi term only
    long long col_limit;   // columns >= col_limit are not stored (ldc may then be smaller than the padded column count)
    // PEARSON mode: the Mrows rows are predictions of the test rows; instead of being stored they are reduced, per
    // 128-row slab and column, to (n, mean p, mean y, sum dp^2, sum dy^2, sum dp dy) in fp64 against the test targets
    // y[pr_rows[i], pr_cols[column]] (NULL lists: i / the column itself; pr_cols[j] < 0: no such column)
    const float* pr_y;
    long long pr_ldy;
    const int* pr_rows;
    const int* pr_cols;
    double* pr_part;       // (slabs, 6, Ncols)
    long long pr_ncols;
};

#define MFMA16(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, acc_, 0, 0, 0)

// STAMP builds (diagnostics only, lc_debug_sweep16_stamps): s_memtime at the phase boundaries, per-wave sums of
// the segments of an iteration / of the tile added into pa.c (unsigned long long[2 groups][16]) -- results are not used.
#define STAMP_T(var_)                                                                          \
    if (STAMP) {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }

// SERMOM (with LIGHTCAP, plain operands): the A rows are the four shared series terms of TWO 32-row validation blocks
// per tile -- wave row wm = 0: [T0 b0, T0 b1, T1 b0, T1 b1] (heavy), wm = 1: [T2 b0, T2 b1, T3 b0, T3 b1] (light) --
// and the epilogue reduces them to the blocks' partial moments (lc::epi_series_block) instead of storing them: the two
// waves that share a column panel swap halves through the (then idle) LDS ring, so that each holds all four terms of
// ONE 32-column block for both validation blocks.
// HI2 (score and series-moments modes; "screening" arithmetic): ONE MFMA per product -- the hi planes
// alone, 11-bit operands, fp32 accumulation -- with TWO K-tiles per ring stage and barrier: the slot of a stage that
// holds a K-tile's lo plane in the three-MFMA modes holds the NEXT K-tile's hi plane (same 32 KB stages, same fragment
// addresses, same four DMA pieces per step and thread, 16 MFMAs per wave and step instead of 24 for twice the depth).
template <bool SCORE, bool STAMP, bool LIGHTCAP = false, bool SERMOM = false, bool PEARSON = false, bool HI2 = false>
__global__ void __launch_bounds__(512, 2)
k_sweep_f16x3(const uint4* __restrict__ At, const uint4* __restrict__ Bt, int KT_tiles, int Mtiles, Score16Args sa,
              Plain16Args pa, BView bv, FoldViews fv) {
    static_assert(!HI2 || ((SCORE || SERMOM) && !PEARSON), "HI2 is a mode of the score / series-moments kernels");
    const int KT = HI2 ? KT_tiles / 2 : KT_tiles;       // ring steps (HI2: two K-tiles each; the host checks K % 64 == 0)
    extern __shared__ __attribute__((aligned(16))) uint4 lds16[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform (scalar)
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    const int tile = xcd_tile_id16(blockIdx.x, gridDim.x);
    const int mt_all = tile % Mtiles, nt = tile / Mtiles;
    constexpr bool FOLDS = SCORE || SERMOM;
    // the refinement's column panel: its capacity is fixed when the launch is queued, how many columns it
    // holds only the device knows -- the tiles behind them leave at once (block-uniform: before any barrier)
    if (FOLDS && sa.live_cols != nullptr && (long long)nt * TN >= (long long)*sa.live_cols) return;
    const int fold = FOLDS ? mt_all / fv.mt_per_fold : 0;
    const int mt = FOLDS ? mt_all - fold * fv.mt_per_fold : mt_all;          // M-tile inside the fold
    if (FOLDS) {
        sa.yv += (long long)fold * sa.M * sa.V;
        sa.ymean += (long long)fold * 3 * sa.V;
        sa.part += (long long)fold * fv.part_stride;
        sa.n_val = fv.n_val[fold];
    }
    const int b_cut = FOLDS ? fv.cut[fold] : bv.cut, b_skip = FOLDS ? fv.skip[fold] : bv.skip;
    int grp = 0;
    if (!SCORE) {
        while (grp + 1 < pa.G && nt >= pa.start[grp + 1]) ++grp;
    }
    // debug builds only (EXPK, tools/debug_kernels): experiment bits in pa.G, which the score modes do not use -- 1: the HI2 step
    // fetches 16 KB contiguous per operand (the hi AND lo plane of K-tile t: wrong numbers, the same bytes) instead of two
    // 8 KB hi planes 16 KB apart; 2: no MFMAs, no fragment reads (every wave only keeps the ring going: what the operand
    // delivery alone takes); 4: every workgroup fetches tile (0, 0) (everything hits L2)
    const int exp_bits = (EXPK && SCORE) ? (LC_SWEEP_EXP_BITS >= 0 ? LC_SWEEP_EXP_BITS : pa.G) : 0;
    const bool exp_contig = exp_bits & 1, exp_dma_only = exp_bits & 2, exp_same = exp_bits & 4, exp_no_dma = exp_bits & 8;
    // 8: no DMA in the main loop (MFMAs and fragment reads alone); 16: every tile starts its K loop at another K-tile and
    // wraps (the workgroups of a round do not walk their slabs at the same offsets)
    const bool exp_no_reads = exp_bits & 64, exp_no_mfma = exp_bits & 128;   // 64: no fragment reads; 128: no MFMAs (reads + DMA)
    const bool exp_dword = exp_bits & 32;          // 32: every 1 KB LDS-DMA piece as four global_load_lds_dword of 256 B
    // 256: every workgroup fetches a private 2 x 32 KB region over and over (L2-resident, spread over the channels)
    const bool exp_private = exp_bits & 256;
    const int exp_rot = (EXPK && (exp_bits & 16)) ? (int)(((long long)(tile / Mtiles) * 13 + (tile % Mtiles) * 7) % KT) : 0;
    const uint4* a_src = At + ((long long)grp * Mtiles + (exp_same ? 0 : mt_all)) * KT_tiles * CHUNK16 + tid;
    const uint4* b_src = Bt + (long long)(exp_same ? 0 : nt) * bv.kt_total * CHUNK16 + tid;
    if (EXPK && exp_private) {
        a_src = Bt + (long long)(blockIdx.x & 255) * 4 * CHUNK16 + tid;
        b_src = Bt + (long long)(256 + (blockIdx.x & 255)) * 4 * CHUNK16 + tid;
    }
#define BKT(kt_) ((kt_) + ((kt_) >= b_cut ? b_skip : 0))
    // "light" slabs (plain mode): rows whose product only needs fp16 accuracy (11-bit operands) -- the higher
    // terms of a series, which enter the caller's result scaled down by >= 2^-11 -- take the hi*hi MFMA alone
    const bool light = HI2 ? false : SERMOM ? wm != 0
                              : (LIGHTCAP && !SCORE && pa.slab_light != nullptr &&
                                 __builtin_amdgcn_readfirstlane((int)pa.slab_light[((long long)grp * Mtiles + mt_all) * 2 + wm]) != 0);

    // ---- main loop: software-pipelined fragments, ONE block barrier per K-tile ----------------------------
    // Every wave keeps two register sets of fragments: while the 24 MFMAs of K-tile j run on one set, the 12
    // LDS reads of tile j+1 fill the other and the four LDS-DMA pieces of tile j+4 are started, all interleaved
    // between the MFMAs -- each wave always has independent MFMAs to issue, so the two waves of a SIMD keep the
    // matrix pipe fed without any phase choreography, and the only bubble left is the barrier crossing.
    // Operand chunks go global -> LDS directly (global_load_lds_dwordx4 through inline asm, so that hipcc does
    // not drain the queue in front of every LDS read; M0 = the wave's LDS base, written in the same statement with the
    // ONE WAIT STATE the hardware needs between an SALU write of M0 and an LDS-DMA reading it -- hipcc pads nothing
    // inside an asm string; without the s_nop a piece now and then landed at the PREVIOUS piece's address whenever
    // another stream's waves shared the SIMD: one stale 64-column slice of one K-tile in ~1 fit of 10, round 3 --
    // and with the compiler's own M0 saved and restored around it); thread t moves the 16-byte units t and t + 512 of each
    // 16 KB chunk, a wave's 64 lanes land contiguously at its wave-uniform LDS base (M0).  Ring discipline
    // (tile t lives in stage t & 3):
    //   prologue: tiles 0..3 -> stages 0..3, fragments of tile 0 -> registers, barrier;
    //   iteration j: MFMAs of tile j (registers) | read tile j+1 from stage (j+1)&3 | DMA tile j+4 -> stage j&3
    //   (every wave read tile j out of it during iteration j-1, i.e. before the last barrier);
    //   end of iteration j: wait (counted vmcnt: tiles j+3, j+4 may still fly) until this wave's share of tile
    //   j+2 has landed, then the barrier publishes it -- two iterations of latency budget per DMA.
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) uint4*)lds16);
#define DMA16_X4(gptr_, unit_)                                                                                \
    {                                                                                                         \
        const unsigned m0_ = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((unit_) + (wave << 6)) * 16u); \
        const uint4* gp_ = (gptr_);                                                                           \
        unsigned keep_;                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(gp_), "s"(m0_) : "memory");                                          \
    }
    // (STAMP experiment: the same 1 KB piece as four 256-byte pieces, one dword per lane)
#define DMA16_D1Q(gp_, m0_)                                                                                   \
    {                                                                                                         \
        unsigned keep_;                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(gp_), "s"(m0_) : "memory");                                          \
    }
#define DMA16_D1(gptr_, unit_)                                                                                \
    {                                                                                                         \
        const unsigned m0b_ = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((unit_) + (wave << 6)) * 16u); \
        const char* gpb_ = reinterpret_cast<const char*>(gptr_) - lane * 12;                                  \
        DMA16_D1Q(gpb_, m0b_);                                                                                \
        DMA16_D1Q(gpb_ + 256, m0b_ + 256u);                                                                   \
        DMA16_D1Q(gpb_ + 512, m0b_ + 512u);                                                                   \
        DMA16_D1Q(gpb_ + 768, m0b_ + 768u);                                                                   \
    }
#define DMA16(gptr_, unit_)                                   \
    { if (EXPK && exp_dword) DMA16_D1(gptr_, unit_) else DMA16_X4(gptr_, unit_) }
    // the two pieces of an operand's share of ring step t: hi and lo plane of K-tile t, or (HI2) the hi planes of the
    // K-tiles 2t and 2t + 1 (a thread's unit of a plane: + tid, in a_src / b_src already; the lo plane follows 512 units on)
#define HI2_STRIDED (HI2 && !(EXPK && exp_contig))
#define ROT_T(t_) (EXPK ? (exp_private ? ((t_) & 1) : ((t_) + exp_rot) >= KT ? (t_) + exp_rot - KT : (t_) + exp_rot) : (t_))
#define SRC_A0(t_) (a_src + (long long)(HI2_STRIDED ? 2 * ROT_T(t_) : ROT_T(t_)) * CHUNK16)
#define SRC_A1(t_) (HI2_STRIDED ? a_src + (long long)(2 * ROT_T(t_) + 1) * CHUNK16 : a_src + (long long)ROT_T(t_) * CHUNK16 + 512)
#define SRC_B0(t_) (b_src + (long long)BKT(HI2_STRIDED ? 2 * ROT_T(t_) : ROT_T(t_)) * CHUNK16)
#define SRC_B1(t_) (HI2_STRIDED ? b_src + (long long)BKT(2 * ROT_T(t_) + 1) * CHUNK16 : b_src + (long long)BKT(ROT_T(t_)) * CHUNK16 + 512)
#define GLDS16(kt_, stg_)                                                           \
    {                                                                               \
        DMA16(SRC_A0(kt_), (stg_) * STAGE16);                                       \
        DMA16(SRC_A1(kt_), (stg_) * STAGE16 + 512);                                 \
        DMA16(SRC_B0(kt_), (stg_) * STAGE16 + CHUNK16);                             \
        DMA16(SRC_B1(kt_), (stg_) * STAGE16 + CHUNK16 + 512);                       \
    }
#define PHASE_BARRIER()                        \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0)

    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
    STAMP_T(tk0);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < NSTAGE && t < KT; ++t) GLDS16(t, t);

    // score epilogue operands that do not depend on the accumulators are fetched up front: the epilogue has
    // nothing to hide a global round trip behind (one block per CU).  Row scales of the tile -> LDS behind the
    // ring, per-column constants -> registers; the first target batches follow in the last K-tiles (below).
    float* lds_rs = reinterpret_cast<float*>(lds16 + NSTAGE * STAGE16);
    const long long V = sa.V;
    const long long col0 = (long long)nt * TN + wn * 64 + li;
    const bool cok[2] = {col0 < V, col0 + 32 < V};
    const long long colc[2] = {cok[0] ? col0 : 0, cok[1] ? col0 + 32 : 0};       // clamped: loads stay in range
    float ymv[2] = {0.f, 0.f}, cscv[2] = {0.f, 0.f};
    if (SERMOM) {                                    // this wave's column block after the swap: ni = wm
        if (tid < TM) lds_rs[tid] = pa.rs_inv[((long long)grp * Mtiles + mt_all) * TM + tid];
        ymv[0] = sa.ymean[colc[wm]];
        cscv[0] = pa.cs_inv[(long long)nt * TN + wn * 64 + wm * 32 + li];
    }
    if (SCORE) {
        if (tid < TM) lds_rs[tid] = sa.rs_inv[mt_all * TM + tid];                  // rs_inv has rows_pad entries per fold
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            ymv[ni] = sa.ymean[colc[ni]];
            cscv[ni] = sa.cs_inv[colc[ni]];
        }
    }
    // targets of epilogue step s = 2 mi + ni (32 rows x this lane's column of panel ni)
    // Epilogue targets.  A wave's four 32-row blocks mi = 0..3 are image blocks mt 8 + wm 4 + mi = (validation block, alpha)
    // pairs in that order, so consecutive mi mostly belong to the SAME validation block (always, with the fit's four
    // factorised alphas): its targets -- 32 rows x this lane's column of the two panels -- are loaded once and kept (round 6;
    // they used to be loaded again for every alpha: eight batches per wave where two do), the next block's only when it changes.
    auto val_block = [&](int mi) {
        // (blocks past the image's last one -- the padding of its last tile -- take the last validation block: unused)
        return min((mt * (TM / 32) + wm * 4 + mi) / sa.A, (sa.M >> 5) - 1);
    };
    auto load_vb = [&](int ib, int ni, lc::EpiTargets& t) { lc::epi_load_targets(sa.yv, V, ib * 32, lh, colc[ni], t); };
    lc::EpiTargets tc0, tc1;                         // the current validation block's targets, panels 0 and 1

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PHASE_BARRIER();

    // fragment addresses (16-byte units) inside a stage: ((plane*KG + lh) * 256 + row)
    const int a_frag = lh * 256 + wm * 128 + li;
    const int b_frag = CHUNK16 + lh * 256 + wn * 64 + li;
    struct Frag {
        h8 ah[4], al[4], bh[2], bl[2];
    };
    // fragment k of the 12 of a stage, in the order the MFMAs consume them (lo*hi terms first)
    auto read_frag = [&](Frag& f, const uint4* st, const int k) {
        if (k < 2) {
            const uint4 v = st[b_frag + k * 32];
            f.bh[k] = *reinterpret_cast<const h8*>(&v);
        } else if (k < 6) {
            const uint4 v = st[a_frag + KG * 256 + (k - 2) * 32];
            f.al[k - 2] = *reinterpret_cast<const h8*>(&v);
        } else if (k < 8) {
            const uint4 v = st[b_frag + KG * 256 + (k - 6) * 32];
            f.bl[k - 6] = *reinterpret_cast<const h8*>(&v);
        } else {
            const uint4 v = st[a_frag + (k - 8) * 32];
            f.ah[k - 8] = *reinterpret_cast<const h8*>(&v);
        }
    };
    Frag fa, fb;
#pragma unroll
    for (int k = 0; k < 12; ++k) read_frag(fa, lds16, k);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PHASE_BARRIER();                             // everybody holds tile 0: stage 0 may be overwritten
    STAMP_T(tk1);
    unsigned long long tr1 = 0, tr2 = 0;         // 100 MHz wall clock around the main loop: in-kernel shader clock
    if (STAMP) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr1)::"memory");

    // one K-tile.  MODE 0 = steady (kt + 4 < KT): no branches between the MFMAs, where an instruction-fetch
    // hiccup is a bubble in the matrix pipe;  1 = the guarded version for the last few tiles;  2 = the last
    // tile (no next fragments, no DMA: its free registers take the first target batches of the epilogue).
    auto kstep = [&](const int kt, const Frag& cur, Frag& nxt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool STEADY = MODE == 0, LAST = MODE == 2;
        const bool has_next = STEADY || (!LAST && kt + 1 < KT);
        const bool do_dma = STEADY || (!LAST && kt + 4 < KT);
        const uint4* stn = lds16 + ((kt + 1) & 3) * STAGE16;
        const int stg = kt & 3;
        const uint4* pa_ = a_src + (long long)(kt + 4) * CHUNK16;
        const uint4* pb_ = b_src + (long long)BKT(kt + 4) * CHUNK16;
        if (LAST && SCORE) { load_vb(val_block(0), 0, tc0); load_vb(val_block(0), 1, tc1); }   // land under the MFMAs of the last tile
        // 12 slots of two MFMAs, term-major (the eight accumulators take the lo*hi terms, then hi*lo, then
        // hi*hi: small terms first, and consecutive MFMAs never wait for each other's result)
#pragma unroll
        for (int sl = 0; sl < 12; ++sl) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * sl + h, term = i >> 3, mi = (i & 7) >> 1, ni = i & 1;
                if (term == 0) MFMA16(acc[mi][ni], cur.al[mi], cur.bh[ni]);
                if (term == 1) MFMA16(acc[mi][ni], cur.ah[mi], cur.bl[ni]);
                if (term == 2) MFMA16(acc[mi][ni], cur.ah[mi], cur.bh[ni]);
            }
            if (has_next) read_frag(nxt, stn, sl);
            if (sl % 3 == 1 && do_dma) {
                if (sl == 1) DMA16(pa_, stg * STAGE16);
                if (sl == 4) DMA16(pa_ + 512, stg * STAGE16 + 512);
                if (sl == 7) DMA16(pb_, stg * STAGE16 + CHUNK16);
                if (sl == 10) DMA16(pb_ + 512, stg * STAGE16 + CHUNK16 + 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LAST) return;
        // publish tile kt+2 (read during iteration kt+1); nothing left to publish in the last two iterations.
        // lgkmcnt(0): this wave's fragment reads of tile kt+1 have RETURNED before it reports at the barrier -- the
        // first DMA piece of the next iteration goes into the stage they came from, and "issued before the barrier" is not
        // "done": the protocol used to lean on a DMA taking longer to land (>= 250 cycles) than a queued ds_read to return,
        // which a second workgroup on the CU (any small LDS-using kernel of another stream fits beside this one) breaks
        // now and then -- one wave then multiplied 32 rows of one K-tile with the NEXT ring turn's bytes (round 3)
        if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (STEADY || kt + 2 < KT) { PHASE_BARRIER(); }
    };
    // the same K-tile for a light slab: 8 MFMAs (hi*hi), the 6 hi fragments of the next tile, the same DMA share
    auto kstep_light = [&](const int kt, const Frag& cur, Frag& nxt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool STEADY = MODE == 0, LAST = MODE == 2;
        const bool has_next = STEADY || (!LAST && kt + 1 < KT);
        const bool do_dma = STEADY || (!LAST && kt + 4 < KT);
        const uint4* stn = lds16 + ((kt + 1) & 3) * STAGE16;
        const int stg = kt & 3;
        const uint4* pa_ = a_src + (long long)(kt + 4) * CHUNK16;
        const uint4* pb_ = b_src + (long long)BKT(kt + 4) * CHUNK16;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
            for (int h = 0; h < 2; ++h) MFMA16(acc[sl][h], cur.ah[sl], cur.bh[h]);
            if (has_next) {
                if (sl < 2) read_frag(nxt, stn, sl);              // bh[0], bh[1]
                read_frag(nxt, stn, 8 + sl);                      // ah[sl]
            }
            if (do_dma) {
                if (sl == 0) DMA16(pa_, stg * STAGE16);
                if (sl == 1) DMA16(pa_ + 512, stg * STAGE16 + 512);
                if (sl == 2) DMA16(pb_, stg * STAGE16 + CHUNK16);
                if (sl == 3) DMA16(pb_ + 512, stg * STAGE16 + CHUNK16 + 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LAST) return;
        if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (STEADY || kt + 2 < KT) { PHASE_BARRIER(); }
    };
    // HI2: one ring step = two K-tiles, hi planes only.  The fragments sit where the three-MFMA step finds them -- "ah / bh"
    // are K-tile 2 kt, "al / bl" K-tile 2 kt + 1 -- so the step is the heavy one without its cross terms: 16 MFMAs in 8
    // slots, the 12 fragment reads of the next step and the four DMA pieces of step kt + 4 spread over them.
    auto kstep_hi2 = [&](const int kt, const Frag& cur, Frag& nxt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool STEADY = MODE == 0, LAST = MODE == 2;
        const bool has_next = STEADY || (!LAST && kt + 1 < KT);
        const bool do_dma = (STEADY || (!LAST && kt + 4 < KT)) && !(EXPK && exp_no_dma);
        const uint4* stn = lds16 + ((kt + 1) & 3) * STAGE16;
        const int stg = kt & 3;
        if (LAST && SCORE) { load_vb(val_block(0), 0, tc0); load_vb(val_block(0), 1, tc1); }
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            if (!(EXPK && exp_no_mfma))
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * sl + h, mi = (i & 7) >> 1, ni = i & 1;
                if (i < 8) MFMA16(acc[mi][ni], cur.ah[mi], cur.bh[ni]);
                else MFMA16(acc[mi][ni], cur.al[mi], cur.bl[ni]);
            }
            if (has_next && !(EXPK && exp_no_reads)) {
                // reads in the order the next step consumes them: bh, ah (K-tile 2 kt' first), then bl, al
                if (sl == 0) { read_frag(nxt, stn, 0); read_frag(nxt, stn, 1); }
                if (sl == 1) { read_frag(nxt, stn, 8); read_frag(nxt, stn, 9); }
                if (sl == 2) { read_frag(nxt, stn, 10); read_frag(nxt, stn, 11); }
                if (sl == 3) { read_frag(nxt, stn, 6); read_frag(nxt, stn, 7); }
                if (sl >= 4) read_frag(nxt, stn, sl - 2);
            }
            if (do_dma) {
                if (sl == 1) DMA16(SRC_A0(kt + 4), stg * STAGE16);
                if (sl == 3) DMA16(SRC_A1(kt + 4), stg * STAGE16 + 512);
                if (sl == 5) DMA16(SRC_B0(kt + 4), stg * STAGE16 + CHUNK16);
                if (sl == 7) DMA16(SRC_B1(kt + 4), stg * STAGE16 + CHUNK16 + 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LAST) return;
        if (EXPK && exp_dword) {
            if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
            else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (STEADY || kt + 2 < KT) { PHASE_BARRIER(); }
    };
    // a 128-row slab with no valid rows (the last M-tile of A x 480 = 1920 score rows is half padding): its waves only
    // keep up their share of the DMA ring and the barriers; the other wave of each SIMD then has the matrix pipe to
    // itself and the tile takes half the time -- 1/16 of a fused launch
    auto kstep_empty = [&](const int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool STEADY = MODE == 0, LAST = MODE == 2;
        if (STEADY || (!LAST && kt + 4 < KT)) {
            const int stg = kt & 3;
            DMA16(SRC_A0(kt + 4), stg * STAGE16);
            DMA16(SRC_A1(kt + 4), stg * STAGE16 + 512);
            DMA16(SRC_B0(kt + 4), stg * STAGE16 + CHUNK16);
            DMA16(SRC_B1(kt + 4), stg * STAGE16 + CHUNK16 + 512);
        }
        if (LAST) return;
        if (EXPK && exp_dword) {
            if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STEADY || kt + 2 < KT) { PHASE_BARRIER(); }
    };
    using Steady = std::integral_constant<int, 0>;
    using Tail = std::integral_constant<int, 1>;
    using Last = std::integral_constant<int, 2>;
    const bool slab_empty = (EXPK && exp_dma_only) || (!SERMOM && !STAMP && mt * TM + wm * 128 >= (SCORE ? sa.Mrows : pa.Mrows));
    int kt = 0;
    if (EXPK && exp_no_reads) fb = fa;
    if (HI2 && !slab_empty) {
        for (; kt + 5 < KT; kt += 2) {
            kstep_hi2(kt, fa, fb, Steady{});
            kstep_hi2(kt + 1, fb, fa, Steady{});
        }
        for (; kt + 2 < KT; kt += 2) {
            kstep_hi2(kt, fa, fb, Tail{});
            kstep_hi2(kt + 1, fb, fa, Tail{});
        }
        kstep_hi2(kt, fa, fb, Tail{});               // KT (steps) is even: K % 64 == 0
        kstep_hi2(kt + 1, fb, fa, Last{});
    } else if (slab_empty) {
        for (; kt + 5 < KT; kt += 2) {
            kstep_empty(kt, Steady{});
            kstep_empty(kt + 1, Steady{});
        }
        for (; kt + 2 < KT; kt += 2) {
            kstep_empty(kt, Tail{});
            kstep_empty(kt + 1, Tail{});
        }
        kstep_empty(kt, Tail{});
        kstep_empty(kt + 1, Last{});
    } else if (LIGHTCAP && light) {
        for (; kt + 5 < KT; kt += 2) {
            kstep_light(kt, fa, fb, Steady{});
            kstep_light(kt + 1, fb, fa, Steady{});
        }
        for (; kt + 2 < KT; kt += 2) {
            kstep_light(kt, fa, fb, Tail{});
            kstep_light(kt + 1, fb, fa, Tail{});
        }
        kstep_light(kt, fa, fb, Tail{});
        kstep_light(kt + 1, fb, fa, Last{});
    } else {
        for (; kt + 5 < KT; kt += 2) {
            kstep(kt, fa, fb, Steady{});
            kstep(kt + 1, fb, fa, Steady{});
        }
        for (; kt + 2 < KT; kt += 2) {
            kstep(kt, fa, fb, Tail{});
            kstep(kt + 1, fb, fa, Tail{});
        }
        kstep(kt, fa, fb, Tail{});                   // KT is even (K % 32 == 0): kt == KT - 2 here
        kstep(kt + 1, fb, fa, Last{});
    }
    STAMP_T(tk2);
    if (STAMP) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr2)::"memory");
        if (lane == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(pa.c) + wm * 16;
            atomicAdd(dbg + 6, tk1 - tk0);
            atomicAdd(dbg + 0, tk2 - tk1);                      // main loop, shader cycles
            atomicAdd(dbg + 14, tr2 - tr1);                     // main loop, 100 MHz ticks
            atomicAdd(dbg + 5, (unsigned long long)KT);
        }
    }
    unsigned long long te0 = 0, te1 = 0, te2 = 0, te3 = 0;
    STAMP_T(te0);

    if (SERMOM) {
        // ---- series-moments epilogue
        const long long colm = colc[wm];
        const int blk0 = 2 * mt;
        lc::EpiTargets tg[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if ((blk0 + b) * 32 < sa.M) lc::epi_load_targets(sa.yv, V, (blk0 + b) * 32, lh, colm, tg[b]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PHASE_BARRIER();                             // every wave is done with the ring: it becomes the swap buffer
        lc::ep_f32x4* xch = reinterpret_cast<lc::ep_f32x4*>(lds16);
        {
            lc::ep_f32x4* dst = xch + (wn * 2 + wm) * 16 * 64 + lane;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const lc::ep_f32x4 rs = *reinterpret_cast<const lc::ep_f32x4*>(lds_rs + wm * 128 + mi * 32 + 8 * q + 4 * lh);
                    lc::ep_f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (wm ? acc[mi][0][4 * q + j] : acc[mi][1][4 * q + j]) * rs[j];
                    dst[(mi * 4 + q) * 64] = v;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PHASE_BARRIER();
        const lc::ep_f32x4* src = xch + (wn * 2 + (1 - wm)) * 16 * 64 + lane;
        const float cs = cscv[0], ym = ymv[0];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int i0 = (blk0 + b) * 32;
            if (i0 >= sa.n_val) continue;
            float own[2][16], oth[2][16];            // [term of the pair][row]
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mi = 2 * h + b;
                    const lc::ep_f32x4 rs = *reinterpret_cast<const lc::ep_f32x4*>(lds_rs + wm * 128 + mi * 32 + 8 * q + 4 * lh);
                    const lc::ep_f32x4 o = src[(mi * 4 + q) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        own[h][4 * q + j] = (wm ? acc[mi][1][4 * q + j] : acc[mi][0][4 * q + j]) * rs[j] * cs;
                        oth[h][4 * q + j] = o[j] * cs;
                    }
                }
            float T[4][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                T[0][r] = wm ? oth[0][r] : own[0][r];
                T[1][r] = wm ? oth[1][r] : own[1][r];
                T[2][r] = wm ? own[0][r] : oth[0][r];
                T[3][r] = wm ? own[1][r] : oth[1][r];
            }
            float* dstp = sa.part + (long long)(blk0 + b) * lc::EPI_SERIES_PARTS * V + colm;
            if (i0 + 32 <= sa.n_val) lc::epi_series_block<false>(T, tg[b], ym, i0, sa.n_val, lh, dstp, V, cok[wm]);
            else lc::epi_series_block<true>(T, tg[b], ym, i0, sa.n_val, lh, dstp, V, cok[wm]);
        }
        return;
    }

    if (PEARSON) {
        // ---- Pearson epilogue (the test rows of the refit, nested_cv.py:151-155, 251-257): the predictions
        // fl32(acc * row scale * column scale) -- the values the plain epilogue would store -- never leave the registers.
        // A lane holds 64 rows of a column (two columns): it sums p - p0, y - y0 and their products in fp64 about ITS first
        // row's values (a shift by a sample of the same column: the centred sums that follow lose two digits of sixteen at
        // most, and a constant column gives exact zeros), turns them into (n, means, centred sums), merges with the lane
        // that holds the other rows of the 128-row slab, and k_pearson_from_parts merges the slabs -- all by the pairwise
        // update formulas.  One pass, no copy of the targets in registers (the accumulators leave room for little else).
        const int colw = wn * 64 + li;
        const int rbase = mt * TM + wm * 128;
        if (rbase >= pa.Mrows) return;                       // a slab past the last test row: no partial of its own
        const float* rsp = pa.rs_inv + (long long)grp * Mtiles * TM;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            __builtin_amdgcn_sched_barrier(0);
            const long long cj = (long long)nt * TN + colw + ni * 32;
            const int src_c = cj < pa.col_limit ? (pa.pr_cols ? pa.pr_cols[cj] : (int)cj) : -1;
            const float csc = pa.cs_inv[cj];
            const float* ycol = pa.pr_y + (src_c >= 0 ? src_c : 0);
            double sa = 0.0, sb = 0.0, qa = 0.0, qb = 0.0, qab = 0.0, p0 = 0.0, y0 = 0.0;
            int cnt = 0;
#pragma unroll
            for (int h8 = 0; h8 < 8; ++h8) {               // eight rows at a time (register room)
                const int mi = h8 >> 1, rr0 = (h8 & 1) * 8;
                float pv[8], yv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = rr0 + k;
                    const int row = rbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    pv[k] = acc[mi][ni][r] * rsp[row] * csc;
                    const bool ok = row < pa.Mrows && src_c >= 0;
                    const long long yr = ok ? (pa.pr_rows ? (long long)pa.pr_rows[row] : (long long)row) : -1;
                    yv[k] = yr >= 0 ? ycol[yr * pa.pr_ldy] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = rr0 + k;
                    const int row = rbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < pa.Mrows && src_c >= 0) {
                        if (cnt == 0) { p0 = (double)pv[k]; y0 = (double)yv[k]; }
                        const double da = (double)pv[k] - p0, db = (double)yv[k] - y0;
                        sa += da; sb += db;
                        qa += da * da; qb += db * db; qab += da * db;
                        ++cnt;
                    }
                }
            }
            // this lane's rows: count, means, centred sums
            double n = (double)cnt;
            const double inv = cnt > 0 ? 1.0 / n : 0.0;
            double ma = p0 + sa * inv, mb = y0 + sb * inv;
            qa -= sa * sa * inv;
            qb -= sb * sb * inv;
            qab -= sa * sb * inv;
            // ... merged with the lane that holds the slab's other rows of this column (lh ^ 1)
            const double n2 = __shfl_xor(n, 32), ma2 = __shfl_xor(ma, 32), mb2 = __shfl_xor(mb, 32);
            const double qa2 = __shfl_xor(qa, 32), qb2 = __shfl_xor(qb, 32), qab2 = __shfl_xor(qab, 32);
            if (lh == 0 && cj < pa.pr_ncols) {
                if (n2 > 0.0) {
                    if (n > 0.0) {
                        const double tot = n + n2, da = ma2 - ma, db = mb2 - mb, w = n * n2 / tot;
                        qa += qa2 + da * da * w;
                        qb += qb2 + db * db * w;
                        qab += qab2 + da * db * w;
                        ma += da * (n2 / tot);
                        mb += db * (n2 / tot);
                        n = tot;
                    } else {
                        n = n2; ma = ma2; mb = mb2; qa = qa2; qb = qb2; qab = qab2;
                    }
                }
                double* dst = pa.pr_part + ((long long)(mt * 2 + wm) * 6) * pa.pr_ncols + cj;
                dst[0] = n;
                dst[pa.pr_ncols] = ma;
                dst[2 * pa.pr_ncols] = mb;
                dst[3 * pa.pr_ncols] = qa;
                dst[4 * pa.pr_ncols] = qb;
                dst[5 * pa.pr_ncols] = qab;
            }
        }
        return;
    }

    if (!SCORE) {
        // ---- plain epilogue: undo the power-of-two scales and store
        const int colw = wn * 64 + li;                       // column inside the 256-wide tile (+ 32 ni)
        float* cbase = pa.c + (long long)nt * TN;
        const long long rstride = pa.ldc;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int rb0 = mt * TM + wm * 128 + mi * 32;
            float rsc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rsc[r] = pa.rs_inv[(long long)grp * Mtiles * TM + rb0 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float csc = pa.cs_inv[(long long)nt * TN + colw + ni * 32];
                const bool col_ok = (long long)nt * TN + colw + ni * 32 < pa.col_limit;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < pa.Mrows && col_ok) cbase[(long long)row * rstride + colw + ni * 32] = acc[mi][ni][r] * rsc[r] * csc;
                }
            }
        }
        return;
    }

    // ---- epilogue: the statistics of lc_epilogue.h after undoing the power-of-two scales.  Two target
    // batches are already in registers, the third is issued first thing; batch s+3 follows step s.
    const bool corr = sa.mode == LC_SCORE_CORR;
    auto reduce = [&](int step, const lc::EpiTargets& t) {
        const int mi = step >> 1, ni = step & 1;
        const int rb0 = mt * TM + wm * 128 + mi * 32;
        if (rb0 >= sa.Mrows) return;
        lc::ep_f32x4 rs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rs[q] = *reinterpret_cast<const lc::ep_f32x4*>(lds_rs + wm * 128 + mi * 32 + 8 * q + 4 * lh);
        const int blk = rb0 >> 5, ib = blk / sa.A, al = blk - ib * sa.A;            // partials stay alpha-major
        lc::epi_block_dispatch<true>(corr, acc[mi][ni], t, rs, cscv[ni], ymv[ni], ib * 32, sa.n_val, lh,
                                     sa.part + (long long)(al * (sa.M >> 5) + ib) * 4 * V + colc[ni], V, cok[ni]);
    };
#define EPI_FENCE() __builtin_amdgcn_sched_barrier(0)
    unsigned long long ter = 0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        reduce(2 * mi, tc0); EPI_FENCE();
        reduce(2 * mi + 1, tc1); EPI_FENCE();
        if (mi < 3) {
            const int ib_next = val_block(mi + 1);
            if (ib_next != val_block(mi)) {                      // (wave-uniform; never with the fit's four factorised alphas)
                load_vb(ib_next, 0, tc0);
                load_vb(ib_next, 1, tc1);
            }
        }
        EPI_FENCE();
        if (mi == 0) { STAMP_T(ter); STAMP_T(te1); }
        if (mi == 2) { STAMP_T(te2); }
    }
    STAMP_T(te3);
#undef EPI_FENCE
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP_T(tk3);
        if (lane == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(pa.c) + wm * 16;
            atomicAdd(dbg + 7, tk3 - tk2);
            atomicAdd(dbg + 8, te0 - tk2);       // group-0 catch-up barrier
            atomicAdd(dbg + 9, ter - te0);       // block mi = 0 (two reduce steps, cold)
            atomicAdd(dbg + 13, te1 - ter);      // (nothing between: the repeated step of rounds 3-5 is gone)
            atomicAdd(dbg + 10, te2 - te1);      // blocks mi = 1, 2
            atomicAdd(dbg + 11, te3 - te2);      // block mi = 3
            atomicAdd(dbg + 12, tk3 - te3);      // store drain
        }
    }
}

// ------------------------------------------------------------------ screening sweeps, two workgroups per CU (round 6)
// The HI2 modes of k_sweep_f16x3 (one MFMA per product) spend a third of a tile outside the main loop -- 8.6 k cycles of
// prologue (the ring's first fill), 32-37 k of epilogue (target batches, packed statistics, partial stores: latency, not
// work) against ~78 k of main loop -- with the matrix pipe idle, because the ONE workgroup of a CU is in the same phase on
// all four SIMDs.  This kernel is the same contraction in workgroups of FOUR waves (2 x 2, wave tile 128 x 64 as before)
// on 256 x 128 tiles -- one M-tile of the operator image against HALF a column tile of the target image (same images,
// no new layout) -- with a ring of THREE 24 KB stages (72 KB + the row scales: two workgroups per CU, one wave of each
// per SIMD, <= 256 VGPRs): whatever one workgroup waits for -- its barrier, its first fill, its epilogue's loads -- the
// other's MFMAs run under.  Ring discipline (step t = K-tiles 2t, 2t+1 lives in stage t % 3):
//   prologue: steps 0..2 -> stages 0..2; fragments of step 0 -> registers; barrier;
//   iteration j: MFMAs of step j (registers) | read step j+1 from stage (j+1) % 3 | DMA step j+3 -> stage j % 3 (read out
//   during iteration j-1, before the last barrier);  end of iteration j: wait until this wave's pieces of step j+2 have
//   landed (counted vmcnt: the six pieces of step j+3 may still fly) and for its own fragment reads, then the barrier.
// A thread moves six 16-byte units per step: four of A (the hi planes of two K-tiles, 512 units each, units tid and tid + 256)
// and two of B (per K-tile the hi plane's two k-groups x 128 columns of this half).
constexpr int H2_THREADS = 256, H2_TN = 128;
constexpr int H2_A16 = 2 * 512;                     // A units per stage: two hi planes of 256 rows x 2 k-groups
constexpr int H2_B16 = 2 * KG * H2_TN;              // B units per stage: two K-tiles x 2 k-groups x 128 columns
constexpr int H2_STAGE16 = H2_A16 + H2_B16;         // 1536 units = 24 KB
constexpr int H2_NSTAGE = 3;
constexpr int H2_LDS_BYTES = H2_NSTAGE * H2_STAGE16 * 16 + TM * 4;

template <bool SCORE>                                // true: fused score epilogue;  false: series-moments epilogue
__global__ void __launch_bounds__(H2_THREADS, 2)
k_sweep_hi2(const uint4* __restrict__ At, const uint4* __restrict__ Bt, int KT_tiles, int Mtiles, Score16Args sa,
            Plain16Args pa, BView bv, FoldViews fv) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds16[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int KT = KT_tiles / 2;                     // ring steps of two K-tiles (the host checks K % 64 == 0)

    const int tile = xcd_tile_id16(blockIdx.x, gridDim.x);
    const int mt_all = tile % Mtiles, nh = tile / Mtiles;       // nh = half column tile
    const int nt = nh >> 1, hb = nh & 1;
    const long long ncol0 = (long long)nt * TN + hb * H2_TN;    // first column of this workgroup
    if (sa.live_cols != nullptr && ncol0 >= (long long)*sa.live_cols) return;
    const int fold = mt_all / fv.mt_per_fold;
    const int mt = mt_all - fold * fv.mt_per_fold;
    sa.yv += (long long)fold * sa.M * sa.V;
    sa.ymean += (long long)fold * 3 * sa.V;
    sa.part += (long long)fold * fv.part_stride;
    sa.n_val = fv.n_val[fold];
    const int b_cut = fv.cut[fold], b_skip = fv.skip[fold];
    const uint4* a_src = At + (long long)mt_all * KT_tiles * CHUNK16 + tid;
    // B: K-tile chunk = [plane][k-group][256 columns]; this thread's unit of a hi plane's half: k-group tid >> 7, column tid & 127
    const uint4* b_src = Bt + (long long)nt * bv.kt_total * CHUNK16 + (tid >> 7) * 256 + hb * H2_TN + (tid & 127);
#define H2_BKT(kt_) ((kt_) + ((kt_) >= b_cut ? b_skip : 0))
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) uint4*)lds16);
#define H2_DMA(gptr_, unit_)                                                                                  \
    {                                                                                                         \
        const unsigned m0_ = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((unit_) + (wave << 6)) * 16u); \
        const uint4* gp_ = (gptr_);                                                                           \
        unsigned keep_;                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(gp_), "s"(m0_) : "memory");                                          \
    }
    // piece q = 0..5 of ring step t into stage s
#define H2_PIECE(q_, t_, s_)                                                                                   \
    {                                                                                                          \
        if ((q_) == 0) H2_DMA(a_src + (long long)(2 * (t_)) * CHUNK16, (s_) * H2_STAGE16);                     \
        if ((q_) == 1) H2_DMA(a_src + (long long)(2 * (t_)) * CHUNK16 + 256, (s_) * H2_STAGE16 + 256);         \
        if ((q_) == 2) H2_DMA(a_src + (long long)(2 * (t_) + 1) * CHUNK16, (s_) * H2_STAGE16 + 512);           \
        if ((q_) == 3) H2_DMA(a_src + (long long)(2 * (t_) + 1) * CHUNK16 + 256, (s_) * H2_STAGE16 + 768);     \
        if ((q_) == 4) H2_DMA(b_src + (long long)H2_BKT(2 * (t_)) * CHUNK16, (s_) * H2_STAGE16 + H2_A16);      \
        if ((q_) == 5) H2_DMA(b_src + (long long)H2_BKT(2 * (t_) + 1) * CHUNK16, (s_) * H2_STAGE16 + H2_A16 + 256); \
    }
#define H2_BARRIER()                           \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < H2_NSTAGE && t < KT; ++t) {
#pragma unroll
        for (int q = 0; q < 6; ++q) H2_PIECE(q, t, t);
    }

    // epilogue operands that do not depend on the accumulators: row scales -> LDS behind the ring, column constants
    float* lds_rs = reinterpret_cast<float*>(lds16 + H2_NSTAGE * H2_STAGE16);
    const long long V = sa.V;
    const long long col0 = ncol0 + wn * 64 + li;
    const bool cok[2] = {col0 < V, col0 + 32 < V};
    const long long colc[2] = {cok[0] ? col0 : 0, cok[1] ? col0 + 32 : 0};
    float ymv[2] = {0.f, 0.f}, cscv[2] = {0.f, 0.f};
    if (SCORE) {
        lds_rs[tid] = sa.rs_inv[mt_all * TM + tid];                              // (256 threads = the tile's 256 rows)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            ymv[ni] = sa.ymean[colc[ni]];
            cscv[ni] = sa.cs_inv[colc[ni]];
        }
    } else {                                            // series moments: this wave's column block after the swap: ni = wm
        lds_rs[tid] = pa.rs_inv[(long long)mt_all * TM + tid];
        ymv[0] = sa.ymean[colc[wm]];
        cscv[0] = pa.cs_inv[ncol0 + wn * 64 + wm * 32 + li];
    }
    auto val_block = [&](int mi) { return min((mt * (TM / 32) + wm * 4 + mi) / sa.A, (sa.M >> 5) - 1); };
    auto load_vb = [&](int ib, int ni, lc::EpiTargets& t) { lc::epi_load_targets(sa.yv, V, ib * 32, lh, colc[ni], t); };
    lc::EpiTargets tc0, tc1;                         // the current validation block's targets, panels 0 and 1 (as in k_sweep_f16x3)

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    H2_BARRIER();

    // fragment addresses (16-byte units) inside a stage.  A: ((K-tile q * KG + lh) * 256 + row);  B: A16 + ((q * KG + lh) * 128 + col)
    const int a_frag = lh * 256 + wm * 128 + li;
    const int b_frag = H2_A16 + lh * H2_TN + wn * 64 + li;
    struct Frag {
        h8 a0[4], a1[4], b0[2], b1[2];                  // K-tile 2t (a0, b0) and 2t+1 (a1, b1)
    };
    auto read_frag = [&](Frag& f, const uint4* st, const int k) {
        if (k < 2) {
            const uint4 v = st[b_frag + k * 32];
            f.b0[k] = *reinterpret_cast<const h8*>(&v);
        } else if (k < 6) {
            const uint4 v = st[a_frag + (k - 2) * 32];
            f.a0[k - 2] = *reinterpret_cast<const h8*>(&v);
        } else if (k < 8) {
            const uint4 v = st[b_frag + KG * H2_TN + (k - 6) * 32];
            f.b1[k - 6] = *reinterpret_cast<const h8*>(&v);
        } else {
            const uint4 v = st[a_frag + KG * 256 + (k - 8) * 32];
            f.a1[k - 8] = *reinterpret_cast<const h8*>(&v);
        }
    };
    Frag fa, fb;
#pragma unroll
    for (int k = 0; k < 12; ++k) read_frag(fa, lds16, k);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    H2_BARRIER();                                       // everybody holds step 0: stage 0 may be overwritten

    // a 128-row slab with no valid rows (the padded half of a fold's last M-tile): its waves keep up their share of the DMA
    // ring and the barriers only
    const bool slab_empty = SCORE && mt * TM + wm * 128 >= sa.Mrows;
    int s_cur = 0;                                      // stage of step kt (kt % 3), carried along
    // MODE 0 = steady (kt + 3 < KT: a DMA step, a next step), 1 = guarded tail, 2 = the last step
    auto kstep = [&](const int kt, const Frag& cur, Frag& nxt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool STEADY = MODE == 0, LAST = MODE == 2;
        const bool has_next = STEADY || (!LAST && kt + 1 < KT);
        const bool do_dma = STEADY || (!LAST && kt + 3 < KT);
        const int s_nxt = s_cur == 2 ? 0 : s_cur + 1;
        const uint4* stn = lds16 + s_nxt * H2_STAGE16;
        if (LAST && SCORE) { load_vb(val_block(0), 0, tc0); load_vb(val_block(0), 1, tc1); }
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            if (!slab_empty) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * sl + h, mi = (i & 7) >> 1, ni = i & 1;
                    if (i < 8) MFMA16(acc[mi][ni], cur.a0[mi], cur.b0[ni]);
                    else MFMA16(acc[mi][ni], cur.a1[mi], cur.b1[ni]);
                }
                if (has_next) {
                    if (sl == 0) { read_frag(nxt, stn, 0); read_frag(nxt, stn, 1); }
                    if (sl == 1) { read_frag(nxt, stn, 2); read_frag(nxt, stn, 3); }
                    if (sl == 2) { read_frag(nxt, stn, 4); read_frag(nxt, stn, 5); }
                    if (sl == 3) { read_frag(nxt, stn, 6); read_frag(nxt, stn, 7); }
                    if (sl >= 4) read_frag(nxt, stn, sl + 4);
                }
            }
            if (do_dma) {
                if (sl == 0) H2_PIECE(0, kt + 3, s_cur);
                if (sl == 1) H2_PIECE(1, kt + 3, s_cur);
                if (sl == 3) H2_PIECE(2, kt + 3, s_cur);
                if (sl == 4) H2_PIECE(3, kt + 3, s_cur);
                if (sl == 5) H2_PIECE(4, kt + 3, s_cur);
                if (sl == 7) H2_PIECE(5, kt + 3, s_cur);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s_cur = s_nxt;
        if (LAST) return;
        // publish step kt+2 (read during iteration kt+1): its pieces were issued in iteration kt-1 (or the prologue); the
        // six pieces of step kt+3, if any were issued now, may still fly.  lgkmcnt(0): this wave's fragment reads of step
        // kt+1 have returned before it reports at the barrier (the next iteration's first DMA piece goes into their stage)
        if (STEADY || kt + 3 < KT) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (STEADY || kt + 2 < KT) { H2_BARRIER(); }
    };
    using Steady = std::integral_constant<int, 0>;
    using Tail = std::integral_constant<int, 1>;
    using Last = std::integral_constant<int, 2>;
    int kt = 0;
    for (; kt + 4 < KT; kt += 2) {                     // (kt + 1) + 3 < KT: both steps issue a DMA step
        kstep(kt, fa, fb, Steady{});
        kstep(kt + 1, fb, fa, Steady{});
    }
    for (; kt + 2 < KT; kt += 2) {
        kstep(kt, fa, fb, Tail{});
        kstep(kt + 1, fb, fa, Tail{});
    }
    kstep(kt, fa, fb, Tail{});                         // KT (steps) is even: kt == KT - 2 here
    kstep(kt + 1, fb, fa, Last{});

    if (!SCORE) {
        // ---- series-moments epilogue (as k_sweep_f16x3<SERMOM>; the swap buffer is the idle ring: 4 waves x 16 KB)
        const long long colm = colc[wm];
        const int blk0 = 2 * mt;
        lc::EpiTargets tg[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if ((blk0 + b) * 32 < sa.M) lc::epi_load_targets(sa.yv, V, (blk0 + b) * 32, lh, colm, tg[b]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        H2_BARRIER();                                   // every wave is done with the ring: it becomes the swap buffer
        lc::ep_f32x4* xch = reinterpret_cast<lc::ep_f32x4*>(lds16);
        {
            lc::ep_f32x4* dst = xch + (wn * 2 + wm) * 16 * 64 + lane;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const lc::ep_f32x4 rs = *reinterpret_cast<const lc::ep_f32x4*>(lds_rs + wm * 128 + mi * 32 + 8 * q + 4 * lh);
                    lc::ep_f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (wm ? acc[mi][0][4 * q + j] : acc[mi][1][4 * q + j]) * rs[j];
                    dst[(mi * 4 + q) * 64] = v;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        H2_BARRIER();
        const lc::ep_f32x4* src = xch + (wn * 2 + (1 - wm)) * 16 * 64 + lane;
        const float cs = cscv[0], ym = ymv[0];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int i0 = (blk0 + b) * 32;
            if (i0 >= sa.n_val) continue;
            float own[2][16], oth[2][16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mi = 2 * h + b;
                    const lc::ep_f32x4 rs = *reinterpret_cast<const lc::ep_f32x4*>(lds_rs + wm * 128 + mi * 32 + 8 * q + 4 * lh);
                    const lc::ep_f32x4 o = src[(mi * 4 + q) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        own[h][4 * q + j] = (wm ? acc[mi][1][4 * q + j] : acc[mi][0][4 * q + j]) * rs[j] * cs;
                        oth[h][4 * q + j] = o[j] * cs;
                    }
                }
            float T[4][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                T[0][r] = wm ? oth[0][r] : own[0][r];
                T[1][r] = wm ? oth[1][r] : own[1][r];
                T[2][r] = wm ? own[0][r] : oth[0][r];
                T[3][r] = wm ? own[1][r] : oth[1][r];
            }
            float* dstp = sa.part + (long long)(blk0 + b) * lc::EPI_SERIES_PARTS * V + colm;
            if (i0 + 32 <= sa.n_val) lc::epi_series_block<false>(T, tg[b], ym, i0, sa.n_val, lh, dstp, V, cok[wm]);
            else lc::epi_series_block<true>(T, tg[b], ym, i0, sa.n_val, lh, dstp, V, cok[wm]);
        }
        return;
    }

    // ---- score epilogue (as k_sweep_f16x3<SCORE>): two target batches are in registers, the third is issued first
    const bool corr = sa.mode == LC_SCORE_CORR;
    auto reduce = [&](int step, const lc::EpiTargets& t) {
        const int mi = step >> 1, ni = step & 1;
        const int rb0 = mt * TM + wm * 128 + mi * 32;
        if (rb0 >= sa.Mrows) return;
        lc::ep_f32x4 rs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rs[q] = *reinterpret_cast<const lc::ep_f32x4*>(lds_rs + wm * 128 + mi * 32 + 8 * q + 4 * lh);
        const int blk = rb0 >> 5, ib = blk / sa.A, al = blk - ib * sa.A;
        lc::epi_block_dispatch<true>(corr, acc[mi][ni], t, rs, cscv[ni], ymv[ni], ib * 32, sa.n_val, lh,
                                     sa.part + (long long)(al * (sa.M >> 5) + ib) * 4 * V + colc[ni], V, cok[ni]);
    };
#define H2_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        reduce(2 * mi, tc0); H2_FENCE();
        reduce(2 * mi + 1, tc1); H2_FENCE();
        if (mi < 3) {
            const int ib_next = val_block(mi + 1);
            if (ib_next != val_block(mi)) {
                load_vb(ib_next, 0, tc0);
                load_vb(ib_next, 1, tc1);
            }
        }
        H2_FENCE();
    }
#undef H2_FENCE
#undef H2_BARRIER
#undef H2_PIECE
#undef H2_DMA
#undef H2_BKT
}

}  // namespace
