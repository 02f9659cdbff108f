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
#include <type_traits>

#include "lc_common.h"
#include "lc_epilogue.h"

namespace {

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
__global__ void __launch_bounds__(256) k_split_rows_f16(const float* __restrict__ h, long long ld, int rows, int K,
                                                        uint4* __restrict__ out, float* __restrict__ rs_inv,
                                                        int rows_pad, int groups, int il_A) {
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
    const float* src = h + ((long long)g * rows + rsrc) * ld;
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
                                                           float* __restrict__ cs, int* __restrict__ flag) {
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
        if (flag != nullptr && mx > 0.f && 2 * n_small > T) atomicOr(flag, 1);
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
                                                        uint4* __restrict__ out) {
    const int nt = blockIdx.x, g = blockIdx.y;              // g = K-tile * KG + k-group
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
template <bool SCORE, bool STAMP, bool LIGHTCAP = false, bool SERMOM = false, bool PEARSON = false>
__global__ void __launch_bounds__(512, 2)
k_sweep_f16x3(const uint4* __restrict__ At, const uint4* __restrict__ Bt, int KT, int Mtiles, Score16Args sa,
              Plain16Args pa, BView bv, FoldViews fv) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds16[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform (scalar)
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    const int tile = xcd_tile_id16(blockIdx.x, gridDim.x);
    const int mt_all = tile % Mtiles, nt = tile / Mtiles;
    constexpr bool FOLDS = SCORE || SERMOM;
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
    const uint4* a_src = At + ((long long)grp * Mtiles + mt_all) * KT * CHUNK16 + tid;
    const uint4* b_src = Bt + (long long)nt * bv.kt_total * CHUNK16 + tid;
#define BKT(kt_) ((kt_) + ((kt_) >= b_cut ? b_skip : 0))
    // "light" slabs (plain mode): rows whose product only needs fp16 accuracy (11-bit operands) -- the higher
    // terms of a series, which enter the caller's result scaled down by >= 2^-11 -- take the hi*hi MFMA alone
    const bool light = SERMOM ? wm != 0
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
#define DMA16(gptr_, unit_)                                                                                   \
    {                                                                                                         \
        const unsigned m0_ = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((unit_) + (wave << 6)) * 16u); \
        const uint4* gp_ = (gptr_);                                                                           \
        unsigned keep_;                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(gp_), "s"(m0_) : "memory");                                          \
    }
#define GLDS16(kt_, stg_)                                                           \
    {                                                                               \
        const uint4* pa_ = a_src + (long long)(kt_) * CHUNK16;                      \
        const uint4* pb_ = b_src + (long long)BKT(kt_) * CHUNK16;                   \
        DMA16(pa_, (stg_) * STAGE16);                                               \
        DMA16(pa_ + 512, (stg_) * STAGE16 + 512);                                   \
        DMA16(pb_, (stg_) * STAGE16 + CHUNK16);                                     \
        DMA16(pb_ + 512, (stg_) * STAGE16 + CHUNK16 + 512);                         \
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
    auto load_t = [&](int step, lc::EpiTargets& t) {
        const int blk = mt * (TM / 32) + wm * 4 + (step >> 1);                      // image block = (validation block, alpha)
        // (blocks past the image's last one -- the padding of its last tile -- load the last validation block: unused)
        lc::epi_load_targets(sa.yv, V, min(blk / sa.A, (sa.M >> 5) - 1) * 32, lh, colc[step & 1], t);
    };
    lc::EpiTargets tb0, tb1, tb2;

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
        if (LAST && SCORE) { load_t(0, tb0); load_t(1, tb1); }   // land under the MFMAs of the last tile
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
    // a 128-row slab with no valid rows (the last M-tile of A x 480 = 1920 score rows is half padding): its waves only
    // keep up their share of the DMA ring and the barriers; the other wave of each SIMD then has the matrix pipe to
    // itself and the tile takes half the time -- 1/16 of a fused launch
    auto kstep_empty = [&](const int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool STEADY = MODE == 0, LAST = MODE == 2;
        if (STEADY || (!LAST && kt + 4 < KT)) {
            const int stg = kt & 3;
            const uint4* pa_ = a_src + (long long)(kt + 4) * CHUNK16;
            const uint4* pb_ = b_src + (long long)BKT(kt + 4) * CHUNK16;
            DMA16(pa_, stg * STAGE16);
            DMA16(pa_ + 512, stg * STAGE16 + 512);
            DMA16(pb_, stg * STAGE16 + CHUNK16);
            DMA16(pb_ + 512, stg * STAGE16 + CHUNK16 + 512);
        }
        if (LAST) return;
        if (STEADY || kt + 4 < KT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 3 < KT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STEADY || kt + 2 < KT) { PHASE_BARRIER(); }
    };
    using Steady = std::integral_constant<int, 0>;
    using Tail = std::integral_constant<int, 1>;
    using Last = std::integral_constant<int, 2>;
    const bool slab_empty = !SERMOM && !STAMP && mt * TM + wm * 128 >= (SCORE ? sa.Mrows : pa.Mrows);
    int kt = 0;
    if (slab_empty) {
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
    load_t(2, tb2); EPI_FENCE();
    unsigned long long ter = 0;
    if (STAMP) {                       // diagnostics: run step 0 twice, the second pass finds code and data warm
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {
            reduce(0, tb0); EPI_FENCE();
            if (rep == 0) { STAMP_T(ter); }
        }
        load_t(3, tb0); EPI_FENCE();
    } else {
        reduce(0, tb0); EPI_FENCE(); load_t(3, tb0); EPI_FENCE();
    }
    STAMP_T(te1);
    reduce(1, tb1); EPI_FENCE(); load_t(4, tb1); EPI_FENCE();
    reduce(2, tb2); EPI_FENCE(); load_t(5, tb2); EPI_FENCE();
    reduce(3, tb0); EPI_FENCE(); load_t(6, tb0); EPI_FENCE();
    reduce(4, tb1); EPI_FENCE(); load_t(7, tb1); EPI_FENCE();
    reduce(5, tb2);
    reduce(6, tb0);
    STAMP_T(te2);
    reduce(7, tb1);
    STAMP_T(te3);
#undef EPI_FENCE
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP_T(tk3);
        if (lane == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(pa.c) + wm * 16;
            atomicAdd(dbg + 7, tk3 - tk2);
            atomicAdd(dbg + 8, te0 - tk2);       // group-0 catch-up barrier
            atomicAdd(dbg + 9, ter - te0);       // first reduce step (cold)
            atomicAdd(dbg + 13, te1 - ter);      // the same step again (warm)
            atomicAdd(dbg + 10, te2 - te1);      // steps 1..6
            atomicAdd(dbg + 11, te3 - te2);      // step 7
            atomicAdd(dbg + 12, tk3 - te3);      // store drain
        }
    }
}


// ------------------------------------------------------------------ experiment: the same contraction on 16x16x32 MFMAs
// v_mfma_f32_16x16x32_f16 holds a higher clock than 32x32x16 at equal flops (tools/mfma_f16_rate.hip: +7 % with the
// kernel's LDS traffic).  Same tiled operand images, same 256 x 256 tile, 8 waves (2 x 4), wave tile 128 x 64 = 8 x 4
// blocks of 16 x 16; one MFMA step is K = 32 = TWO ring stages (lane groups 0, 1 read the first, 2, 3 the second).
// Fragments cannot be double-buffered (96 VGPRs a set + 128 accumulators), so the three terms rotate:
//     term 0 (lo*hi)  ||  read ah, bl of THIS pair          -- al, bh were read during the previous iteration
//     [barrier: pair p's stages are free, pair p+1 is published; DMA of pair p+2 starts]
//     term 2 (hi*hi)  ||  read al of the next pair
//     term 1 (hi*lo)  ||  read bh of the next pair
// Plain (store) mode only: diagnostics (lc_debug_gemm_f16x3_wide), not the product path.
typedef float f32x4w __attribute__((ext_vector_type(4)));
#define MFMA16W(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, acc_, 0, 0, 0)

__global__ void __launch_bounds__(512, 2)
k_sweep16w_plain(const uint4* __restrict__ At, const uint4* __restrict__ Bt, int KT, int Mtiles, Plain16Args pa) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds16[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;              // lane group = 8-k group of the K = 32 step
    const int tile = xcd_tile_id16(blockIdx.x, gridDim.x);
    const int mt = tile % Mtiles, nt = tile / Mtiles;
    const uint4* a_src = At + (long long)mt * KT * CHUNK16 + tid;
    const uint4* b_src = Bt + (long long)nt * KT * CHUNK16 + tid;
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) uint4*)lds16);
#define DMA16W(gptr_, unit_)                                                                                  \
    {                                                                                                         \
        const unsigned m0_ = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((unit_) + (wave << 6)) * 16u); \
        const uint4* gp_ = (gptr_);                                                                           \
        unsigned keep_;                                                                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(gp_), "s"(m0_) : "memory");                                          \
    }
#define GLDS16W(kt_, stg_)                                                          \
    {                                                                               \
        const uint4* pa_ = a_src + (long long)(kt_) * CHUNK16;                      \
        const uint4* pb_ = b_src + (long long)(kt_) * CHUNK16;                      \
        DMA16W(pa_, (stg_) * STAGE16);                                              \
        DMA16W(pa_ + 512, (stg_) * STAGE16 + 512);                                  \
        DMA16W(pb_, (stg_) * STAGE16 + CHUNK16);                                    \
        DMA16W(pb_ + 512, (stg_) * STAGE16 + CHUNK16 + 512);                        \
    }
#define BARRIER16W()                           \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0)

    f32x4w acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
    const int P = KT / 2;                                    // K = 32 steps (KT is even)
    for (int t = 0; t < NSTAGE && t < KT; ++t) GLDS16W(t, t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BARRIER16W();
    // fragment addresses (16-byte units) of this lane inside pair q (stages 2q & 3 and (2q + 1) & 3): the lane's
    // k-group picks the stage (lg >> 1) and the group inside it (lg & 1)
    const int a_off = (lg & 1) * 256 + wm * 128 + l15;
    const int b_off = CHUNK16 + (lg & 1) * 256 + wn * 64 + l15;
    h8 ah[8], al[8], bh[4], bl[4];
    auto stage_of = [&](int q) { return lds16 + ((2 * q + (lg >> 1)) & 3) * STAGE16; };
    auto rd_ah = [&](const uint4* st, int mi) { const uint4 v = st[a_off + mi * 16]; ah[mi] = *reinterpret_cast<const h8*>(&v); };
    auto rd_al = [&](const uint4* st, int mi) { const uint4 v = st[a_off + KG * 256 + mi * 16]; al[mi] = *reinterpret_cast<const h8*>(&v); };
    auto rd_bh = [&](const uint4* st, int ni) { const uint4 v = st[b_off + ni * 16]; bh[ni] = *reinterpret_cast<const h8*>(&v); };
    auto rd_bl = [&](const uint4* st, int ni) { const uint4 v = st[b_off + KG * 256 + ni * 16]; bl[ni] = *reinterpret_cast<const h8*>(&v); };
    {
        const uint4* st = stage_of(0);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) rd_al(st, mi);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) rd_bh(st, ni);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int p = 0; p < P; ++p) {
        const uint4* st = stage_of(p);
        const uint4* stn = stage_of(p + 1);
        const bool has_next = p + 1 < P;
        // ---- term 0: al * bh, reading ah and bl of this pair  (slots of four MFMAs and one or two reads: two MFMAs
        // and one read per slot, with the DMA pieces spread over the slots, measured 13-20 % slower)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) MFMA16W(acc[mi][ni], al[mi], bh[ni]);
            rd_ah(st, mi);
            if (mi < 4) rd_bl(st, mi);
            __builtin_amdgcn_sched_barrier(0);
        }
        // pair p's stages are read out; pair p + 1 must be complete and visible from here on
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        BARRIER16W();
        if (2 * p + 4 < KT) GLDS16W(2 * p + 4, (2 * p) & 3);
        if (2 * p + 5 < KT) GLDS16W(2 * p + 5, (2 * p + 1) & 3);
        // ---- term 2: ah * bh, reading al of the next pair
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) MFMA16W(acc[mi][ni], ah[mi], bh[ni]);
            if (has_next) rd_al(stn, mi);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- term 1: ah * bl, reading bh of the next pair
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) MFMA16W(acc[mi][ni], ah[mi], bl[ni]);
            if (has_next && mi < 4) rd_bh(stn, mi);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- plain epilogue: accumulator r of a lane = row 4 lg + r, column l15 of the 16 x 16 block
    float* cbase = pa.c + (long long)nt * TN;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int rb0 = mt * TM + wm * 128 + mi * 16 + 4 * lg;
        float rsc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rsc[r] = pa.rs_inv[rb0 + r];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + l15;
            const float csc = pa.cs_inv[(long long)nt * TN + col];
            const bool col_ok = (long long)nt * TN + col < pa.col_limit;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rb0 + r < pa.Mrows && col_ok) cbase[(long long)(rb0 + r) * pa.ldc + col] = acc[mi][ni][r] * rsc[r] * csc;
        }
    }
#undef DMA16W
#undef GLDS16W
#undef BARRIER16W
}

}  // namespace

// defined in lc_gemm.hip: combines the per-block partial moments into scores (F folds: fp32 sum in fold order)
int lc_score_finalize_launch(const float* d_part, const float* d_ystat, const float* d_yblk, int A, int M,
                             const int* h_n_val, int F, long long V, int mode, float* d_scores, int accumulate,
                             hipStream_t s);

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
                       (long long)ld, (int)rows, (int)K, (uint4*)d_tiled, d_rowscale_inv, (int)rows_pad, groups, 0);
    return lc::launched("k_split_rows_f16");
}

extern "C" int lc_split_rows_f16_alphas(const float* d_h, int64_t ld, int groups, int A, int64_t M, int64_t K, void* d_tiled,
                                        float* d_rowscale_inv, lc_stream_t stream) {
    LC_REQUIRE(d_h && d_tiled && d_rowscale_inv, LC_E_BADARG, "lc_split_rows_f16_alphas: null pointer");
    LC_REQUIRE(groups > 0 && A > 0 && M > 0 && M % LC_MB == 0 && K > 0 && K % TK == 0 && ld % 4 == 0 && ld >= K, LC_E_SHAPE,
               "lc_split_rows_f16_alphas: need M %% %d == 0, K %% %d == 0 and ld %% 4 == 0", LC_MB, TK);
    const long long rows = (long long)A * M;
    const long long rows_pad = lc::ceil_div<long long>(rows, TM) * TM;
    LC_REQUIRE(rows_pad * groups < (1ll << 31), LC_E_SHAPE, "lc_split_rows_f16_alphas: too many rows");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_split_rows_f16, dim3((unsigned)(rows_pad * groups / 4)), dim3(256), 0, lc::as_stream(stream), d_h,
                       (long long)ld, (int)rows, (int)K, (uint4*)d_tiled, d_rowscale_inv, (int)rows_pad, groups, A);
    return lc::launched("k_split_rows_f16");
}

extern "C" int lc_split_rows_f16(const float* d_h, int64_t ld, int64_t rows, int64_t K, void* d_tiled,
                                 float* d_rowscale_inv, lc_stream_t stream) {
    return lc_split_rows_f16_groups(d_h, ld, 1, rows, K, d_tiled, d_rowscale_inv, stream);
}

extern "C" int lc_col_scales_f16(const float* d_y, int64_t ldy, int64_t T, int64_t V, float* d_cscale,
                                 int32_t* d_flag, lc_stream_t stream) {
    LC_REQUIRE(d_y && d_cscale, LC_E_BADARG, "lc_col_scales_f16: null pointer");     // d_flag may be NULL: scales only
    LC_REQUIRE(T > 0 && V > 0 && ldy >= V, LC_E_SHAPE, "lc_col_scales_f16: bad shape");
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    hipLaunchKernelGGL(k_col_scales, dim3((unsigned)lc::ceil_div<long long>(V, 64)), dim3(64, CS_RG), 0,
                       lc::as_stream(stream), d_y, (long long)ldy, (int)T, (long long)V, d_cscale, d_flag);
    return lc::launched("k_col_scales");
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
                                 const float* d_cscale, void* d_tiled, lc_stream_t stream) {
    LC_REQUIRE(d_y && d_rows && d_cscale && d_tiled, LC_E_BADARG, "lc_split_cols_f16: null pointer");
    LC_REQUIRE(V > 0 && K > 0 && K % TK == 0 && ldy >= V, LC_E_SHAPE, "lc_split_cols_f16: need K %% %d == 0", TK);
    lc::ScopedTimer timer_(lc::T_SPLIT16, lc::as_stream(stream));
    dim3 grid((unsigned)lc::ceil_div<long long>(V, TN), (unsigned)(K / 8));
    hipLaunchKernelGGL(k_split_cols_f16, grid, dim3(256), 0, lc::as_stream(stream), d_y, (long long)ldy, (long long)V,
                       d_rows, K, d_cscale, (uint4*)d_tiled);
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
                                                 const int64_t* h_gap_rows, lc_stream_t stream) {
    LC_REQUIRE(d_ht && d_rowscale_inv && d_yt && d_cscale_inv && d_yv && d_ystat && d_yblk && d_part && d_scores &&
                   h_n_val, LC_E_BADARG, "lc_alpha_sweep_scores_f16x3: null pointer");
    LC_REQUIRE(A > 0 && M > 0 && M % LC_MB == 0 && N > 0 && N % (2 * TK) == 0, LC_E_SHAPE,
               "lc_alpha_sweep_scores_f16x3: need M %% %d == 0, N %% %d == 0", LC_MB, 2 * TK);
    LC_REQUIRE(V > 0 && V % 128 == 0, LC_E_SHAPE, "lc_alpha_sweep_scores_f16x3: V must be a multiple of 128");
    LC_REQUIRE(mode == LC_SCORE_CORR || mode == LC_SCORE_R2, LC_E_BADARG, "lc_alpha_sweep_scores_f16x3: bad mode");
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_sweep_f16x3<true, false>), LDS16_BYTES)) return rc;
    hipStream_t s = lc::as_stream(stream);
    BView bv;
    FoldViews fv{};
    if (int rc = fill_fold_views("lc_alpha_sweep_scores_f16x3", F, N, b_rows, h_gap_begin, h_gap_rows, h_n_val, M, &bv, &fv))
        return rc;
    const int Mrows = A * M;
    const int Mtiles = lc::ceil_div(Mrows, TM);            // per fold: the folds' images are stacked tile-aligned
    const long long Ntiles = lc::ceil_div<long long>(V, TN);
    LC_REQUIRE((long long)F * Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_alpha_sweep_scores_f16x3: grid too large");
    fv.mt_per_fold = Mtiles;
    fv.part_stride = (long long)(Mrows / LC_MB) * 4 * V;
    Score16Args sa{d_yv, d_ystat, d_rowscale_inv, d_cscale_inv, d_part, (long long)V, M, 0, mode, Mrows, A};
    {
        lc::ScopedTimer timer_(lc::T_SWEEP_GEMM, s);
        Plain16Args pa{};
        hipLaunchKernelGGL((k_sweep_f16x3<true, false>), dim3((unsigned)(F * Mtiles * Ntiles)), dim3(512), LDS16_BYTES, s,
                           (const uint4*)d_ht, (const uint4*)d_yt, N / TK, F * Mtiles, sa, pa, bv, fv);
    }
    if (int rc = lc::launched("k_sweep_f16x3")) return rc;
    return lc_score_finalize_launch(d_part, d_ystat, d_yblk, A, M, h_n_val, F, (long long)V, mode, d_scores, accumulate, s);
}

extern "C" int lc_alpha_sweep_scores_f16x3(const void* d_ht, const float* d_rowscale_inv, int A, int M, int N,
                                           const void* d_yt, const float* d_cscale_inv, const float* d_yv,
                                           int64_t V, int n_val, const float* d_ystat,
                                           const float* d_yblk, int mode, float* d_part, float* d_scores,
                                           int accumulate, int64_t b_rows, int64_t b_gap_begin, int64_t b_gap_rows,
                                           lc_stream_t stream) {
    const int32_t nv = n_val;
    return lc_alpha_sweep_scores_f16x3_folds(d_ht, d_rowscale_inv, 1, A, M, N, d_yt, d_cscale_inv, d_yv, V, &nv, d_ystat,
                                             d_yblk, mode, d_part, d_scores, accumulate, b_rows, &b_gap_begin, &b_gap_rows,
                                             stream);
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
                              int accumulate, hipStream_t s);

extern "C" int lc_series_sweep_scores_f16x3_folds(const void* d_pt, const float* d_rowscale_inv, int F, int M,
                                                  const int32_t* h_n_val, int64_t K, const void* d_yt,
                                                  const float* d_cscale_inv, int64_t Ncols, const float* d_yv, int64_t V,
                                                  const float* d_ystat, const float* d_yblk, const double* d_coef,
                                                  const int32_t* d_aidx, int S, float* d_part, float* d_scores,
                                                  int accumulate, int64_t b_rows, const int64_t* h_gap_begin,
                                                  const int64_t* h_gap_rows, lc_stream_t stream) {
    LC_REQUIRE(d_pt && d_rowscale_inv && d_yt && d_cscale_inv && d_yv && d_ystat && d_yblk && d_coef && d_aidx &&
                   d_part && d_scores && h_n_val, LC_E_BADARG, "lc_series_sweep_scores_f16x3: null pointer");
    LC_REQUIRE(M > 0 && M % LC_MB == 0 && K > 0 && K % (2 * TK) == 0 && S > 0, LC_E_SHAPE,
               "lc_series_sweep_scores_f16x3: need M %% %d == 0, K %% %d == 0", LC_MB, 2 * TK);
    LC_REQUIRE(V > 0 && V % 128 == 0 && Ncols >= V && Ncols % TN == 0, LC_E_SHAPE,
               "lc_series_sweep_scores_f16x3: V must be a multiple of 128, Ncols >= V a multiple of %d", TN);
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_sweep_f16x3<false, false, true, true>), LDS16_BYTES))
        return rc;
    hipStream_t s = lc::as_stream(stream);
    BView bv;
    FoldViews fv{};
    if (int rc = fill_fold_views("lc_series_sweep_scores_f16x3", F, K, b_rows, h_gap_begin, h_gap_rows, h_n_val, M, &bv, &fv))
        return rc;
    for (int f = 0; f < F; ++f) LC_REQUIRE(h_n_val[f] > 1, LC_E_SHAPE, "lc_series_sweep_scores_f16x3: need n_val > 1");
    const int nblk = M / LC_MB;
    const int Mtiles = (nblk + 1) / 2;                     // two 32-row validation blocks x four terms per tile
    const long long Ntiles = Ncols / TN;
    LC_REQUIRE((long long)F * Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_series_sweep_scores_f16x3: grid too large");
    fv.mt_per_fold = Mtiles;
    fv.part_stride = (long long)nblk * lc::EPI_SERIES_PARTS * V;
    Score16Args sa{d_yv, d_ystat, nullptr, nullptr, d_part, (long long)V, M, 0, LC_SCORE_CORR, Mtiles * TM, 1};
    Plain16Args pa{};
    pa.rs_inv = d_rowscale_inv;
    pa.cs_inv = d_cscale_inv;
    pa.Mrows = Mtiles * TM;
    pa.G = 1;
    pa.start[0] = 0;
    pa.start[1] = (int)Ntiles;
    pa.slab_light = nullptr;
    {
        lc::ScopedTimer timer_(lc::T_GROUPED_GEMM, s);
        hipLaunchKernelGGL((k_sweep_f16x3<false, false, true, true>), dim3((unsigned)(F * Mtiles * Ntiles)), dim3(512),
                           LDS16_BYTES, s, (const uint4*)d_pt, (const uint4*)d_yt, (int)(K / TK), F * Mtiles, sa, pa, bv, fv);
    }
    if (int rc = lc::launched("k_sweep_f16x3<series moments>")) return rc;
    return lc_series_finalize_launch(d_part, d_ystat, d_yblk, M, h_n_val, F, (long long)V, d_coef, d_aidx, S, d_scores,
                                     accumulate, s);
}

extern "C" int lc_series_sweep_scores_f16x3(const void* d_pt, const float* d_rowscale_inv, int M, int n_val, int64_t K,
                                            const void* d_yt, const float* d_cscale_inv, int64_t Ncols,
                                            const float* d_yv, int64_t V, const float* d_ystat, const float* d_yblk,
                                            const double* d_coef, const int32_t* d_aidx, int S, float* d_part,
                                            float* d_scores, int accumulate, int64_t b_rows, int64_t b_gap_begin,
                                            int64_t b_gap_rows, lc_stream_t stream) {
    const int32_t nv = n_val;
    return lc_series_sweep_scores_f16x3_folds(d_pt, d_rowscale_inv, 1, M, &nv, K, d_yt, d_cscale_inv, Ncols, d_yv, V,
                                              d_ystat, d_yblk, d_coef, d_aidx, S, d_part, d_scores, accumulate, b_rows,
                                              &b_gap_begin, &b_gap_rows, stream);
}

// Diagnostics: the score kernel with s_memtime stamps (not part of the product path; see tools/gpu_kernel_bench.py).
// d_stamps: 32 x uint64, zeroed by the caller: [wave group][main loop cycles, -, -, -, -, K-tiles, prologue, epilogue,
// -, epilogue step 0, steps 1-6, step 7, store drain, step 0 repeated, main loop 100 MHz ticks, -].
extern "C" int lc_debug_sweep16_stamps(const void* d_ht, const float* d_rowscale_inv, int A, int M, int N, const void* d_yt,
                                       const float* d_cscale_inv, const float* d_yv, int64_t V, int n_val,
                                       const float* d_ystat, float* d_part,
                                       unsigned long long* d_stamps, lc_stream_t stream) {
    LC_REQUIRE(d_ht && d_yt && d_stamps, LC_E_BADARG, "lc_debug_sweep16_stamps: null pointer");
    LC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_f16x3<true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS16_BYTES));
    const int Mrows = A * M;
    const int Mtiles = lc::ceil_div(Mrows, TM);
    const long long Ntiles = lc::ceil_div<long long>(V, TN);
    Score16Args sa{d_yv, d_ystat, d_rowscale_inv, d_cscale_inv, d_part, (long long)V, M, n_val, LC_SCORE_CORR, Mrows, A};
    Plain16Args pa{};
    FoldViews fv{};
    fv.mt_per_fold = Mtiles;
    fv.n_val[0] = n_val;
    fv.cut[0] = N / TK;
    pa.c = reinterpret_cast<float*>(d_stamps);
    hipLaunchKernelGGL((k_sweep_f16x3<true, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512), LDS16_BYTES,
                       lc::as_stream(stream), (const uint4*)d_ht, (const uint4*)d_yt, N / TK, Mtiles, sa, pa,
                       BView{N / TK, N / TK, 0}, fv);
    return lc::launched("k_sweep_f16x3<stamp>");
}

// Diagnostics: the single-group plain contraction on the 16x16x32 MFMA variant (see k_sweep16w_plain); same operands
// and output as lc_gemm_grouped_f16x3 with one group.
extern "C" int lc_debug_gemm_f16x3_wide(const void* d_at, const float* d_rowscale_inv, int64_t Mrows, const void* d_bt,
                                        const float* d_cscale_inv, float* d_c, int64_t ldc, int64_t Ncols, int64_t K,
                                        lc_stream_t stream) {
    LC_REQUIRE(d_at && d_rowscale_inv && d_bt && d_cscale_inv && d_c, LC_E_BADARG, "lc_debug_gemm_f16x3_wide: null pointer");
    LC_REQUIRE(Mrows > 0 && K > 0 && K % (2 * TK) == 0 && K / TK >= 4 && Ncols > 0 && Ncols % TN == 0 && ldc > Ncols - TN,
               LC_E_SHAPE, "lc_debug_gemm_f16x3_wide: need K %% %d == 0, K >= %d, Ncols %% %d == 0", 2 * TK, 4 * TK, TN);
    if (int rc = lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_sweep16w_plain), LDS16_BYTES)) return rc;
    const int Mtiles = (int)lc::ceil_div<long long>(Mrows, TM);
    const long long Ntiles = Ncols / TN;
    Plain16Args pa{};
    pa.c = d_c;
    pa.ldc = ldc;
    pa.rs_inv = d_rowscale_inv;
    pa.cs_inv = d_cscale_inv;
    pa.Mrows = (int)Mrows;
    pa.G = 1;
    pa.col_limit = ldc < Ncols ? ldc : Ncols;
    hipLaunchKernelGGL(k_sweep16w_plain, dim3((unsigned)(Mtiles * Ntiles)), dim3(512), LDS16_BYTES, lc::as_stream(stream),
                       (const uint4*)d_at, (const uint4*)d_bt, (int)(K / TK), Mtiles, pa);
    return lc::launched("k_sweep16w_plain");
}
