// The V-wide contractions of the ridge fit on the f32-input MFMA (v_mfma_f32_32x32x2_f32):
//   * fused alpha sweep: pred_a = H_a . Y[train rows] with the z-score / correlation (or R2)
//     reduction done in the epilogue -- predictions never reach HBM;
//   * grouped plain GEMM for the refit weights and the test predictions.
//
// Tile: 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA blocks.
// Operand feeding: the MFMA consumes two k per issue (lane half h = lane>>5 supplies one).  We
// let half h own k = 4h..4h+3 of every 8-k group, so ONE ds_read_b128 per operand block feeds
// four MFMAs (component t of the float4 pairs k = t (h=0) with k = 4+t (h=1)).  A is staged
// row-major with a 36-float row stride (conflict-free b128 reads).  B is staged as it comes from
// memory ([k][n], float4 rows, no register shuffles) and its fragment is four conflict-free
// ds_read_b32 -- the f32 MFMA is slow enough (64 cycles each) that LDS read width is irrelevant,
// while shuffles in the staging path would force early waits on the prefetched global loads.
#include "lc_common.h"
#include "lc_epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int AS_LD = BK + 4;                     // floats per staged A row
constexpr int AS_SZ = BM * AS_LD;                 // floats per A stage
constexpr int BS_SZ = BK * BN;                    // floats per B stage
constexpr int GEMM_LDS_BYTES = 2 * (AS_SZ + BS_SZ) * 4;
constexpr int MAX_GROUPS = 64;

struct GroupTiles {
    int start[MAX_GROUPS + 1];   // first column tile of each group; start[G] = number of tiles
    int G;
};

struct ScoreArgs {
    const float* yv;       // gathered validation targets (M, V), zero padding rows
    const float* ymean;    // (V)
    float* part;           // (rowblocks, 4, V)
    int M;                 // padded validation rows per alpha
    int n_val;
    int mode;
};

// XCD-aware block id: blocks that share an XCD (equal blockIdx % 8) get consecutive tile ids,
// so the M tiles that re-read one Y column panel run on one L2.  Bijective for any grid size.
__device__ inline int xcd_tile_id(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#define READ_FRAG(g, fa0, fa1, fb0, fb1)                                             \
    fa0 = *reinterpret_cast<const float4*>(as + (g) * 8);                            \
    fa1 = *reinterpret_cast<const float4*>(as + 32 * AS_LD + (g) * 8);               \
    fb0 = make_float4(bs[((g) * 8 + 0) * BN], bs[((g) * 8 + 1) * BN], bs[((g) * 8 + 2) * BN], bs[((g) * 8 + 3) * BN]); \
    fb1 = make_float4(bs[((g) * 8 + 0) * BN + 32], bs[((g) * 8 + 1) * BN + 32], bs[((g) * 8 + 2) * BN + 32],         \
                      bs[((g) * 8 + 3) * BN + 32])

#define MFMA4(acc_, fa_, fb_)                                                        \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_.x, fb_.x, acc_, 0, 0, 0);        \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_.y, fb_.y, acc_, 0, 0, 0);        \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_.z, fb_.z, acc_, 0, 0, 0);        \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_.w, fb_.w, acc_, 0, 0, 0)

#define MFMA_GROUP(fa0, fa1, fb0, fb1) \
    MFMA4(acc00, fa0, fb0);            \
    MFMA4(acc01, fa0, fb1);            \
    MFMA4(acc10, fa1, fb0);            \
    MFMA4(acc11, fa1, fb1)

template <bool SCORE, bool HAS_ROWS>
__global__ void __launch_bounds__(256, 2)
k_gemm_f32(const float* __restrict__ A, long long lda, long long a_group_stride, int Mrows,
           const float* __restrict__ B, long long ldb, const int* __restrict__ brows, int K, int Mtiles,
           GroupTiles gt, float* __restrict__ C, long long ldc, ScoreArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                    // [2][BM][AS_LD]
    float* Bs = lds + 2 * AS_SZ;        // [2][BK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int tile = xcd_tile_id(blockIdx.x, gridDim.x);
    const int mt = tile % Mtiles, nt = tile / Mtiles;
    const int m0 = mt * BM;
    const long long n0 = (long long)nt * BN;

    if (!SCORE) {
        int g = 0;
        while (g + 1 < gt.G && nt >= gt.start[g + 1]) ++g;
        A += (long long)g * a_group_stride;
    }

    // ---- global -> register staging maps (branch-free: out-of-range rows are clamped to a valid
    // row; their products are either multiplied by zero-padded operand entries or never stored)
    // A: 128 rows x 8 float4 (along k); thread handles (row = idx>>3, kq = idx&7), idx = tid + 256 q
    const int a_kq = (tid & 7) * 4;
    const float* a_p0 = A + (long long)min(m0 + (tid >> 3), Mrows - 1) * lda + a_kq;
    const float* a_p1 = A + (long long)min(m0 + 32 + (tid >> 3), Mrows - 1) * lda + a_kq;
    const float* a_p2 = A + (long long)min(m0 + 64 + (tid >> 3), Mrows - 1) * lda + a_kq;
    const float* a_p3 = A + (long long)min(m0 + 96 + (tid >> 3), Mrows - 1) * lda + a_kq;
    // B: thread handles k = 4*kg .. 4*kg+3 (kg = tid>>5) and columns n = 4*(tid&31) .. +3
    const int b_kg = tid >> 5, b_nq = tid & 31;
    const float* b_col = B + n0 + b_nq * 4;
    const int* b_rows = HAS_ROWS ? brows + b_kg * 4 : nullptr;

    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    int4 bidx = make_int4(0, 0, 0, 0);     // raw B row indices of the tile to be loaded next (one tile ahead)
#define LOAD_IDX(kt_) \
    if (HAS_ROWS) bidx = *reinterpret_cast<const int4*>(b_rows + (kt_) * BK)
    // -1 (padding) -> row 0: the matching A column is zero, so the product vanishes
#define B_ROW(e_, raw_) (HAS_ROWS ? max((raw_), 0) : k0_ + b_kg * 4 + (e_))
#define LOAD_TILE(kt_)                                                                   \
    {                                                                                    \
        const int k0_ = (kt_) * BK;                                                      \
        ra0 = *reinterpret_cast<const float4*>(a_p0 + k0_);                              \
        ra1 = *reinterpret_cast<const float4*>(a_p1 + k0_);                              \
        ra2 = *reinterpret_cast<const float4*>(a_p2 + k0_);                              \
        ra3 = *reinterpret_cast<const float4*>(a_p3 + k0_);                              \
        rb0 = *reinterpret_cast<const float4*>(b_col + (long long)B_ROW(0, bidx.x) * ldb); \
        rb1 = *reinterpret_cast<const float4*>(b_col + (long long)B_ROW(1, bidx.y) * ldb); \
        rb2 = *reinterpret_cast<const float4*>(b_col + (long long)B_ROW(2, bidx.z) * ldb); \
        rb3 = *reinterpret_cast<const float4*>(b_col + (long long)B_ROW(3, bidx.w) * ldb); \
    }
    const int a_woff = (tid >> 3) * AS_LD + a_kq;
    const int b_woff = b_kg * 4 * BN + b_nq * 4;
#define STORE_TILE(buf_)                                              \
    {                                                                 \
        float* as_ = As + (buf_) * AS_SZ + a_woff;                    \
        float* bs_ = Bs + (buf_) * BS_SZ + b_woff;                    \
        *reinterpret_cast<float4*>(as_) = ra0;                        \
        *reinterpret_cast<float4*>(as_ + 32 * AS_LD) = ra1;           \
        *reinterpret_cast<float4*>(as_ + 64 * AS_LD) = ra2;           \
        *reinterpret_cast<float4*>(as_ + 96 * AS_LD) = ra3;           \
        *reinterpret_cast<float4*>(bs_) = rb0;                        \
        *reinterpret_cast<float4*>(bs_ + BN) = rb1;                   \
        *reinterpret_cast<float4*>(bs_ + 2 * BN) = rb2;               \
        *reinterpret_cast<float4*>(bs_ + 3 * BN) = rb3;               \
    }

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

    const int KT = K / BK;
    LOAD_IDX(0);
    LOAD_TILE(0);
    if (KT > 1) { LOAD_IDX(1); }
    STORE_TILE(0);
    __syncthreads();

    const int a_frag = (wm * 64 + li) * AS_LD + lh * 4;          // + mi*32*AS_LD + g*8
    const int b_frag = lh * 4 * BN + wn * 64 + li;               // + (g*8 + t)*BN + ni*32

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) LOAD_TILE(kt + 1);
        if (kt + 2 < KT) { LOAD_IDX(kt + 2); }
        const float* as = As + cur * AS_SZ + a_frag;
        const float* bs = Bs + cur * BS_SZ + b_frag;
        // fragment reads run one 8-k group ahead of the MFMAs that consume them
        float4 a0, a1, b0, b1, c0, c1, d0, d1;
        READ_FRAG(0, a0, a1, b0, b1);
        READ_FRAG(1, c0, c1, d0, d1);
        MFMA_GROUP(a0, a1, b0, b1);
        READ_FRAG(2, a0, a1, b0, b1);
        MFMA_GROUP(c0, c1, d0, d1);
        READ_FRAG(3, c0, c1, d0, d1);
        MFMA_GROUP(a0, a1, b0, b1);
        MFMA_GROUP(c0, c1, d0, d1);
        // pin the issue order: two groups of fragment reads up front, then {16 MFMA, one group of
        // reads} x 2, then 32 MFMA (a group = 2 ds_read_b128 + 8 ds_read_b32, possibly paired)
        __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
        if (kt + 1 < KT) STORE_TILE(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  Accumulator map (32x32 block): column = lane & 31,
    // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const f32x16 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};
    if (!SCORE) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const long long col = n0 + wn * 64 + ni * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < Mrows) C[(long long)row * ldc + col] = acc[mi][ni][r];
                }
            }
        return;
    }

    const long long V = ldc;   // score mode: ldc carries the padded voxel count
    const bool corr = sa.mode == LC_SCORE_CORR;
    const lc::ep_f32x4 no_scale[4] = {};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rb0 = m0 + wm * 64 + mi * 32;          // first row of this 32-row block
        if (rb0 >= Mrows) continue;
        const int i0 = rb0 % sa.M;                       // offset inside the alpha's validation rows
        lc::EpiTargets t[2];
        float ym[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const long long col = n0 + wn * 64 + ni * 32 + li;
            lc::epi_load_targets(sa.yv, V, i0, lh, col, t[ni]);
            ym[ni] = sa.ymean[col];
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const long long col = n0 + wn * 64 + ni * 32 + li;
            lc::epi_block_dispatch<false>(corr, acc[mi][ni], t[ni], no_scale, 1.f, ym[ni], i0, sa.n_val, lh,
                                          sa.part + (long long)(rb0 >> 5) * 4 * V + col, V, true);
        }
    }
}

// Combine the per-32-row-block partial moments of one (alpha, voxel) in fixed order (fp64) and
// turn them into the reference's score:  corr = mean(z(y) z(pred)) with unbiased stds and the
// +1e-8 in both denominators (ridge_utils.py:6-15, ridge_regression.py:124-125), or signed
// sqrt|R2| (:126-130); then nan_to_num (:133) and accumulate over inner folds.
struct FoldCounts {
    int n_val[64];
};

// One thread per column walks the folds itself (the form for many columns; k_score_finalize_fw below for few).
__global__ void __launch_bounds__(256) k_score_finalize(const float* __restrict__ part_all, const float* __restrict__ ystat_all,
                                                        const float* __restrict__ yblk_all, int A, int M, FoldCounts fc,
                                                        int F, long long V, int mode, float* __restrict__ scores,
                                                        int accumulate, const int* __restrict__ live_cols = nullptr) {
    if (live_cols && (long long)blockIdx.x * blockDim.x >= (((long long)*live_cols + 255) & ~255ll)) return;   // (the refinement's panel: no voxel here)
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (v >= V) return;
    const int nblk = M / LC_MB;
  for (int fold = 0; fold < F; ++fold) {
    const int n_val = fc.n_val[fold];
    const float* part = part_all + (long long)fold * A * nblk * 4 * V;
    const float* ystat = ystat_all + (long long)fold * 3 * V;
    const float* yblk = yblk_all + (long long)fold * nblk * V;
    const float* p0 = part + (long long)a * nblk * 4 * V + v;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const int nb = min(LC_MB, n_val - b * LC_MB);
        if (nb <= 0) break;
        const double s1 = p0[(long long)b * 4 * V], q = p0[(long long)b * 4 * V + V];
        const double mb = s1 / nb, nn = n + nb, d = mb - mean;
        m2 += q + d * d * n * nb / nn;
        mean += d * nb / nn;
        n = nn;
    }
    float score;
    if (mode == LC_SCORE_CORR) {
        double cov = 0.0;
        for (int b = 0; b < nblk; ++b) {
            const int nb = min(LC_MB, n_val - b * LC_MB);
            if (nb <= 0) break;
            const double s1 = p0[(long long)b * 4 * V], s3 = p0[(long long)b * 4 * V + 2 * V];
            cov += s3 + (s1 / nb - mean) * (double)yblk[(long long)b * V + v];
        }
        const double sp = sqrt(m2 / (double)(n_val - 1));
        const double sy = (double)ystat[V + v];
        score = (float)(cov / ((double)n_val * (sy + 1e-8) * (sp + 1e-8)));
    } else {
        // moments of the raw targets, formed and merged exactly like those of the residual (lc_epilogue.h)
        double ny = 0.0, meany = 0.0, m2y = 0.0;
        for (int b = 0; b < nblk; ++b) {
            const int nb = min(LC_MB, n_val - b * LC_MB);
            if (nb <= 0) break;
            const double s1 = p0[(long long)b * 4 * V + 2 * V], q = p0[(long long)b * 4 * V + 3 * V];
            const double mb = s1 / nb, nn = ny + nb, d = mb - meany;
            m2y += q + d * d * ny * nb / nn;
            meany += d * nb / nn;
            ny = nn;
        }
        const float resvar = (float)(m2 / (double)(n_val - 1));
        const float yvar_same = (float)(m2y / (double)(n_val - 1));
        // residual statistically identical to the targets under identical arithmetic: the reference's two
        // torch.var calls agree bit for bit as well and give Rsq = 0 (or 0/0 -> NaN -> 0 for constant targets)
        const float yvar = ystat[2 * V + v];
        const float rsq = resvar == yvar_same ? (yvar == 0.f ? __builtin_nanf("") : 0.f) : 1.f - resvar / yvar;
        const float sgn = rsq > 0.f ? 1.f : (rsq < 0.f ? -1.f : rsq);   // sign(NaN) = NaN, sign(0) = 0
        score = sqrtf(fabsf(rsq)) * sgn;
    }
    // torch.nan_to_num defaults: NaN -> 0, +-inf -> +-FLT_MAX
    if (score != score) score = 0.f;
    else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
    else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
    float* dst = scores + (long long)a * V + v;
    *dst = (accumulate || fold > 0) ? *dst + score : score;
  }
}


// F folds (slices f of part / ystat / yblk): the folds' scores are added in fp32, fold order (nested_cv.py:373-380).
// Block = 64 columns x FIN_FW fold workers: a worker merges the blocks of ITS fold (a latency chain of ~45 strided
// loads), the scores meet in LDS and are summed in fold order -- one thread per column doing all folds one after the
// other left most of the chip idle at 10 000 voxels per rank.
constexpr int FIN_FW = 8;
constexpr long long FIN_WIDE_V = 32768;
// <FW, COLS>: <8, 64> below ~32 000 columns (the fold workers fill the chip), <1, 256> above (one thread per column
// walks the folds itself: with enough columns that is the cheaper form -- no idle workers when F < 8, no barriers)
template <int FW, int COLS>
__global__ void __launch_bounds__(COLS * FW) k_score_finalize_fw(const float* __restrict__ part_all, const float* __restrict__ ystat_all,
                                                        const float* __restrict__ yblk_all, int A, int M, FoldCounts fc,
                                                        int F, long long V, int mode, float* __restrict__ scores,
                                                        int accumulate, const int* __restrict__ live_cols = nullptr) {
    if (live_cols && (long long)blockIdx.x * COLS >= (((long long)*live_cols + 255) & ~255ll)) return;   // (the refinement's panel: no voxel here)
    __shared__ float sc_lds[FW][COLS];
    const long long v = (long long)blockIdx.x * COLS + threadIdx.x;
    const int a = blockIdx.y, fy = threadIdx.y;
    const int nblk = M / LC_MB;
  for (int fc0 = 0; fc0 < F; fc0 += FW) {
    const int fold = fc0 + fy;
    float score = 0.f;
    if (fold < F && v < V) {
    const int n_val = fc.n_val[fold];
    const float* part = part_all + (long long)fold * A * nblk * 4 * V;
    const float* ystat = ystat_all + (long long)fold * 3 * V;
    const float* yblk = yblk_all + (long long)fold * nblk * V;
    const float* p0 = part + (long long)a * nblk * 4 * V + v;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const int nb = min(LC_MB, n_val - b * LC_MB);
        if (nb <= 0) break;
        const double s1 = p0[(long long)b * 4 * V], q = p0[(long long)b * 4 * V + V];
        const double mb = s1 / nb, nn = n + nb, d = mb - mean;
        m2 += q + d * d * n * nb / nn;
        mean += d * nb / nn;
        n = nn;
    }
    if (mode == LC_SCORE_CORR) {
        double cov = 0.0;
        for (int b = 0; b < nblk; ++b) {
            const int nb = min(LC_MB, n_val - b * LC_MB);
            if (nb <= 0) break;
            const double s1 = p0[(long long)b * 4 * V], s3 = p0[(long long)b * 4 * V + 2 * V];
            cov += s3 + (s1 / nb - mean) * (double)yblk[(long long)b * V + v];
        }
        const double sp = sqrt(m2 / (double)(n_val - 1));
        const double sy = (double)ystat[V + v];
        score = (float)(cov / ((double)n_val * (sy + 1e-8) * (sp + 1e-8)));
    } else {
        // moments of the raw targets, formed and merged exactly like those of the residual (lc_epilogue.h)
        double ny = 0.0, meany = 0.0, m2y = 0.0;
        for (int b = 0; b < nblk; ++b) {
            const int nb = min(LC_MB, n_val - b * LC_MB);
            if (nb <= 0) break;
            const double s1 = p0[(long long)b * 4 * V + 2 * V], q = p0[(long long)b * 4 * V + 3 * V];
            const double mb = s1 / nb, nn = ny + nb, d = mb - meany;
            m2y += q + d * d * ny * nb / nn;
            meany += d * nb / nn;
            ny = nn;
        }
        const float resvar = (float)(m2 / (double)(n_val - 1));
        const float yvar_same = (float)(m2y / (double)(n_val - 1));
        // residual statistically identical to the targets under identical arithmetic: the reference's two
        // torch.var calls agree bit for bit as well and give Rsq = 0 (or 0/0 -> NaN -> 0 for constant targets)
        const float yvar = ystat[2 * V + v];
        const float rsq = resvar == yvar_same ? (yvar == 0.f ? __builtin_nanf("") : 0.f) : 1.f - resvar / yvar;
        const float sgn = rsq > 0.f ? 1.f : (rsq < 0.f ? -1.f : rsq);   // sign(NaN) = NaN, sign(0) = 0
        score = sqrtf(fabsf(rsq)) * sgn;
    }
    // torch.nan_to_num defaults: NaN -> 0, +-inf -> +-FLT_MAX
    if (score != score) score = 0.f;
    else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
    else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
    }
    if (FW == 1) {                                       // one thread per column: add the fold's score directly
        if (v < V) {
            float* dst = scores + (long long)a * V + v;
            *dst = (accumulate || fc0 > 0) ? *dst + score : score;
        }
        continue;
    }
    sc_lds[fy][threadIdx.x] = score;
    __syncthreads();
    if (fy == 0 && v < V) {
        float* dst = scores + (long long)a * V + v;
        float acc = (accumulate || fc0 > 0) ? *dst : 0.f;            // 0 + x = x: the first fold's score as it is
        for (int f = 0; f < FW && fc0 + f < F; ++f) acc += sc_lds[f][threadIdx.x];
        *dst = acc;
    }
    __syncthreads();
  }
}

// ---- correlation scores of the series alphas from the moments of the shared terms ------------------------------
// For the alphas on the polynomial series the prediction is a fixed linear combination of `terms` matrices
//     pred_s = sum_j c_sj T_j,   T_j = P'_j Y (M x V, one GEMM for all those alphas),   c_sj given per alpha
// (minimax polynomial of 1 / (x + alpha^2), series.py), so every statistic the score needs is a linear or quadratic form in the moments of the
// T_j over the validation rows:  mean_s = c.m,  M2_s = c' S c,  cov_s = c.C  (S the terms' scatter matrix, C their
// co-moments with y).  One pass over T in fp64 (shifted by the first row) gives all of them for every alpha.
// Block: 64 columns x 4 row groups; same shift in every group, so the groups' raw sums simply add.
template <int TERMS, bool MAPPED>
__global__ void __launch_bounds__(256) k_series_scores(const float* __restrict__ T, long long ldt, int M, int n_val,
                                                       long long V, const float* __restrict__ yv,
                                                       const float* __restrict__ ystat, const double* __restrict__ coefs,
                                                       const int* __restrict__ aidx, int S,
                                                       const int* __restrict__ rowmap, float* __restrict__ scores,
                                                       int accumulate) {
    constexpr int NB2 = TERMS * (TERMS + 1) / 2, NACC = 2 * TERMS + NB2 + 1;
    __shared__ double red[3][NACC][64];
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const long long c = (long long)blockIdx.x * 64 + lane;
    const bool live = c < V;
    const long long cc = live ? c : 0;
    const float* tcol = T + cc;
    const long long tstride = ldt;
    double sh[TERMS], a[TERMS], cy[TERMS], b[NB2], ay = 0.0;
#pragma unroll
    for (int j = 0; j < TERMS; ++j) {
        sh[j] = (double)tcol[(long long)(MAPPED ? rowmap[j * M] : j * M) * tstride];
        a[j] = 0.0;
        cy[j] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < NB2; ++k) b[k] = 0.0;
    const double shy = (double)yv[lc::yv_index(0, cc, V)];
    // rows g, g+4, ...: four rows per trip with all their loads issued first (the loop is latency bound)
    constexpr int UR = 4;
    for (int i0 = g; i0 < n_val; i0 += 4 * UR) {
        float tv[UR][TERMS], yvv[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int i = min(i0 + 4 * u, n_val - 1);        // clamped: the extra rows are masked below
            yvv[u] = yv[lc::yv_index(i, cc, V)];
#pragma unroll
            for (int j = 0; j < TERMS; ++j)                  // row index: wave-uniform, a scalar load
                tv[u][j] = tcol[(long long)(MAPPED ? rowmap[j * M + i] : j * M + i) * tstride];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            if (i0 + 4 * u >= n_val) break;
            const double dy = (double)yvv[u] - shy;
            double d[TERMS];
#pragma unroll
            for (int j = 0; j < TERMS; ++j) d[j] = (double)tv[u][j] - sh[j];
            ay += dy;
            int k = 0;
#pragma unroll
            for (int j = 0; j < TERMS; ++j) {
                a[j] += d[j];
                cy[j] += d[j] * dy;
#pragma unroll
                for (int l = j; l < TERMS; ++l) b[k++] += d[j] * d[l];
            }
        }
    }
    if (g > 0) {
#pragma unroll
        for (int j = 0; j < TERMS; ++j) { red[g - 1][j][lane] = a[j]; red[g - 1][TERMS + j][lane] = cy[j]; }
#pragma unroll
        for (int k = 0; k < NB2; ++k) red[g - 1][2 * TERMS + k][lane] = b[k];
        red[g - 1][NACC - 1][lane] = ay;
    }
    __syncthreads();
    if (g != 0 || !live) return;
    for (int q = 0; q < 3; ++q) {                        // fixed order: deterministic
#pragma unroll
        for (int j = 0; j < TERMS; ++j) { a[j] += red[q][j][lane]; cy[j] += red[q][TERMS + j][lane]; }
#pragma unroll
        for (int k = 0; k < NB2; ++k) b[k] += red[q][2 * TERMS + k][lane];
        ay += red[q][NACC - 1][lane];
    }
    // centred moments
    const double n = (double)n_val;
    {
        int k = 0;
#pragma unroll
        for (int j = 0; j < TERMS; ++j) {
            cy[j] -= a[j] * ay / n;
#pragma unroll
            for (int l = j; l < TERMS; ++l) { b[k] -= a[j] * a[l] / n; ++k; }
        }
    }
    const double sy = (double)ystat[V + c];
    for (int s = 0; s < S; ++s) {
        double coef[TERMS];
#pragma unroll
        for (int j = 0; j < TERMS; ++j) coef[j] = coefs[s * TERMS + j];
        double m2 = 0.0, cov = 0.0;
        int k = 0;
#pragma unroll
        for (int j = 0; j < TERMS; ++j) {
            cov += coef[j] * cy[j];
#pragma unroll
            for (int l = j; l < TERMS; ++l) { m2 += (l == j ? 1.0 : 2.0) * coef[j] * coef[l] * b[k]; ++k; }
        }
        const double sp = sqrt(fmax(m2, 0.0) / (n - 1.0));
        float score = (float)(cov / (n * (sy + 1e-8) * (sp + 1e-8)));
        if (score != score) score = 0.f;
        else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
        else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
        float* dst = scores + (long long)aidx[s] * V + c;
        *dst = accumulate ? *dst + score : score;
    }
}

// The same scores from the per-32-row-block partial moments the series-moments epilogue of the fp16x3 kernel writes
// (lc::epi_series_block; part: (M / 32, 18, V) f32): blocks merged in fixed order in fp64 (Chan), then the forms.
__global__ void __launch_bounds__(256) k_series_scores_part(const float* __restrict__ part_all, const float* __restrict__ ystat_all,
                                                            const float* __restrict__ yblk_all, int M, FoldCounts fc, int F,
                                                            long long V, const double* __restrict__ coefs,
                                                            const int* __restrict__ aidx, int S,
                                                            float* __restrict__ scores, int accumulate, const int* __restrict__ live_cols = nullptr) {
    if (live_cols && (long long)blockIdx.x * blockDim.x >= (((long long)*live_cols + 255) & ~255ll)) return;   // (the refinement's panel: no voxel here)
    constexpr int NP = lc::EPI_SERIES_PARTS;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // 64 columns per workgroup: one thread per
    if (c >= V) return;                                                      // column is a dependent chain over the blocks,
    const int nblk = M / LC_MB;                                              // so the chip is filled with MANY small workgroups
  for (int fold = 0; fold < F; ++fold) {
    const int n_val = fc.n_val[fold];
    const float* part = part_all + (long long)fold * nblk * NP * V;
    const float* ystat = ystat_all + (long long)fold * 3 * V;
    const float* yblk = yblk_all + (long long)fold * nblk * V;
    double n = 0.0, mean[4] = {0.0, 0.0, 0.0, 0.0}, sc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) sc[k] = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const int nb = min(LC_MB, n_val - b * LC_MB);
        if (nb <= 0) break;
        const float* p0 = part + (long long)b * NP * V + c;
        const double nn = n + nb, w = n * nb / nn;
        double d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (double)p0[(long long)j * V] / nb - mean[j];
        int k = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int l = j; l < 4; ++l) {
                sc[k] += (double)p0[(long long)(8 + k) * V] + d[j] * d[l] * w;
                ++k;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) mean[j] += d[j] * nb / nn;
        n = nn;
    }
    double cy[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < nblk; ++b) {
        const int nb = min(LC_MB, n_val - b * LC_MB);
        if (nb <= 0) break;
        const float* p0 = part + (long long)b * NP * V + c;
        const double yb = (double)yblk[(long long)b * V + c];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cy[j] += (double)p0[(long long)(4 + j) * V] + ((double)p0[(long long)j * V] / nb - mean[j]) * yb;
    }
    const double nv = (double)n_val, sy = (double)ystat[V + c];
    for (int s = 0; s < S; ++s) {
        double coef[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) coef[j] = coefs[s * 4 + j];
        double m2 = 0.0, cov = 0.0;
        int k = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cov += coef[j] * cy[j];
#pragma unroll
            for (int l = j; l < 4; ++l) { m2 += (l == j ? 1.0 : 2.0) * coef[j] * coef[l] * sc[k]; ++k; }
        }
        const double sp = sqrt(fmax(m2, 0.0) / (nv - 1.0));
        float score = (float)(cov / (nv * (sy + 1e-8) * (sp + 1e-8)));
        if (score != score) score = 0.f;
        else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
        else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
        float* dst = scores + (long long)aidx[s] * V + c;
        *dst = (accumulate || fold > 0) ? *dst + score : score;
    }
  }
}


template <int FW, int COLS>
__global__ void __launch_bounds__(COLS * FW) k_series_scores_part_fw(const float* __restrict__ part_all, const float* __restrict__ ystat_all,
                                                            const float* __restrict__ yblk_all, int M, FoldCounts fc, int F,
                                                            long long V, const double* __restrict__ coefs,
                                                            const int* __restrict__ aidx, int S,
                                                            float* __restrict__ scores, int accumulate, const int* __restrict__ live_cols = nullptr) {
    if (live_cols && (long long)blockIdx.x * COLS >= (((long long)*live_cols + 255) & ~255ll)) return;   // (the refinement's panel: no voxel here)
    constexpr int NP = lc::EPI_SERIES_PARTS;
    __shared__ float sc_lds[FW][FIN_FW][COLS];             // [fold worker][alpha of the chunk][column]
    const long long c = (long long)blockIdx.x * COLS + threadIdx.x;
    const int fy = threadIdx.y;
    const int nblk = M / LC_MB;
  for (int fc0 = 0; fc0 < F; fc0 += FW) {
    const int fold = fc0 + fy;
    const bool live = fold < F && c < V;
    double mean[4] = {0.0, 0.0, 0.0, 0.0}, sc[10], cy[4] = {0.0, 0.0, 0.0, 0.0}, nv = 1.0, sy = 0.0;
#pragma unroll
    for (int k = 0; k < 10; ++k) sc[k] = 0.0;
    if (live) {
        const int n_val = fc.n_val[fold];
        const float* part = part_all + (long long)fold * nblk * NP * V;
        const float* ystat = ystat_all + (long long)fold * 3 * V;
        const float* yblk = yblk_all + (long long)fold * nblk * V;
        double n = 0.0;
        for (int b = 0; b < nblk; ++b) {
            const int nb = min(LC_MB, n_val - b * LC_MB);
            if (nb <= 0) break;
            const float* p0 = part + (long long)b * NP * V + c;
            const double nn = n + nb, w = n * nb / nn;
            double d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = (double)p0[(long long)j * V] / nb - mean[j];
            int k = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = j; l < 4; ++l) {
                    sc[k] += (double)p0[(long long)(8 + k) * V] + d[j] * d[l] * w;
                    ++k;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) mean[j] += d[j] * nb / nn;
            n = nn;
        }
        for (int b = 0; b < nblk; ++b) {
            const int nb = min(LC_MB, n_val - b * LC_MB);
            if (nb <= 0) break;
            const float* p0 = part + (long long)b * NP * V + c;
            const double yb = (double)yblk[(long long)b * V + c];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                cy[j] += (double)p0[(long long)(4 + j) * V] + ((double)p0[(long long)j * V] / nb - mean[j]) * yb;
        }
        nv = (double)n_val;
        sy = (double)ystat[V + c];
    }
    for (int s0 = 0; s0 < S; s0 += FIN_FW) {
#pragma unroll
        for (int q = 0; q < FIN_FW; ++q) {
            const int s = s0 + q;
            float score = 0.f;
            if (live && s < S) {
                double coef[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) coef[j] = coefs[s * 4 + j];
                double m2 = 0.0, cov = 0.0;
                int k = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cov += coef[j] * cy[j];
#pragma unroll
                    for (int l = j; l < 4; ++l) { m2 += (l == j ? 1.0 : 2.0) * coef[j] * coef[l] * sc[k]; ++k; }
                }
                const double sp = sqrt(fmax(m2, 0.0) / (nv - 1.0));
                score = (float)(cov / (nv * (sy + 1e-8) * (sp + 1e-8)));
                if (score != score) score = 0.f;
                else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
                else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
            }
            if (FW == 1) {
                if (live && s < S) {
                    float* dst = scores + (long long)aidx[s] * V + c;
                    *dst = (accumulate || fc0 > 0) ? *dst + score : score;
                }
            } else {
                sc_lds[fy][q][threadIdx.x] = score;
            }
        }
        if (FW == 1) continue;
        __syncthreads();
        for (int q = fy; q < FIN_FW; q += FW) {             // worker fy sums alphas s0 + fy, + FW, ... over the chunk's folds
            const int s = s0 + q;
            if (s < S && c < V) {
                float* dst = scores + (long long)aidx[s] * V + c;
                float acc = (accumulate || fc0 > 0) ? *dst : 0.f;
                for (int f = 0; f < FW && fc0 + f < F; ++f) acc += sc_lds[f][q][threadIdx.x];
                *dst = acc;
            }
        }
        __syncthreads();
    }
  }
}

int check_gemm_shapes(const char* who, const void* a, long long lda, const void* b, long long ldb, long long Ncols,
                      long long K) {
    LC_REQUIRE(K > 0 && K % BK == 0, LC_E_SHAPE, "%s: K=%lld must be a positive multiple of %d", who, K, BK);
    LC_REQUIRE(Ncols > 0 && Ncols % BN == 0, LC_E_SHAPE, "%s: column count %lld must be a positive multiple of %d", who,
               Ncols, BN);
    LC_REQUIRE(lda % 4 == 0 && ldb % 4 == 0, LC_E_SHAPE, "%s: leading dimensions must be multiples of 4", who);
    LC_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0), LC_E_BADARG, "%s: operands must be 16-byte aligned",
               who);
    return LC_OK;
}

template <bool SCORE, bool HAS_ROWS>
int set_lds_attr() {
    return lc::ensure_dynamic_lds(reinterpret_cast<const void*>(k_gemm_f32<SCORE, HAS_ROWS>), GEMM_LDS_BYTES);
}

}  // namespace

// shared with the fp16x3 variant (lc_gemm16.hip)
int lc_score_finalize_launch(const float* d_part, const float* d_ystat, const float* d_yblk, int A, int M,
                             const int* h_n_val, int F, long long V, int mode, float* d_scores, int accumulate,
                             hipStream_t s, const int* d_live_cols) {
    FoldCounts fc{};
    for (int f = 0; f < F && f < 64; ++f) fc.n_val[f] = h_n_val[f];
    lc::ScopedTimer timer_(lc::T_SWEEP_FINALIZE, s);
    if (V < FIN_WIDE_V && F > 1)
        hipLaunchKernelGGL((k_score_finalize_fw<FIN_FW, 64>), dim3((unsigned)lc::ceil_div<long long>(V, 64), (unsigned)A),
                           dim3(64, FIN_FW), 0, s, d_part, d_ystat, d_yblk, A, M, fc, F, V, mode, d_scores, accumulate, d_live_cols);
    else
        hipLaunchKernelGGL(k_score_finalize, dim3((unsigned)lc::ceil_div<long long>(V, 256), (unsigned)A), dim3(256), 0, s,
                           d_part, d_ystat, d_yblk, A, M, fc, F, V, mode, d_scores, accumulate, d_live_cols);
    return lc::launched("k_score_finalize");
}

int lc_series_finalize_launch(const float* d_part, const float* d_ystat, const float* d_yblk, int M, const int* h_n_val,
                              int F, long long V, const double* d_coef, const int* d_aidx, int S, float* d_scores,
                              int accumulate, hipStream_t s, const int* d_live_cols) {
    FoldCounts fc{};
    for (int f = 0; f < F && f < 64; ++f) fc.n_val[f] = h_n_val[f];
    lc::ScopedTimer timer_(lc::T_SWEEP_FINALIZE, s);
    if (V < FIN_WIDE_V && F > 1)
        hipLaunchKernelGGL((k_series_scores_part_fw<FIN_FW, 64>), dim3((unsigned)lc::ceil_div<long long>(V, 64)), dim3(64, FIN_FW),
                           0, s, d_part, d_ystat, d_yblk, M, fc, F, V, d_coef, d_aidx, S, d_scores, accumulate, d_live_cols);
    else
        hipLaunchKernelGGL(k_series_scores_part, dim3((unsigned)lc::ceil_div<long long>(V, 64)), dim3(64), 0, s, d_part,
                           d_ystat, d_yblk, M, fc, F, V, d_coef, d_aidx, S, d_scores, accumulate, d_live_cols);
    return lc::launched("k_series_scores_part");
}

extern "C" int lc_alpha_sweep_scores(const float* d_h, int A, int M, int N, const float* d_y, int64_t ldy, int64_t V,
                                     const int32_t* d_tr, const float* d_yv, int n_val, const float* d_ystat,
                                     const float* d_yblk, int mode, float* d_part, float* d_scores, int accumulate,
                                     lc_stream_t stream) {
    LC_REQUIRE(d_h && d_y && d_tr && d_yv && d_ystat && d_yblk && d_part && d_scores, LC_E_BADARG,
               "lc_alpha_sweep_scores: null pointer");
    LC_REQUIRE(A > 0 && M > 0 && M % LC_MB == 0 && n_val > 0 && n_val <= M, LC_E_SHAPE,
               "lc_alpha_sweep_scores: need M %% %d == 0 and 0 < n_val <= M", LC_MB);
    LC_REQUIRE(mode == LC_SCORE_CORR || mode == LC_SCORE_R2, LC_E_BADARG, "lc_alpha_sweep_scores: bad mode %d", mode);
    LC_REQUIRE(ldy >= V, LC_E_SHAPE, "lc_alpha_sweep_scores: ldy < V");
    if (int rc = check_gemm_shapes("lc_alpha_sweep_scores", d_h, N, d_y, ldy, V, N)) return rc;
    if (int rc = set_lds_attr<true, true>()) return rc;
    hipStream_t s = lc::as_stream(stream);
    const int Mrows = A * M;
    const int Mtiles = lc::ceil_div(Mrows, BM);
    const long long Ntiles = V / BN;
    LC_REQUIRE((long long)Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_alpha_sweep_scores: grid too large");
    GroupTiles gt;
    gt.G = 1;
    gt.start[0] = 0;
    gt.start[1] = (int)Ntiles;
    ScoreArgs sa{d_yv, d_ystat, d_part, M, n_val, mode};
    {
        lc::ScopedTimer timer_(lc::T_SWEEP_GEMM, s);
        hipLaunchKernelGGL((k_gemm_f32<true, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(256), GEMM_LDS_BYTES, s, d_h,
                           (long long)N, 0ll, Mrows, d_y, (long long)ldy, d_tr, N, Mtiles, gt, (float*)nullptr,
                           (long long)V, sa);
    }
    if (int rc = lc::launched("k_gemm_f32<score>")) return rc;
    return lc_score_finalize_launch(d_part, d_ystat, d_yblk, A, M, &n_val, 1, (long long)V, mode, d_scores, accumulate, s, nullptr);
}

extern "C" int lc_series_scores(const float* d_t, int64_t ldt, int terms, int M, int n_val, int64_t V, const float* d_yv,
                                const float* d_ystat, const double* d_coef, const int32_t* d_aidx,
                                int S, const int32_t* d_rowmap, float* d_scores, int accumulate, lc_stream_t stream) {
    LC_REQUIRE(d_t && d_yv && d_ystat && d_coef && d_aidx && d_scores, LC_E_BADARG,
               "lc_series_scores: null pointer");
    LC_REQUIRE(terms >= 1 && terms <= 8 && M > 0 && M % LC_MB == 0 && n_val > 1 && n_val <= M && V > 0 && ldt >= V &&
                   S > 0,
               LC_E_SHAPE, "lc_series_scores: need 1 <= terms <= 8, M %% %d == 0, 1 < n_val <= M, ldt >= V", LC_MB);
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_SWEEP_FINALIZE, s);
    const dim3 grid((unsigned)lc::ceil_div<long long>(V, 64)), block(256);
#define LC_SERIES_CASE(t_)                                                                                             \
    case t_:                                                                                                           \
        if (d_rowmap)                                                                                                  \
            hipLaunchKernelGGL((k_series_scores<t_, true>), grid, block, 0, s, d_t, (long long)ldt, M, n_val,          \
                               (long long)V, d_yv, d_ystat, d_coef, d_aidx, S, d_rowmap, d_scores, accumulate);        \
        else                                                                                                           \
            hipLaunchKernelGGL((k_series_scores<t_, false>), grid, block, 0, s, d_t, (long long)ldt, M, n_val,         \
                               (long long)V, d_yv, d_ystat, d_coef, d_aidx, S, d_rowmap, d_scores, accumulate);        \
        break;
    switch (terms) {
        LC_SERIES_CASE(1) LC_SERIES_CASE(2) LC_SERIES_CASE(3) LC_SERIES_CASE(4)
        LC_SERIES_CASE(5) LC_SERIES_CASE(6) LC_SERIES_CASE(7) LC_SERIES_CASE(8)
    }
#undef LC_SERIES_CASE
    return lc::launched("k_series_scores");
}

extern "C" int lc_gemm_grouped_f32(const float* d_a, int64_t lda, int64_t a_group_stride, const float* d_b, int64_t ldb,
                                   const int32_t* d_brows, float* d_c, int64_t ldc, int64_t Mrows, int64_t Ncols,
                                   int64_t K, const int32_t* h_group_tiles, int G, lc_stream_t stream) {
    LC_REQUIRE(d_a && d_b && d_c && h_group_tiles, LC_E_BADARG, "lc_gemm_grouped_f32: null pointer");
    LC_REQUIRE(G >= 1 && G <= MAX_GROUPS, LC_E_SHAPE, "lc_gemm_grouped_f32: G must be in 1..%d", MAX_GROUPS);
    LC_REQUIRE(Mrows > 0 && Mrows < (1ll << 31) && ldc >= Ncols, LC_E_SHAPE, "lc_gemm_grouped_f32: bad M / ldc");
    if (int rc = check_gemm_shapes("lc_gemm_grouped_f32", d_a, lda, d_b, ldb, Ncols, K)) return rc;
    const long long Ntiles = Ncols / BN;
    GroupTiles gt;
    gt.G = G;
    for (int g = 0; g <= G; ++g) {
        gt.start[g] = h_group_tiles[g];
        LC_REQUIRE(g == 0 ? gt.start[0] == 0 : gt.start[g] >= gt.start[g - 1], LC_E_SHAPE,
                   "lc_gemm_grouped_f32: group tile offsets must start at 0 and be non-decreasing");
    }
    LC_REQUIRE(gt.start[G] == Ntiles, LC_E_SHAPE, "lc_gemm_grouped_f32: last group offset %d != %lld column tiles",
               gt.start[G], Ntiles);
    const int Mtiles = (int)lc::ceil_div<long long>(Mrows, BM);
    LC_REQUIRE((long long)Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_gemm_grouped_f32: grid too large");
    ScoreArgs sa{};
    lc::ScopedTimer timer_(lc::T_GROUPED_GEMM, lc::as_stream(stream));
    if (d_brows) {
        if (int rc = set_lds_attr<false, true>()) return rc;
        hipLaunchKernelGGL((k_gemm_f32<false, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(256), GEMM_LDS_BYTES,
                           lc::as_stream(stream), d_a, (long long)lda, (long long)a_group_stride, (int)Mrows, d_b,
                           (long long)ldb, d_brows, (int)K, Mtiles, gt, d_c, (long long)ldc, sa);
    } else {
        if (int rc = set_lds_attr<false, false>()) return rc;
        hipLaunchKernelGGL((k_gemm_f32<false, false>), dim3((unsigned)(Mtiles * Ntiles)), dim3(256), GEMM_LDS_BYTES,
                           lc::as_stream(stream), d_a, (long long)lda, (long long)a_group_stride, (int)Mrows, d_b,
                           (long long)ldb, d_brows, (int)K, Mtiles, gt, d_c, (long long)ldc, sa);
    }
    return lc::launched("k_gemm_f32<plain>");
}
