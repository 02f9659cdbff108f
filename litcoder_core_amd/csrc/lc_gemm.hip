// The V-wide contractions of the ridge fit on the f32-input MFMA (v_mfma_f32_32x32x2_f32):
//   * fused alpha sweep: pred_a = H_a . Y[train rows] with the z-score / correlation (or R2)
//     reduction done in the epilogue -- predictions never reach HBM;
//   * grouped plain GEMM for the refit weights and the test predictions.
//
// Tile: 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA blocks.
// Operand feeding: the MFMA consumes two k per issue (lane half h = lane>>5 supplies one).  We
// let half h own k = 4h..4h+3 of every 8-k group, so ONE ds_read_b128 per operand block feeds
// four MFMAs (component t of the float4 pairs k = t (h=0) with k = 4+t (h=1)).  A is staged
// row-major with a 36-float row stride (conflict-free b128 reads), B is staged k-interleaved
// [k/4][n][k%4] via a 4x4 register transpose so its fragment is also one b128.
#include "lc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int AS_LD = BK + 4;                     // floats per staged A row
constexpr int AS_SZ = BM * AS_LD;                 // floats per A stage
constexpr int BS_SZ = (BK / 4) * BN * 4;          // floats per B stage
constexpr int GEMM_LDS_BYTES = 2 * (AS_SZ + BS_SZ) * 4;
constexpr int MAX_GROUPS = 64;

struct GroupTiles {
    int start[MAX_GROUPS + 1];   // first column tile of each group; start[G] = number of tiles
    int G;
};

struct ScoreArgs {
    const float* y;        // targets (T, ldy)
    long long ldy;
    const int* va;         // validation rows (M entries, first n_val valid)
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

template <bool SCORE>
__global__ void __launch_bounds__(256, 2)
k_gemm_f32(const float* __restrict__ A, long long lda, long long a_group_stride, int Mrows,
           const float* __restrict__ B, long long ldb, const int* __restrict__ brows, int K, int Mtiles,
           GroupTiles gt, float* __restrict__ C, long long ldc, ScoreArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                    // [2][BM][AS_LD]
    float* Bs = lds + 2 * AS_SZ;        // [2][BK/4][BN][4]

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

    // ---- global -> register staging maps
    // A: 128 rows x 8 float4 (along k); thread handles (row = idx>>3, kq = idx&7), idx = tid + 256 q
    const float* a_ptr[4];
    bool a_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        const int row = m0 + (idx >> 3);
        a_ok[q] = row < Mrows;
        a_ptr[q] = A + (long long)(a_ok[q] ? row : 0) * lda + (idx & 7) * 4;
    }
    // B: thread handles k = 4*kg .. 4*kg+3 (kg = tid>>5) and columns n = 4*(tid&31) .. +3
    const int b_kg = tid >> 5, b_nq = tid & 31;
    const float* b_col = B + n0 + b_nq * 4;

    float4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ra[q] = a_ok[q] ? *reinterpret_cast<const float4*>(a_ptr[q] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + b_kg * 4 + e;
            const int r = brows ? brows[k] : k;
            rb[e] = r >= 0 ? *reinterpret_cast<const float4*>(b_col + (long long)r * ldb)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * AS_SZ;
        float* bs = Bs + buf * BS_SZ;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            *reinterpret_cast<float4*>(as + (idx >> 3) * AS_LD + (idx & 7) * 4) = ra[q];
        }
        float* dst = bs + (b_kg * BN + b_nq * 4) * 4;
        *reinterpret_cast<float4*>(dst + 0) = make_float4(rb[0].x, rb[1].x, rb[2].x, rb[3].x);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(rb[0].y, rb[1].y, rb[2].y, rb[3].y);
        *reinterpret_cast<float4*>(dst + 8) = make_float4(rb[0].z, rb[1].z, rb[2].z, rb[3].z);
        *reinterpret_cast<float4*>(dst + 12) = make_float4(rb[0].w, rb[1].w, rb[2].w, rb[3].w);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int a_frag = (wm * 64 + li) * AS_LD + lh * 4;          // + mi*32*AS_LD + g*8
    const int b_frag = (lh * BN + wn * 64 + li) * 4;             // + g*2*BN*4 + ni*32*4

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_tile(kt + 1);
        const float* as = As + cur * AS_SZ + a_frag;
        const float* bs = Bs + cur * BS_SZ + b_frag;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 fa[2], fb[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fa[mi] = *reinterpret_cast<const float4*>(as + mi * 32 * AS_LD + g * 8);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[ni] = *reinterpret_cast<const float4*>(bs + g * 2 * BN * 4 + ni * 32 * 4);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].x, fb[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].y, fb[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].z, fb[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].w, fb[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        if (kt + 1 < KT) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  Accumulator map (32x32 block): column = lane & 31,
    // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
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
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rb0 = m0 + wm * 64 + mi * 32;          // first row of this 32-row block
        if (rb0 >= Mrows) continue;
        const int alpha = rb0 / sa.M;
        const int i0 = rb0 - alpha * sa.M;               // offset inside the alpha's validation rows
        const int nb = min(32, sa.n_val - i0);           // valid rows in the block (<= 0: padding only)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const long long col = n0 + wn * 64 + ni * 32 + li;
            float p[16], yc[16];
            const float ym = sa.ymean[col];
            float s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bool ok = il < sa.n_val;
                const float yraw = ok ? sa.y[(long long)sa.va[il] * sa.ldy + col] : 0.f;
                const float pv = acc[mi][ni][r];
                yc[r] = ok ? yraw - ym : 0.f;
                // corr: statistics of pred; R2: statistics of the residual fl32(y - pred), formed from
                // the raw target exactly as ``(Presp - pred).var()`` does (ridge_regression.py:128)
                p[r] = ok ? (sa.mode == LC_SCORE_CORR ? pv : yraw - pv) : 0.f;
                s1 += p[r];
            }
            s1 += __shfl_xor(s1, 32);
            const float mean_b = nb > 0 ? s1 / (float)nb : 0.f;
            float m2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float d = il < sa.n_val ? p[r] - mean_b : 0.f;
                m2 += d * d;
                s3 += d * yc[r];
            }
            m2 += __shfl_xor(m2, 32);
            s3 += __shfl_xor(s3, 32);
            if (lh == 0) {
                float* out = sa.part + (long long)(rb0 >> 5) * 4 * V + col;
                out[0] = s1;
                out[V] = m2;
                out[2 * V] = s3;
            }
        }
    }
}

// Combine the per-32-row-block partial moments of one (alpha, voxel) in fixed order (fp64) and
// turn them into the reference's score:  corr = mean(z(y) z(pred)) with unbiased stds and the
// +1e-8 in both denominators (ridge_utils.py:6-15, ridge_regression.py:124-125), or signed
// sqrt|R2| (:126-130); then nan_to_num (:133) and accumulate over inner folds.
__global__ void __launch_bounds__(256) k_score_finalize(const float* __restrict__ part, const float* __restrict__ ystat,
                                                        const float* __restrict__ yblk, int A, int M, int n_val,
                                                        long long V, int mode, float* __restrict__ scores,
                                                        int accumulate) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (v >= V) return;
    const int nblk = M / LC_MB;
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
        const float resvar = (float)(m2 / (double)(n_val - 1));
        const float rsq = 1.f - resvar / ystat[2 * V + v];
        const float sgn = rsq > 0.f ? 1.f : (rsq < 0.f ? -1.f : rsq);   // sign(NaN) = NaN, sign(0) = 0
        score = sqrtf(fabsf(rsq)) * sgn;
    }
    // torch.nan_to_num defaults: NaN -> 0, +-inf -> +-FLT_MAX
    if (score != score) score = 0.f;
    else if (score > 3.4028234663852886e38f) score = 3.4028234663852886e38f;
    else if (score < -3.4028234663852886e38f) score = -3.4028234663852886e38f;
    float* dst = scores + (long long)a * V + v;
    *dst = accumulate ? *dst + score : score;
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

template <bool SCORE>
int set_lds_attr() {
    static thread_local bool done = false;
    if (!done) {
        LC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f32<SCORE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        done = true;
    }
    return LC_OK;
}

}  // namespace

extern "C" int lc_alpha_sweep_scores(const float* d_h, int A, int M, int N, const float* d_y, int64_t ldy, int64_t V,
                                     const int32_t* d_tr, const int32_t* d_va, int n_val, const float* d_ystat,
                                     const float* d_yblk, int mode, float* d_part, float* d_scores, int accumulate,
                                     lc_stream_t stream) {
    LC_REQUIRE(d_h && d_y && d_tr && d_va && d_ystat && d_yblk && d_part && d_scores, LC_E_BADARG,
               "lc_alpha_sweep_scores: null pointer");
    LC_REQUIRE(A > 0 && M > 0 && M % LC_MB == 0 && n_val > 0 && n_val <= M, LC_E_SHAPE,
               "lc_alpha_sweep_scores: need M %% %d == 0 and 0 < n_val <= M", LC_MB);
    LC_REQUIRE(mode == LC_SCORE_CORR || mode == LC_SCORE_R2, LC_E_BADARG, "lc_alpha_sweep_scores: bad mode %d", mode);
    LC_REQUIRE(ldy >= V, LC_E_SHAPE, "lc_alpha_sweep_scores: ldy < V");
    if (int rc = check_gemm_shapes("lc_alpha_sweep_scores", d_h, N, d_y, ldy, V, N)) return rc;
    if (int rc = set_lds_attr<true>()) return rc;
    hipStream_t s = lc::as_stream(stream);
    const int Mrows = A * M;
    const int Mtiles = lc::ceil_div(Mrows, BM);
    const long long Ntiles = V / BN;
    LC_REQUIRE((long long)Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_alpha_sweep_scores: grid too large");
    GroupTiles gt;
    gt.G = 1;
    gt.start[0] = 0;
    gt.start[1] = (int)Ntiles;
    ScoreArgs sa{d_y, (long long)ldy, d_va, d_ystat, d_part, M, n_val, mode};
    {
        lc::ScopedTimer timer_(lc::T_SWEEP_GEMM, s);
        hipLaunchKernelGGL(k_gemm_f32<true>, dim3((unsigned)(Mtiles * Ntiles)), dim3(256), GEMM_LDS_BYTES, s, d_h,
                           (long long)N, 0ll, Mrows, d_y, (long long)ldy, d_tr, N, Mtiles, gt, (float*)nullptr,
                           (long long)V, sa);
    }
    if (int rc = lc::launched("k_gemm_f32<score>")) return rc;
    lc::ScopedTimer timer_(lc::T_SWEEP_FINALIZE, s);
    hipLaunchKernelGGL(k_score_finalize, dim3((unsigned)lc::ceil_div<long long>(V, 256), (unsigned)A), dim3(256), 0, s,
                       d_part, d_ystat, d_yblk, A, M, n_val, (long long)V, mode, d_scores, accumulate);
    return lc::launched("k_score_finalize");
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
    if (int rc = set_lds_attr<false>()) return rc;
    const int Mtiles = (int)lc::ceil_div<long long>(Mrows, BM);
    LC_REQUIRE((long long)Mtiles * Ntiles < (1ll << 31), LC_E_SHAPE, "lc_gemm_grouped_f32: grid too large");
    ScoreArgs sa{};
    lc::ScopedTimer timer_(lc::T_GROUPED_GEMM, lc::as_stream(stream));
    hipLaunchKernelGGL(k_gemm_f32<false>, dim3((unsigned)(Mtiles * Ntiles)), dim3(256), GEMM_LDS_BYTES,
                       lc::as_stream(stream), d_a, (long long)lda, (long long)a_group_stride, (int)Mrows, d_b,
                       (long long)ldb, d_brows, (int)K, Mtiles, gt, d_c, (long long)ldc, sa);
    return lc::launched("k_gemm_f32<plain>");
}
