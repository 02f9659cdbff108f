// Diagnostic builds of the dominant kernel, OUTSIDE the product library: the score kernel with
// s_memtime / s_memrealtime stamps (in-kernel clock, per-phase cycles) and the experiment kernel on
// v_mfma_f32_16x16x32_f16 (plain mode only; measured 1-2 % faster than the 32x32x16 kernel, not adopted: HISTORY.md 4.1).
// Built by tools/debug_kernels/build.py into tools/bin/liblitcoder_debug.so; tools/gpu_kernel_bench.py loads it.
#define LC_SWEEP_EXPERIMENTS 1
#include "../../litcoder_core_amd/csrc/lc_gemm16_kernel.h"
#include "lc_debug.h"

// (lc::fail / lc::ensure_dynamic_lds / the event timers live in the product library: this one links against it)

namespace {

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

// ... the same for the HI2 mode (round 6: the screening pass; one MFMA per product, two K-tiles per ring step): d_stamps[.][5]
// counts ring STEPS (N / 32 per tile and wave)
extern "C" int lc_debug_sweep16_stamps_hi2(const void* d_ht, const float* d_rowscale_inv, int A, int M, int N, const void* d_yt,
                                           const float* d_cscale_inv, const float* d_yv, int64_t V, int n_val,
                                           const float* d_ystat, float* d_part,
                                           unsigned long long* d_stamps, lc_stream_t stream) {
    LC_REQUIRE(d_ht && d_yt && d_stamps && N % 64 == 0, LC_E_BADARG, "lc_debug_sweep16_stamps_hi2: null pointer / N %% 64");
    LC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_f16x3<true, true, false, false, false, true>),
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
    hipLaunchKernelGGL((k_sweep_f16x3<true, true, false, false, false, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512),
                       LDS16_BYTES, lc::as_stream(stream), (const uint4*)d_ht, (const uint4*)d_yt, N / TK, Mtiles, sa, pa,
                       BView{N / TK, N / TK, 0}, fv);
    return lc::launched("k_sweep_f16x3<stamp, HI2>");
}

// ... and with the kernel's experiment bits (round 6, lc_gemm16_kernel.h: 1 = contiguous 16 KB fetches per operand and HI2
// step, 2 = operand delivery alone (no MFMAs, no fragment reads), 4 = every workgroup fetches tile (0, 0)); hi2 = 0: the
// three-MFMA form (bit 1 has no meaning there)
extern "C" int lc_debug_sweep16_stamps_exp(const void* d_ht, const float* d_rowscale_inv, int A, int M, int N, const void* d_yt,
                                           const float* d_cscale_inv, const float* d_yv, int64_t V, int n_val,
                                           const float* d_ystat, float* d_part, unsigned long long* d_stamps, int hi2,
                                           int exp_bits, lc_stream_t stream) {
    LC_REQUIRE(d_ht && d_yt && N % 64 == 0, LC_E_BADARG, "lc_debug_sweep16_stamps_exp: null pointer / N %% 64");
    if (d_stamps == nullptr) {
        // no stamps: the PRODUCT instantiation (this translation unit's copy of it, with the experiment bits live)
        LC_REQUIRE(hi2, LC_E_BADARG, "lc_debug_sweep16_stamps_exp: the un-stamped experiment build is the HI2 form");
        LC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_f16x3<true, false, false, false, false, true>),
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
        pa.G = exp_bits;
        hipLaunchKernelGGL((k_sweep_f16x3<true, false, false, false, false, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512),
                           LDS16_BYTES, lc::as_stream(stream), (const uint4*)d_ht, (const uint4*)d_yt, N / TK, Mtiles, sa, pa,
                           BView{N / TK, N / TK, 0}, fv);
        return lc::launched("k_sweep_f16x3<HI2, exp>");
    }
    const void* fn = hi2 ? reinterpret_cast<const void*>(k_sweep_f16x3<true, true, false, false, false, true>)
                         : reinterpret_cast<const void*>(k_sweep_f16x3<true, true>);
    LC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS16_BYTES));
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
    pa.G = exp_bits;
    if (hi2)
        hipLaunchKernelGGL((k_sweep_f16x3<true, true, false, false, false, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512),
                           LDS16_BYTES, lc::as_stream(stream), (const uint4*)d_ht, (const uint4*)d_yt, N / TK, Mtiles, sa, pa,
                           BView{N / TK, N / TK, 0}, fv);
    else
        hipLaunchKernelGGL((k_sweep_f16x3<true, true>), dim3((unsigned)(Mtiles * Ntiles)), dim3(512), LDS16_BYTES,
                           lc::as_stream(stream), (const uint4*)d_ht, (const uint4*)d_yt, N / TK, Mtiles, sa, pa,
                           BView{N / TK, N / TK, 0}, fv);
    return lc::launched("k_sweep_f16x3<stamp, exp>");
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
