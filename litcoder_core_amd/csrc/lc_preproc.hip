// FIR delay stacking and Lanczos downsampling: HBM-bound, coalesced along the feature axis.
#include "lc_common.h"

namespace {

constexpr int FIR_MAX_DELAYS = 16;

struct FirDelays {
    long long d[FIR_MAX_DELAYS];
    int n;
};

// One thread per output element; grid.x = time row, grid.y covers (delay, feature).
// out[r, k*ndim + c] = stim[r - d_k, c] when that row exists, else 0 (or wrapped).
template <typename T>
__global__ void __launch_bounds__(256) k_fir_delay(const T* __restrict__ stim, long long nt, long long ndim,
                                                   long long ld_in, FirDelays dl, int circpad,
                                                   double* __restrict__ out, long long ld_out, long long col0) {
    const long long r = blockIdx.x;
    const long long e = (long long)blockIdx.y * blockDim.x + threadIdx.x;   // index into (k, c)
    if (e >= (long long)dl.n * ndim) return;
    const int k = (int)(e / ndim);
    const long long c = e - (long long)k * ndim;
    const long long d = dl.d[k];
    long long src = r - d;
    double v = 0.0;
    const bool far = (d >= nt) || (-d >= nt);
    if (far) {
        // Slice arithmetic of the reference: |d| >= nt copies everything unshifted when
        // wrapping, and leaves zeros otherwise (oracle/fir.py).
        if (circpad) v = (double)stim[r * ld_in + c];
    } else if (src >= 0 && src < nt) {
        v = (double)stim[src * ld_in + c];
    } else if (circpad) {
        src = src < 0 ? src + nt : src - nt;
        v = (double)stim[src * ld_in + c];
    }
    out[r * ld_out + col0 + e] = v;
}

constexpr int LZ_THREADS = 256;
constexpr int LZ_CHUNK = 1024;  // input samples per weight chunk

enum { WK_LANCZOS = 0, WK_SINC = 1 };

// weight of one (output time, input sample) pair; t is the time difference, c the cutoff frequency
template <int KIND>
__device__ inline double interp_weight(double dt, double c, double window, int causal) {
    constexpr double PI = 3.141592653589793;
    if (KIND == WK_LANCZOS) {
        // interpdata.py:59-63
        const double t = dt * c;
        if (t == 0.0) return 1.0;
        if (fabs(t) > window) return 0.0;
        // sin(pi x) as sinpi(x): the argument reduction is exact and a fraction of sin's (the weights are a third of the
        // resampler's arithmetic); the value differs from sin(PI * x) by the rounding of PI * x, ~1e-16 relative
        return window * sinpi(t) * sinpi(t / window) / ((PI * PI) * (t * t));
    } else {
        // interpdata.py:31-36 (sincfun, array branch): 2B sin(2 pi B t) / (2 pi B t + 1e-20), windowed, causal
        if (fabs(dt) > window / (2 * c)) return 0.0;
        if (causal && dt < 0) return 0.0;
        return 2 * c * sin(2 * PI * c * dt) / (2 * PI * c * dt + 1e-20);
    }
}

// One block per (output time point, 256-column slab).  Pass 1: every thread evaluates the weights of its
// input samples (fp64 sin) and the non-zero ones are compacted, in increasing sample order, into an LDS list
// (ballot + prefix: deterministic).  Pass 2: each thread streams its column down that short list -- for a
// Lanczos window of 3 lobes only ~60 of the 2500 samples of a story carry weight.
template <typename T, bool RECTIFY, int KIND>
__global__ void __launch_bounds__(LZ_THREADS) k_lanczos(const T* __restrict__ data, long long n_old, long long D,
                                                        long long ld_in, const double* __restrict__ oldtime,
                                                        const double* __restrict__ newtime, double cutoff,
                                                        double window, int causal, int renorm,
                                                        double* __restrict__ out, long long ld_out) {
    __shared__ double w[LZ_CHUNK];
    __shared__ int wj[LZ_CHUNK];
    __shared__ double red[LZ_THREADS];
    __shared__ int wave_cnt[LZ_THREADS / 64];
    const long long i = blockIdx.x;
    const long long c = (long long)blockIdx.y * LZ_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double tn = newtime[i];
    double scale = 1.0;
    if (KIND == WK_SINC && renorm) {
        // sincfun's ``val / np.sum(val)`` (skipped when the sum is exactly 0), interpdata.py:37-38
        double s = 0.0;
        for (long long j = threadIdx.x; j < n_old; j += LZ_THREADS) s += interp_weight<KIND>(tn - oldtime[j], cutoff, window, causal);
        red[threadIdx.x] = s;
        __syncthreads();
        for (int h = LZ_THREADS / 2; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (red[0] != 0.0) scale = 1.0 / red[0];
        __syncthreads();
    }
    double acc = 0.0, accp = 0.0;
    int fill = 0;                                   // entries in the LDS list (block-uniform)
    for (long long j0 = 0; j0 < n_old || fill > 0; j0 += LZ_THREADS) {
        // ---- pass 1 on samples j0 .. j0+255: weight, then ordered compaction
        const long long j = j0 + threadIdx.x;
        double wt = 0.0;
        if (j < n_old) wt = interp_weight<KIND>(tn - oldtime[j], cutoff, window, causal) * scale;
        const unsigned long long m = __ballot(wt != 0.0);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int base = fill, total = 0;
#pragma unroll
        for (int q = 0; q < LZ_THREADS / 64; ++q) {
            if (q < wave) base += wave_cnt[q];
            total += wave_cnt[q];
        }
        if (wt != 0.0) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            w[pos] = wt;
            wj[pos] = (int)j;
        }
        fill += total;
        __syncthreads();
        // ---- pass 2 whenever the list could overflow on the next round, or at the end
        const bool last = j0 + LZ_THREADS >= n_old;
        if (fill > LZ_CHUNK - LZ_THREADS || last) {
            if (c < D)
                for (int e = 0; e < fill; ++e) {
                    const double x = (double)data[(long long)wj[e] * ld_in + c];
                    if (RECTIFY) {
                        acc += w[e] * fmin(x, 0.0);
                        accp += w[e] * fmax(x, 0.0);
                    } else {
                        acc += w[e] * x;
                    }
                }
            fill = 0;
            __syncthreads();
            if (last) break;
        }
    }
    if (c < D) {
        out[i * ld_out + c] = acc;
        if (RECTIFY) out[i * ld_out + D + c] = accp;
    }
}

// All stories of a training run in ONE launch (trainer.py:125-157 calls the downsampler once per story: 27 launches of
// 26 us each, every block evaluating the fp64 sines of ALL 2500 word times of its story).  A story table gives each
// story's slice of the concatenated inputs / outputs.  When a story's sample times are non-decreasing (``sorted``: word
// onsets always are) the samples that can carry weight -- |t| <= window lobes around the output time -- are a contiguous
// index range found by bisection with the SAME comparison lanczosfun applies (interpdata.py:62), so only ~64 instead of
// ~2500 weights are evaluated per output row; an unsorted story scans everything.  (Round 4's kernel, one block per
// output row, is gone: k_lanczos_rows below.)
struct LzStory {
    long long old_off, n_old, new_off;
    double cutoff;
    int sorted;
};

// Round 5: LZR consecutive output rows per block.  Neighbouring TRs share most of their words (a window of +-3 lobes is
// ~60 words, a TR step ~7): the block-per-row kernel above re-read every input row ~9 times (from L2) with one 4-byte load
// per thread and trip behind two 11-step bisections -- 0.95 TB/s of algorithmic bytes on the 27 stories of a LeBel run,
// 0.12 of the HBM roof.  Here a block finds the windows of its LZR rows at once (one lane per row), builds
// the dense (row, sample) weight table of their UNION range in LDS (zeros outside a row's own window), and every thread
// streams its column down the union ONCE, feeding LZR accumulators: ~14 instead of ~64 loads per output value.  Weights
// are lanczosfun's (interpdata.py:59-63) evaluated exactly as above, every output adds its non-zero terms in increasing
// sample order: the same bits as k_lanczos / k_lanczos_stories.  A story whose times are not sorted (or whose cutoff is not
// a positive finite number: fewer than two output times, decreasing output times) takes its whole sample range as the
// window -- correct, slow, and not what word onsets look like.
constexpr int LZR = 8;            // output rows per block
constexpr int LZ_SPAN = 256;      // samples of the union range per pass, a multiple of the batch (LDS: LZR x LZ_SPAN doubles)

// VEC columns per thread (16-byte loads: 4 float / 2 double columns -- a LeBel story's 768 float columns are ONE pass of 192
// threads; 1 = any alignment)
template <typename T, bool RECTIFY, int VEC>
__global__ void __launch_bounds__(LZ_THREADS) k_lanczos_rows(const T* __restrict__ data, long long D, long long ld_in,
                                                             const double* __restrict__ oldtime,
                                                             const double* __restrict__ newtime, long long n_new_total,
                                                             const int* __restrict__ row_story,
                                                             const LzStory* __restrict__ stories, double window,
                                                             double* __restrict__ out, long long ld_out) {
    __shared__ double W[LZR][LZ_SPAN];                                 // [row][sample of the pass]
    __shared__ long long s_lo[LZR], s_hi[LZR];                         // windows as sample indices of the CONCATENATION
    __shared__ double s_tn[LZR], s_cut[LZR];
    const long long i0 = (long long)blockIdx.x * LZR;
    const int nrows = (int)min((long long)LZR, n_new_total - i0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- the rows' windows: wave w finds those of rows 2 w and 2 w + 1 by a 64-way search (the block-per-row kernel ran two
    // bisections of ~11 dependent loads in every thread; here a search is 2-3 rounds of one load per lane)
    for (int q = 0; q < LZR / (LZ_THREADS / 64); ++q) {
        const int r = wave * (LZR / (LZ_THREADS / 64)) + q;
        long long lo = 0, hi = 0;
        double tn = 0.0, cut = 0.0;
        if (r < nrows) {
            const LzStory st = stories[row_story[i0 + r]];
            tn = newtime[i0 + r];
            cut = st.cutoff;
            const double* ot = oldtime + st.old_off;
            if (st.sorted) {
                // (tn - ot[j]) * cutoff is non-increasing in j.  first(pred): the first j in [a, b) whose pred is false,
                // pred true on a prefix -- narrowed 64-fold per round by one probe per lane
                auto first = [&](long long a, long long b, auto pred) {
                    while (b - a > 0) {
                        const long long step = (b - a + 63) / 64;
                        const long long m = a + (long long)lane * step;             // probes a, a + step, ...
                        const bool p = m < b ? pred(ot[m]) : false;
                        const unsigned long long mask = __ballot(p);                // a prefix of ones
                        const int k = mask == ~0ull ? 64 : __ffsll((long long)~mask) - 1;   // first lane whose probe fails
                        // the answer lies in (probe k-1, probe k]: all probes true -> beyond the last one
                        const long long na = k == 0 ? a : a + (long long)(k - 1) * step + 1;
                        const long long nb = k == 64 ? b : min(b, a + (long long)k * step);
                        if (step == 1) { a = nb; break; }
                        a = na;
                        b = nb;
                    }
                    return a;
                };
                lo = first(0, st.n_old, [&](double o) { return (tn - o) * cut > window; });
                hi = first(lo, st.n_old, [&](double o) { return (tn - o) * cut >= -window; });
            } else {
                hi = st.n_old;
            }
            lo += st.old_off;
            hi += st.old_off;
        }
        if (lane == 0) { s_lo[r] = lo; s_hi[r] = hi; s_tn[r] = tn; s_cut[r] = cut; }
    }
    __syncthreads();
    long long gmin = 0, gmax = 0;
    bool any = false;
#pragma unroll
    for (int r = 0; r < LZR; ++r)
        if (r < nrows && s_hi[r] > s_lo[r]) {
            gmin = any ? min(gmin, s_lo[r]) : s_lo[r];
            gmax = any ? max(gmax, s_hi[r]) : s_hi[r];
            any = true;
        }
    constexpr int U = 8;                                               // samples in flight per thread
    // the weight table of one pass over the union range (zero outside a row's own window; the tail of the last batch of U
    // samples zero too)
    auto build = [&](long long g0, int span) {
        const int span_u = (span + U - 1) / U * U;
        for (int e = threadIdx.x; e < span_u * LZR; e += LZ_THREADS) {
            const int r = e / span_u, jj = e - r * span_u;
            const long long g = g0 + jj;
            double wt = 0.0;
            if (r < nrows && jj < span && g >= s_lo[r] && g < s_hi[r])
                wt = interp_weight<WK_LANCZOS>(s_tn[r] - oldtime[g], s_cut[r], window, 0);
            W[r][jj] = wt;
        }
    };
    // one column of the slab down the pass: a row takes part in a batch of U samples only when its window reaches into it -- a
    // block-uniform (scalar) test; inside such a batch the weights outside the window are zeros and add nothing:
    // acc + 0 * x == acc for finite x (the first version tested every weight: a compare and two selects per fp64 FMA made the
    // kernel VALU-bound; a NON-finite sample now spoils the rows whose batches cover it, up to U - 1 samples outside their
    // windows -- the reference's dense np.dot spoils the whole output column)
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    typedef double vecD __attribute__((ext_vector_type(VEC)));
    auto stream = [&](long long g0, int span, long long c, vecD (&acc)[LZR], vecD (&accp)[LZR]) {
        int rlo[LZR], rhi[LZR];
#pragma unroll
        for (int r = 0; r < LZR; ++r) {
            rlo[r] = __builtin_amdgcn_readfirstlane((int)max(-1ll, min((long long)span, s_lo[r] - g0)));
            rhi[r] = __builtin_amdgcn_readfirstlane(r < nrows ? (int)max(0ll, min((long long)span, s_hi[r] - g0)) : 0);
        }
        const T* col = data + g0 * ld_in + c;
        for (int j0 = 0; j0 < span; j0 += U) {
            vecT xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u < span) xv[u] = *reinterpret_cast<const vecT*>(col + (long long)(j0 + u) * ld_in);
                else xv[u] = vecT(0);
            }
#pragma unroll
            for (int r = 0; r < LZR; ++r) {
                if (j0 < rhi[r] && j0 + U > rlo[r]) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double w = W[r][j0 + u];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const double x = (double)xv[u][q];
                            if (RECTIFY) {
                                acc[r][q] += w * fmin(x, 0.0);
                                accp[r][q] += w * fmax(x, 0.0);
                            } else {
                                acc[r][q] += w * x;
                            }
                        }
                    }
                }
            }
        }
    };
    auto store = [&](long long c, const vecD (&acc)[LZR], const vecD (&accp)[LZR]) {
#pragma unroll
        for (int r = 0; r < LZR; ++r)
            if (r < nrows) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    out[(i0 + r) * ld_out + c + q] = acc[r][q];
                    if (RECTIFY) out[(i0 + r) * ld_out + D + c + q] = accp[r][q];
                }
            }
    };
    const bool one_pass = gmax - gmin <= LZ_SPAN;                      // (sorted stories: always -- ~120 samples)
    const long long cstep = (long long)gridDim.y * LZ_THREADS * VEC;   // grid.y > 1 only when there are few row groups
    if (one_pass) {
        // the weights ONCE for all the block's columns of the rows (the first version computed them per 256-column slab --
        // two fp64 sines per weight, a third of the kernel's arithmetic)
        const int span = (int)(gmax - gmin);
        build(gmin, span);
        __syncthreads();
        for (long long c = ((long long)blockIdx.y * LZ_THREADS + threadIdx.x) * VEC; c < D; c += cstep) {
            vecD acc[LZR], accp[LZR];
#pragma unroll
            for (int r = 0; r < LZR; ++r) { acc[r] = vecD(0.0); accp[r] = vecD(0.0); }
            if (span > 0) stream(gmin, span, c, acc, accp);
            store(c, acc, accp);
        }
        return;
    }
    for (long long c0 = (long long)blockIdx.y * LZ_THREADS * VEC; c0 < D; c0 += cstep) {   // (block-uniform trips: the barriers)
        const long long c = c0 + (long long)threadIdx.x * VEC;
        vecD acc[LZR], accp[LZR];
#pragma unroll
        for (int r = 0; r < LZR; ++r) { acc[r] = vecD(0.0); accp[r] = vecD(0.0); }
        for (long long g0 = gmin; g0 < gmax; g0 += LZ_SPAN) {
            const int span = (int)min((long long)LZ_SPAN, gmax - g0);
            __syncthreads();
            build(g0, span);
            __syncthreads();
            if (c < D) stream(g0, span, c, acc, accp);
        }
        if (c < D) store(c, acc, accp);
    }
}

// The design matrix of a story-structured fit in one launch (trainer.py:203-209 FIR.make_delayed per story, then
// :235-257 per story zs(features[start:end]), np.vstack, np.nan_to_num; nested_cv.py:99 the float32 cast):
//   X[row0_s + (t - a_s), k ndim + c] = fl32( nan_to_num( zs_s( delayed_s )[t, k ndim + c] ) ),  a_s <= t < b_s,
// delayed_s[t, k ndim + c] = feat_s[t - d_k, c] (0 outside the story; circpad is not used by the trainer).  One thread
// per (story, output column) walks the trimmed rows three times -- mean, squared deviations, normalise -- adding in row
// order without fused multiply-adds, i.e. numpy's own axis-0 reduction: the same bits as the reference's float64
// pipeline followed by its cast.  Neighbouring threads read neighbouring features of one row: coalesced.
struct StoryRows {
    long long in_off, n_in;      // the story's rows in the concatenated (downsampled) features
    long long a, b;              // trimmed row range [a, b) of the delayed story
    long long out_row0;          // first row of the story in X
};

__global__ void __launch_bounds__(256) k_story_design(const double* __restrict__ feat, long long ndim, long long ld_in,
                                                      const StoryRows* __restrict__ stories, FirDelays dl,
                                                      float* __restrict__ X, long long ldx) {
#pragma clang fp contract(off)
    const StoryRows st = stories[blockIdx.y];
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)dl.n * ndim) return;
    const int k = (int)(e / ndim);
    const long long c = e - (long long)k * ndim;
    const long long d = dl.d[k];
    const double* f = feat + st.in_off * ld_in + c;
    const long long n = st.b - st.a;
    auto val = [&](long long t) -> double {
        const long long src = t - d;
        return (src >= 0 && src < st.n_in && d < st.n_in && -d < st.n_in) ? f[src * ld_in] : 0.0;
    };
    // (round 5: SD_U rows are loaded before any of them is added -- the sums still take the rows one by one, numpy's order,
    // but a thread no longer waits a memory round trip per row: the kernel was a chain of ~1000 dependent loads per thread
    // at little over one wave per SIMD, 0.47 ms for 27 stories = 0.05 of the HBM roof on its algorithmic bytes; 0.19 ms
    // with 16 rows in flight.  A version that staged 16 input columns of a story in LDS -- read once for all delays and
    // passes -- was built and measured SLOWER, 0.30-0.36 ms: 64 of its 256 threads ran the sequential sums while the
    // others waited, 78 % of its wave cycles parked)
    constexpr int SD_U = 48;
    double mean = 0.0;
    for (long long t0 = st.a; t0 < st.b; t0 += SD_U) {
        double v[SD_U];
#pragma unroll
        for (int u = 0; u < SD_U; ++u) v[u] = t0 + u < st.b ? val(t0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < SD_U; ++u)
            if (t0 + u < st.b) mean = mean + v[u];
    }
    mean = mean / (double)n;
    double ss = 0.0;
    for (long long t0 = st.a; t0 < st.b; t0 += SD_U) {
        double v[SD_U];
#pragma unroll
        for (int u = 0; u < SD_U; ++u) v[u] = t0 + u < st.b ? val(t0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < SD_U; ++u)
            if (t0 + u < st.b) {
                const double dv = v[u] - mean;
                ss = ss + dv * dv;
            }
    }
    const double sd = sqrt(ss / (double)n);
    float* x = X + st.out_row0 * ldx + e;
    for (long long t0 = st.a; t0 < st.b; t0 += SD_U) {
        double v[SD_U];
#pragma unroll
        for (int u = 0; u < SD_U; ++u) v[u] = t0 + u < st.b ? val(t0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < SD_U; ++u)
            if (t0 + u < st.b) {
                double m = v[u] - mean;
                if (sd != 0.0) m = m / sd;
                // np.nan_to_num: NaN -> 0, +-inf -> +-DBL_MAX (which the float32 cast turns back into +-inf: nothing to do)
                if (m != m) m = 0.0;
                x[(t0 + u - st.a) * ldx] = (float)m;
            }
    }
}

// Per-TR reducers (downsampling.py:24-136,180-319): out[s] = mean | sum | last of the rows idx[seg[s] .. seg[s+1]),
// zero for an empty segment.  One block row per segment, lanes across the feature axis.
enum { SR_MEAN = 0, SR_SUM = 1, SR_LAST = 2 };

template <typename T>
__global__ void __launch_bounds__(256) k_segment_reduce(const T* __restrict__ data, long long D, long long ld_in,
                                                        const long long* __restrict__ seg, const int* __restrict__ idx,
                                                        int how, double* __restrict__ out, long long ld_out) {
    const long long s = blockIdx.x;
    const long long lo = seg[s], hi = seg[s + 1];
    const long long stride = (long long)gridDim.y * blockDim.x;
    for (long long c = (long long)blockIdx.y * blockDim.x + threadIdx.x; c < D; c += stride) {
        double acc = 0.0;
        if (hi > lo) {
            if (how == SR_LAST) {
                acc = (double)data[(long long)idx[hi - 1] * ld_in + c];
            } else {
                for (long long e = lo; e < hi; ++e) acc += (double)data[(long long)idx[e] * ld_in + c];
                if (how == SR_MEAN) acc /= (double)(hi - lo);
            }
        }
        out[s * ld_out + c] = acc;
    }
}

}  // namespace

extern "C" int lc_fir_delay(const void* d_stim, int dtype, int64_t nt, int64_t ndim, int64_t ld_in,
                            const int64_t* h_delays, int nd, int circpad, double* d_out, int64_t ld_out,
                            lc_stream_t stream) {
    LC_REQUIRE(d_stim && d_out && h_delays, LC_E_BADARG, "lc_fir_delay: null pointer");
    LC_REQUIRE(nt >= 0 && ndim >= 0 && nd >= 0, LC_E_BADARG, "lc_fir_delay: negative size");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, LC_E_BADARG, "lc_fir_delay: dtype %d unsupported", dtype);
    LC_REQUIRE(ld_in >= ndim && ld_out >= ndim * nd, LC_E_SHAPE, "lc_fir_delay: leading dimension too small");
    if (nt == 0 || ndim == 0 || nd == 0) return LC_OK;
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_FIR, s);
    for (int k0 = 0; k0 < nd; k0 += FIR_MAX_DELAYS) {
        FirDelays dl;
        dl.n = nd - k0 < FIR_MAX_DELAYS ? nd - k0 : FIR_MAX_DELAYS;
        for (int k = 0; k < dl.n; ++k) dl.d[k] = h_delays[k0 + k];
        dim3 grid((unsigned)nt, (unsigned)lc::ceil_div<long long>((long long)dl.n * ndim, 256));
        const long long col0 = (long long)k0 * ndim;
        if (dtype == LC_F32)
            hipLaunchKernelGGL(k_fir_delay<float>, grid, dim3(256), 0, s, (const float*)d_stim, nt, ndim, ld_in, dl,
                               circpad, d_out, ld_out, col0);
        else
            hipLaunchKernelGGL(k_fir_delay<double>, grid, dim3(256), 0, s, (const double*)d_stim, nt, ndim, ld_in,
                               dl, circpad, d_out, ld_out, col0);
        if (int rc = lc::launched("k_fir_delay")) return rc;
    }
    return LC_OK;
}

extern "C" int lc_lanczos_interp(const void* d_data, int dtype, int64_t n_old, int64_t D, int64_t ld_in,
                                 const double* d_oldtime, const double* d_newtime, int64_t n_new, double cutoff,
                                 double window, int rectify, double* d_out, int64_t ld_out, lc_stream_t stream) {
    LC_REQUIRE(d_data && d_oldtime && d_newtime && d_out, LC_E_BADARG, "lc_lanczos_interp: null pointer");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, LC_E_BADARG, "lc_lanczos_interp: dtype %d unsupported", dtype);
    LC_REQUIRE(n_old >= 0 && D >= 0 && n_new >= 0, LC_E_BADARG, "lc_lanczos_interp: negative size");
    LC_REQUIRE(ld_in >= D && ld_out >= (rectify ? 2 * D : D), LC_E_SHAPE, "lc_lanczos_interp: leading dimension too small");
    if (n_new == 0 || D == 0) return LC_OK;
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LANCZOS, s);
    dim3 grid((unsigned)n_new, (unsigned)lc::ceil_div<long long>(D, LZ_THREADS));
#define LC_LZ(T, R)                                                                                              \
    hipLaunchKernelGGL((k_lanczos<T, R, WK_LANCZOS>), grid, dim3(LZ_THREADS), 0, s, (const T*)d_data, n_old, D, ld_in, \
                       d_oldtime, d_newtime, cutoff, window, 0, 0, d_out, ld_out)
    if (dtype == LC_F32) { if (rectify) LC_LZ(float, true); else LC_LZ(float, false); }
    else                 { if (rectify) LC_LZ(double, true); else LC_LZ(double, false); }
#undef LC_LZ
    return lc::launched("k_lanczos");
}

extern "C" int lc_sinc_interp(const void* d_data, int dtype, int64_t n_old, int64_t D, int64_t ld_in,
                              const double* d_oldtime, const double* d_newtime, int64_t n_new, double cutoff,
                              double window, int causal, int renorm, double* d_out, int64_t ld_out,
                              lc_stream_t stream) {
    LC_REQUIRE(d_data && d_oldtime && d_newtime && d_out, LC_E_BADARG, "lc_sinc_interp: null pointer");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, LC_E_BADARG, "lc_sinc_interp: dtype %d unsupported", dtype);
    LC_REQUIRE(n_old >= 0 && D >= 0 && n_new >= 0 && ld_in >= D && ld_out >= D, LC_E_SHAPE, "lc_sinc_interp: bad shape");
    if (n_new == 0 || D == 0) return LC_OK;
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LANCZOS, s);
    dim3 grid((unsigned)n_new, (unsigned)lc::ceil_div<long long>(D, LZ_THREADS));
    if (dtype == LC_F32)
        hipLaunchKernelGGL((k_lanczos<float, false, WK_SINC>), grid, dim3(LZ_THREADS), 0, s, (const float*)d_data, n_old, D,
                           ld_in, d_oldtime, d_newtime, cutoff, window, causal, renorm, d_out, ld_out);
    else
        hipLaunchKernelGGL((k_lanczos<double, false, WK_SINC>), grid, dim3(LZ_THREADS), 0, s, (const double*)d_data, n_old,
                           D, ld_in, d_oldtime, d_newtime, cutoff, window, causal, renorm, d_out, ld_out);
    return lc::launched("k_lanczos<sinc>");
}

extern "C" int lc_segment_reduce(const void* d_data, int dtype, int64_t D, int64_t ld_in, const int64_t* d_seg,
                                 const int32_t* d_idx, int64_t n_seg, int how, double* d_out, int64_t ld_out,
                                 lc_stream_t stream) {
    LC_REQUIRE(d_data && d_seg && d_idx && d_out, LC_E_BADARG, "lc_segment_reduce: null pointer");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, LC_E_BADARG, "lc_segment_reduce: dtype %d unsupported", dtype);
    LC_REQUIRE(how >= SR_MEAN && how <= SR_LAST, LC_E_BADARG, "lc_segment_reduce: how must be 0 (mean), 1 (sum), 2 (last)");
    LC_REQUIRE(D >= 0 && n_seg >= 0 && ld_in >= D && ld_out >= D, LC_E_SHAPE, "lc_segment_reduce: bad shape");
    if (n_seg == 0 || D == 0) return LC_OK;
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LANCZOS, s);
    dim3 grid((unsigned)n_seg, (unsigned)lc::imin(lc::ceil_div<long long>(D, 256), 64));
    if (dtype == LC_F32)
        hipLaunchKernelGGL(k_segment_reduce<float>, grid, dim3(256), 0, s, (const float*)d_data, (long long)D,
                           (long long)ld_in, (const long long*)d_seg, d_idx, how, d_out, (long long)ld_out);
    else
        hipLaunchKernelGGL(k_segment_reduce<double>, grid, dim3(256), 0, s, (const double*)d_data, (long long)D,
                           (long long)ld_in, (const long long*)d_seg, d_idx, how, d_out, (long long)ld_out);
    return lc::launched("k_segment_reduce");
}

extern "C" int lc_lanczos_interp_stories(const void* d_data, int dtype, int64_t D, int64_t ld_in, const double* d_oldtime,
                                         const double* d_newtime, int64_t n_new_total, const int32_t* d_row_story,
                                         const void* d_stories, int n_stories, double window, int rectify, double* d_out,
                                         int64_t ld_out, lc_stream_t stream) {
    LC_REQUIRE(d_data && d_oldtime && d_newtime && d_row_story && d_stories && d_out, LC_E_BADARG,
               "lc_lanczos_interp_stories: null pointer");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, LC_E_BADARG, "lc_lanczos_interp_stories: dtype %d unsupported", dtype);
    LC_REQUIRE(D >= 0 && n_new_total >= 0 && n_stories > 0 && ld_in >= D && ld_out >= (rectify ? 2 * D : D), LC_E_SHAPE,
               "lc_lanczos_interp_stories: bad shape");
    if (n_new_total == 0 || D == 0) return LC_OK;
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_LANCZOS, s);
    // a block: LZR rows x all columns (the weights are evaluated once) -- unless that leaves the chip idle: with few row
    // groups (one story) the 256-column slabs are dealt out over grid.y
    const int esz = dtype == LC_F32 ? 4 : 8, vec = 16 / esz;
    const bool wide = D % vec == 0 && ld_in % vec == 0 && (reinterpret_cast<uintptr_t>(d_data) & 15) == 0;
    const long long per_block = (long long)LZ_THREADS * (wide ? vec : 1);
    const long long gx = lc::ceil_div<long long>(n_new_total, LZR), slabs = lc::ceil_div<long long>(D, per_block);
    const long long gy = lc::imin(slabs, gx >= 1024 ? 1 : lc::ceil_div<long long>(1024, gx));
    dim3 grid((unsigned)gx, (unsigned)gy);
#define LC_LZS(T, R, V)                                                                                               \
    hipLaunchKernelGGL((k_lanczos_rows<T, R, V>), grid, dim3(LZ_THREADS), 0, s, (const T*)d_data, (long long)D,      \
                       (long long)ld_in, d_oldtime, d_newtime, (long long)n_new_total, d_row_story,                   \
                       (const LzStory*)d_stories, window, d_out, (long long)ld_out)
    if (dtype == LC_F32) {
        if (wide) { if (rectify) LC_LZS(float, true, 4); else LC_LZS(float, false, 4); }
        else      { if (rectify) LC_LZS(float, true, 1); else LC_LZS(float, false, 1); }
    } else {
        if (wide) { if (rectify) LC_LZS(double, true, 2); else LC_LZS(double, false, 2); }
        else      { if (rectify) LC_LZS(double, true, 1); else LC_LZS(double, false, 1); }
    }
#undef LC_LZS
    return lc::launched("k_lanczos_rows");
}

extern "C" int lc_story_design_f32(const double* d_feat, int64_t ndim, int64_t ld_in, const void* d_stories, int n_stories,
                                   const int64_t* h_delays, int nd, float* d_x, int64_t ldx, lc_stream_t stream) {
    LC_REQUIRE(d_feat && d_stories && h_delays && d_x, LC_E_BADARG, "lc_story_design_f32: null pointer");
    LC_REQUIRE(ndim > 0 && n_stories > 0 && nd > 0 && nd <= FIR_MAX_DELAYS && ld_in >= ndim && ldx >= ndim * nd, LC_E_SHAPE,
               "lc_story_design_f32: bad shape (at most %d delays)", FIR_MAX_DELAYS);
    FirDelays dl;
    dl.n = nd;
    for (int k = 0; k < nd; ++k) dl.d[k] = h_delays[k];
    hipStream_t s = lc::as_stream(stream);
    lc::ScopedTimer timer_(lc::T_FIR, s);
    dim3 grid((unsigned)lc::ceil_div<long long>((long long)nd * ndim, 256), (unsigned)n_stories);
    hipLaunchKernelGGL(k_story_design, grid, dim3(256), 0, s, d_feat, (long long)ndim, (long long)ld_in,
                       (const StoryRows*)d_stories, dl, d_x, (long long)ldx);
    return lc::launched("k_story_design");
}
