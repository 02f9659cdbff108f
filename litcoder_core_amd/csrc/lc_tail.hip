// Statistics tail on the device: Fisher's combination of the per-fold p-values and the
// Benjamini-Hochberg step-up procedure.  O(V) work on vectors that are already in HBM; the sort is a least-significant-
// digit radix sort written here (4-bit digits over the bit pattern of the non-negative doubles, stable, index payload).
//   reference: nested_cv.py:441-477 (_combine_pvalues_across_folds), statsmodels fdrcorrection(method="indep")
//   at nested_cv.py:158,263,282 -- host twins with the same arithmetic: litcoder_core_amd/stats.py.
#include "lc_common.h"

namespace {

// p = chi2.sf(-2 sum_k ln p_k, 2k) = exp(-L) sum_{i<k} L^i / i!,  L = -sum_k ln p_k  (even degrees of freedom);
// inf L -> 0, all-ones rows -> exactly 1 (the reference's shortcut), result clipped to 1.
__global__ void __launch_bounds__(256) k_fisher(const double* __restrict__ p, int k, long long V,
                                                double* __restrict__ out) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double L = 0.0;
    bool ones = true;
    for (int f = 0; f < k; ++f) {
        const double x = p[(long long)f * V + v];
        ones = ones && x == 1.0;
        L -= log(x);
    }
    double term = 1.0, acc = 1.0;
    for (int i = 1; i < k; ++i) {
        term = term * L / (double)i;
        acc += term;
    }
    double r = exp(-L) * acc;
    if (isinf(L)) r = 0.0;
    if (ones) r = 1.0;
    out[v] = fmin(r, 1.0);                      // fmin drops a NaN operand like the clip never sees one: p is NaN-free
}

// ------------------------------------------------------------------ LSD radix sort of (double key >= 0, int index)
// Non-negative IEEE doubles order like their bit patterns read as unsigned integers, so 16 passes over 4-bit digits sort
// them; every pass is stable (a block keeps its keys in input order: thread t owns RS_ITEMS consecutive keys, and the
// (digit, block)-major exclusive scan of the block histograms places equal digits of earlier blocks first), hence so is
// the whole sort.  Per pass: k_rs_hist (per-block digit histogram), k_rs_scan (one workgroup: exclusive scan of the
// 16 x blocks table; a digit that is the same for ALL keys is flagged and the scatter degenerates to a copy),
// k_rs_scatter.  n <= 2^31; 640 000 keys = 313 blocks.  (hipCUB's DeviceRadixSort did this until round 3.)
constexpr int RS_THREADS = 256, RS_ITEMS = 8, RS_TILE = RS_THREADS * RS_ITEMS, RS_BINS = 16;

__device__ inline unsigned long long rs_key(const double* __restrict__ keys, long long i) {
    return (unsigned long long)__double_as_longlong(keys[i]);
}

__global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const double* __restrict__ keys, long long n, int shift,
                                                        int* __restrict__ hist, int nblk) {
    __shared__ int h[RS_BINS];
    if (threadIdx.x < RS_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)threadIdx.x * RS_ITEMS;
    int mine[RS_BINS] = {};
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j)
        if (base + j < n) mine[(rs_key(keys, base + j) >> shift) & (RS_BINS - 1)] += 1;
#pragma unroll
    for (int b = 0; b < RS_BINS; ++b)
        if (mine[b]) atomicAdd(&h[b], mine[b]);
    __syncthreads();
    if (threadIdx.x < RS_BINS) hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of hist (RS_BINS x nblk, digit-major) in place; *trivial = 1 when one digit holds all n keys
__global__ void __launch_bounds__(1024) k_rs_scan(int* __restrict__ hist, int nblk, long long n, int* __restrict__ trivial) {
    __shared__ int part[1024];
    __shared__ int carry, triv;
    const int t = threadIdx.x, total = RS_BINS * nblk;
    if (t == 0) { carry = 0; triv = 0; }
    __syncthreads();
    if (t < RS_BINS) {                               // digit totals (before the scan overwrites them)
        long long c = 0;
        for (int b = 0; b < nblk; ++b) c += hist[t * nblk + b];
        if (c == n) triv = 1;
    }
    __syncthreads();
    for (int base = 0; base < total; base += 1024) {
        const int i = base + t;
        const int v = i < total ? hist[i] : 0;
        part[t] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {         // Hillis-Steele inclusive scan of the chunk
            const int x = t >= o ? part[t - o] : 0;
            __syncthreads();
            part[t] += x;
            __syncthreads();
        }
        if (i < total) hist[i] = carry + part[t] - v;
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) *trivial = triv;
}

// vals_in == nullptr: the payload is the key's own position (first pass)
__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(const double* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                           long long n, int shift, const int* __restrict__ offs, int nblk,
                                                           const int* __restrict__ trivial, double* __restrict__ keys_out,
                                                           int* __restrict__ vals_out) {
    __shared__ int cnt[RS_BINS * RS_THREADS];        // [digit][thread]: keys of that digit the thread holds
    __shared__ int tot[RS_THREADS];
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)t * RS_ITEMS;
    unsigned long long k[RS_ITEMS];
    int v[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        k[j] = base + j < n ? rs_key(keys_in, base + j) : 0ull;
        v[j] = base + j < n ? (vals_in ? vals_in[base + j] : (int)(base + j)) : 0;
    }
    if (*trivial) {                                  // every key has the same digit: the pass is the identity
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j)
            if (base + j < n) { keys_out[base + j] = __longlong_as_double((long long)k[j]); vals_out[base + j] = v[j]; }
        return;
    }
    int mine[RS_BINS] = {};
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j)
        if (base + j < n) mine[(k[j] >> shift) & (RS_BINS - 1)] += 1;
#pragma unroll
    for (int b = 0; b < RS_BINS; ++b) cnt[b * RS_THREADS + t] = mine[b];
    __syncthreads();
    // exclusive scan of the flattened [digit][thread] table: thread t scans 16 consecutive entries, then the 256 partial
    // sums are scanned; cnt[d][u] becomes the number of the block's keys with a smaller digit, or digit d in threads < u
    int run = 0, first[RS_BINS];
#pragma unroll
    for (int e = 0; e < RS_BINS; ++e) { first[e] = run; run += cnt[t * RS_BINS + e]; }
    tot[t] = run;
    __syncthreads();
    for (int o = 1; o < RS_THREADS; o <<= 1) {
        const int x = t >= o ? tot[t - o] : 0;
        __syncthreads();
        tot[t] += x;
        __syncthreads();
    }
    const int before = tot[t] - run;
#pragma unroll
    for (int e = 0; e < RS_BINS; ++e) cnt[t * RS_BINS + e] = before + first[e];
    __syncthreads();
    // key j of this thread goes to  offs[digit][block] + (rank among the block's keys of that digit)
    int next[RS_BINS];
#pragma unroll
    for (int b = 0; b < RS_BINS; ++b) next[b] = offs[b * nblk + blockIdx.x] + cnt[b * RS_THREADS + t] - cnt[b * RS_THREADS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j)
        if (base + j < n) {
            const int d = (int)((k[j] >> shift) & (RS_BINS - 1));
            int pos = 0;
#pragma unroll
            for (int b = 0; b < RS_BINS; ++b)
                if (b == d) { pos = next[b]; next[b] += 1; }
            keys_out[pos] = __longlong_as_double((long long)k[j]);
            vals_out[pos] = v[j];
        }
}

// Sorted p-values -> adjusted p-values and rejections, scattered back to input order, in two launches over blocks of
// BH_THREADS consecutive ranks (coalesced; a fold's 80 000 - 640 000 p-values are 79 - 625 blocks):
//   k_bh_block_min  per block: min of q_i = p_(i) / frac_i and the largest i with p_(i) <= frac_i alpha;
//   k_bh_apply      per block: the minimum over the blocks to its right and the largest passing rank of all blocks,
//                   an inclusive suffix-minimum scan of its own q_i in LDS, then the scatter.
// Same expressions per element as the one-workgroup version it replaces (0.56 ms for 80 000 values, 4.5 ms for 640 000).
constexpr int BH_THREADS = 1024;

__device__ inline void bh_element(const double* __restrict__ ps, long long i, long long n, double alpha, double& q,
                                  long long& pass) {
    q = INFINITY;
    pass = -1;
    if (i < n) {
        const double frac = (double)(i + 1) / (double)n;
        q = ps[i] / frac;
        if (ps[i] <= frac * alpha) pass = i;
    }
}

__global__ void __launch_bounds__(BH_THREADS) k_bh_block_min(const double* __restrict__ ps, long long n, double alpha,
                                                             double* __restrict__ bmin, long long* __restrict__ blast) {
    __shared__ double smin[BH_THREADS / 64];
    __shared__ long long smax[BH_THREADS / 64];
    const int t = threadIdx.x;
    double q;
    long long pass;
    bh_element(ps, (long long)blockIdx.x * BH_THREADS + t, n, alpha, q, pass);
    for (int o = 32; o > 0; o >>= 1) {
        q = fmin(q, __shfl_xor(q, o));
        const long long other = __shfl_xor(pass, o);
        pass = other > pass ? other : pass;
    }
    if ((t & 63) == 0) { smin[t >> 6] = q; smax[t >> 6] = pass; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < BH_THREADS / 64; ++w) {
            q = fmin(q, smin[w]);
            pass = smax[w] > pass ? smax[w] : pass;
        }
        bmin[blockIdx.x] = q;
        blast[blockIdx.x] = pass;
    }
}

__global__ void __launch_bounds__(BH_THREADS) k_bh_apply(const double* __restrict__ ps, const int* __restrict__ order,
                                                         long long n, double alpha, const double* __restrict__ bmin,
                                                         const long long* __restrict__ blast, int nb,
                                                         unsigned char* __restrict__ reject, double* __restrict__ padj) {
    __shared__ double scan[2][BH_THREADS];
    __shared__ double smin[BH_THREADS / 64];
    __shared__ long long smax[BH_THREADS / 64];
    const int t = threadIdx.x, b = blockIdx.x;
    // minimum of the blocks to the right, largest passing rank anywhere
    double right = INFINITY;
    long long kmax = -1;
    for (int u = t; u < nb; u += BH_THREADS) {
        if (u > b) right = fmin(right, bmin[u]);
        kmax = blast[u] > kmax ? blast[u] : kmax;
    }
    for (int o = 32; o > 0; o >>= 1) {
        right = fmin(right, __shfl_xor(right, o));
        const long long other = __shfl_xor(kmax, o);
        kmax = other > kmax ? other : kmax;
    }
    if ((t & 63) == 0) { smin[t >> 6] = right; smax[t >> 6] = kmax; }
    __syncthreads();
    for (int w = 0; w < BH_THREADS / 64; ++w) {
        right = fmin(right, smin[w]);
        kmax = smax[w] > kmax ? smax[w] : kmax;
    }
    // inclusive suffix minimum of this block's q_i (Hillis-Steele, double buffered)
    const long long i = (long long)b * BH_THREADS + t;
    double q;
    long long pass;
    bh_element(ps, i, n, alpha, q, pass);
    int cur = 0;
    scan[0][t] = q;
    __syncthreads();
    for (int o = 1; o < BH_THREADS; o <<= 1) {
        double v = scan[cur][t];
        if (t + o < BH_THREADS) v = fmin(v, scan[cur][t + o]);
        scan[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    if (i < n) {
        const double run = fmin(right, scan[cur][t]);
        const int o = order[i];
        padj[o] = run > 1.0 ? 1.0 : run;
        reject[o] = i <= kmax ? 1 : 0;
    }
}

// The rejections ALONE need no sort.  With T(c) = (c / n) alpha, Benjamini-Hochberg rejects the k* smallest p-values,
// k* = max{i : p_(i) <= T(i)}, and k* is the largest fixed point of  c -> #{p <= T(c)}:  from c = n the iteration falls
// monotonically and stops exactly at k* (T is monotone, so c >= k* is kept, and a fixed point c has p_(c) <= T(c), i.e.
// c <= k*); no tie straddles the boundary (p_(k*+1) = p_(k*) would pass at k* + 1), so {p <= T(k*)} is the sorted
// routine's set, element for element -- same T(c) expression as bh_element.  One workgroup, a handful of counting passes
// over values that sit in L2: the per-fold masks of a cross-validated fit (only they enter its majority vote) cost
// ~0.1 ms instead of a 16-pass radix sort each.  The iteration is geometric on real p-values, but p-values that hug the
// BH line from above (p_(i) ~ (i + 1/2) alpha / n) make it fall one step at a time: after BH_REJECT_MAX_PASSES passes
// without a fixed point *status is set to 1 and the mask is left unwritten: the caller then takes the sort-based path
// (lc_bh_fdr) for that vector, so the worst case is the sort's ~2 ms, not n passes.  (Queueing the sort's 50 launches
// behind every call with an early exit was measured: each still waits ~35 us for a CU beside the sweeps.)
constexpr int BH_REJECT_MAX_PASSES = 48;
__global__ void __launch_bounds__(1024) k_bh_reject(const double* __restrict__ p, long long n, double alpha,
                                                    unsigned char* __restrict__ reject, int* __restrict__ status) {
    __shared__ long long wsum[16];
    __shared__ long long total;
    const int t = threadIdx.x;
    long long c = n;
    double thr = 0.0;
    bool fixed = false;
    for (int pass = 0; pass < BH_REJECT_MAX_PASSES; ++pass) {
        thr = ((double)c / (double)n) * alpha;
        long long mine = 0;
        for (long long i = t; i < n; i += 1024) mine += p[i] <= thr ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if ((t & 63) == 0) wsum[t >> 6] = mine;
        __syncthreads();
        if (t == 0) {
            long long s = 0;
            for (int w = 0; w < 16; ++w) s += wsum[w];
            total = s;
        }
        __syncthreads();
        const long long cn = total;
        __syncthreads();
        if (cn == c) { fixed = true; break; }
        c = cn;
    }
    if (t == 0) *status = fixed ? 0 : 1;
    if (fixed)
        for (long long i = t; i < n; i += 1024) reject[i] = (c > 0 && p[i] <= thr) ? 1 : 0;
}

}  // namespace

extern "C" int lc_fisher_combine(const double* d_p, int k, int64_t V, double* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_p && d_out, LC_E_BADARG, "lc_fisher_combine: null pointer");
    LC_REQUIRE(k > 0 && V > 0, LC_E_SHAPE, "lc_fisher_combine: bad shape");
    hipLaunchKernelGGL(k_fisher, dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0, lc::as_stream(stream), d_p, k,
                       (long long)V, d_out);
    return lc::launched("k_fisher");
}

// workspace layout: keys A | keys B (n doubles each) | indices A | indices B (n ints each, 8-byte padded) | digit
// histograms (16 x blocks ints) + flag | per-block minimum and last passing rank of the step-up pass
static size_t bh_ints(int64_t n) { return (size_t)((n * 4 + 7) / 8) * 8; }

extern "C" int64_t lc_bh_fdr_work_bytes(int64_t n) {
    if (n <= 0 || n >= (1ll << 31)) return -1;
    const int64_t nblk = lc::ceil_div<int64_t>(n, RS_TILE);
    return (int64_t)(2 * n * 8 + 2 * bh_ints(n) + ((size_t)(RS_BINS * nblk + 2) * 4 + 7) / 8 * 8 +
                     lc::ceil_div<int64_t>(n, BH_THREADS) * 16 + 64);
}

extern "C" int lc_bh_fdr(const double* d_p, int64_t n, double alpha, uint8_t* d_reject, double* d_padj, void* d_work,
                         int64_t work_bytes, lc_stream_t stream) {
    LC_REQUIRE(d_p && d_reject && d_padj && d_work, LC_E_BADARG, "lc_bh_fdr: null pointer");
    LC_REQUIRE(n > 0 && n < (1ll << 31), LC_E_SHAPE, "lc_bh_fdr: bad length");
    LC_REQUIRE(work_bytes >= lc_bh_fdr_work_bytes(n), LC_E_SHAPE, "lc_bh_fdr: workspace too small (lc_bh_fdr_work_bytes)");
    hipStream_t s = lc::as_stream(stream);
    char* w = static_cast<char*>(d_work);
    const size_t ints = bh_ints(n);
    const int nblk = (int)lc::ceil_div<int64_t>(n, RS_TILE);
    double* kbuf[2] = {reinterpret_cast<double*>(w), reinterpret_cast<double*>(w + n * 8)};
    int* vbuf[2] = {reinterpret_cast<int*>(w + 2 * n * 8), reinterpret_cast<int*>(w + 2 * n * 8 + ints)};
    int* hist = reinterpret_cast<int*>(w + 2 * n * 8 + 2 * ints);
    int* trivial = hist + (size_t)RS_BINS * nblk;
    const size_t hist_bytes = ((size_t)(RS_BINS * nblk + 2) * 4 + 7) / 8 * 8;
    const int nb = (int)lc::ceil_div<int64_t>(n, BH_THREADS);
    double* bmin = reinterpret_cast<double*>(w + 2 * n * 8 + 2 * ints + hist_bytes);
    long long* blast = reinterpret_cast<long long*>(w + 2 * n * 8 + 2 * ints + hist_bytes + (size_t)nb * 8);
    // 16 passes; pass 0 reads the input (payload = position) and writes buffer 1, odd passes write buffer 0: the sorted
    // pairs end in buffer 0
    const double* kin = d_p;
    const int* vin = nullptr;
    for (int pass = 0; pass < 16; ++pass) {
        const int out = (pass + 1) & 1;
        hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)nblk), dim3(RS_THREADS), 0, s, kin, (long long)n, 4 * pass, hist, nblk);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, s, hist, nblk, (long long)n, trivial);
        hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)nblk), dim3(RS_THREADS), 0, s, kin, vin, (long long)n, 4 * pass, hist,
                           nblk, trivial, kbuf[out], vbuf[out]);
        kin = kbuf[out];
        vin = vbuf[out];
    }
    if (int rc = lc::launched("radix sort")) return rc;
    const double* keys = kbuf[0];
    const int* order = vbuf[0];
    hipLaunchKernelGGL(k_bh_block_min, dim3((unsigned)nb), dim3(BH_THREADS), 0, s, keys, (long long)n, alpha, bmin, blast);
    hipLaunchKernelGGL(k_bh_apply, dim3((unsigned)nb), dim3(BH_THREADS), 0, s, keys, order, (long long)n, alpha, bmin, blast, nb,
                       d_reject, d_padj);
    return lc::launched("k_bh_apply");
}

extern "C" int lc_bh_reject(const double* d_p, int64_t n, double alpha, uint8_t* d_reject, int32_t* d_status,
                            lc_stream_t stream) {
    LC_REQUIRE(d_p && d_reject && d_status, LC_E_BADARG, "lc_bh_reject: null pointer");
    LC_REQUIRE(n > 0 && n < (1ll << 31), LC_E_SHAPE, "lc_bh_reject: bad length");
    hipLaunchKernelGGL(k_bh_reject, dim3(1), dim3(1024), 0, lc::as_stream(stream), d_p, (long long)n, alpha, d_reject, d_status);
    return lc::launched("k_bh_reject");
}
