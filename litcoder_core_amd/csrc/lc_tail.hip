// Statistics tail on the device: Fisher's combination of the per-fold p-values and the
// Benjamini-Hochberg step-up procedure.  O(V) work on vectors that are already in HBM; the sort is hipCUB's radix
// sort (rocPRIM), everything else is a single workgroup.
//   reference: nested_cv.py:441-477 (_combine_pvalues_across_folds), statsmodels fdrcorrection(method="indep")
//   at nested_cv.py:158,263,282 -- host twins with the same arithmetic: litcoder_core_amd/stats.py.
#include <hipcub/hipcub.hpp>

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

__global__ void k_iota(int* __restrict__ idx, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (int)i;
}

// Sorted p-values -> adjusted p-values and rejections, scattered back to input order.  One workgroup: thread t owns
// the contiguous chunk [t*c, (t+1)*c) of the sorted vector; the suffix minimum and the largest passing rank are
// combined across threads through LDS.
constexpr int BH_THREADS = 1024;
__global__ void __launch_bounds__(BH_THREADS) k_bh_sorted(const double* __restrict__ ps, const int* __restrict__ order,
                                                          long long n, double alpha, unsigned char* __restrict__ reject,
                                                          double* __restrict__ padj) {
    __shared__ double smin[BH_THREADS];
    __shared__ long long smax[BH_THREADS];
    const int t = threadIdx.x;
    const long long c = (n + BH_THREADS - 1) / BH_THREADS;
    const long long lo = (long long)t * c, hi = lo + c < n ? lo + c : n;
    const double dn = (double)n;
    // chunk minimum of p_(i) / frac_i and the largest i in the chunk with p_(i) <= frac_i * alpha
    double m = INFINITY;
    long long last = -1;
    for (long long i = hi - 1; i >= lo; --i) {
        const double frac = (double)(i + 1) / dn;
        m = fmin(m, ps[i] / frac);
        if (last < 0 && ps[i] <= frac * alpha) last = i;
    }
    smin[t] = m;
    smax[t] = last;
    __syncthreads();
    // exclusive suffix minimum over the chunks to the right, global maximum passing rank (serial: 1024 entries)
    double right = INFINITY;
    for (int u = BH_THREADS - 1; u > t; --u) right = fmin(right, smin[u]);
    long long kmax = -1;
    for (int u = 0; u < BH_THREADS; ++u) kmax = smax[u] > kmax ? smax[u] : kmax;
    double run = right;
    for (long long i = hi - 1; i >= lo; --i) {
        const double frac = (double)(i + 1) / dn;
        run = fmin(run, ps[i] / frac);
        const int o = order[i];
        padj[o] = run > 1.0 ? 1.0 : run;
        reject[o] = i <= kmax ? 1 : 0;
    }
}

}  // namespace

extern "C" int lc_fisher_combine(const double* d_p, int k, int64_t V, double* d_out, lc_stream_t stream) {
    LC_REQUIRE(d_p && d_out, LC_E_BADARG, "lc_fisher_combine: null pointer");
    LC_REQUIRE(k > 0 && V > 0, LC_E_SHAPE, "lc_fisher_combine: bad shape");
    hipLaunchKernelGGL(k_fisher, dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0, lc::as_stream(stream), d_p, k,
                       (long long)V, d_out);
    return lc::launched("k_fisher");
}

// workspace layout: sorted keys (n doubles) | sorted indices (n ints) | input indices (n ints) | hipCUB temporary
extern "C" int64_t lc_bh_fdr_work_bytes(int64_t n) {
    if (n <= 0 || n >= (1ll << 31)) return -1;
    size_t tmp = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const double*)nullptr, (double*)nullptr, (const int*)nullptr,
                                           (int*)nullptr, (int)n) != hipSuccess)
        return -1;
    return (int64_t)(n * 8 + ((n * 4 + 7) / 8) * 8 * 2 + tmp + 64);
}

extern "C" int lc_bh_fdr(const double* d_p, int64_t n, double alpha, uint8_t* d_reject, double* d_padj, void* d_work,
                         int64_t work_bytes, lc_stream_t stream) {
    LC_REQUIRE(d_p && d_reject && d_padj && d_work, LC_E_BADARG, "lc_bh_fdr: null pointer");
    LC_REQUIRE(n > 0 && n < (1ll << 31), LC_E_SHAPE, "lc_bh_fdr: bad length");
    LC_REQUIRE(work_bytes >= lc_bh_fdr_work_bytes(n), LC_E_SHAPE, "lc_bh_fdr: workspace too small (lc_bh_fdr_work_bytes)");
    hipStream_t s = lc::as_stream(stream);
    char* w = static_cast<char*>(d_work);
    const size_t ints = (size_t)((n * 4 + 7) / 8) * 8;
    double* keys = reinterpret_cast<double*>(w);
    int* order = reinterpret_cast<int*>(w + n * 8);
    int* iota = reinterpret_cast<int*>(w + n * 8 + ints);
    void* tmp = w + n * 8 + 2 * ints;
    size_t tmp_bytes = (size_t)work_bytes - (n * 8 + 2 * ints);
    hipLaunchKernelGGL(k_iota, dim3((unsigned)lc::ceil_div<long long>(n, 256)), dim3(256), 0, s, iota, (long long)n);
    LC_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, d_p, keys, iota, order, (int)n, 0, 64, s));
    hipLaunchKernelGGL(k_bh_sorted, dim3(1), dim3(BH_THREADS), 0, s, keys, order, (long long)n, alpha, d_reject, d_padj);
    return lc::launched("k_bh_sorted");
}
