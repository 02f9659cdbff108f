// Voxel shards: the small device-side pieces around the exchanges of a sharded fit (gfx950).
//
// A fold's per-voxel results leave a rank as ONE packed (4, ld) f64 block in natural voxel order -- r, p, chosen
// alpha index, and the Cholesky pivot flags -- so that one all-gather (RCCL over xGMI; torch.distributed on the host
// side) moves everything the global statistics need; lc_fold_unpack turns the gathered blocks of all ranks back into
// V_total-long vectors.  With one rank the same two kernels run (no collective in between): one code path.
#include "lc_common.h"

namespace {

__global__ void __launch_bounds__(256) k_fold_pack(const double* __restrict__ r_s, const double* __restrict__ p_s,
                                                   const int* __restrict__ perm, long long Vs,
                                                   const int* __restrict__ best, long long V,
                                                   const int* __restrict__ info_a, int n_a,
                                                   const int* __restrict__ info_b, int n_b,
                                                   double* __restrict__ out, long long ld, long long col0) {
    // out points at column col0 of the block (this panel's voxels); the flags live in columns 0, 1 of row 3
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j < Vs) {
        const int v = perm[j];
        if (v >= 0) {
            out[v] = r_s[j];
            out[ld + v] = p_s[j];
        }
    }
    if (j < V) out[2 * ld + j] = (double)best[j];
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // any pivot failure in the inner-fold systems / in the refit systems: one wave, ballot
        int bad_a = 0, bad_b = 0;
        for (int i = threadIdx.x; i < n_a; i += 64) bad_a |= info_a[i] != 0;
        for (int i = threadIdx.x; i < n_b; i += 64) bad_b |= info_b[i] != 0;
        const unsigned long long ma = __ballot(bad_a), mb = __ballot(bad_b);
        if (threadIdx.x == 0) {                       // OR into the block's flags (cleared by the first panel's memset)
            if (ma) out[3 * ld - col0 + 0] = 1.0;
            if (mb) out[3 * ld - col0 + 1] = 1.0;
        }
    }
}

__global__ void __launch_bounds__(256) k_fold_unpack(const double* __restrict__ src, int world, long long ld,
                                                     const long long* __restrict__ lo, double* __restrict__ r,
                                                     double* __restrict__ p, int* __restrict__ idx,
                                                     double* __restrict__ p_clean, int* __restrict__ bad) {
    const int rk = blockIdx.y;
    const long long w = lo[rk + 1] - lo[rk];
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    const double* blk = src + (long long)rk * 4 * ld;
    if (c < w) {
        const long long v = lo[rk] + c;
        const double rv = blk[c], pv = blk[ld + c];
        r[v] = rv;
        p[v] = pv;
        idx[v] = (int)blk[2 * ld + c];
        p_clean[v] = (rv != rv) ? 1.0 : pv;           // NaN r -> p = 1 (nested_cv.py:436)
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 && blk[3 * ld + threadIdx.x] != 0.0) atomicOr(bad + threadIdx.x, 1);
}

__global__ void __launch_bounds__(256) k_fill_argmax(const double* __restrict__ rowsum, int A, int* __restrict__ best,
                                                     long long V) {
    int k = 0;
    double m = rowsum[0];
    for (int a = 1; a < A; ++a) {                       // first maximum, like torch.argmax; NaN never wins -- not
        const double x = rowsum[a];                     // as the seed either (all NaN: index 0)
        if (x > m || (m != m && x == x)) { m = x; k = a; }
    }
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v < V) best[v] = k;
}

__global__ void __launch_bounds__(256) k_accumulate_f64(const double* __restrict__ x, double* __restrict__ acc, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

}  // namespace

extern "C" int lc_accumulate_f64(const double* d_x, double* d_acc, int64_t n, lc_stream_t stream) {
    LC_REQUIRE(d_x && d_acc && n >= 0, LC_E_BADARG, "lc_accumulate_f64: bad argument");
    if (n == 0) return LC_OK;
    hipLaunchKernelGGL(k_accumulate_f64, dim3((unsigned)lc::ceil_div<long long>(n, 256)), dim3(256), 0, lc::as_stream(stream),
                       d_x, d_acc, (long long)n);
    return lc::launched("k_accumulate_f64");
}

extern "C" int lc_fold_pack_at(const double* d_r_sorted, const double* d_p_sorted, const int32_t* d_perm, int64_t Vs,
                               const int32_t* d_best, int64_t V, const int32_t* d_info_a, int n_a, const int32_t* d_info_b,
                               int n_b, double* d_out, int64_t ld, int64_t col0, int clear, lc_stream_t stream) {
    LC_REQUIRE(d_r_sorted && d_p_sorted && d_perm && d_best && d_out, LC_E_BADARG, "lc_fold_pack: null pointer");
    LC_REQUIRE(Vs >= 0 && V >= 0 && col0 >= 0 && ld >= col0 + V && ld >= 2 && (n_a == 0 || d_info_a) && (n_b == 0 || d_info_b),
               LC_E_SHAPE, "lc_fold_pack: need ld >= max(col0 + V, 2)");
    hipStream_t s = lc::as_stream(stream);
    if (clear) LC_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * 4 * ld, s));
    const long long n = Vs > V ? Vs : V;
    hipLaunchKernelGGL(k_fold_pack, dim3((unsigned)lc::ceil_div<long long>(n > 0 ? n : 1, 256)), dim3(256), 0, s, d_r_sorted,
                       d_p_sorted, d_perm, (long long)Vs, d_best, (long long)V, d_info_a, n_a, d_info_b, n_b, d_out + col0,
                       (long long)ld, (long long)col0);
    return lc::launched("k_fold_pack");
}

extern "C" int lc_fold_pack(const double* d_r_sorted, const double* d_p_sorted, const int32_t* d_perm, int64_t Vs,
                            const int32_t* d_best, int64_t V, const int32_t* d_info_a, int n_a, const int32_t* d_info_b,
                            int n_b, double* d_out, int64_t ld, lc_stream_t stream) {
    return lc_fold_pack_at(d_r_sorted, d_p_sorted, d_perm, Vs, d_best, V, d_info_a, n_a, d_info_b, n_b, d_out, ld, 0, 1, stream);
}

extern "C" int lc_fold_unpack(const double* d_src, int world, int64_t ld, const int64_t* d_lo, int64_t w_max, double* d_r,
                              double* d_p, int32_t* d_idx, double* d_p_clean, int32_t* d_bad, lc_stream_t stream) {
    LC_REQUIRE(d_src && d_lo && d_r && d_p && d_idx && d_p_clean && d_bad, LC_E_BADARG, "lc_fold_unpack: null pointer");
    LC_REQUIRE(world >= 1 && world <= 65535 && ld >= 2 && w_max >= 0 && w_max <= ld, LC_E_SHAPE, "lc_fold_unpack: bad shape");
    hipStream_t s = lc::as_stream(stream);
    LC_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t) * 2, s));
    hipLaunchKernelGGL(k_fold_unpack, dim3((unsigned)lc::ceil_div<long long>(w_max > 0 ? w_max : 1, 256), (unsigned)world),
                       dim3(256), 0, s, d_src, world, (long long)ld, reinterpret_cast<const long long*>(d_lo), d_r, d_p,
                       d_idx, d_p_clean, d_bad);
    return lc::launched("k_fold_unpack");
}

extern "C" int lc_fill_argmax(const double* d_rowsum, int A, int32_t* d_best, int64_t V, lc_stream_t stream) {
    LC_REQUIRE(d_rowsum && d_best && A > 0 && V >= 0, LC_E_BADARG, "lc_fill_argmax: bad argument");
    if (V == 0) return LC_OK;
    hipLaunchKernelGGL(k_fill_argmax, dim3((unsigned)lc::ceil_div<long long>(V, 256)), dim3(256), 0, lc::as_stream(stream),
                       d_rowsum, A, d_best, (long long)V);
    return lc::launched("k_fill_argmax");
}
