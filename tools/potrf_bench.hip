// Micro-benchmark of the 64 x 64 diagonal tile of the batched Cholesky (potrf_diag_tile of csrc/lc_chol.hip, included as a
// translation unit): B systems, the product kernel against variants that stop after the factorisation / skip it, to see
// where its ~40 us go.   hipcc -O3 --offload-arch=gfx950 tools/potrf_bench.hip -o /tmp/potrf_bench && /tmp/potrf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define LC_POTRF_BENCH 1
#include "../litcoder_core_amd/csrc/lc_chol.hip"

namespace lc {
void set_error(const char*, ...) {}
int ensure_dynamic_lds(const void*, int) { return 0; }
bool timing_on(int) { return false; }
void timing_begin(int, hipStream_t) {}
void timing_end(int, hipStream_t) {}
}  // namespace lc

namespace {
template <int MODE>
__global__ void __launch_bounds__(256) k_variant(double* aug, int N, int M, int k, double* linv, int* info) {
    __shared__ double L[NB * PD_LD];
    __shared__ double rdiag[NB + 768];
    potrf_diag_tile_v<false, MODE>(aug, N, M, k, blockIdx.x, linv, info, L, rdiag);
}
}  // namespace

int main() {
    const int B = 20, N = 64, M = 32, R = N + M;
    std::vector<double> h((size_t)B * R * N, 0.0);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j <= i; ++j) {
                double v = 0.0;
                for (int q = 0; q < 8; ++q) v += std::sin(0.37 * (i + 1) * (q + 1) + b) * std::sin(0.37 * (j + 1) * (q + 1) + b);
                h[((size_t)b * R + i) * N + j] = h[((size_t)b * R + j) * N + i] = v / 8 + (i == j ? 2.0 : 0.0);
            }
    double *d_aug, *d_ref, *d_linv; int* d_info;
    hipMalloc(&d_aug, h.size() * 8); hipMalloc(&d_ref, h.size() * 8); hipMalloc(&d_linv, (size_t)B * N * N * 8); hipMalloc(&d_info, B * 4);
    hipMemcpy(d_ref, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(d_info, 0, B * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemcpy(d_aug, d_ref, h.size() * 8, hipMemcpyDeviceToDevice);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(B), dim3(256), 0, 0, d_aug, N, M, 0, d_linv, d_info);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        std::printf("%-44s %7.1f us\n", name, best * 1e3f);
    };
    run(k_variant<0>, "product (factor + Linv + stores)");
    run(k_variant<1>, "factor only (no Linv)");
    run(k_variant<2>, "load + store only");
    run(k_variant<3>, "Linv only (on the unfactored tile)");
    std::vector<double> li((size_t)N * N);
    hipLaunchKernelGGL(k_variant<0>, dim3(B), dim3(256), 0, 0, d_aug, N, M, 0, d_linv, d_info);
    hipMemcpy(li.data(), d_linv, li.size() * 8, hipMemcpyDeviceToHost);
    std::printf("Linv[0][0] = %.6f  Linv[63][63] = %.6f\n", li[0], li[63 * 64 + 63]);
    return 0;
}
