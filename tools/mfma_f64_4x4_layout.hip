// Diagnostic: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 and its CBSZ / ABID broadcast, found by
// one-hot operands.   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_4x4_layout.hip -o tools/bin/mfma_f64_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void k_onehot(double* out) {
    const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    out[((long long)la * 64 + lb) * 64 + lane] = d;
}

template <int CBSZ, int ABID>
static void run(const char* name) {
    double* d;
    (void)hipMalloc(&d, 64 * 64 * 64 * 8);
    hipLaunchKernelGGL((k_onehot<CBSZ, ABID>), dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%s\n", name);
    for (int la : {0, 1, 2, 4, 8, 16, 17, 32, 63}) {
        printf("  A one-hot at lane %2d:", la);
        int n = 0;
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((long long)la * 64 + lb) * 64 + l] != 0.0 && n++ < 20) printf(" (B%d->D%d)", lb, l);
        printf("  [%d pairs]\n", n);
    }
    (void)hipFree(d);
}

int main() {
    run<0, 0>("cbsz 0");
    run<1, 0>("cbsz 1 abid 0");
    run<1, 1>("cbsz 1 abid 1");
    run<2, 0>("cbsz 2 abid 0");
    run<2, 1>("cbsz 2 abid 1");
    run<2, 3>("cbsz 2 abid 3");
    return 0;
}
