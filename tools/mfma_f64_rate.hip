// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double* d;
    hipMalloc(&d, 1024 * 256 * 8 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpb : {1, 2}) {
        const int blocks = 256 * wpb, iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 * iters * 8 * 2048.0;
            if (rep) printf("mfma_f64 16x16x4: blocks/CU=%d  %.2f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpb, ms,
                            flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * wpb));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            flops = (double)blocks * 256 * iters * 16 * 2.0;
            if (rep) printf("v_fma_f64: blocks/CU=%d  %.2f ms  %.1f TFLOP/s\n", wpb, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
