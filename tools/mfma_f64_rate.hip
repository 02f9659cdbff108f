// Diagnostic: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on gfx950 (cycles per instruction, TFLOP/s).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o tools/bin/mfma_f64_rate && tools/bin/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(int iters, double* out, unsigned long long* cyc) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x & 2047] = t1 - t0; cyc[2048 + (blockIdx.x & 2047)] = r1 - r0; }
}


// accumulators pinned to AGPRs (the "a" register class), as a compiler does under register pressure
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_agpr(int iters, double* out, unsigned long long* cyc) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x & 2047] = t1 - t0; cyc[2048 + (blockIdx.x & 2047)] = r1 - r0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) k_fma(int iters, double* out, unsigned long long* cyc) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x & 2047] = t1 - t0; cyc[2048 + (blockIdx.x & 2047)] = r1 - r0; }
}


// variants of the bare MFMA loop: distinct operand registers, fillers between MFMAs, the 4x4x4 (4-block) shape
template <int MODE>
__global__ void __launch_bounds__(256) k_mfma_var(int iters, double* out, unsigned long long* cyc) {
    f64x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 0.5 + threadIdx.x * 1e-4 - i; }
    int filler = threadIdx.x;
    double acc4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 4) {
                acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[(i >> 1) & 3], acc4[i], 0, 0, 0);
            } else {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
                if (MODE == 2) asm volatile("s_nop 1");
                if (MODE == 3) asm volatile("v_add_u32 %0, %0, 1" : "+v"(filler));
            }
        }
    }
    double s = filler;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x & 2047] = 1; cyc[2048 + (blockIdx.x & 2047)] = 1; }
}

// 4x4x4 MFMA with a GEMM-like register pattern: 64 accumulators, A fragment shared by 8 consecutive MFMAs
template <int MODE>
__global__ void __launch_bounds__(256, 1) k_mfma44(int iters, double* out, unsigned long long* cyc) {
    double acc[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = 0;
    double a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 0.5 + threadIdx.x * 1e-4 - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(i + j) & 7], b[j], acc[i][j], 0, 0, 0);
            }
        // keep the operands live and changing so that nothing is hoisted
        for (int i = 0; i < 8; ++i) { a[i] += 1e-9; b[i] -= 1e-9; }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x & 2047] = 1; cyc[2048 + (blockIdx.x & 2047)] = 1; }
}

// half of the waves issue MFMAs, the other half vector FMAs: do the two share one fp64 datapath?
__global__ void __launch_bounds__(256) k_mix(int iters, double* out, unsigned long long* cyc) {
    const bool mf = (threadIdx.x >> 6) & 1;
    f64x4 acc[8];
    double v[16];
    for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) v[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    if (mf) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else {
        for (int it = 0; it < iters * 10; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x & 2047] = 1; cyc[2048 + (blockIdx.x & 2047)] = 1; }
}

static void run_mix(int blocks) {
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 4096);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, 100, out, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per block: 2 MFMA waves x iters x 8 x 2048 flop, 2 FMA waves x iters x 10 x 16 x 128 flop
    const double fm = 2.0 * iters * 8 * 2048.0 * blocks, fv = 2.0 * iters * 10.0 * 16 * 128.0 * blocks;
    printf("mixed, %d blocks (half the waves MFMA, half FMA; the longer half sets the time): %.3f ms; MFMA part alone "
           "would be %.1f TFLOP/s, FMA part %.1f TFLOP/s, sum %.1f TFLOP/s\n", blocks, ms, fm / (ms * 1e9), fv / (ms * 1e9),
           (fm + fv) / (ms * 1e9));
    (void)hipFree(out); (void)hipFree(cyc);
}

template <typename K>
static void run(K kern, const char* name, int blocks, int per_iter, double flop_per_inst) {
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 4096);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, 100, out, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4096];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double insts = (double)iters * per_iter;
    const double total = insts * 4.0 * blocks * flop_per_inst;   // 4 waves per block
    printf("%s: %d blocks: %.1f cycles per instruction and wave (clock %.2f GHz), %.1f TFLOP/s\n", name, blocks,
           (double)h[0] / insts, (double)h[0] / ((double)h[2048] * 10.0), total / (ms * 1e9));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int m = 1; m <= 8; m *= 2) {
        char name[128];
        snprintf(name, sizeof name, "v_mfma_f64_16x16x4_f64, %d wave(s)/SIMD, 8 accumulators", m);
        run(k_mfma<8>, name, 256 * m, 8, 2048.0);
    }
    run(k_mfma<1>, "v_mfma_f64_16x16x4_f64, 1 wave/SIMD, dependent chain", 256, 1, 2048.0);
    for (int m = 1; m <= 4; m *= 2) {
        char name[128];
        snprintf(name, sizeof name, "v_mfma_f64_16x16x4_f64 AGPR accumulators, %d wave(s)/SIMD, 8 accumulators", m);
        run(k_mfma_agpr<8>, name, 256 * m, 8, 2048.0);
        snprintf(name, sizeof name, "v_mfma_f64_16x16x4_f64 AGPR accumulators, %d wave(s)/SIMD, 16 accumulators", m);
        run(k_mfma_agpr<16>, name, 256 * m, 16, 2048.0);
    }
    run(k_mfma<4>, "v_mfma_f64_16x16x4_f64, 4 waves/SIMD, 4 accumulators", 1024, 4, 2048.0);
    for (int m = 1; m <= 8; m *= 2) {
        char name[128];
        snprintf(name, sizeof name, "v_fma_f64, %d wave(s)/SIMD, 16 accumulators", m);
        run(k_fma<16>, name, 256 * m, 16, 128.0);
    }
    for (int m = 1; m <= 4; m *= 2) {
        char name[160];
        snprintf(name, sizeof name, "MFMA 16x16x4, 4 operand pairs, %d wave(s)/SIMD", m);
        run(k_mfma_var<1>, name, 256 * m, 8, 2048.0);
        snprintf(name, sizeof name, "MFMA 16x16x4, 4 operand pairs + s_nop, %d wave(s)/SIMD", m);
        run(k_mfma_var<2>, name, 256 * m, 8, 2048.0);
        snprintf(name, sizeof name, "MFMA 16x16x4, 4 operand pairs + VALU filler, %d wave(s)/SIMD", m);
        run(k_mfma_var<3>, name, 256 * m, 8, 2048.0);
        snprintf(name, sizeof name, "MFMA 4x4x4 (4 blocks), %d wave(s)/SIMD", m);
        run(k_mfma_var<4>, name, 256 * m, 8, 512.0);
    }
    run(k_mfma44<0>, "MFMA 4x4x4, 64 accumulators, A shared by 8 consecutive, 1 wave/SIMD", 256, 64, 512.0);
    run(k_mfma44<1>, "MFMA 4x4x4, 64 accumulators, A rotating, 1 wave/SIMD", 256, 64, 512.0);
    run_mix(512);
    run_mix(1024);
    run_mix(2048);
    return 0;
}
