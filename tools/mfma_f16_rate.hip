// Diagnostic: sustained v_mfma_f32_32x32x16_f16 rate on gfx950 and the frequency of the s_memtime counter.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f16_rate.hip -o tools/bin/mfma_f16_rate && tools/bin/mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(512) k_rate(int iters, float* out, unsigned long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(threadIdx.x * 0.001f + r); b[r] = (_Float16)(r * 0.5f); }
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = r1 - r0; cyc[512 + blockIdx.x] = r0; }
}

// the same flops per wave with v_mfma_f32_16x16x32_f16 (4 accumulator VGPRs per block, 16 cycles per MFMA)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(512) k_rate16(int iters, float* out, unsigned long long* cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(threadIdx.x * 0.001f + r); b[r] = (_Float16)(r * 0.5f); }
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = r1 - r0; cyc[512 + blockIdx.x] = r0; }
}

void run16(int threads) {
    const int blocks = 256, iters = 4000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, 3 * blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate16<32>, dim3(blocks), dim3(threads), 0, 0, 10, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate16<32>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[768]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { mean += h[i]; real += h[256 + i]; }
    mean /= blocks; real /= blocks;
    const double flops = (double)iters * 3 * 32 * 16384.0 * blocks * (threads / 64);
    printf("16x16x32, %d waves/SIMD, 32 accumulators: %.3f ms  %.0f TFLOP/s | shader clock %.2f GHz (s_memtime / s_memrealtime)\n",
           threads / 256, ms, flops / ms / 1e9, mean / real * 0.1);
}

template <int NACC>
void run(int threads, const char* name) {
    const int blocks = 256, iters = 4000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, 3 * blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, 10, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[768]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0, real = 0; unsigned long long rmin = ~0ull, rmax = 0;
    for (int i = 0; i < blocks; ++i) { mean += h[i]; real += h[256 + i]; rmin = h[512 + i] < rmin ? h[512 + i] : rmin; rmax = h[512 + i] > rmax ? h[512 + i] : rmax; }
    mean /= blocks; real /= blocks;
    printf("   per-wave s_memrealtime span %.0f ticks (= %.3f ms at 100 MHz); block start times spread over %.3f ms; shader clock %.2f GHz\n", real, real / 1e5, (rmax - rmin) / 1e5, mean / real * 0.1);
    const double mfmas_per_wave = (double)iters * 3 * NACC;
    const double flops = mfmas_per_wave * 32768.0 * blocks * (threads / 64);
    printf("%s: %.3f ms  %.0f TFLOP/s  | s_memtime delta %.0f -> counter %.3f GHz, %.1f counts per MFMA per SIMD-wave-slot\n", name, ms,
           flops / ms / 1e9, mean, mean / (ms * 1e6), mean / mfmas_per_wave / (threads / 256.0));
}

// The two shapes with the fp16x3 kernel's LDS traffic beside them: per K-step of a 128 x 64 wave tile the fragments
// come from LDS as conflict-free ds_read_b128 -- 12 reads per 24 MFMAs (32x32x16, K = 16) or 24 reads per 96 MFMAs
// (16x16x32, K = 32) -- with the three-term product hi*hi + hi*lo + lo*hi.  Same flops per wave in both.
template <bool WIDE>
__global__ void __launch_bounds__(512) k_rate_lds(int iters, float* out, unsigned long long* cyc) {
    __shared__ uint4 lds[2 * 4096];                       // 128 KB like the kernel's ring
    for (int i = threadIdx.x; i < 2 * 4096; i += 512) lds[i] = uint4{0x3c003c00u + i, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0, t1, r0, r1;
    float s = 0.f;
    if (!WIDE) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
            const uint4* st = lds + ((it & 3) * 2048) + wave * 64 + lane;
            h8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint4 v = st[q * 512]; ah[q] = *(h8*)&v; uint4 u = st[q * 512 + 256]; al[q] = *(h8*)&u; }
#pragma unroll
            for (int q = 0; q < 2; ++q) { uint4 v = st[q * 512 + 128]; bh[q] = *(h8*)&v; uint4 u = st[q * 512 + 384]; bl[q] = *(h8*)&u; }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi * 2 + ni], 0, 0, 0);
                    acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi * 2 + ni], 0, 0, 0);
                    acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi * 2 + ni], 0, 0, 0);
                }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; it += 2) {           // one K = 32 step = two of the other kernel's iterations
            const uint4* st = lds + ((it & 2) * 2048) + wave * 64 + lane;
            h8 ah[8], al[8], bh[4], bl[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) { uint4 v = st[q * 256]; ah[q] = *(h8*)&v; uint4 u = st[q * 256 + 2048]; al[q] = *(h8*)&u; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint4 v = st[q * 512 + 128]; bh[q] = *(h8*)&v; uint4 u = st[q * 512 + 2176]; bl[q] = *(h8*)&u; }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh[ni], acc[mi * 4 + ni], 0, 0, 0);
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl[ni], acc[mi * 4 + ni], 0, 0, 0);
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh[ni], acc[mi * 4 + ni], 0, 0, 0);
                }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = r1 - r0; }
}

template <bool WIDE>
void run_lds(const char* name) {
    const int blocks = 256, iters = 4000, threads = 512;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, 3 * blocks * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate_lds<WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate_lds<WIDE>, dim3(blocks), dim3(threads), 0, 0, 16, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate_lds<WIDE>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[768]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { mean += h[i]; real += h[256 + i]; }
    const double flops = (double)iters * 24 * 32768.0 * blocks * (threads / 64);
    printf("%s: %.3f ms  %.0f TFLOP/s of fp16 MFMA | shader clock %.2f GHz\n", name, ms, flops / ms / 1e9, mean / real * 0.1);
}

int main() {
    run_lds<false>("32x32x16 with the kernel's LDS fragment reads (12 b128 per 24 MFMAs), 2 waves/SIMD");
    run_lds<true>("16x16x32 with the kernel's LDS fragment reads (24 b128 per 96 MFMAs), 2 waves/SIMD");
    run<8>(256, "1 wave/SIMD, 8 accumulators");
    run<8>(512, "2 waves/SIMD, 8 accumulators");
    run<1>(512, "2 waves/SIMD, 1 accumulator (dependent chain)");
    run16(256);
    run16(512);
    return 0;
}
