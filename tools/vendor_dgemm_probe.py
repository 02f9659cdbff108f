"""Diagnostic: what the vendor BLAS reaches on fp64 products of the Cholesky's deep-update shapes (for comparison
with k_mm64v; nothing in the library calls it).  Run under rocprofv3 --kernel-trace to see the kernels it picks."""
import time
import torch

dev = torch.device("cuda:0")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for (B, M, N, K) in ((1, 4096, 4096, 4096), (20, 2144, 1664, 256), (20, 2144, 1664, 512), (4, 5856, 2176, 256),
                     (4, 3680, 1024, 256), (20, 480, 1024, 256)):
    a = torch.randn((B, M, K), dtype=torch.float64, device=dev)
    b = torch.randn((B, N, K), dtype=torch.float64, device=dev)
    c = torch.randn((B, M, N), dtype=torch.float64, device=dev)
    ms = timeit(lambda: torch.baddbmm(c, a, b.transpose(1, 2), beta=1.0, alpha=-1.0, out=c))
    print(f"baddbmm fp64 B={B} M={M} N={N} K={K} (N x T): {ms:.3f} ms -> {2.0 * B * M * N * K / ms / 1e9:.1f} TFLOP/s")
