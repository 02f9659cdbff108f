#!/usr/bin/env python3
"""One batch shape of the batched Cholesky solve, alone, a few times (for rocprofv3 --kernel-trace --stats):
    python tools/chol_profile.py [B N M [inverse]]       default 80 1920 480 (the inner folds of outer folds 1-4 at cfg2)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops
dev = ops.device(0)
B, N, M = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (80, 1920, 480)))
inverse = "inverse" in sys.argv[4:]
g = torch.Generator(device=dev); g.manual_seed(B + N)
X = torch.randn((B, N, N + 8), dtype=torch.float64, device=dev, generator=g)
base = torch.empty((B, N + M, N), dtype=torch.float64, device=dev)
base[:, :N] = X @ X.transpose(1, 2) / N + 0.05 * torch.eye(N, dtype=torch.float64, device=dev)
if inverse:
    base[:, N:] = torch.eye(N, dtype=torch.float64, device=dev)
else:
    base[:, N:] = torch.randn((B, M, N), dtype=torch.float64, device=dev, generator=g)
del X
aug = base.clone()
H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
fl = B * (N ** 3 / 3 + 2.0 * N * N * M)
for i in range(4):
    aug.copy_(base)
    torch.cuda.synchronize()
    t = time.perf_counter()
    (ops.batch_chol_inverse(aug, B, N, H) if inverse else ops.batch_chol_solve(aug, B, N, M, H))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"B={B} N={N} M={M}{' inverse' if inverse else ''}: {dt * 1e3:.2f} ms ({fl / dt / 1e12:.1f} TF)")
