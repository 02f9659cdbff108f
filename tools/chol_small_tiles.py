#!/usr/bin/env python3
"""The deep updates of the batched Cholesky in 64 x 64 tiles (k_mm64s) against the 128 x 128 kernel (k_mm64q), by batch size:
times, and whether the results are the same bits.      python tools/chol_small_tiles.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops
dev = ops.device(0)
for (B, N, M) in ((1, 1920, 480), (3, 1920, 480), (6, 1920, 480), (10, 1920, 480), (20, 1920, 480), (40, 1920, 480), (80, 1920, 480),
                  (1, 2432, 1920), (1, 2432, 2432), (4, 2432, 2432), (12, 2432, 2432), (4, 2432, 3680), (3, 3072, 1856)):
    g = torch.Generator(device=dev); g.manual_seed(B + N)
    X = torch.randn((B, N, N + 8), dtype=torch.float64, device=dev, generator=g)
    base = torch.empty((B, N + M, N), dtype=torch.float64, device=dev)
    base[:, :N] = X @ X.transpose(1, 2) / N + 0.05 * torch.eye(N, dtype=torch.float64, device=dev)
    base[:, N:] = torch.randn((B, M, N), dtype=torch.float64, device=dev, generator=g)
    del X
    res = {}
    for bk in (4, 3):
        copt = ops.chol_options(big_kernel=bk)
        aug = base.clone()
        H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
        ops.batch_chol_solve(aug, B, N, M, H, options=copt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            aug.copy_(base); torch.cuda.synchronize(); t = time.perf_counter()
            ops.batch_chol_solve(aug, B, N, M, H, options=copt); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res[bk] = (min(ts), H.clone())
    print(f"B={B} N={N} M={M}: 128-tiles {1e3 * res[4][0]:.2f} ms, 64-tiles {1e3 * res[3][0]:.2f} ms; same bits: {torch.equal(res[4][1], res[3][1])}", flush=True)
