import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from litcoder_core_amd import _lib, ops
dev = ops.device(0)
for (B, N, M) in ((1, 1920, 480), (3, 1920, 480), (10, 1920, 480), (20, 1920, 480), (80, 1920, 480), (1, 2432, 1920), (4, 2432, 3680), (4, 2432, 2432)):
    g = torch.Generator(device=dev); g.manual_seed(B + N)
    X = torch.randn((B, N, N + 8), dtype=torch.float64, device=dev, generator=g)
    base = torch.empty((B, N + M, N), dtype=torch.float64, device=dev)
    base[:, :N] = X @ X.transpose(1, 2) / N + 0.05 * torch.eye(N, dtype=torch.float64, device=dev)
    base[:, N:] = torch.randn((B, M, N), dtype=torch.float64, device=dev, generator=g)
    del X
    res = []
    for ob in (128, 256, 384, 512, 768):
        copt = ops.chol_options(outer_block=ob)
        aug = base.clone()
        H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
        ops.batch_chol_solve(aug, B, N, M, H, options=copt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            aug.copy_(base); torch.cuda.synchronize(); t = time.perf_counter()
            ops.batch_chol_solve(aug, B, N, M, H, options=copt); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res.append(f"ob{ob} {1e3*min(ts):.2f}")
    print(f"B={B} N={N} M={M}: " + "  ".join(res), flush=True)
