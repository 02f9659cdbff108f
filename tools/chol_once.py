#!/usr/bin/env python3
"""One batched Cholesky solve (or inverse) at a fit's batch shape, for kernel traces:
   python tools/chol_once.py B N M [inverse] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops
B, N, M = (int(x) for x in sys.argv[1:4])
inverse = len(sys.argv) > 4 and sys.argv[4] == "inverse"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = ops.device(0)
g = torch.Generator(device=dev); g.manual_seed(B + N)
X = torch.randn((B, N, N + 8), dtype=torch.float64, device=dev, generator=g)
base = torch.empty((B, N + M, N), dtype=torch.float64, device=dev)
base[:, :N] = X @ X.transpose(1, 2) / N + 0.05 * torch.eye(N, dtype=torch.float64, device=dev)
if inverse:
    base[:, N:] = torch.eye(N, dtype=torch.float64, device=dev)
else:
    base[:, N:] = torch.randn((B, M, N), dtype=torch.float64, device=dev, generator=g)
del X
H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for r in range(reps):
    aug = base.clone()
    torch.cuda.synchronize()
    ev[0].record()
    if inverse:
        ops.batch_chol_inverse(aug, B, N, H)
    else:
        ops.batch_chol_solve(aug, B, N, M, H)
    ev[1].record()
    torch.cuda.synchronize()
    fl = B * (N ** 3 if inverse else (N ** 3 / 3 + 2.0 * N * N * M))
    ms = ev[0].elapsed_time(ev[1])
    print(f"B={B} N={N} M={M} inverse={inverse}: {ms:.2f} ms  {fl / ms / 1e9:.1f} TF", flush=True)
