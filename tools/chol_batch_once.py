#!/usr/bin/env python3
"""One batched Cholesky solve at the inner-fold shape of cfg2 (B = 20, N = 1920, M = 480), twice (warm-up + one), for
counter collection:  rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/chol_batch_once.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402

dev = ops.device(0)
B, N, M = 20, 1920, 480
aug = torch.randn((B, N + M, N), dtype=torch.float64, device=dev)
aug[:, :N] = torch.eye(N, dtype=torch.float64, device=dev) * (4.0 * N) + 1.0
base = aug.clone()
H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
for _ in range(2):
    aug.copy_(base)
    ops.batch_chol_solve(aug, B, N, M, H)
torch.cuda.synchronize()
