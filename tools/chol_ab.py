#!/usr/bin/env python3
"""Batched Cholesky solve at the fit's batch shapes: fused left-looking steps (k_lstep / k_bstep) against the first
version's three launches per step; results must agree to rounding.   python tools/chol_ab.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import _lib, ops
dev = ops.device(0)
lib = _lib.load()
for (B, N, M, label) in ((3, 1920, 480, "inner, one rank of 8"), (20, 1920, 480, "inner folds, fold 0"), (80, 1920, 480, "inner folds 1-4"),
                         (4, 2432, 3680, "refit"), (1, 2432, 1920, "refit job, one rank of 8"), (100, 64, 1440, "primal p<=64")):
    g = torch.Generator(device=dev); g.manual_seed(B + N)
    X = torch.randn((B, N, N + 8), dtype=torch.float64, device=dev, generator=g)
    base = torch.empty((B, N + M, N), dtype=torch.float64, device=dev)
    base[:, :N] = X @ X.transpose(1, 2) / N + 0.05 * torch.eye(N, dtype=torch.float64, device=dev)
    base[:, N:] = torch.randn((B, M, N), dtype=torch.float64, device=dev, generator=g)
    del X
    fl = B * (N ** 3 / 3 + 2.0 * N * N * M)
    out = {}
    for fused in (0, 1, 2):                     # 0 = round 1, 1 = fused steps + right-looking deep updates, 2 = + left-looking deep
        copt = ops.chol_options(fused_steps=bool(fused), left_deep=fused == 2)
        aug = base.clone()
        H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
        ops.batch_chol_solve(aug, B, N, M, H, options=copt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            aug.copy_(base)
            torch.cuda.synchronize()
            t = time.perf_counter()
            info = ops.batch_chol_solve(aug, B, N, M, H, options=copt)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        out[fused] = (min(ts), H.clone(), int(info.abs().max()))
    ref = torch.linalg.solve(base[:1, :N].transpose(1, 2), base[:1, N:].transpose(1, 2)).transpose(1, 2)   # H A = G, A symmetric
    err_new = float((out[2][1][:1].double() - ref).abs().max() / ref.abs().max())
    diff = float((out[0][1].double() - out[2][1].double()).abs().max() / out[0][1].double().abs().max())
    print(f"{label}: B={B} N={N} M={M}: round 1 {1e3 * out[0][0]:.2f} ms ({fl / out[0][0] / 1e12:.1f} TF), fused steps "
          f"{1e3 * out[1][0]:.2f} ms ({fl / out[1][0] / 1e12:.1f} TF), + left-looking deep updates {1e3 * out[2][0]:.2f} ms "
          f"({fl / out[2][0] / 1e12:.1f} TF); new vs fp64 solve {err_new:.1e}, new vs round 1 {diff:.1e}, "
          f"info {out[0][2]}/{out[2][2]}", flush=True)
