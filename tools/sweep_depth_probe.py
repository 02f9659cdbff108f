#!/usr/bin/env python3
"""Where the PRODUCT sweep kernel's time goes, without touching it: the fused score sweep (lc_alpha_sweep_scores_f16x3_folds,
4 alphas x 480 validation rows, 80 000 voxels) timed at several contraction depths N.  The slope of time over depth is the
main loop's cost per ring step, the intercept is prologue + epilogue (+ launch); both forms (screening: terms=101, one MFMA
per product; terms=3).  A/M variants: 4 alphas = 7.5 M-tiles, 8 alphas = 15.
    python tools/sweep_depth_probe.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from litcoder_core_amd import ops  # noqa: E402
from litcoder_core_amd._lib import LC_MB, LC_NB, LC_SCORE_CORR  # noqa: E402

dev = ops.device()
V, n_v = int(os.environ.get("PROBE_V", 80000)), 480
g = torch.Generator(device=dev); g.manual_seed(0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def setup(A, N):
    T = N + n_v
    M = ops.pad_to(n_v, LC_MB)
    H = torch.randn((A * M, N), generator=g, device=dev, dtype=torch.float32) * 0.02
    Y = torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    tr = ops.idx_tensor(np.r_[0:N], N, dev)
    va = ops.idx_tensor(np.r_[N:T], M, dev)
    ystat = torch.empty((3, V), dtype=torch.float32, device=dev)
    yblk = torch.empty((M // LC_MB, V), dtype=torch.float32, device=dev)
    part = torch.empty((A * M // LC_MB, 4, V), dtype=torch.float32, device=dev)
    scores = torch.empty((A, V), dtype=torch.float32, device=dev)
    yv = torch.empty((M, V), dtype=torch.float32, device=dev)
    ops.val_stats(Y, V, va, M, n_v, ystat, yblk, yv)
    rows_pad = ops.pad_to(A * M, 256)
    Ht = torch.empty(rows_pad * N * 2, dtype=torch.float16, device=dev)
    rs_inv = torch.empty(rows_pad, dtype=torch.float32, device=dev)
    Yt = torch.empty(ops.pad_to(V, 256) * N * 2, dtype=torch.float16, device=dev)
    cs, _flag = ops.col_scales_f16(Y, T, V)
    ops.split_rows_f16_alphas(H, 1, A, M, N, Ht, rs_inv)
    ops.split_cols_f16(Y, V, tr, N, cs, Yt)
    return lambda terms: ops.alpha_sweep_scores_f16x3(Ht, rs_inv, A, M, N, Yt, cs[V:], yv, V, n_v, ystat, yblk, LC_SCORE_CORR,
                                                      part, scores, False, terms=terms)


for A in (4, 8):
    tiles = -(-A * 480 // 256) * -(-V // 256)
    rounds = tiles / 256.0
    res = {}
    for N in (640, 1280, 1920, 2560):
        fn = setup(A, N)
        for terms in (101, 3):
            res[(terms, N)] = timeit(lambda: fn(terms))
    for terms in (101, 3):
        ns = sorted(n for t, n in res if t == terms)
        x = np.asarray(ns, dtype=np.float64)
        y = np.asarray([res[(terms, n)] for n in ns])
        slope, icpt = np.polyfit(x, y, 1)
        step_k = 32 if terms == 101 else 16
        per_step_us = slope * step_k * 1e3 / rounds
        print(f"A = {A} ({tiles} tiles = {rounds:.2f} rounds), terms = {terms}: " +
              ", ".join(f"N {n}: {res[(terms, n)]:.3f} ms" for n in ns) +
              f" | slope {slope * 1e3:.4f} us per unit of depth -> {per_step_us * 1e3:.0f} ns per ring step and tile "
              f"(MFMA floor {1024 if terms == 101 else 1536} cycles), intercept {icpt:.3f} ms = {icpt / rounds * 1e3:.1f} us per tile "
              f"outside the main loop", flush=True)
