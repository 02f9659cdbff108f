#!/usr/bin/env python3
"""Do the fp64 Cholesky chains (auxiliary stream) and the MFMA sweeps (main stream) actually overlap?
Times 12 sweep launches alone, 2 inner-fold Cholesky batches alone, and both at once on two streams
(default and high priority).   python tools/overlap_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402
from litcoder_core_amd._lib import LC_MB, LC_SCORE_CORR  # noqa: E402

dev = ops.device(0)
g = torch.Generator(device=dev); g.manual_seed(0)
A, n_v, n_i, V, T = 20, 480, 1920, 80000, 3000
M, N = n_v, n_i
H = torch.randn((A * M, N), generator=g, device=dev, dtype=torch.float32) * 0.02
Y = torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
tr = ops.idx_tensor(np.r_[0:1920], N, dev)
va = ops.idx_tensor(np.r_[1920:2400], M, dev)
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
cs, _ = ops.col_scales_f16(Y, T, V)
ops.split_rows_f16_alphas(H, 1, A, M, N, Ht, rs_inv)
ops.split_cols_f16(Y, V, tr, N, cs, Yt)

B_, N_, M_ = 20, 1920, 480
base = torch.randn((B_, N_ + M_, N_), dtype=torch.float64, device=dev)
base[:, :N_] = torch.eye(N_, dtype=torch.float64, device=dev) * (4.0 * N_) + 1.0
aug = base.clone()
H2 = torch.empty((B_, M_, N_), dtype=torch.float32, device=dev)


def sweeps(n=12):
    for _ in range(n):
        ops.alpha_sweep_scores_f16x3(Ht, rs_inv, A, M, N, Yt, cs[V:], yv, V, n_v, ystat, yblk, LC_SCORE_CORR, part, scores,
                                     False)


def chols(n=6):
    for _ in range(n):
        aug.copy_(base)
        ops.batch_chol_solve(aug, B_, N_, M_, H2)


def wall(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t)


sweeps(2); chols(1)
t_s = wall(sweeps)
t_c = wall(chols)
print(f"12 sweeps alone {t_s:.1f} ms; 6 Cholesky batches (B=20, N=1920, M=480) alone {t_c:.1f} ms; sum {t_s + t_c:.1f} ms")
# two independent Cholesky chains on two streams (do the serial diagonal steps of one hide behind the other?)
aug_b = base.clone()
H2b = torch.empty_like(H2)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def chols_two_streams(n=6):
    for i in range(n):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            tgt, out = (aug, H2) if i % 2 == 0 else (aug_b, H2b)
            tgt.copy_(base)
            ops.batch_chol_solve(tgt, B_, N_, M_, out)


chols_two_streams(2)
print(f"6 Cholesky batches alternating between two streams: {wall(chols_two_streams):.1f} ms")
for prio in (0, -1):
    aux = torch.cuda.Stream(device=dev, priority=prio)

    def both():
        with torch.cuda.stream(aux):
            chols()
        sweeps()
    both()
    print(f"both at once, aux priority {prio}: {wall(both):.1f} ms")


# main stream restricted to a subset of the CUs (lc_stream_create_cu_mask): do the short fp64 kernels of the
# auxiliary stream then run beside the sweeps instead of waiting for sweep workgroups to retire?
masks = {
    "all but CUs 224-255": [0xFFFFFFFF] * 7 + [0],
    "all but every 8th CU": [0xFEFEFEFE] * 8,
    "all but every 16th CU": [0xFFFEFFFE] * 8,
    "all but CUs 0-15": [0xFFFF0000] + [0xFFFFFFFF] * 7,
}
aux = torch.cuda.Stream(device=dev)
for name, words in masks.items():
    try:
        ms = ops.masked_stream(words)
    except Exception as e:   # noqa: BLE001
        print(f"masked stream ({name}): {e}")
        continue

    def masked_sweeps():
        with torch.cuda.stream(ms):
            sweeps()

    def both_masked():
        with torch.cuda.stream(aux):
            chols()
        with torch.cuda.stream(ms):
            sweeps()
    masked_sweeps()
    t_m = wall(masked_sweeps)
    both_masked()
    print(f"main stream on {name}: 12 sweeps alone {t_m:.1f} ms; with the Cholesky batches on an unrestricted stream "
          f"{wall(both_masked):.1f} ms")
