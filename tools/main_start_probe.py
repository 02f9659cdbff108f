#!/usr/bin/env python3
"""HIP events on the main stream around the first V-wide launches of a resident cfg2 fit: when, on the device, the index
upload, the operand split and the first sweep of step 0 ran (and when the host queued them).   python tools/main_start_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

dev = ops.device(0)
dX, dY, p = bench.synth_inputs(80000, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
for _ in range(3):
    model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
torch.cuda.synchronize()
marks = []
main = torch.cuda.current_stream()
t0 = [0.0]


def wrap(name, limit):
    fn = getattr(ops, name)
    n = [0]

    def inner(*a, **k):
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != main.cuda_stream or n[0] >= limit:
            return fn(*a, **k)
        n[0] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h = time.perf_counter() - t0[0]
        out = fn(*a, **k)
        e1.record()
        marks.append((name, h, e0, e1))
        return out
    setattr(ops, name, inner)


for nm, lim in (("idx_tensor", 6), ("upload", 6), ("split_cols_f16", 3), ("val_stats_folds", 3), ("series_sweep_scores_f16x3", 3),
                ("alpha_sweep_scores_f16x3", 2), ("col_scales_f16", 2), ("gram", 2), ("zeros", 4)):
    if hasattr(ops, nm):
        wrap(nm, lim)
start = torch.cuda.Event(enable_timing=True)
start.record()
t0[0] = time.perf_counter()
model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
torch.cuda.synchronize()
for name, h, e0, e1 in sorted(marks, key=lambda m: m[1]):
    print(f"host {h * 1e3:6.2f} ms   device {start.elapsed_time(e0):7.2f} -> {start.elapsed_time(e1):7.2f} ms   {name}")
