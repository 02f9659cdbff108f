#!/usr/bin/env python3
"""Host-side timeline of one cfg2 fit: wall time the Python thread spends in each engine phase (enqueue cost and
sync waits, no added synchronisation).  python tools/host_timeline.py [V]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops, stats  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
log = []


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((label or name, t, time.perf_counter()))
    setattr(obj, name, inner)


for n in ("fold_prepare", "fold_begin", "fold_select", "fold_finish", "fold_collect", "precompute_lmax", "weights"):
    wrap(ncv.RidgeCVEngine, n)
for n in ("choose", "_refit_groups", "_refit_systems", "_sweeps", "_hat_matrices", "_refit_apply", "_fold_data",
          "_series_by_moments", "_shared_image", "begin_fit", "fold_choose", "fold_speculate"):
    wrap(ncv.RidgeCVEngine, n, "    . " + n)
for n in ("fdrcorrection", "fisher_combine", "full_cv_metrics"):
    wrap(stats, n)
wrap(ncv, "_fold_lists")


def run():
    return model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)


run()
torch.cuda.synchronize()
log.clear()
t0 = time.perf_counter()
run()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"fit wall {1e3 * (t1 - t0):.1f} ms")
for name, a, b in log:
    print(f"  {1e3 * (a - t0):8.2f} -> {1e3 * (b - t0):8.2f}  ({1e3 * (b - a):7.2f} ms)  {name}")
tot = {}
for name, a, b in log:
    tot[name] = tot.get(name, 0.0) + (b - a)
print("totals:", {k: round(1e3 * v, 1) for k, v in tot.items()})
