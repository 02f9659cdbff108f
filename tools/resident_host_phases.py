#!/usr/bin/env python3
"""Host wall time of the engine phases of a RESIDENT cfg2 fit (when the Python driver has issued what): the start of a fit is
bound by the host issuing ~400 launches before the main stream has its first V-wide kernel.   python tools/resident_host_phases.py [fits]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops  # noqa: E402

fits = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(80000, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
log = []


def wrap(name):
    fn = getattr(ncv.RidgeCVEngine, name)

    def inner(self, *a, **k):
        t = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            log.append((name, t, time.perf_counter()))
    setattr(ncv.RidgeCVEngine, name, inner)


for n in ("__init__", "begin_fit", "precompute_lmax", "prepare_folds", "fold_speculate", "refit_ahead", "fold_begin", "fold_sweeps_finish",
          "fold_choose", "fold_select", "fold_finish", "fold_collect", "_hat_matrices", "_mean_operator_weights", "combined_significance_begin"):
    wrap(n)
model = NestedCVModel("ridge_regression")
for i in range(fits):
    del log[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"last fit: {(t1 - t0) * 1e3:.1f} ms")
for name, a, b in sorted(log, key=lambda r: r[1]):
    print(f"  {(a - t0) * 1e3:8.2f} -> {(b - t0) * 1e3:8.2f}  ({(b - a) * 1e3:6.2f} ms)  {name}")
