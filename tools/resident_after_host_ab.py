#!/usr/bin/env python3
"""Resident cfg2 fits by a model that has just done host-to-host fits (bench.py's resident_path leg) against a fresh model's:
does anything the host-to-host fits leave behind (the model's plan, the allocator's state) slow the resident ones?
    python tools/resident_after_host_ab.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
X, Y = bench.host_arrays(dX, dY, p, V)
alphas = np.logspace(-1, 8, bench.A)


def resident(model, n, sync_between):
    ts = []
    out = None
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for _ in range(n):
        out = None
        t0 = time.perf_counter()
        out = model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
        if sync_between:
            torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t_all) / n, ts


m0 = NestedCVModel("ridge_regression")
resident(m0, 2, True)
avg, ts = resident(m0, 4, True)
print(f"before any host-to-host fit in this process: {avg:.1f} ms per fit; host view {[round(t, 1) for t in ts]}")
m1 = NestedCVModel("ridge_regression")
for _ in range(3):
    o = m1.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
    o = None
torch.cuda.synchronize()
for label, model in (("the model that did host-to-host fits", m1), ("a fresh model", NestedCVModel("ridge_regression"))):
    resident(model, 1, True)
    for sync in (False, True):
        avg, ts = resident(model, 4, sync)
        print(f"{label}, {'a device sync after every fit' if sync else 'fits back to back'}: {avg:.1f} ms per fit; host view {[round(t, 1) for t in ts]}")
