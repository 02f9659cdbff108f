#!/usr/bin/env python3
"""When do the first fold's events fire on the device?  Timing events at: fit start (main), end of Lanczos / data_ready
position (aux), _sweeps entry (main, before any wait), first V-wide kernel (main), end of prepare (aux).
    python tools/startup_probe.py [host]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops
dev = ops.device(0)
V = 80000
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
marks = []
def ev(label):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((label, e, time.perf_counter()))
def wrap(obj, name, before=None, after=None):
    fn = getattr(obj, name)
    def inner(*a, **k):
        if before: ev(before)
        out = fn(*a, **k)
        if after: ev(after)
        return out
    setattr(obj, name, inner)
wrap(ncv.RidgeCVEngine, "_fold_design", after="aux: data_ready position")
wrap(ncv.RidgeCVEngine, "_hat_matrices", before="aux: hat matrices begin", after="aux: hat matrices end")
wrap(ncv.RidgeCVEngine, "_sweeps", before="main: _sweeps entry", after="main: _sweeps queued end")
wrap(ops, "split_cols_f16", before="main: before split_cols", after="main: after split_cols")
wrap(ops, "val_stats_folds", before="main: before val_stats (after series_ready wait)")
wrap(ncv.RidgeCVEngine, "fold_speculate", before="cur: fold_speculate entry")
wrap(ncv.RidgeCVEngine, "_refit_rhs", before="aux2: refit_rhs begin")
def run():
    return model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
run(); torch.cuda.synchronize(); marks.clear()
base = torch.cuda.Event(enable_timing=True); base.record(); t0 = time.perf_counter()
run(); torch.cuda.synchronize()
for label, e, th in marks[:40]:
    print(f"dev {base.elapsed_time(e):8.2f} ms   host {1e3 * (th - t0):8.2f} ms   {label}")
