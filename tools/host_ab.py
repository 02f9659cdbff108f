#!/usr/bin/env python3
"""A/B of FitOptions variants on the host-to-host cfg2 fit, interleaved on one box (boxes differ by 2-3 %):
    python tools/host_ab.py "tail_panels_geometric=False" "tail_folds=1" ...     (the defaults always run as variant 0)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from litcoder_core_amd import NestedCVModel, ops
from litcoder_core_amd.nested_cv import FitOptions
dev = ops.device(0)
V = 80000
resident = "--resident" in sys.argv
specs = [a for a in sys.argv[1:] if not a.startswith("--")]
dX, dY, p = bench.synth_inputs(V, 0, dev)
X, Y = bench.host_arrays(dX, dY, p, V)
if not resident:
    del dX, dY
alphas = np.logspace(-1, 8, bench.A)
variants = [("defaults", FitOptions())]
for spec in specs:
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        kw[k] = eval(v)
    variants.append((spec, FitOptions(**kw)))
models = [(name, NestedCVModel("ridge_regression", options=o)) for name, o in variants]
times = {name: [] for name, _ in variants}
for rep in range(7):
    for name, m in models:
        out = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if resident:
            out = m.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
        else:
            out = m.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
        torch.cuda.synchronize()
        if rep >= 2:
            times[name].append(1e3 * (time.perf_counter() - t0))
for name, ts in times.items():
    print(f"{name:50s} median {np.median(ts):7.2f} ms   min {min(ts):7.2f}   all {[round(t, 1) for t in ts]}")
