#!/usr/bin/env python3
"""cfg2 fit time with a library debug switch on / off, interleaved.   python tools/fit_ab.py <lc_debug_function> [V]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from litcoder_core_amd import NestedCVModel, _lib, ops
fn = getattr(_lib.load(), sys.argv[1])
V = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
fit = lambda: model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
for setting in (1, 0, 1, 0):
    fn(setting)
    fit(); fit(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(6):
        fit()
    torch.cuda.synchronize()
    print(f"{sys.argv[1]}({setting}): {1e3 * (time.perf_counter() - t) / 6:.1f} ms per fit", flush=True)
fn(1)
