#!/usr/bin/env python3
"""Where one more alpha in use costs a resident cfg2 fit ~8 ms: the bench fit against the same fit in which ONE voxel is
pure noise (it takes the grid's largest alphas, which nobody else does) -- the library's per-class event timers, the
mean-operator refit's counters and the alphas in use.   python tools/noise_voxel_profile.py"""
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
dYn = dY.clone()
dYn[:, 40123] = torch.randn(dY.shape[0], device=dY.device, generator=torch.Generator(device=dY.device).manual_seed(3))
alphas = np.logspace(-1, 8, bench.A)
res = {}
for name, Y in (("clean", dY), ("one noise voxel", dYn)):
    model = NestedCVModel("ridge_regression")
    for _ in range(3):
        model.fit_predict_device(dX, Y, p, V, alphas=alphas, **bench.FIT_KW)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.fit_predict_device(dX, Y, p, V, alphas=alphas, **bench.FIT_KW)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    ops.timing_read()
    ops.timing_enable(True)
    out = model.fit_predict_device(dX, Y, p, V, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    kern = ops.timing_read()
    ops.timing_enable(False)
    res[name] = kern
    lf = model.last_fit
    print(f"{name}: median {np.median(ts):.1f} ms; alphas in use {sorted(set(np.round(np.log10(np.asarray(out[2])), 2).tolist()))[:8]}...; "
          f"mean-operator refit {lf.get('mean_operator')}; used_all {lf.get('used_all')}")
names = sorted(set(res["clean"]) | set(res["one noise voxel"]))
print(f"{'class':32s} {'clean ms':>9s} {'launches':>8s} {'noise ms':>9s} {'launches':>8s} {'diff':>7s}")
for n in names:
    a, b = res["clean"].get(n, (0.0, 0)), res["one noise voxel"].get(n, (0.0, 0))
    if abs(b[0] - a[0]) >= 0.15 or b[1] != a[1]:
        print(f"{n:32s} {a[0]:9.2f} {a[1]:8d} {b[0]:9.2f} {b[1]:8d} {b[0] - a[0]:7.2f}")
