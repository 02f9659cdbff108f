#!/usr/bin/env python3
"""The mean-operator refit where it cannot pay: cfg2's shape with weak / mixed signal (every fold another alpha for most
voxels), option on against off, interleaved resident fits.     python tools/weak_signal_ab.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402
from litcoder_core_amd.engine.common import FitOptions  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
alphas = np.logspace(-1, 8, bench.A)
cases = {}
for name, wscale, frac_noise in (("weak signal (W x 0.1)", 0.002, 0.0), ("half the voxels pure noise", 0.02, 0.5)):
    Y = torch.zeros_like(dY)
    W = wscale * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
    W[:, torch.rand(V, generator=g, device=dev) < frac_noise] = 0.0
    Y[:, :V] = dX[:, :p] @ W + torch.randn((dY.shape[0], V), generator=g, device=dev, dtype=torch.float32)
    cases[name] = Y
    del W
for name, Y in cases.items():
    models = {"mean operator on": NestedCVModel("ridge_regression"),
              "off": NestedCVModel("ridge_regression", options=FitOptions(mean_operator_refit=False))}
    times = {k: [] for k in models}
    for k, m in models.items():
        m.fit_predict_device(dX, Y, p, V, alphas=alphas, **bench.FIT_KW)
    for _ in range(rounds):
        for k, m in models.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = m.fit_predict_device(dX, Y, p, V, alphas=alphas, **bench.FIT_KW)
            torch.cuda.synchronize()
            times[k].append(1e3 * (time.perf_counter() - t0))
    mo = models["mean operator on"].last_fit.get("mean_operator")
    print(f"{name}: on {np.median(times['mean operator on']):.1f} ms, off {np.median(times['off']):.1f} ms; median score "
          f"{out[0]['median_score']:.3f}; screening undecided {models['off'].last_fit.get('undecided')} of {models['off'].last_fit.get('screened')}; {mo}", flush=True)
