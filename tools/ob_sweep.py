#!/usr/bin/env python3
"""cfg2 fit time against the outer-block width of the batched Cholesky (FitOptions.chol_outer_block).  python tools/ob_sweep.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from litcoder_core_amd import NestedCVModel, ops
from litcoder_core_amd.nested_cv import FitOptions
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(80000, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
for ob in [int(a) for a in sys.argv[1:]] or [512, 256, 128, 1024, 512]:
    model = NestedCVModel("ridge_regression", options=FitOptions(chol_outer_block=ob))
    fit = lambda: model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
    fit(); fit(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(6):
        fit()
    torch.cuda.synchronize()
    print(f"outer block {ob}: {1e3 * (time.perf_counter() - t) / 6:.1f} ms per fit", flush=True)
