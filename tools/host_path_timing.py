#!/usr/bin/env python3
"""PCIe-inclusive time of the reference-compatible entry point at cfg2: host float64 arrays in, host weights out
(never the bench `value`).   python tools/host_path_timing.py [V]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
X = dX[:, :p].double().cpu().numpy()
Y = dY[:, :V].double().cpu().numpy()
del dX, dY
torch.cuda.empty_cache()
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
kw = dict(bench.FIT_KW, alphas=alphas)
model.fit_predict(X, Y, **kw)                     # warm-up (allocator, first launches)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    metrics, W, a = model.fit_predict(X, Y, **kw)
    t1 = time.perf_counter()
    print(f"fit_predict(host f64 X {X.shape}, Y {Y.shape}) -> host weights {W.shape} {W.dtype}: {1e3 * (t1 - t0):.0f} ms "
          f"= {V / (t1 - t0):.0f} voxels/s, median score {metrics['median_score']:.5f}")
