#!/usr/bin/env python3
"""n device-resident cfg2 fits in a row (fp32 inputs in HBM, weights left there): the workload under a profiler.
    python tools/resident_fit_loop.py [n] [V]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
V = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    out = None
