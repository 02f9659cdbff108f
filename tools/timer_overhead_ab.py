#!/usr/bin/env python3
"""What the bench's launch timers cost a host-to-host cfg2 fit: the same fits with the event timers of the two screening
contractions on (as inside bench.py's timed region) and off, interleaved.   python tools/timer_overhead_ab.py [rounds] [resident]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
resident = "resident" in sys.argv[2:]
if not resident:
    X, Y = bench.host_arrays(dX, dY, p, V)
    del dX, dY
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
ts = {True: [], False: []}
for i in range(2 * rounds + 2):
    on = bool(i % 2)
    ops.timing_read()
    ops.timing_enable(on, only=["alpha_sweep_gemm", "series_sweep_gemm"] if on else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = (model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW) if resident
           else model.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW))
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    ops.timing_enable(False)
    ops.timing_read()
    if i >= 2:
        ts[on].append(dt)
    out = None
for on in (True, False):
    v = sorted(ts[on])
    print(f"timers {'on ' if on else 'off'}: median {v[len(v) // 2]:.1f} ms (min {v[0]:.1f}, max {v[-1]:.1f})")
