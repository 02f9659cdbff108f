#!/usr/bin/env python3
"""cfg2 fits in one process, host-to-host first then resident or the other way round: median fit time of each kind.
    python tools/host_then_resident.py host|resident"""
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
model = NestedCVModel("ridge_regression")


def loop(kind, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = (model.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW) if kind == "host"
               else model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW))
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
        out = None
    v = sorted(ts[2:])
    return v[len(v) // 2]


first = sys.argv[1] if len(sys.argv) > 1 else "host"
for kind in ((first, "resident" if first == "host" else "host")):
    print(f"{kind} {loop(kind):.1f} ms;")
