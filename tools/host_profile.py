#!/usr/bin/env python3
"""cProfile of the Python side of ONE warm host-to-host cfg2 fit (where the interpreter spends the call's time).
    python tools/host_profile.py [n_lines]"""
import cProfile, io, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from litcoder_core_amd import NestedCVModel, ops
dev = ops.device(0)
V = 80000
dX, dY, p = bench.synth_inputs(V, 0, dev)
X, Y = bench.host_arrays(dX, dY, p, V)
del dX, dY
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
for _ in range(3):
    out = model.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW); out = None
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
out = model.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 35)
print(s.getvalue())
