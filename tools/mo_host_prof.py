import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops
from litcoder_core_amd.engine import mean_refit as mr
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(80000, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
import cProfile, pstats
orig = ncv.RidgeCVEngine._mean_operator_weights
prof = cProfile.Profile()
def wrapped(self, rg):
    prof.enable()
    try:
        return orig(self, rg)
    finally:
        prof.disable()
ncv.RidgeCVEngine._mean_operator_weights = wrapped
for i in range(3):
    if i == 2:
        prof.clear()
    model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
pstats.Stats(prof).sort_stats("cumulative").print_stats(25)
