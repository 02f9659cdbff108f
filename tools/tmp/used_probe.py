import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(80000, 0, dev)
orig = ncv.RidgeCVEngine._refit_groups
def patched(self, best, split):
    out = orig(self, best, split)
    perm, used, tiles, Vs = out
    cnt = np.diff(tiles)
    print("used alphas", used, "tiles per group", cnt.tolist(), flush=True)
    return out
ncv.RidgeCVEngine._refit_groups = patched
m = NestedCVModel("r").fit_predict_device(dX, dY, p, 80000, alphas=np.logspace(-1, 8, 20), **bench.FIT_KW)
