#!/usr/bin/env python3
"""Diagnostic: cfg4 fixture volume, full fit vs a 25 000-voxel block fitted alone -- where do they differ?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import litcoder_core_amd as lc
from litcoder_core_amd import ops
from litcoder_core_amd.dist import shard_bounds
import test_gpu_configs as tg
V = 200000
for variant in ("fixture", "plain"):
    X, Y, kw, dX, dY, p = tg._fixture_volume(lc, "cfg4", V, seed=4)
    if variant == "plain":
        dY[:, :256] = dY[:, 256:512]
    T = len(X)
    model = lc.NestedCVModel("r")
    m, W, a = model.fit_predict_device(dX, dY, p, V, weights_on_host=False, **kw)
    fa = [np.asarray(x) for x in model.last_fold_alphas]
    print(variant, "full fit:", model.last_fit.get("precision"), "used_all", model.last_fit.get("used_all"))
    lo, hi = shard_bounds(V, 8, 5)
    blk = torch.zeros((T, 25088), dtype=torch.float32, device=dY.device)
    blk[:, : hi - lo] = dY[:, lo:hi]
    mb = lc.NestedCVModel("r")
    m_b, W_b, a_b = mb.fit_predict_device(dX, blk, p, hi - lo, weights_on_host=False, **kw)
    fb = [np.asarray(x) for x in mb.last_fold_alphas]
    print(variant, "block fit:", mb.last_fit.get("precision"), "used_all", mb.last_fit.get("used_all"))
    r, rb = np.asarray(m["correlations"])[lo:hi], np.asarray(m_b["correlations"])
    print("  r differ:", int((r != rb).sum()), "max", float(np.abs(r - rb).max()), "| mean alphas differ:", int((a[lo:hi] != a_b).sum()))
    for f in range(len(fa)):
        d = fa[f][lo:hi] != fb[f]
        print(f"  fold {f}: alphas differ {int(d.sum())}; distinct alphas full {len(np.unique(fa[f]))}, block {len(np.unique(fb[f]))}")
    dW = (W[:, lo:hi] != W_b)
    print("  W columns differing:", int(dW.any(0).sum()), "of", hi - lo)
