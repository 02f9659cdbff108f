"""Determinism stress: the same cfg2-size fit (resident inputs) repeated in this process; every repetition must equal
the first bit for bit -- per-fold r / p / alpha index of every voxel and the weights.  Any mismatch is printed with the
fold, the quantity and the voxel columns (a race between streams shows up as a run-to-run difference).

    python tools/race_hunt.py [reps] [voxels] [host | spike | cfg3]
spike: one outlier-dominated target column (the f32 side panel's stream beside the fit, round 5); cfg3: the story pipeline
host to host (StoryPipeline.fit_words: staging threads with z-scoring, voxel panels, the single-alpha guess).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402
import litcoder_core_amd.nested_cv as ncv  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
V = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
mode = sys.argv[3] if len(sys.argv) > 3 else "resident"
host_mode = mode == "host"
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
if mode == "spike":
    dY[7, V // 2 + 123] = 1e6
if mode == "cfg3":
    from litcoder_core_amd import StoryPipeline
    words, wtimes, trtimes, brain = bench.synth_stories(V, dev)
alphas = np.logspace(-1, 8, bench.A)
captured = []
real = ncv.RidgeCVEngine.fold_collect


def spy(self, pend):
    f = real(self, pend)
    captured.append((f.r.copy(), f.p.copy(), f.best_idx.copy()))
    return f


ncv.RidgeCVEngine.fold_collect = spy
host = bench.host_arrays(dX, dY, p, V) if host_mode else None
ref = None
bad = 0
for it in range(reps):
    captured.clear()
    model = NestedCVModel("r")
    if mode == "cfg3":
        m, W, a = StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM, model=model).fit_words(words, wtimes, trtimes, brain, **bench.CFG3_KW)
    elif host_mode:
        m, W, a = model.fit_predict(host[0], host[1], alphas=alphas, **bench.FIT_KW)
    else:
        m, W, a = model.fit_predict_device(dX, dY, p, V, weights_on_host=True, alphas=alphas, **bench.FIT_KW)
    cur = dict(folds=[tuple(x) for x in captured], W=np.array(W, copy=True), a=np.array(a, copy=True),
               c=np.asarray(m["correlations"]).copy())
    if ref is None:
        ref = cur
        continue
    for f, (x, y) in enumerate(zip(cur["folds"], ref["folds"])):
        for name, u, v in zip(("r", "p", "idx"), x, y):
            d = np.nonzero(~((u == v) | ((u != u) & (v != v))))[0]
            if d.size:
                bad += 1
                print(f"rep {it} fold {f} {name}: {d.size} voxels differ, first {d[:8]}, last {d[-3:]}, "
                      f"max |d| {np.nanmax(np.abs(u[d] - v[d])):.3e}", flush=True)
    dW = np.nonzero((cur["W"] != ref["W"]).any(axis=0))[0]
    if dW.size:
        bad += 1
        print(f"rep {it} W: {dW.size} columns differ, first {dW[:8]}, last {dW[-3:]}", flush=True)
print(f"{reps} repetitions, V={V}, {mode}: {bad} mismatching quantities" + (f"; side panel columns "
      f"{model.last_fit.get('side_panel_cols')}" if mode == "spike" else ""), flush=True)
