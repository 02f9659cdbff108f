#!/usr/bin/env python3
"""How the chosen alphas of a voxel vary over the outer folds of a fit (bench workload, cfg2): voxels with the same alpha in
every fold, distinct (alpha_0 .. alpha_4) tuples and how many voxels the commonest ones cover -- what a refit that applies
the MEAN of the folds' operators once per tuple (instead of every fold's operator) would have to work with.
    python tools/alpha_tuple_probe.py [V]
"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
dev = ops.device()
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
m = NestedCVModel("ridge_regression")
out = m.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW) if hasattr(m, "fit_predict_device") else None
fa = np.stack(m.last_fold_alphas)                      # (folds, V)
idx = np.searchsorted(alphas, fa * (1 - 1e-9))
same = (idx == idx[0]).all(axis=0)
print(f"V = {V}: {same.mean() * 100:.1f} % of the voxels take the same alpha in all {idx.shape[0]} folds")
cnt = collections.Counter(map(tuple, idx.T))
order = cnt.most_common()
cum = np.cumsum([c for _, c in order]) / V
print(f"{len(order)} distinct tuples; the commonest 1 / 4 / 16 / 64 cover {cum[0]:.3f} / {cum[min(3, len(cum) - 1)]:.3f} / "
      f"{cum[min(15, len(cum) - 1)]:.3f} / {cum[min(63, len(cum) - 1)]:.3f}")
tiles = sum(-(-c // 256) for _, c in order)
print(f"256-column tiles if every tuple is a column group: {tiles} (ungrouped: {-(-V // 256)}); per-fold alpha histogram:")
for f in range(idx.shape[0]):
    print("   fold", f, np.bincount(idx[f], minlength=len(alphas)).tolist())
