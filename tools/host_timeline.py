#!/usr/bin/env python3
"""Host-side timeline of one cfg2 fit: wall time the Python thread spends in each engine phase (enqueue cost and
sync waits, no added synchronisation).  python tools/host_timeline.py [V_total [world rank]]  -- with world > 1 one
rank of a simulated world-rank job (ShardContext.simulated: collectives are local copies) on its block of V_total."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops, stats  # noqa: E402

from litcoder_core_amd import ShardContext  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402
V_total = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
world, rank = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 0)
dev = ops.device(0)
lo, hi = shard_bounds(V_total, world, rank)
V = hi - lo
dX, dY, p = bench.synth_inputs(V, rank, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression", shard=ShardContext.simulated(world, rank, device=dev) if world > 1 else None)
log = []


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((label or name, t, time.perf_counter()))
    setattr(obj, name, inner)


for n in ("prepare_folds", "fold_begin", "fold_select", "fold_finish", "fold_collect", "precompute_lmax", "weights"):
    wrap(ncv.RidgeCVEngine, n)
for n in ("choose", "_refit_groups", "_refit_systems", "_sweeps", "_hat_matrices", "_refit_apply", "_fold_targets",
          "_shared_image", "begin_fit", "fold_choose", "fold_speculate", "_sharded_solve", "_refit_chol",
          "combined_significance", "refit_ahead", "_refit_rhs"):
    wrap(ncv.RidgeCVEngine, n, "    . " + n)
for n in ("fdrcorrection", "fisher_combine", "full_cv_metrics"):
    wrap(stats, n)
from litcoder_core_amd import dist as _dist  # noqa: E402
wrap(_dist.ShardContext, "all_gather", "        . all_gather")
for n in ("batch_chol_solve", "batch_assemble_sel", "penalties", "transpose_rows"):
    wrap(ops, n, "        . ops." + n)
wrap(ncv, "_fold_lists")


def run():
    return model.fit_predict_device(dX, dY, p, V, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)


run()
torch.cuda.synchronize()
log.clear()
t0 = time.perf_counter()
run()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"fit wall {1e3 * (t1 - t0):.1f} ms")
for name, a, b in log:
    print(f"  {1e3 * (a - t0):8.2f} -> {1e3 * (b - t0):8.2f}  ({1e3 * (b - a):7.2f} ms)  {name}")
tot = {}
for name, a, b in log:
    tot[name] = tot.get(name, 0.0) + (b - a)
print("totals:", {k: round(1e3 * v, 1) for k, v in tot.items()})
