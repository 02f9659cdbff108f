#!/usr/bin/env python3
"""cfg2 fit time with a boolean FitOptions field on / off, interleaved on one box.
    python tools/flag_ab.py FIELD [V [world rank]]      e.g. folds_in_one_launch, refit_by_inverse, series_fused_moments
(world > 1: one rank of a simulated sharded job, ShardContext.simulated)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ShardContext, nested_cv as ncv, ops  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402

flag = sys.argv[1]
V_total = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
world, rank = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 0)
dev = ops.device(0)
lo, hi = shard_bounds(V_total, world, rank)
V = hi - lo
dX, dY, p = bench.synth_inputs(V, rank, dev)
alphas = np.logspace(-1, 8, bench.A)
shard = ShardContext.simulated(world, rank, device=dev, global_lists=False) if world > 1 else None
for setting in (True, False, True, False):
    model = NestedCVModel("ridge_regression", shard=shard, options=ncv.FitOptions(**{flag: setting}))
    fit = lambda: model.fit_predict_device(dX, dY, p, V, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)  # noqa: E731
    fit(); fit(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(6):
        fit()
    torch.cuda.synchronize()
    print(f"{flag}={setting}: {1e3 * (time.perf_counter() - t) / 6:.1f} ms per fit", flush=True)
