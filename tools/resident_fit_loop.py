#!/usr/bin/env python3
"""n device-resident cfg2 fits in a row (fp32 inputs in HBM, weights left there): the workload under a profiler.
    python tools/resident_fit_loop.py [n] [V] [G r]      (G r: as simulated rank r of G -- its V / G voxels, its share of the systems)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ShardContext, ops  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
V = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
G, rank = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 0)
dev = ops.device(0)
V_total = V
if G > 1:
    lo, hi = shard_bounds(V_total, G, rank)
    V = hi - lo
dX, dY, p = bench.synth_inputs(V, rank, dev)
if os.environ.get("FIT_SPIKE"):                          # one outlier-dominated column: the f32 side panel's path
    dY[7, V // 2 + 123] = 1e6
alphas = np.logspace(-1, 8, bench.A)
shard = ShardContext.simulated(G, rank, device=dev, global_lists=False) if G > 1 else None
opts = None
if os.environ.get("FIT_OPTS"):                           # e.g. FIT_OPTS="screen_inner=0" or "screen_two_workgroups=0": FitOptions fields
    from litcoder_core_amd.engine.common import FitOptions
    kv = dict(item.split("=") for item in os.environ["FIT_OPTS"].split(","))
    opts = FitOptions(**{k: type(getattr(FitOptions(), k))(float(v)) for k, v in kv.items()})
model = NestedCVModel("ridge_regression", shard=shard, options=opts)
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.fit_predict_device(dX, dY, p, V, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    out = None
