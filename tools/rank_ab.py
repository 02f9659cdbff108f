#!/usr/bin/env python3
"""A/B of FitOptions variants on ONE simulated rank of a G-rank strong-scaled cfg2 job (80 000 voxels in total),
interleaved on one box:   python tools/rank_ab.py G rank "second_fold_own_batch=False" ..."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from litcoder_core_amd import NestedCVModel, ShardContext, ops
from litcoder_core_amd.dist import shard_bounds
from litcoder_core_amd.nested_cv import FitOptions
G, rank = int(sys.argv[1]), int(sys.argv[2])
specs = sys.argv[3:]
dev = ops.device(0)
V_total = 80000
lo, hi = shard_bounds(V_total, G, rank)
V = hi - lo
dX, dY, p = bench.synth_inputs(V, rank, dev)
alphas = np.logspace(-1, 8, bench.A)
variants = [("defaults", FitOptions())]
for spec in specs:
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        kw[k] = eval(v)
    variants.append((spec, FitOptions(**kw)))
models = [(name, NestedCVModel("ridge_regression", options=o,
                               shard=ShardContext.simulated(G, rank, device=dev, global_lists=False))) for name, o in variants]
times = {name: [] for name, _ in variants}
for rep in range(8):
    for name, m in models:
        out = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.fit_predict_device(dX, dY, p, V, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)
        torch.cuda.synchronize()
        if rep >= 2:
            times[name].append(1e3 * (time.perf_counter() - t0))
for name, ts in times.items():
    print(f"G={G} rank {rank}  {name:46s} median {np.median(ts):7.2f} ms   min {min(ts):7.2f}   all {[round(t, 1) for t in ts]}")
