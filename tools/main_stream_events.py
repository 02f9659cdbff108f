#!/usr/bin/env python3
"""GPU-side timeline of the MAIN stream of one cfg2 fit from HIP events recorded around its phases (no profiler, no
added synchronisation): when each phase started and ended on the device, and the idle time between them.
    python tools/main_stream_events.py [V_total [world rank]]   (world > 1: one rank of a simulated sharded job; the
    events of the auxiliary streams' phases are listed too, each on the stream it ran on)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops  # noqa: E402

from litcoder_core_amd import ShardContext  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402
V_total = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
world, rank = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 0)
dev = ops.device(0)
lo, hi = shard_bounds(V_total, world, rank)
V = hi - lo
dX, dY, p = bench.synth_inputs(V, rank, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression", shard=ShardContext.simulated(world, rank, device=dev, global_lists="--global-lists" in sys.argv)
                      if world > 1 else None)
marks = []


def wrap(name, label):
    fn = getattr(ncv.RidgeCVEngine, name)

    def inner(self, *a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        out = fn(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append((label, e0, e1))
        return out
    setattr(ncv.RidgeCVEngine, name, inner)


for n, lab in (("_sweeps", "MAIN sweeps (with a look-ahead: their first part)"), ("fold_sweeps_finish", "MAIN sweeps, fused part"),
               ("fold_finish", "MAIN refit apply + statistics"), ("fold_choose", "MAIN choose + group"),
               ("lmax_systems", "aux  lanczos"), ("_hat_matrices", "aux  hat matrices (series chain + cholesky)"),
               ("_refit_chol", "aux2 refit cholesky"), ("_refit_systems", "aux2 refit systems (poly chain, copies)"),
               ("refit_ahead", "aux2 refit_ahead (entry/exit on the issuing stream)")):
    wrap(n, lab)


def run():
    return model.fit_predict_device(dX, dY, p, V, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)


run(); torch.cuda.synchronize(); marks.clear()
start = torch.cuda.Event(enable_timing=True); start.record()
run()
end = torch.cuda.Event(enable_timing=True); end.record()
torch.cuda.synchronize()
print(f"fit (device time on the main stream): {start.elapsed_time(end):.1f} ms")
rows = sorted(((start.elapsed_time(e0), start.elapsed_time(e1), lab) for lab, e0, e1 in marks))
prev = 0.0
busy = 0.0
for a, b, lab in rows:
    if lab.startswith("MAIN"):
        print(f"  {a:7.2f} -> {b:7.2f}  ({b - a:6.2f} ms)  {lab}   [idle before: {a - prev:5.2f} ms]")
        busy += b - a
        prev = b
    else:
        print(f"  {a:7.2f} -> {b:7.2f}  ({b - a:6.2f} ms)      {lab}")
print(f"main-stream phases {busy:.1f} ms, between them {start.elapsed_time(end) - busy:.1f} ms")
