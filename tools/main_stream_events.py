#!/usr/bin/env python3
"""GPU-side timeline of the MAIN stream of one cfg2 fit from HIP events recorded around its phases (no profiler, no
added synchronisation): when each phase started and ended on the device, and the idle time between them.
    python tools/main_stream_events.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops  # noqa: E402

V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
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


for n, lab in (("_sweeps", "sweeps"), ("fold_finish", "refit apply + statistics"), ("fold_choose", "choose + group")):
    wrap(n, lab)


def run():
    return model.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)


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
    print(f"  {a:7.2f} -> {b:7.2f}  ({b - a:6.2f} ms)  {lab}   [idle before: {a - prev:5.2f} ms]")
    busy += b - a
    prev = b
print(f"phases {busy:.1f} ms, between them {start.elapsed_time(end) - busy:.1f} ms")
