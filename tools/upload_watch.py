#!/usr/bin/env python3
"""When the upload panels of a host-to-host cfg2 fit are resident (a watcher thread waits for every job of the fit's own
PanelUploader on a stream of its own; nothing else is instrumented), and when the first V-wide kernels of the steps ran.
    python tools/upload_watch.py [fits]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
X, Y = bench.host_arrays(dX, dY, p, V)
del dX, dY
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
t0 = [0.0]
marks = []
watch_stream = torch.cuda.Stream()
orig_init = ops.PanelUploader.__init__


def init(self, jobs, dev_, after=None):
    orig_init(self, jobs, dev_, after=after)
    up = self

    def watch():
        for j in range(len(up.jobs)):
            up.wait(j, watch_stream)
            watch_stream.synchronize()
            marks.append((j, 1e3 * (time.perf_counter() - t0[0])))
    threading.Thread(target=watch, daemon=True).start()


ops.PanelUploader.__init__ = init
for i in range(n):
    del marks[:]
    torch.cuda.synchronize()
    t0[0] = time.perf_counter()
    out = model.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0[0])
    time.sleep(0.05)
    print(f"fit {i}: {dt:.1f} ms; upload jobs resident after {[round(m, 1) for _, m in sorted(marks)]} ms", flush=True)
    out = None
