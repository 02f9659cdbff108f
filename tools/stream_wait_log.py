#!/usr/bin/env python3
"""Which cross-stream waits the MAIN stream of a resident cfg2 fit is given, in host order, with the caller of each
(the main stream's first V-wide kernel runs ~10 ms after the fit began: behind which wait?).   python tools/stream_wait_log.py"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

dev = ops.device(0)
dX, dY, p = bench.synth_inputs(80000, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
model = NestedCVModel("ridge_regression")
for _ in range(2):
    model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
torch.cuda.synchronize()
log = []
t0 = [0.0]
orig_wait = torch.cuda.Stream.wait_event
main_id = torch.cuda.current_stream().cuda_stream


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if "litcoder_core_amd" in f.filename]
    return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])


def wait_event(self, ev):
    log.append((time.perf_counter() - t0[0], "main" if self.cuda_stream == main_id else hex(self.cuda_stream), where()))
    return orig_wait(self, ev)


torch.cuda.Stream.wait_event = wait_event
t0[0] = time.perf_counter()
model.fit_predict_device(dX, dY, p, 80000, alphas=alphas, **bench.FIT_KW)
torch.cuda.synchronize()
for t, s, w in log:
    if t < 0.012:
        print(f"{t * 1e3:7.2f} ms  {s:>14s} waits   {w}")
