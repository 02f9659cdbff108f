#!/usr/bin/env python3
"""The cfg2 fit's own upload (float64 design + targets in the fit's four column panels) by itself, beside a host thread that
queues small launches as the fit's driver does, and beside fp64 Cholesky batches on another stream: inside a host-to-host fit
the second panel (24 576 columns) takes 17 ms where the link alone would need 6 -- which of the fit's activities does that?
    python tools/upload_in_fit_probe.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import ops  # noqa: E402

V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
X, Y = bench.host_arrays(dX, dY, p, V)
T = X.shape[0]
panels = [(0, 12288), (12288, 36864), (36864, 67584), (67584, V)]
dXu = torch.zeros_like(dX)
dYu = torch.empty_like(dY)
print(f"X {X.shape} {X.dtype}, Y {Y.shape} {Y.dtype}: {Y.nbytes / 1e9:.2f} GB on the host, {Y.nbytes / 2e9:.2f} GB on the link")


def one(label, beside=None):
    for rep in range(3):
        torch.cuda.synchronize()
        stop = threading.Event()
        side = None
        if beside is not None:
            side = beside(stop)
        t0 = time.perf_counter()
        up = ops.PanelUploader([(ops.HostRows([X]), dXu, 0, p)] + [(ops.HostRows([Y]), dYu, a, b) for a, b in panels], dev)
        marks = []
        s = torch.cuda.Stream()
        for j in range(len(panels) + 1):
            up.wait(j, s)
            s.synchronize()
            marks.append(1e3 * (time.perf_counter() - t0))
        up.join()
        stop.set()
        if side is not None:
            side()
        torch.cuda.synchronize()
        print(f"  {label}: design + panels resident after {[round(m, 1) for m in marks]} ms", flush=True)


def launches_in_this_thread(stop):
    """The fit's driver: thousands of small launches from the Python main thread while the upload runs -- here from a second
    Python thread (the main one blocks in up.wait, lock released)."""
    n = [0]

    def run():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            buf = torch.empty(1 << 16, dtype=torch.float32, device=dev)
            while not stop.is_set():
                ops.zero_cols(buf.view(256, 256), 0, 256)
                n[0] += 1
    th = threading.Thread(target=run)
    th.start()

    def done():
        th.join()
        print(f"      ({n[0]} small launches queued meanwhile)")
    return done


def chains(stop):
    """fp64 Cholesky batches (the hat-matrix chains' kernels) back to back on another stream."""
    st = torch.cuda.Stream()
    B, N, M = 20, 1920, 480
    a2 = torch.full((B,), 50.0, dtype=torch.float64, device=dev)
    K = torch.randn((N + M, N + M), dtype=torch.float64, device=dev)
    K = K @ K.T / (N + M) + torch.eye(N + M, dtype=torch.float64, device=dev)
    aug = torch.empty((B, N + M, N), dtype=torch.float64, device=dev)
    n = [0]

    def run():
        with torch.cuda.stream(st):
            while not stop.is_set():
                for b in range(B):
                    aug[b].copy_(K[:, :N])
                h = torch.empty((B, M, N), dtype=torch.float32, device=dev)
                ops.batch_chol_solve(aug, B, N, M, h)
                st.synchronize()
                n[0] += 1
    th = threading.Thread(target=run)
    th.start()

    def done():
        th.join()
        print(f"      ({n[0]} Cholesky batches of {B} x {N} meanwhile)")
    return done


one("alone")
one("beside a thread queueing small launches", launches_in_this_thread)
try:
    one("beside fp64 Cholesky batches", chains)
except Exception as e:   # (signature drift of the probe's helper is not the measurement's problem)
    print("  chains arm failed:", repr(e)[:200])
