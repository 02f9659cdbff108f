#!/usr/bin/env python3
"""Timeline of one HOST-TO-HOST cfg2 fit (float64 numpy in -> metrics + host float32 weights out): host wall time of
the engine phases, when each upload panel had been staged / had landed in HBM, when the V-wide phases ran on the
device, when each weight panel had reached the host.   python tools/host_path_timeline.py [V] [cfg3]
cfg3: the story pipeline (StoryPipeline.fit_words at BASELINE configs[2]'s size) instead, with the V-wide launches of the
primal form marked one by one (runs of one kind merged)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, nested_cv as ncv, ops  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
CFG3 = "cfg3" in sys.argv[2:]
dev = ops.device(0)
model = NestedCVModel("ridge_regression")
if CFG3:
    from litcoder_core_amd import StoryPipeline
    words, wtimes, trtimes, brain = bench.synth_stories(V, dev)
    pipe = StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM, model=model)
else:
    dX, dY, p = bench.synth_inputs(V, 0, dev)
    X, Y = bench.host_arrays(dX, dY, p, V)
    alphas = np.logspace(-1, 8, bench.A)
host_log, dev_marks = [], []
T0 = [0.0]


def wrap_host(obj, name, label=None):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            host_log.append((label or name, t, time.perf_counter()))
    setattr(obj, name, inner)


def wrap_dev(name, label):
    fn = getattr(ncv.RidgeCVEngine, name)

    def inner(self, *a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        out = fn(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        dev_marks.append((label, e0, e1))
        return out
    setattr(ncv.RidgeCVEngine, name, inner)


for n, lab in (("_sweeps", "MAIN sweeps"), ("fold_finish", "MAIN refit + statistics"), ("fold_choose", "MAIN choose + group")):
    wrap_dev(n, lab)
for n in ("begin_fit", "precompute_lmax", "prepare_folds", "fold_begin", "fold_choose", "fold_select", "fold_speculate",
          "fold_finish", "fold_collect", "_wait_targets", "_range_finished", "combined_significance", "weights"):
    wrap_host(ncv.RidgeCVEngine, n)

import threading
real_init = ncv.RidgeCVEngine.__init__
watch_stream = torch.cuda.Stream()


def init_and_watch(self, *a, **k):
    real_init(self, *a, **k)
    up = self.uploader
    if up is None:
        return

    def watch():
        torch.cuda.set_device(0)
        for j in range(len(up.jobs)):
            up.wait(j, watch_stream)                       # host: issued; device: watch_stream behind the panel's copies
            host_log.append((f"upload job {j} staged+issued", time.perf_counter(), time.perf_counter()))
            e = torch.cuda.Event(enable_timing=True)
            e.record(watch_stream)
            dev_marks.append((f"UPLOAD job {j} cols {up.jobs[j][2]}:{up.jobs[j][3]} landed", e, e))

    threading.Thread(target=watch, daemon=True).start()


ncv.RidgeCVEngine.__init__ = init_and_watch
real_dl = ops.download_cols


def dl(src, host, c0, Vc, stream):
    real_dl(src, host, c0, Vc, stream)
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    dev_marks.append((f"DOWNLOAD cols {c0}:{c0 + Vc} on the host", e, e))


ops.download_cols = dl


def wrap_op(name, label):
    fn = getattr(ops, name)

    def inner(*a, **k):
        s = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
        out = fn(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
        dev_marks.append((label, e0, e1))
        return out
    setattr(ops, name, inner)


if CFG3:
    for n, lab in (("lanczos_interp_stories", "  features: Lanczos, all stories"), ("story_design", "  features: design matrix"),
                   ("gemm_grouped_f16x3", "  . plain f16x3 launch (block product / series chain / refit)"),
                   ("series_sweep_scores_f16x3", "  . series sweep"), ("alpha_sweep_scores_f16x3", "  . fused sweep"),
                   ("val_stats", "  . validation statistics"), ("combine_colmax", "  . B_f + column maxima"),
                   ("split_cols_f16", "  . column image"), ("batch_chol_solve", "  p x p: Cholesky batch"),
                   ("lambda_max_masked", "  p x p: Lanczos run for lambda_max")):
        if hasattr(ops, n):
            wrap_op(n, lab)


def run():
    if CFG3:
        out = pipe.fit_words(words, wtimes, trtimes, brain, **bench.CFG3_KW)
    else:
        out = model.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    return out


for _ in range(2):
    run()
host_log.clear(); dev_marks.clear()
torch.cuda.synchronize()
slow_calls, stack = [], []
if "trace" in sys.argv[2:]:
    # every Python / C call of the caller's thread that took >= 0.5 ms, with its depth: what the host thread waits in
    def prof(frame, event, arg):
        now = time.perf_counter()
        if event in ("call", "c_call"):
            name = (frame.f_code.co_name + " @" + os.path.basename(frame.f_code.co_filename) + ":" + str(frame.f_lineno)
                    if event == "call" else "C " + getattr(arg, "__qualname__", str(arg)) + " <- " + frame.f_code.co_name +
                    ":" + str(frame.f_lineno))
            stack.append((name, now))
        elif stack:
            name, t = stack.pop()
            if now - t >= 5e-4:
                slow_calls.append((t, now, len(stack), name))
    sys.setprofile(prof)
start = torch.cuda.Event(enable_timing=True); start.record()
t0 = time.perf_counter()
run()
t1 = time.perf_counter()
sys.setprofile(None)
print(f"host-to-host fit: {1e3 * (t1 - t0):.1f} ms   panels {model.last_fit.get('panels')}")
print("---- host thread(s)")
for name, a, b in sorted(host_log, key=lambda x: x[1]):
    print(f"  {1e3 * (a - t0):8.2f} -> {1e3 * (b - t0):8.2f}  ({1e3 * (b - a):7.2f} ms)  {name}")
if slow_calls:
    print("---- calls of the host thread >= 0.5 ms (depth-indented)")
    for a, b, d, name in sorted(slow_calls):
        print(f"  {1e3 * (a - t0):8.2f} -> {1e3 * (b - t0):8.2f}  ({1e3 * (b - a):7.2f} ms)  {'  ' * min(d, 12)}{name}")
print("---- device")
rows = sorted(((start.elapsed_time(e0), start.elapsed_time(e1), lab) for lab, e0, e1 in dev_marks))
merged = []
for a, b, lab in rows:
    if merged and merged[-1][2] == lab and lab.startswith("  .") and a - merged[-1][1] < 0.05:
        merged[-1][1] = b; merged[-1][3] += 1
    else:
        merged.append([a, b, lab, 1])
for a, b, lab, n in merged:
    print(f"  {a:8.2f} -> {b:8.2f}  ({b - a:7.2f} ms)  {lab}{'' if n == 1 else ' x%d' % n}")
