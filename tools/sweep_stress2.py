"""Fold 0's sweeps of the cfg2 fit repeated while the auxiliary stream runs what it runs beside them in a real fit: the
V-independent preparation of folds 1-4 (series chain; its Cholesky part gated behind the sweeps).  Scores compared with
an undisturbed run.  STRESS_SKIP=name[,name]: ops replaced by no-ops on the aux side (to find the culprit).
    python tools/sweep_stress2.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import ops  # noqa: E402
import litcoder_core_amd.nested_cv as ncv  # noqa: E402
from litcoder_core_amd.folding import create_folds  # noqa: E402

V = int(os.environ.get("STRESS_V", "80000"))
reps = int(os.environ.get("STRESS_REPS", "16"))
skip = [s for s in os.environ.get("STRESS_SKIP", "").split(",") if s]
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
eng = ncv.RidgeCVEngine(ncv._DeviceShapes(dX, p), ncv._DeviceShapes(dY, V), alphas, True, True, False, False)
eng.begin_fit(5)
outer = []
for tr, te in create_folds(bench.T, "kfold", 5):
    outer.append((tr, te, create_folds(len(tr), "kfold", 5)))
lmax_pre = eng.precompute_lmax(outer)
base = eng.prepare_folds(outer[:1], lmax_pre[:1])[0]
torch.cuda.synchronize()


def sweep(done=None):
    hat = dict(base["hat"])
    cs, split = eng._target_scales(eng.dY_full, eng.full)
    hat.update(cs=cs, split=split)
    return eng._sweeps(hat, eng.dY_full, done)


N, M, B = 1920, 480, 20
aug0 = torch.randn((B, N + M, N), dtype=torch.float64, device=dev) * 0.01
aug0[:, :N] += torch.eye(N, dtype=torch.float64, device=dev) * 50.0


ref = sweep()
torch.cuda.synchronize()
ref = ref.clone()
real = {n: getattr(ops, n) for n in skip}
bad = 0
for it in range(reps):
    gate = torch.cuda.Event()
    for n in skip:
        setattr(ops, n, lambda *a, **k: None)
    try:
        done = None
        if os.environ.get("STRESS_NO_AUX") != "1":
            # like fold 0 of a fit: the auxiliary stream is still in fold 0's own Cholesky chain while the series pass runs,
            # then (event `done`: the fused pass may start) goes straight on with the preparation of folds 1-4
            with torch.cuda.stream(eng.aux):
                aug = aug0.clone()
                H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
                ops.batch_chol_solve(aug, B, N, M, H)
                done = torch.cuda.Event()
                done.record()
            eng.prepare_folds(outer[1:], lmax_pre[1:], chol_after=gate)
    finally:
        for n in skip:
            setattr(ops, n, real[n])
    s = sweep(done)
    gate.record()
    torch.cuda.synchronize()
    d = (s[:, :V] != ref[:, :V])
    if bool(d.any()):
        bad += 1
        rows = torch.nonzero(d.any(dim=1)).flatten().tolist()
        cols = torch.nonzero(d.any(dim=0)).flatten()
        print(f"  rep {it}: {int(d.sum())} entries differ, alpha rows {rows}, {cols.numel()} voxels, "
              f"tiles {sorted(set((cols // 256).tolist()))[:10]}", flush=True)
print(f"skip={skip}: {bad} of {reps} sweeps differ from the undisturbed run", flush=True)
