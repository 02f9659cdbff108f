"""Like race_hunt.py, with the inner-CV scores and the V-independent operators captured too: per fold the (A, V) score
table, and checksums (fp64 sums) of the hat matrices H / series terms P of the fold -- to tell WHICH stage differs when
a repetition of the same fit is not bit-identical.    python tools/race_hunt2.py [reps] [voxels]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402
import litcoder_core_amd.nested_cv as ncv  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
V = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
cap = {"scores": [], "ops": []}
real_choose = ncv.RidgeCVEngine.fold_choose
pending = []


def spy_choose(self, st, single_alpha):
    out = real_choose(self, st, single_alpha)
    pending.append((st["scores"], st["hat"], st["best"]))       # looked at after the fit (no extra synchronisation)
    return out


ncv.RidgeCVEngine.fold_choose = spy_choose
ref = None
bad = 0
for it in range(reps):
    pending.clear()
    model = NestedCVModel("r")
    m, W, a = model.fit_predict_device(dX, dY, p, V, weights_on_host=False, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    cur = []
    for scores, hat, best in pending:
        sums = []
        for f0, fc, H, P in hat["Hs"]:
            sums.append((None if H is None else H.double().sum(dim=(1, 2)).cpu().numpy(),
                         None if P is None else P.double().sum(dim=(1, 2)).cpu().numpy()))
        cur.append((scores[:, :V].cpu().numpy(), sums, best[:V].cpu().numpy()))
    if ref is None:
        ref = cur
        continue
    for f, (c, r) in enumerate(zip(cur, ref)):
        d = c[0] != r[0]
        if d.any():
            bad += 1
            rows = np.nonzero(d.any(axis=1))[0]
            cols = np.nonzero(d.any(axis=0))[0]
            print(f"rep {it} fold {f} scores: {int(d.sum())} entries differ, alpha rows {rows.tolist()}, "
                  f"{cols.size} voxels (first {cols[:6].tolist()}, tiles {sorted(set((cols // 256).tolist()))[:12]}), "
                  f"max |d| {np.abs(c[0] - r[0])[d].max():.3e}", flush=True)
        for k, ((h1, p1), (h2, p2)) in enumerate(zip(c[1], r[1])):
            if h1 is not None and not np.array_equal(h1, h2):
                print(f"rep {it} fold {f} H checksums differ at systems {np.nonzero(h1 != h2)[0].tolist()}", flush=True)
            if p1 is not None and not np.array_equal(p1, p2):
                print(f"rep {it} fold {f} P checksums differ at inner folds {np.nonzero(p1 != p2)[0].tolist()}", flush=True)
        if not np.array_equal(c[2], r[2]):
            print(f"rep {it} fold {f} best: voxels {np.nonzero(c[2] != r[2])[0].tolist()[:10]}", flush=True)
print(f"{reps} repetitions, V={V}: {bad} folds with differing scores", flush=True)
