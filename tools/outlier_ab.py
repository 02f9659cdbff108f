#!/usr/bin/env python3
"""VERDICT r4 item 7 at the bench shape: a cfg2 fit (resident inputs, 80 000 voxels) whose targets hold ONE column with a
1e6 spike, against the same fit without it -- interleaved in one process.  Before round 5 the spike moved the whole fit to
the f32 MFMA path (5x slower); now that column alone is recomputed on the f32 side path.  Also the host-to-host call
(float64 arrays in voxel panels: the flag of a panel is looked at after its sweeps, so the fit is repeated once with the
targets resident -- 2x, not 5x).
    python tools/outlier_ab.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
dYs = dY.clone()
dYs[7, 40123] = 1e6
# the spiked voxel is noise to the model and takes the largest alphas of the grid, which no other voxel of the synthetic
# targets does: its refit brings the series alphas' operator chain into every fold (~1.5 ms per fold on the second
# auxiliary stream).  That is the price of one more alpha in use, not of the side path: the third arm is a clean fit in
# which the same voxel is ordinary-sized noise (it takes the same alphas, through the main path)
dYn = dY.clone()
dYn[:, 40123] = torch.randn(dY.shape[0], device=dY.device, generator=torch.Generator(device=dY.device).manual_seed(3))
alphas = np.logspace(-1, 8, bench.A)
models = {"clean": NestedCVModel("ridge_regression"), "clean, one noise voxel": NestedCVModel("ridge_regression"),
          "spike": NestedCVModel("ridge_regression"), "spike, precision=f32": NestedCVModel("ridge_regression", precision="f32")}
data = {"clean": dY, "clean, one noise voxel": dYn, "spike": dYs, "spike, precision=f32": dYs}
times = {k: [] for k in models}
res = {}
for k in models:
    models[k].fit_predict_device(dX, data[k], p, V, alphas=alphas, **bench.FIT_KW)
for _ in range(rounds):
    for k in models:
        if k.endswith("f32") and len(times[k]) >= 2:
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = models[k].fit_predict_device(dX, data[k], p, V, alphas=alphas, **bench.FIT_KW)
        torch.cuda.synchronize()
        times[k].append(1e3 * (time.perf_counter() - t0))
        res[k] = (np.asarray(out[0]["correlations"]), np.asarray(out[2]), out[1][:, 40123].cpu().numpy())
for k, t in times.items():
    print(f"{k:22s}: median {np.median(t):7.2f} ms (min {min(t):.2f}, max {max(t):.2f}); arithmetic {models[k].last_fit['precision']}, "
          f"side panel columns {models[k].last_fit.get('side_panel_cols')}")
c, s, f = res["clean"], res["spike"], res["spike, precision=f32"]
others = np.ones(V, dtype=bool)
others[40123] = False
print(f"spike / clean fit time: {np.median(times['spike']) / np.median(times['clean']):.4f}; against the clean fit whose voxel "
      f"40123 is ordinary noise (alpha {res['clean, one noise voxel'][1][40123]:g}; the spiked voxel: {res['spike'][1][40123]:g}): "
      f"{np.median(times['spike']) / np.median(times['clean, one noise voxel']):.4f}")
print(f"other voxels bit-identical to the clean fit: correlations {np.array_equal(c[0][others], s[0][others])}, alphas "
      f"{np.array_equal(c[1][others], s[1][others])}")
print(f"the spiked voxel against the exact-f32 fit: |dcorr| {abs(s[0][40123] - f[0][40123]):.2e}, alpha {s[1][40123]:g} vs "
      f"{f[1][40123]:g}, max |dW| / max |W| {np.abs(s[2] - f[2]).max() / np.abs(f[2]).max():.2e}")
# host to host
X, Y = bench.host_arrays(dX, dYs, p, V)
mh = NestedCVModel("ridge_regression")
mh.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = mh.fit_predict(X, Y, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(f"host to host with the spike: median {np.median(ts):.1f} ms; arithmetic {mh.last_fit['precision']}, side panel columns "
      f"{mh.last_fit.get('side_panel_cols')}; spiked voxel's correlation equal to the resident fit's: "
      f"{out[0]['correlations'][40123] == float(s[0][40123])}")
