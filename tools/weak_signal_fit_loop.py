#!/usr/bin/env python3
"""A few resident fits at cfg2's shape with half the voxels pure noise (for rocprofv3 --kernel-trace --stats).
    python tools/weak_signal_fit_loop.py [fits]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from litcoder_core_amd import NestedCVModel, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
V = 80000
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
W = 0.02 * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
W[:, torch.rand(V, generator=g, device=dev) < 0.5] = 0.0
dY[:, :V] = dX[:, :p] @ W + torch.randn((dY.shape[0], V), generator=g, device=dev, dtype=torch.float32)
del W
alphas = np.logspace(-1, 8, bench.A)
m = NestedCVModel("ridge_regression")
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
lf = m.last_fit
print({k: lf.get(k) for k in ("undecided", "screened", "screen_overflows", "used_all", "refine_launch_cols", "mean_operator")})
print([np.bincount(np.searchsorted(alphas, a * (1 - 1e-9)), minlength=20).tolist() for a in m.last_fold_alphas][:2])
