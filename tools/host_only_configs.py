#!/usr/bin/env python3
"""Host-to-host fits ONLY (the reference's call: float64 numpy in, metrics + float32 host weights out) at cfg2's, cfg4's and
cfg5's shapes, one process per config, no resident fit before them: what a caller of the reference's API sees.
    python tools/host_only_configs.py cfg2|cfg4|cfg5 [fits]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = ops.device(0)
KW = dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, single_alpha=False, normalpha=True, use_corr=True)
shape = {"cfg2": dict(T=3000, F0=768, delays=[1, 2, 3, 4], V=80000, A=20),
         "cfg4": dict(T=2226, F0=768, delays=[1, 2, 3, 4], V=200000, A=20),
         "cfg5": dict(T=3000, F0=1280, delays=[1, 2, 3, 4, 5, 6], V=80000, A=32)}[name]
T, V = shape["T"], shape["V"]
rng = np.random.default_rng(0)
Xd = ops.fir_delay(torch.from_numpy(rng.standard_normal((T, shape["F0"]))).to(dev), shape["delays"], False).to(torch.float32)
p = Xd.shape[1]
if name == "cfg5":
    Xd /= torch.as_tensor(np.r_[np.full(p // 2, 1.0), np.full(p - p // 2, 2.0)], dtype=torch.float32, device=dev)
g = torch.Generator(device=dev); g.manual_seed(7)
Wt = 0.02 * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
Xh = Xd.cpu().numpy().astype(np.float64)
Yh = np.empty((T, V), dtype=np.float64)
for c in range(0, V, 16384):
    c1 = min(V, c + 16384)
    Yh[:, c:c1] = (Xd @ Wt[:, c:c1] + torch.randn((T, c1 - c), generator=g, device=dev, dtype=torch.float32)).cpu().numpy()
del Xd, Wt
torch.cuda.empty_cache()
alphas = np.logspace(-1, 8, shape["A"])
model = NestedCVModel("ridge_regression")
ts = []
for i in range(fits):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.fit_predict(Xh, Yh, alphas=alphas, **KW)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    out = None
v = sorted(ts[2:])
print(f"{name} host to host: median {v[len(v) // 2]:.1f} ms (min {v[0]:.1f}, max {v[-1]:.1f}); streams' order "
      f"'{os.environ.get('LITCODER_AMD_STREAM_ORDER', '')}'")
