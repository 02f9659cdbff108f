#!/usr/bin/env python3
"""How many Lanczos steps S[0]^2 of the bench designs needs: lambda_max of the cfg2 Gram blocks (dual form, masked runs)
and of the cfg3 p x p Gram (primal form) after 8 .. 128 steps, relative to the 128-step value.
    python tools/lanczos_convergence.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import StoryPipeline, ops  # noqa: E402

dev = ops.device(0)
STEPS = (8, 12, 16, 20, 24, 32, 48, 64, 128)


def report(name, fn):
    vals = {s: fn(s).cpu().numpy() for s in STEPS}
    ref = vals[128]
    print(name)
    for s in STEPS:
        print(f"   {s:4d} steps: max rel. difference from 128 steps {np.abs(vals[s] / ref - 1).max():.2e}")


dX, dY, p = bench.synth_inputs(8192, 0, dev)
T = dX.shape[0]
K = ops.gram(dX, T, p)
tr = np.arange(T)[: int(0.8 * T)]
sets = [np.setdiff1d(tr, tr[i::5]) for i in range(5)] + [tr]
rows = ops.idx_matrix(sets, ops.pad_to(len(tr), 64), dev)
report(f"cfg2 synthetic design (T {T}, p {p}): K[I, I] of 5 inner training sets + the outer one",
       lambda s: ops.lambda_max(K, rows, len(sets), rows.shape[1], s))

words, wtimes, trtimes, brain = bench.synth_stories(256, dev)
pipe = StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM)
names = list(words)
dW = torch.cat([torch.from_numpy(words[s]).to(dev) for s in names])
feat, off = ops.lanczos_interp_stories(dW, [wtimes[s] for s in names], [trtimes[s] for s in names], 3, 1.0, False)
dXs, Ts, Tt, ps, _ = pipe.design(feat, off, [len(trtimes[s]) for s in names], names)
Xtr = dXs[:Ts, :ps].double()
G = (Xtr.T @ Xtr).contiguous().reshape(1, ps, ps)
ident = ops.idx_matrix([np.arange(ps)], ps, dev)
report(f"cfg3 synthetic design (T {Ts}, p {ps}): G = X'X of the outer training block",
       lambda s: ops.lambda_max_strided(G, ps, ps * ps, ident, 1, ps, s))
