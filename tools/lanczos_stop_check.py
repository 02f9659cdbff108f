#!/usr/bin/env python3
"""lambda_max of the row sets of a fit from the masked multi-system Lanczos run, with and without the convergence stop,
against numpy's eigvalsh: the design of tools/fuzz_vs_oracle.py's seed 1234 / case 4 (large) -- centred features, p > n --
where the stop of round 5 went wrong.     python tools/lanczos_stop_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402

dev = ops.device(0)
rng = np.random.default_rng(7)
T, p = 1288, 1000
X = (rng.standard_normal((T, p)) * rng.uniform(0.5, 2.0, p)).astype(np.float32).astype(np.float64)
idx = np.arange(T)
outer = np.array_split(idx, 3)
for fo in range(3):
    tr = np.concatenate([outer[j] for j in range(3) if j != fo])
    Xn = (X - X[tr].mean(0)) / (X[tr].std(0) + 1e-8)
    K = Xn @ Xn.T
    inner = np.array_split(tr, 3)
    sets = [np.concatenate([inner[j] for j in range(3) if j != fi]) for fi in range(3)] + [tr]
    want = np.asarray([np.linalg.eigvalsh(K[np.ix_(s, s)])[-1] for s in sets])
    dK = torch.from_numpy(K).to(dev)
    bits = np.zeros(T, dtype=np.uint32)
    for f, rows in enumerate(sets):
        bits[rows] |= np.uint32(1 << f)
    member = ops.upload(bits.view(np.int32), dev)
    for tol in (0.0, 1e-6, 1e-9):
        for mfma in (True, False):
            got = ops.lambda_max_masked(dK, T, member, len(sets), 64, use_mfma=mfma, tol=tol).cpu().numpy()
            print(f"outer fold {fo} tol {tol:g} mfma {int(mfma)}: relative errors {np.array2string((want - got) / want, precision=2)}", flush=True)
