#!/usr/bin/env python3
"""Interleaved A/B of whole fits in ONE process (cdna_hip_programming.md rule 24: never rank builds by timings taken in
different processes or on different boxes): every arm is a set of FitOptions overrides, the arms take turns for `rounds`
rounds, and the distribution per arm is printed (median, min, max) with the results' equality across arms.

    python tools/ab_fits.py <cfg2h|cfg2r|cfg3> <rounds> [<name>=<value>[,<name>=<value>...] | default] ...
e.g. python tools/ab_fits.py cfg3 8 default lanczos_dense=0 lanczos_tol=1e-6
     cfg2h = the bench headline (host float64 arrays in, host weights out), cfg2r = resident inputs, cfg3 = the story pipeline
     AB_RANK=G,r (cfg2r): as simulated rank r of G (its AB_VOXELS / G voxels, its share of the V-independent systems)
"""
import dataclasses
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ShardContext, StoryPipeline, ops  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402
from litcoder_core_amd.engine.common import FitOptions  # noqa: E402

work, rounds = sys.argv[1], int(sys.argv[2])
arms = sys.argv[3:] or ["default"]
V = int(os.environ.get("AB_VOXELS", "80000"))
dev = ops.device(0)


def options(spec):
    if spec == "default":
        return FitOptions()
    kw = {}
    fields = {f.name: f for f in dataclasses.fields(FitOptions)}
    for item in spec.split(","):
        k, v = item.split("=")
        t = fields[k].type if not isinstance(fields[k].type, str) else {"int": int, "float": float, "bool": bool}[fields[k].type]
        kw[k] = (v not in ("0", "False", "false")) if t is bool else t(v)
    return FitOptions(**kw)


G, rank = (int(x) for x in os.environ.get("AB_RANK", "1,0").split(","))
V_total = V
if G > 1:
    lo, hi = shard_bounds(V_total, G, rank)
    V = hi - lo
shard = lambda: ShardContext.simulated(G, rank, device=dev, global_lists=False) if G > 1 else None
models = {a: NestedCVModel("ridge_regression", options=options(a), shard=shard()) for a in arms}
if work == "cfg3":
    words, wtimes, trtimes, brain = bench.synth_stories(V, dev)
    pipes = {a: StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM, model=m) for a, m in models.items()}
    run = lambda a: pipes[a].fit_words(words, wtimes, trtimes, brain, **bench.CFG3_KW)
else:
    dX, dY, p = bench.synth_inputs(V, rank, dev)
    alphas = np.logspace(-1, 8, bench.A)
    if work == "cfg2h":
        host = bench.host_arrays(dX, dY, p, V)
        run = lambda a: models[a].fit_predict(host[0], host[1], alphas=alphas, **bench.FIT_KW)
    else:
        run = lambda a: models[a].fit_predict_device(dX, dY, p, V, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)

times = {a: [] for a in arms}
res = {}
for a in arms:
    run(a); run(a)
for _ in range(rounds):
    for a in arms:
        out = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(a)
        torch.cuda.synchronize()
        times[a].append(1e3 * (time.perf_counter() - t0))
        res[a] = (np.asarray(out[0]["correlations"]), np.asarray(out[2]))
base = arms[0]
for a in arms:
    t = np.asarray(times[a])
    dc = float(np.abs(res[a][0] - res[base][0]).max())
    same_alpha = float(np.mean(res[a][1] == res[base][1]))
    print(f"{work} {a:40s}: median {np.median(t):7.2f} ms  min {t.min():7.2f}  max {t.max():7.2f}  ({rounds} interleaved rounds); "
          f"vs {base}: max |dcorr| {dc:.2e}, alphas equal {same_alpha:.5f}", flush=True)
