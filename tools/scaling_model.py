#!/usr/bin/env python3
"""Strong-scaling model of the cfg2 fit from per-rank timelines measured on ONE GPU.

For G in {1, 2, 4, 8}: run rank r of a G-rank job ALONE (ShardContext.simulated: every collective is a local copy,
so the rank executes exactly its share of the V-independent fp64 systems and its block of V_total / G voxels) and
time whole fits.  An G-GPU job takes max over ranks of that time plus the wire time of the all-gathers, which this
box cannot measure: it is added from the bytes each rank receives at a stated xGMI rate.  Results are meaningless
numerically (other ranks' operators are copies of this rank's), only the time is read.

    python tools/scaling_model.py [V_total] [--ranks all|first]  ->  JSON on stdout
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ShardContext, ops  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402

V_total = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 80000
all_ranks = "--ranks" in sys.argv and sys.argv[sys.argv.index("--ranks") + 1] == "all"
XGMI_ALLGATHER_GBPS = 300.0          # assumed all-gather rate INTO one rank (7 links x ~153 GB/s peak; RCCL ~1/3)
dev = ops.device(0)
alphas = np.logspace(-1, 8, bench.A)
n_o = bench.T - bench.T // bench.N_OUTER
n_v = n_o // bench.N_INNER
n_i = n_o - n_v
out = {"V_total": V_total, "assumed_allgather_GBps_into_a_rank": XGMI_ALLGATHER_GBPS, "per_world": {}}
t1 = None
for G in (1, 2, 4, 8):
    ranks = range(G) if all_ranks else sorted({0, G // 2, G - 1})
    per_rank = {}
    for r in ranks:
        lo, hi = shard_bounds(V_total, G, r)
        dX, dY, p = bench.synth_inputs(hi - lo, r, dev)
        shard = ShardContext.simulated(G, r, device=dev, global_lists=False) if G > 1 else None
        model = NestedCVModel("ridge_regression", shard=shard)
        fit = lambda: model.fit_predict_device(dX, dY, p, hi - lo, n_voxels_total=V_total, alphas=alphas, **bench.FIT_KW)
        fit(); fit()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fit()
        torch.cuda.synchronize()
        per_rank[r] = 1e3 * (time.perf_counter() - t0) / 5
        del dX, dY
        torch.cuda.empty_cache()
    # bytes a rank RECEIVES per fit: hat matrices of the inner folds (4 Cholesky alphas x 5 inner folds per outer fold),
    # refit operators (4 alphas x (p_pad + pad(n_t)) rows), 7/8 of each at G = 8
    hat = 5 * 20 * ops.pad_to(n_v, 32) * ops.pad_to(n_i, 64) * 4
    refit = 5 * 4 * (3072 + 768) * ops.pad_to(n_o, 64) * 4
    wire_ms = 0.0 if G == 1 else 1e3 * (hat + refit) * (G - 1) / G / (XGMI_ALLGATHER_GBPS * 1e9)
    t = max(per_rank.values())
    if G == 1:
        t1 = t
        out["alphas_in_use_last_fold_1gpu"] = model.last_fit.get("used_all")
    out["per_world"][G] = {"ms_per_rank_alone": {str(k): round(v, 2) for k, v in per_rank.items()}, "max_ms": round(t, 2),
                           "allgather_bytes_received": int((hat + refit) * (G - 1) / G), "wire_ms_if_not_hidden": round(wire_ms, 2),
                           "predicted_ms": round(t + wire_ms, 2), "predicted_speedup": round(t1 / (t + wire_ms), 2),
                           "voxels_per_sec": round(V_total / (1e-3 * (t + wire_ms)))}
    print(f"G={G}: {out['per_world'][G]}", file=sys.stderr, flush=True)
# weak scaling: every rank 80 000 voxels (bench.py's default mode): rank 0 of G alone on its 80 000 of 80 000 G voxels
out["weak"] = {}
for G in (2, 4, 8):
    dX, dY, p = bench.synth_inputs(80000, 0, dev)
    model = NestedCVModel("ridge_regression", shard=ShardContext.simulated(G, 0, device=dev, global_lists=False))
    fit = lambda: model.fit_predict_device(dX, dY, p, 80000, n_voxels_total=80000 * G, alphas=alphas, **bench.FIT_KW)
    fit(); fit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        fit()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 4
    out["weak"][G] = {"ms_rank0_alone": round(ms, 1), "voxels_per_sec": round(80000 * G / (1e-3 * ms)),
                      "speedup_vs_1gpu": round(80000 * G / (1e-3 * ms) / (V_total / (1e-3 * t1)), 2)}
    print(f"weak G={G}: {out['weak'][G]}", file=sys.stderr, flush=True)
    del dX, dY
    torch.cuda.empty_cache()
print(json.dumps(out))
