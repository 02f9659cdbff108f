#!/usr/bin/env python3
"""Scaling model of a fit from per-rank timelines measured on ONE GPU, for the configs BASELINE.json puts on 8 GPUs.

For G in {1, 2, 4, 8}: run rank r of a G-rank job ALONE (ShardContext.simulated: every collective is a local copy,
so the rank executes exactly its share of the V-independent fp64 systems and its block of V_total / G voxels) and
time whole fits.  A G-GPU job takes max over ranks of that time plus the wire time of the all-gathers, which this
box cannot measure: it is added from the bytes each rank receives at a stated xGMI rate.  Results are meaningless
numerically (other ranks' operators are copies of this rank's), only the time is read.

    python tools/scaling_model.py [cfg2 cfg4 cfg5] [--ranks all|first]  ->  JSON on stdout

cfg2: T 3000, p 3072, 20 alphas, 80 000 voxels in total (north_star's strong-scaling job) + the weak-scaled job;
cfg4: Narratives-like T 2226, p 3072, 20 alphas, 200 000 voxels sharded over the ranks (BASELINE configs[3]);
cfg5: Whisper-like T 3000, p 7680 (1280 x 6 delays), 32 alphas, 80 000 voxels (configs[4]; the band scales only rescale
the design: the fit is the same work).
cfg3: the LeBel-style story pipeline (BASELINE configs[2], the config north_star's target sentence names), HOST TO HOST
through harness.StoryPipeline.fit_words: 26 + 1 stories, T ~ 9 200 + 291, p 3072, 10 alphas, single_alpha, 80 000 voxels;
a rank stages, z-scores and uploads only its V / G columns of every story (its share of the link), with the staging
threads a rank of G gets on this box (cores / 2 / G), and downloads its block of the weights.  What the model cannot see:
G ranks' staging threads competing for the host's memory bandwidth at the same time.
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
from litcoder_core_amd import NestedCVModel, ShardContext, ops, series  # noqa: E402
from litcoder_core_amd.dist import shard_bounds  # noqa: E402
from litcoder_core_amd.engine.common import FitOptions  # noqa: E402

# A simulated rank computes with copies of ITS OWN operators where the other ranks' would be, so its voxels choose arbitrary
# alphas -- the grid's smallest among them, which no voxel of these targets chooses in a real fit and which (below
# FitOptions.refit_ahead_min_alpha) is formed only once chosen.  The simulated ranks therefore keep EVERY factorised alpha
# ahead, as all fits did until round 5: an upper bound on what a real rank does.
SIM_OPTIONS = dict(refit_ahead_min_alpha=0.0)

CONFIGS = {
    "cfg2": dict(T=3000, F0=768, DELAYS=[1, 2, 3, 4], A=20, V_total=80000, weak=True),
    "cfg4": dict(T=2226, F0=768, DELAYS=[1, 2, 3, 4], A=20, V_total=200000, weak=False),
    "cfg5": dict(T=3000, F0=1280, DELAYS=[1, 2, 3, 4, 5, 6], A=32, V_total=80000, weak=False),
}
which = [a for a in sys.argv[1:] if a in CONFIGS or a == "cfg3"] or ["cfg2"]
all_ranks = "--ranks" in sys.argv and sys.argv[sys.argv.index("--ranks") + 1] == "all"
XGMI_ALLGATHER_GBPS = 300.0          # assumed all-gather rate INTO one rank (7 links x ~153 GB/s peak; RCCL ~1/3)
dev = ops.device(0)
result = {"assumed_allgather_GBps_into_a_rank": XGMI_ALLGATHER_GBPS, "configs": {}}


def timed(fit, n):
    fit(); fit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fit()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def cfg3_model(V_total=80000):
    """BASELINE configs[2] at 1 / 2 / 4 / 8 simulated ranks, host to host (see the module docstring)."""
    from litcoder_core_amd import StoryPipeline
    words, wtimes, trtimes, brain = bench.synth_stories(V_total, dev)
    out = {"shape": dict(stories=len(words), V_total=V_total, alphas=10, single_alpha=True,
                         host_brain_bytes_float64=int(sum(b.nbytes for b in brain.values()))), "per_world": {}}
    t1 = None
    cores = os.cpu_count() or 4
    for G in (1, 2, 4, 8):
        per_rank, wire, info = {}, {}, {}
        os.environ["LITCODER_AMD_UPLOAD_THREADS"] = str(max(2, min(24, cores // 2 // G)))
        for r in (range(G) if all_ranks else sorted({0, G - 1})):
            shard = ShardContext.simulated(G, r, device=dev, global_lists=False) if G > 1 else None
            model = NestedCVModel("ridge_regression", shard=shard, options=FitOptions(**SIM_OPTIONS) if G > 1 else None)
            pipe = StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM, model=model)

            def fit():
                pipe.fit_words(words, wtimes, trtimes, brain, **bench.CFG3_KW)
                torch.cuda.synchronize()

            per_rank[r] = timed(fit, 5)
            b0 = shard.bytes_received if shard is not None else 0
            fit()
            wire[r] = (shard.bytes_received - b0) if shard is not None else 0
            info[r] = {k: model.last_fit.get(k) for k in ("form", "precision", "single_alpha_guess", "panels")}
        t = max(per_rank.values())
        rx = max(wire.values())
        wire_ms = 1e3 * rx / (XGMI_ALLGATHER_GBPS * 1e9)
        if G == 1:
            t1 = t
        lo, hi = shard_bounds(V_total, G, 0)
        out["per_world"][G] = {"ms_per_rank_alone": {str(k): round(v, 2) for k, v in per_rank.items()}, "max_ms": round(t, 2),
                               "staging_threads_per_rank": int(os.environ["LITCODER_AMD_UPLOAD_THREADS"]),
                               "allgather_bytes_received": int(rx), "wire_ms_if_not_hidden": round(wire_ms, 2),
                               "predicted_ms": round(t + wire_ms, 2), "predicted_speedup": round(t1 / (t + wire_ms), 2),
                               "voxels_per_sec": round(V_total / (1e-3 * (t + wire_ms))),
                               "rank0_link_bytes": {"up_float32": int(sum(b.shape[0] for b in brain.values()) * (hi - lo) * 4),
                                                    "down": int(3072 * (hi - lo) * 4)},
                               "rank0_fit": info[0]}
        print(f"cfg3 G={G}: {out['per_world'][G]}", file=sys.stderr, flush=True)
    os.environ.pop("LITCODER_AMD_UPLOAD_THREADS", None)
    return out


for name in which:
    if name == "cfg3":
        result["configs"][name] = cfg3_model()
        continue
    c = CONFIGS[name]
    T, V_total = c["T"], c["V_total"]
    alphas = np.logspace(-1, 8, c["A"])
    p = c["F0"] * len(c["DELAYS"])
    n_o = T - T // bench.N_OUTER
    n_v = n_o // bench.N_INNER
    n_i = n_o - n_v
    n_cho = sum(1 for a in alphas if series.residual_bound(float(a), 4) > 2e-9)       # alphas that need a factorisation
    out = {"shape": dict(T=T, p=p, alphas=c["A"], factorised_alphas=n_cho, V_total=V_total), "per_world": {}}
    t1 = None
    for G in (1, 2, 4, 8):
        ranks = range(G) if all_ranks else sorted({0, G // 2, G - 1})
        per_rank = {}
        for r in ranks:
            lo, hi = shard_bounds(V_total, G, r)
            dX, dY, p_ = bench.synth_inputs(hi - lo, r, dev, T=T, F0=c["F0"], DELAYS=c["DELAYS"])
            shard = ShardContext.simulated(G, r, device=dev, global_lists=False) if G > 1 else None
            model = NestedCVModel("ridge_regression", shard=shard, options=FitOptions(**SIM_OPTIONS) if G > 1 else None)
            per_rank[r] = timed(lambda: model.fit_predict_device(dX, dY, p_, hi - lo, n_voxels_total=V_total, alphas=alphas,
                                                                 **bench.FIT_KW), 4)
            del dX, dY
            torch.cuda.empty_cache()
        # bytes a rank RECEIVES per fit: hat matrices of the inner folds (the factorised alphas x 5 inner folds per outer
        # fold), refit operators / inverses (the factorised alphas x (p_pad + pad(n_t)) rows), (G - 1) / G of each
        hat = bench.N_OUTER * bench.N_INNER * n_cho * ops.pad_to(n_v, 32) * ops.pad_to(n_i, 64) * 4
        refit = bench.N_OUTER * n_cho * min(ops.pad_to(n_o, 64), ops.pad_to(p, 32) + ops.pad_to(T - n_o, 32)) * ops.pad_to(n_o, 64) * 4
        wire_ms = 0.0 if G == 1 else 1e3 * (hat + refit) * (G - 1) / G / (XGMI_ALLGATHER_GBPS * 1e9)
        t = max(per_rank.values())
        if G == 1:
            t1 = t
            out["alphas_in_use_last_fold_1gpu"] = model.last_fit.get("used_all")
        out["per_world"][G] = {"ms_per_rank_alone": {str(k): round(v, 2) for k, v in per_rank.items()}, "max_ms": round(t, 2),
                               "allgather_bytes_received": int((hat + refit) * (G - 1) / G),
                               "wire_ms_if_not_hidden": round(wire_ms, 2), "predicted_ms": round(t + wire_ms, 2),
                               "predicted_speedup": round(t1 / (t + wire_ms), 2),
                               "voxels_per_sec": round(V_total / (1e-3 * (t + wire_ms)))}
        print(f"{name} G={G}: {out['per_world'][G]}", file=sys.stderr, flush=True)
    if c["weak"]:
        # weak scaling: every rank V_total voxels (bench.py's default mode): rank 0 of G alone on its V_total of V_total G
        out["weak"] = {}
        for G in (2, 4, 8):
            dX, dY, p_ = bench.synth_inputs(V_total, 0, dev, T=T, F0=c["F0"], DELAYS=c["DELAYS"])
            model = NestedCVModel("ridge_regression", shard=ShardContext.simulated(G, 0, device=dev, global_lists=False),
                                  options=FitOptions(**SIM_OPTIONS))
            ms = timed(lambda: model.fit_predict_device(dX, dY, p_, V_total, n_voxels_total=V_total * G, alphas=alphas,
                                                        **bench.FIT_KW), 4)
            out["weak"][G] = {"ms_rank0_alone": round(ms, 1), "voxels_per_sec": round(V_total * G / (1e-3 * ms)),
                              "speedup_vs_1gpu": round(V_total * G / (1e-3 * ms) / (V_total / (1e-3 * t1)), 2)}
            print(f"{name} weak G={G}: {out['weak'][G]}", file=sys.stderr, flush=True)
            del dX, dY
            torch.cuda.empty_cache()
    result["configs"][name] = out
print(json.dumps(result))
