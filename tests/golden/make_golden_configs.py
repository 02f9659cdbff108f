#!/usr/bin/env python3
"""Config-shape fixtures: the REFERENCE itself (/root/reference, imported unmodified through _refimport.py) run in the
build container on BASELINE.json's configs at their own shapes, 256 voxels each.  Minutes of CPU (30 SVDs per config):

    python tests/golden/make_golden_configs.py [cfg2 cfg3 cfg4 cfg5 cfg2_r2 cfg2_single]

Inputs are rebuilt from seeds (tests/_config_problems.py), so ``configs.npz`` stores only fingerprints of the inputs and
the reference's OUTPUTS: per outer fold the chosen alphas, the fold-mean inner-CV score table (what a differing alpha
is proven a near-tie against), the test correlations; the mean alphas / correlations; 32 columns of the mean weights.
Per-fold intermediates are captured by wrapping three module-level functions of the imported reference in memory
(ridge_corr_torch, _find_best_alphas, _calculate_correlations_pvalues) -- call-through recorders, nothing is changed.
cfg3 additionally runs the reference's Downsampler (Lanczos), FIR and AbstractTrainer._create_train_test_split on the
synthetic stories and keeps samples of the structured matrices.  Data only: no reference source text is stored.
"""
import contextlib
import io
import json
import logging
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refimport  # noqa: E402
import _config_problems as cp  # noqa: E402
from oracle import stats as ostats  # noqa: E402

ref = _refimport.load(ostats.bh_fdr)
logging.disable(logging.CRITICAL)
W_COLS = 32


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


class Recorder:
    """Call-through wrappers around the reference's inner-CV sweep, alpha choice and per-fold scoring."""

    def __enter__(self):
        ncv = ref.nested_cv
        self.saved = (ncv.ridge_corr_torch, ncv._find_best_alphas, ncv._calculate_correlations_pvalues)
        self.sweeps, self.fold_alphas, self.fold_tables, self.fold_r = [], [], [], []
        sweep, best, pear = self.saved

        def rec_sweep(*a, **k):
            out = sweep(*a, **k)
            self.sweeps.append(out.detach().clone())
            return out

        def rec_best(*a, **k):
            n0 = len(self.sweeps)
            out = best(*a, **k)
            self.fold_alphas.append(out.numpy().copy())
            self.fold_tables.append(torch.stack(self.sweeps[n0:]).mean(dim=0).numpy())      # nested_cv.py:391-393
            del self.sweeps[n0:]
            return out

        def rec_pear(*a, **k):
            out = pear(*a, **k)
            self.fold_r.append(np.asarray(out[0], dtype=np.float32))
            return out

        ncv.ridge_corr_torch, ncv._find_best_alphas, ncv._calculate_correlations_pvalues = rec_sweep, rec_best, rec_pear
        return self

    def __exit__(self, *exc):
        ncv = ref.nested_cv
        ncv.ridge_corr_torch, ncv._find_best_alphas, ncv._calculate_correlations_pvalues = self.saved


def run_fit(tag, out, spec, args, kwargs):
    model = ref.nested_cv.NestedCVModel("ridge_regression")
    random.seed(7)
    np.random.seed(7)
    t0 = time.time()
    with Recorder() as rec:
        metrics, W, best = quiet(model.fit_predict, *args, use_gpu=False, **kwargs)
    out[f"{tag}__fold_alphas"] = np.stack(rec.fold_alphas)
    out[f"{tag}__fold_tables"] = np.stack(rec.fold_tables).astype(np.float32)
    out[f"{tag}__fold_r"] = np.stack(rec.fold_r)
    out[f"{tag}__W"] = np.ascontiguousarray(W[:, :W_COLS])
    out[f"{tag}__alphas"] = np.asarray(best)
    out[f"{tag}__correlations"] = np.asarray(metrics["correlations"])
    spec[tag] = {"median_score": float(metrics["median_score"]), "seconds_reference_cpu": round(time.time() - t0, 1),
                 "alphas_dtype": str(np.asarray(best).dtype), "w_cols": W_COLS,
                 "distinct_alphas_per_fold": [int(len(np.unique(a))) for a in rec.fold_alphas]}
    print(tag, spec[tag], flush=True)


VARIANTS = {"r2": dict(use_corr=False), "single": dict(single_alpha=True)}     # cfg2_r2, cfg2_single: round 6


def gen_matrix(name, out, spec):
    """``name``: a config of _config_problems.CONFIGS, or config_variant (VARIANTS: the same inputs, another scoring rule /
    alpha rule: ridge_regression.py:126-130, nested_cv.py:396-403)."""
    base, _, var = name.partition("_")
    X, Y, kw = cp.matrix_problem(base)
    if var:
        kw.update(VARIANTS[var])
    out[f"{name}__checks"] = cp.checks(X, Y)
    run_fit(name, out, spec, (X, Y), kw)


def gen_cfg3(out, spec):
    pr = cp.story_problem()
    names = list(pr["words"])
    ds = ref.downsampling.Downsampler()
    feats = {}
    for s in names:                                     # trainer.py:174-209: downsample, then FIR, per story
        with np.errstate(all="ignore"):
            d = quiet(ds.downsample, pr["words"][s], pr["wtimes"][s], pr["trtimes"][s], method="lanczos", window=3,
                      cutoff_mult=1.0, split_indices=None)
        feats[s] = ref.FIR_expander.FIR.make_delayed(d, pr["delays"])
    T = ref.trainer.AbstractTrainer
    tr = T.__new__(T)                                   # the reference's own method, no assembly / loggers needed
    tr.trimming_config = pr["trimming"]
    tr.stories_to_process = names
    mats = quiet(tr._create_train_test_split, feats, pr["brain"])
    out["cfg3__checks"] = cp.checks(*[pr["brain"][s] for s in names[:3]], pr["words"][names[0]])
    out["cfg3__mats_checks"] = cp.checks(mats["Rstim"], mats["Rresp"], mats["Pstim"], mats["Presp"])
    out["cfg3__shapes"] = np.asarray([mats[k].shape for k in ("Rstim", "Rresp", "Pstim", "Presp")])
    out["cfg3__Rstim_sample"] = mats["Rstim"][::41, ::53].copy()
    out["cfg3__Rresp_sample"] = mats["Rresp"][::41, ::7].copy()
    out["cfg3__Pstim_sample"] = mats["Pstim"][::11, ::53].copy()
    out["cfg3__Presp_sample"] = mats["Presp"][::11, ::7].copy()
    args = (mats["Rstim"], mats["Rresp"])
    for tag, single in (("cfg3s", True), ("cfg3v", False)):
        run_fit(tag, out, spec, args, dict(pr["kw"], X_test=mats["Pstim"], y_test=mats["Presp"], single_alpha=single,
                                          normalpha=True, use_corr=True))


if __name__ == "__main__":
    ALL = ["cfg2", "cfg3", "cfg4", "cfg5", "cfg2_r2", "cfg2_single"]
    which = sys.argv[1:] or ALL
    path = os.path.join(HERE, "configs.npz")
    out, spec = {}, {}
    if os.path.exists(path) and len(which) < len(ALL):  # partial regeneration keeps the other configs
        out = dict(np.load(path))
        spec = json.load(open(os.path.join(HERE, "configs.json")))["fits"]
    for name in which:
        for k in [k for k in out if k.split("__")[0].rstrip("sv") == name or k.split("__")[0] == name]:
            del out[k]
        if name == "cfg3":
            gen_cfg3(out, spec)
        else:
            gen_matrix(name, out, spec)
    np.savez_compressed(path, **out)
    with open(os.path.join(HERE, "configs.json"), "w") as f:
        json.dump({"versions": {"numpy": np.__version__, "torch": torch.__version__}, "fits": spec}, f, indent=1)
    print(f"configs.npz: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")
