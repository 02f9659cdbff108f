#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE
(/root/reference, imported unmodified through tests/golden/_refimport.py) on
small seeded inputs.  Run in the build container only:

    python tests/golden/make_golden.py

Each .npz holds the inputs and the reference's outputs (data only -- no
reference source).  Library versions in the container when these were made:
numpy 2.2.6, scipy 1.15.3, torch 2.10.0 (CPU), sklearn 1.7.2.  The BH-FDR step
inside the reference runs through oracle.stats.bh_fdr (statsmodels is absent),
so ``*significant*`` / ``corrected_p_values`` entries are NOT an independent
pin of that step.
"""
import contextlib
import io
import json
import logging
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refimport  # noqa: E402
from oracle import stats as ostats  # noqa: E402

ref = _refimport.load(ostats.bh_fdr)
logging.disable(logging.CRITICAL)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ---------------------------------------------------------------- FIR
def gen_fir():
    rng = np.random.default_rng(11)
    out = {}
    cases = [
        ("a", rng.standard_normal((37, 5)), [1, 2, 3, 4], False),
        ("b", rng.standard_normal((37, 5)), [-1, 0, 2], True),
        ("c", rng.standard_normal((20, 3)).astype(np.float32), [0, 3, -2, 25, -20], False),
        ("d", rng.integers(-50, 50, size=(16, 4)).astype(np.int64), [2, -3], True),
        ("e", rng.standard_normal((12, 7)).astype(np.float32), [0, 0], False),
        ("f", rng.standard_normal((9, 2)), list(range(1, 5)), True),
        ("g", rng.standard_normal((6, 2)), [6, -6, 7], True),
    ]
    for tag, stim, delays, circ in cases:
        out[f"{tag}_stim"] = stim
        out[f"{tag}_delays"] = np.array(delays, dtype=np.int64)
        out[f"{tag}_circpad"] = np.array(circ)
        out[f"{tag}_out"] = ref.FIR_expander.FIR.make_delayed(stim, delays, circpad=circ)
    save("fir.npz", **out)


# ---------------------------------------------------------------- downsampling
def gen_downsample():
    rng = np.random.default_rng(12)
    out = {}
    with np.errstate(all="ignore"):
        out["kat_t"] = np.array([0, 2, 6, 6.0001, -6, 1.0])
        out["kat_val"] = ref.interpdata.lanczosfun(0.5, out["kat_t"].copy(), 3)
    ds = ref.downsampling.Downsampler()
    n_old, D, n_new = 260, 12, 40
    oldtime = np.sort(rng.uniform(0, 80, n_old))
    oldtime[5] = 7.0           # an exact hit on a TR time -> the t==0 branch
    oldtime.sort()
    newtime = 1.0 + 2.0 * np.arange(n_new)
    data = rng.standard_normal((n_old, D))
    out.update(oldtime=oldtime, newtime=newtime, data=data, data_f32=data.astype(np.float32))
    with np.errstate(all="ignore"):
        out["lanczos_w3"] = quiet(ds.downsample, data, oldtime, newtime, method="lanczos", window=3, cutoff_mult=1.0,
                                  split_indices=[1, 2])
        out["lanczos_w2_c05"] = quiet(ds.downsample, data, oldtime, newtime, method="lanczos", window=2,
                                      cutoff_mult=0.5)
        out["lanczos_w3_rect"] = quiet(ds.downsample, data, oldtime, newtime, method="lanczos", window=3,
                                       cutoff_mult=1.0, rectify=True)
        out["lanczos_f32"] = quiet(ds.downsample, data.astype(np.float32), oldtime, newtime, method="lanczos",
                                   window=3, cutoff_mult=1.0)
        out["sinc_w3"] = quiet(ds.downsample, data, oldtime, newtime, method="sinc", window=3, cutoff_mult=1.0)
    out["rect"] = quiet(ds.downsample, data, oldtime, newtime)
    labels = np.minimum((oldtime // 2).astype(int), n_new - 1)
    labels[labels == 3] = 4    # leave one TR empty
    out["labels"] = labels
    for m in ("average", "sum", "last"):
        out[m] = quiet(ds.downsample, data, oldtime, newtime, method=m, split_indices=list(labels))
    bounds = np.array([10, 10, 30, 75, 200])
    out["bounds"] = bounds
    for m in ("legacy_average", "legacy_sum", "legacy_last"):
        out[m] = quiet(ds.downsample, data, oldtime, newtime, method=m, split_indices=bounds)
    save("downsample.npz", **out)


# ---------------------------------------------------------------- folds
def gen_folds():
    cases = []
    specs = [
        (3000, "kfold", 5, 20, None), (3000, "kfold_trimmed", 5, 20, None), (3000, "chunked_trimmed", 5, 20, None),
        (3000, "timeseries", 5, 20, None), (3005, "chunked_contiguous", 5, 20, None), (3005, "chunked", 5, 20, None),
        (247, "chunked", 4, 20, None), (247, "chunked_trimmed", 3, 20, 3), (50, "chunked", 5, 20, None),
        (50, "chunked_trimmed", 5, 20, None), (103, "kfold", 4, None, None), (103, "kfold_trimmed", 4, None, 30),
        (64, "timeseries", 3, None, None),
    ]
    for i, (n, ft, k, cl, trim) in enumerate(specs):
        random.seed(100 + i)
        np.random.seed(100 + i)
        sp = quiet(ref.folding.create_folds, n, ft, k, cl, trim)
        cases.append({"n": n, "fold_type": ft, "n_folds": k, "chunk_length": cl, "trim_size": trim, "seed": 100 + i,
                      "splits": [[list(map(int, a)), list(map(int, b))] for a, b in sp]})
    rng = np.random.default_rng(5)
    groups = rng.integers(0, 7, size=90)
    sp = quiet(ref.folding.create_folds, 90, "group", 3, None, None, groups)
    cases.append({"n": 90, "fold_type": "group", "n_folds": 3, "chunk_length": None, "trim_size": None, "seed": None,
                  "groups": groups.tolist(), "splits": [[list(map(int, a)), list(map(int, b))] for a, b in sp]})
    with open(os.path.join(HERE, "folds.json"), "w") as f:
        json.dump(cases, f)
    print("folds.json:", len(cases), "cases")


# ---------------------------------------------------------------- ridge solvers
def synth(rng, T, p, V, ar=0.0, noise=1.0, wscale=None):
    X = rng.standard_normal((T, p))
    if ar:
        for t in range(1, T):
            X[t] = ar * X[t - 1] + np.sqrt(1 - ar * ar) * X[t]
    W = rng.standard_normal((p, V)) * (wscale if wscale is not None else 1.0 / np.sqrt(p))
    Y = X @ W + noise * rng.standard_normal((T, V))
    return X, Y


def gen_ridge():
    import torch
    rng = np.random.default_rng(21)
    out = {}
    alphas = np.logspace(-1, 4, 6)
    for tag, p in (("wide", 160), ("tall", 24)):
        X, Y = synth(rng, 120, p, 48, ar=0.5)
        Y[:, 7] = 3.0           # constant voxel
        Y[:, 9] += 50.0         # large-mean voxel
        Xt = torch.tensor(X, dtype=torch.float32)
        Yt = torch.tensor(Y, dtype=torch.float32)
        tr, va = np.r_[0:60, 84:120], np.r_[60:84]
        out[f"{tag}_X"], out[f"{tag}_Y"] = X, Y
        out[f"{tag}_tr"], out[f"{tag}_va"] = tr, va
        out["alphas"] = alphas
        for uc in (True, False):
            for na in (True, False):
                r = ref.ridge_regression.ridge_corr_torch(Xt[tr], Xt[va], Yt[tr], Yt[va], alphas, singcutoff=1e-10,
                                                           use_corr=uc, normalpha=na)
                out[f"{tag}_scores_corr{int(uc)}_norm{int(na)}"] = r.numpy()
        va_al = torch.tensor(alphas[rng.integers(0, len(alphas), size=48)], dtype=torch.float32)
        out[f"{tag}_valphas"] = va_al.numpy()
        for na in (True, False):
            w = ref.ridge_regression.ridge_torch(Xt[tr], Yt[tr], va_al, singcutoff=1e-10, normalpha=na)
            out[f"{tag}_W_norm{int(na)}"] = w.numpy()
        w = ref.ridge_regression.ridge_torch(Xt[tr], Yt[tr], 2.5, singcutoff=1e-10, normalpha=True)
        out[f"{tag}_W_scalar"] = w.numpy()
        s = torch.linalg.svd(Xt[tr], full_matrices=False)[1]
        out[f"{tag}_s0"] = np.array(s[0].item())
    save("ridge.npz", **out)


# ---------------------------------------------------------------- singcutoff on a rank-deficient design
def gen_singcutoff():
    """ridge_corr_torch / ridge_torch / a full fit of the reference on a design of rank 25 < p = 40 (so the thin SVD
    carries 15 noise-level singular values) for singcutoff in {1e-30, 1e-10, 1e-6}: where the truncation
    (ridge_utils.py:44-63) acts, and how little it moves the result at the penalties the callers use."""
    import torch
    rng = np.random.default_rng(51)
    T, r, p, V = 150, 25, 40, 32
    X = rng.standard_normal((T, r)) @ rng.standard_normal((r, p)) / np.sqrt(r)
    Y = X @ (rng.standard_normal((p, V)) * 0.3) + rng.standard_normal((T, V))
    alphas = np.logspace(-1, 3, 5)
    Xt, Yt = torch.tensor(X, dtype=torch.float32), torch.tensor(Y, dtype=torch.float32)
    tr, va = np.r_[0:80, 110:150], np.r_[80:110]
    out = {"X": X, "Y": Y, "alphas": alphas, "tr": tr, "va": va, "cutoffs": np.array([1e-30, 1e-10, 1e-6])}
    S = torch.linalg.svd(Xt[tr], full_matrices=False)[1].numpy()
    out["singular_values"] = S
    model = ref.nested_cv.NestedCVModel("ridge_regression")
    for i, sc in enumerate(out["cutoffs"]):
        sc = float(sc)
        out[f"kept_{i}"] = np.array(int((S > sc).sum()))
        for na in (True, False):
            out[f"scores_{i}_norm{int(na)}"] = ref.ridge_regression.ridge_corr_torch(
                Xt[tr], Xt[va], Yt[tr], Yt[va], alphas, singcutoff=sc, use_corr=True, normalpha=na).numpy()
            out[f"W_{i}_norm{int(na)}"] = ref.ridge_regression.ridge_torch(
                Xt[tr], Yt[tr], 3.0, singcutoff=sc, normalpha=na).numpy()
        random.seed(7)
        np.random.seed(7)
        m, W, best = quiet(model.fit_predict, X, Y, alphas=alphas, use_gpu=False, folding_type="kfold",
                           n_outer_folds=3, n_inner_folds=3, singcutoff=sc)
        out[f"fit_{i}_W"], out[f"fit_{i}_alphas"] = W, best
        out[f"fit_{i}_correlations"] = np.asarray(m["correlations"])
    save("singcutoff.npz", **out)


# ---------------------------------------------------------------- full fits
def flatten_metrics(m):
    flat = {}
    for k, v in m.items():
        flat["m_" + k] = np.asarray(v)
    return flat


def gen_fits():
    rng = np.random.default_rng(31)
    model = ref.nested_cv.NestedCVModel("ridge_regression")
    X0 = rng.standard_normal((240, 24))
    Xw = ref.FIR_expander.FIR.make_delayed(X0, [1, 2, 3, 4])          # p = 96 (< n)
    Wt = rng.standard_normal((96, 64)) * 0.08
    Yw = Xw @ Wt + rng.standard_normal((240, 64))
    Yw[:, 5] = 1.25                      # constant voxel
    Yw[:, 6] = rng.standard_normal(240)  # pure-noise voxel
    X1 = rng.standard_normal((240, 80))
    Xp = ref.FIR_expander.FIR.make_delayed(X1, [1, 2, 3, 4])          # p = 320 (> n)
    Wp = rng.standard_normal((320, 64)) * 0.05
    Yp = Xp @ Wp + rng.standard_normal((240, 64))
    Xs = rng.standard_normal((240, 4))                                  # wordrate-like p = 4... x4 delays
    Xs = ref.FIR_expander.FIR.make_delayed(Xs, [1, 2, 3, 4])
    Ys = Xs @ (rng.standard_normal((16, 64)) * 0.3) + rng.standard_normal((240, 64))
    alphas = np.logspace(-1, 4, 6)
    data = {"w": (Xw, Yw), "p": (Xp, Yp), "s": (Xs, Ys)}
    cases = {
        "cv_kfold_p": ("p", dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3)),
        "cv_kfold_p_single": ("p", dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, single_alpha=True)),
        "cv_kfold_w": ("w", dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3)),
        "cv_kfoldtrim_w_r2": ("w", dict(folding_type="kfold_trimmed", n_outer_folds=3, n_inner_folds=3, use_corr=False)),
        "cv_chunked_p": ("p", dict(folding_type="chunked", n_outer_folds=3, n_inner_folds=3, chunk_length=10)),
        "cv_contig_p_norm": ("p", dict(folding_type="chunked_contiguous", n_outer_folds=3, n_inner_folds=3,
                                       chunk_length=10, normalize_features=True, normalize_targets=True)),
        "cv_kfold_p_rawalpha": ("p", dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, normalpha=False)),
        "cv_kfold_s": ("s", dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, single_alpha=True)),
        "cv_timeseries_s": ("s", dict(folding_type="timeseries", n_outer_folds=3, n_inner_folds=2)),
        "tt_kfold_p": ("p", dict(folding_type="kfold", n_inner_folds=4, _tt=True)),
        "tt_chunked_w_single": ("w", dict(folding_type="chunked", n_inner_folds=3, chunk_length=10, single_alpha=True,
                                          _tt=True)),
        "tt_kfold_w_normy": ("w", dict(folding_type="kfold", n_inner_folds=3, normalize_targets=True, _tt=True)),
    }
    out = {"alphas": alphas}
    for k, (x, y) in data.items():
        out[f"X_{k}"], out[f"Y_{k}"] = x, y
    spec = {}
    for name, (dk, kw) in cases.items():
        kw = dict(kw)
        tt = kw.pop("_tt", False)
        X, Y = data[dk]
        random.seed(7)
        np.random.seed(7)
        if tt:
            res = quiet(model.fit_predict, X[:180], Y[:180], X_test=X[180:], y_test=Y[180:], alphas=alphas,
                        use_gpu=False, **kw)
        else:
            res = quiet(model.fit_predict, X, Y, alphas=alphas, use_gpu=False, **kw)
        metrics, W, best = res
        out[f"{name}__W"] = W
        out[f"{name}__alphas"] = best
        for k2, v in flatten_metrics(metrics).items():
            out[f"{name}__{k2}"] = v
        spec[name] = {"data": dk, "train_test": tt, "kwargs": kw, "random_seed": 7,
                      "types": {"W": str(W.dtype), "alphas": str(best.dtype),
                                "corr_elem": type(metrics["correlations"][0]).__name__,
                                "p_elem": type(metrics["p_values"][0]).__name__}}
        print(name, "median", metrics["median_score"], "alphas dtype", best.dtype)
    save("fits.npz", **out)
    with open(os.path.join(HERE, "fits.json"), "w") as f:
        json.dump(spec, f, indent=1)


# ---------------------------------------------------------------- trainer-side harness
def gen_harness():
    rng = np.random.default_rng(41)
    out = {}
    stories = ["s0", "s1", "s2", "s3"]
    trimming = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0,
                "train_targets_end": None, "test_features_start": 50, "test_features_end": -5,
                "test_targets_start": 40, "test_targets_end": None}
    feats, brain = {}, {}
    for s in stories:
        nt = int(rng.integers(120, 140))
        f = rng.standard_normal((nt + 15, 3))
        f[:, 2] = 0.5                      # zero-variance column -> zs leaves it un-divided
        feats[s] = ref.FIR_expander.FIR.make_delayed(f, [1, 2, 3, 4])
        brain[s] = rng.standard_normal((nt, 10)) * 3 + 100
        out[f"feat_{s}"] = f
        out[f"brain_{s}"] = brain[s]
    T = ref.trainer.AbstractTrainer
    tr = T.__new__(T)                       # the reference's own methods, no assembly / loggers needed
    tr.trimming_config = trimming
    tr.stories_to_process = stories
    d = quiet(tr._create_train_test_split, feats, brain)
    for k in ("Rstim", "Rresp", "Pstim", "Presp"):
        out[k] = d[k]
    tr.trimming_config = {"features_start": 10, "features_end": -5, "targets_start": 3, "targets_end": -12}
    d = quiet(tr._create_concatenated_data, feats, brain)
    out["cat_X"], out["cat_Y"] = d["X"], d["Y"]
    zs = ref.utils.zs
    out["zs_in"] = rng.standard_normal((30, 6)) * np.array([1, 2, 0, 4, 5, 6.0]) + 3
    out["zs_out"] = zs(out["zs_in"].copy())
    save("harness.npz", **out)


def gen_saver():
    """The reference's ModelSaver run on a small result: directory-name hash and file contents."""
    import json
    import pickle
    import tempfile
    hp = {"fir_delays": [1, 2, 3, 4], "downsample_config": {"method": "lanczos", "window": 3, "cutoff_mult": 1.0},
          "n_outer_folds": 5, "single_alpha": True, "modality": "language_model", "layer_idx": 9, "note": None,
          "alphas": [0.1, 1.0, 10.0]}
    metrics = {"median_score": 0.25, "correlations": [np.float32(0.5), 0.0, np.float32(-0.125)],
               "best_alphas": [1.0, 1.0, 10.0], "significant_mask": [True, False, False]}
    W = np.arange(12, dtype=np.float32).reshape(4, 3)
    with tempfile.TemporaryDirectory() as d:
        sv = ref.utils.ModelSaver(base_dir=d)
        run = quiet(sv.save_encoding_model, W, np.array([1.0, 1.0, 10.0]), hp, metrics, True)
        out = {"hyperparams": hp, "dir_suffix": run.name.split("_")[-1], "files": sorted(p.name for p in run.iterdir()),
               "hyperparams_json_text": (run / "hyperparams.json").read_text(),
               "metrics_keys": sorted(pickle.load(open(run / "metrics.pkl", "rb")).keys()),
               "weights_shape": list(np.load(run / "weights.npy").shape)}
    with open(os.path.join(HERE, "saver.json"), "w") as f:
        json.dump(out, f, indent=1)


# ---------------------------------------------------------------- alpha = 0 and a singcutoff that bites
def gen_spectral():
    """The reference where the Cholesky route of the HIP path cannot follow it: alpha = 0 in the grid (ridge_regression.py:
    56,117: D = S / (S^2 + a^2) at a = 0, the pseudo-inverse of the kept directions) and a singcutoff that really drops
    singular values (ridge_utils.py:44-63).  Three designs: rank-deficient (rank 25 < p = 40; cutoffs chosen clear of
    every singular value), wide full rank (p = 320 > n: alpha = 0 interpolates the training rows), tall (p = 16 < n:
    alpha = 0 is ordinary least squares).  ridge_corr_torch / ridge_torch outputs and full fits."""
    import torch
    rng = np.random.default_rng(77)
    out = {}
    model = ref.nested_cv.NestedCVModel("ridge_regression")

    def clear_cutoff(S, rel):
        """rel * S[0], moved off any singular value by at least 2 %."""
        c = rel * S[0]
        for _ in range(50):
            if np.all(np.abs(S - c) > 0.02 * c):
                return float(c)
            c *= 1.03
        raise RuntimeError("no clear cutoff")

    # rank-deficient
    T, r, p, V = 150, 25, 40, 32
    X = rng.standard_normal((T, r)) @ rng.standard_normal((r, p)) / np.sqrt(r)
    Y = X @ (rng.standard_normal((p, V)) * 0.3) + rng.standard_normal((T, V))
    tr, va = np.r_[0:80, 110:150], np.r_[80:110]
    Xt, Yt = torch.tensor(X, dtype=torch.float32), torch.tensor(Y, dtype=torch.float32)
    S = torch.linalg.svd(Xt[tr], full_matrices=False)[1].numpy().astype(np.float64)
    cut = [clear_cutoff(S, 1e-3), clear_cutoff(S, 0.3)]
    al = np.array([0.0, 0.1, 1.0, 10.0, 100.0])
    out.update(rd_X=X, rd_Y=Y, rd_tr=tr, rd_va=va, rd_alphas=al, rd_cutoffs=np.array(cut), rd_singular_values=S)
    for i, sc in enumerate(cut):
        out[f"rd_kept_{i}"] = np.array(int((S > sc).sum()))
        for na in (True, False):
            out[f"rd_scores_{i}_norm{int(na)}"] = ref.ridge_regression.ridge_corr_torch(
                Xt[tr], Xt[va], Yt[tr], Yt[va], al, singcutoff=sc, use_corr=True, normalpha=na).numpy()
            for a_ in (0.0, 3.0):
                out[f"rd_W_{i}_norm{int(na)}_a{int(a_)}"] = ref.ridge_regression.ridge_torch(
                    Xt[tr], Yt[tr], a_, singcutoff=sc, normalpha=na).numpy()
        for single in (False, True):
            random.seed(7)
            np.random.seed(7)
            m, W, best = quiet(model.fit_predict, X, Y, alphas=al, use_gpu=False, folding_type="kfold", n_outer_folds=3,
                               n_inner_folds=3, singcutoff=sc, single_alpha=single)
            tag = f"rd_fit_{i}_s{int(single)}"
            out[tag + "_W"], out[tag + "_alphas"], out[tag + "_correlations"] = W, best, np.asarray(m["correlations"])
    # wide, full rank: alpha = 0 with the default cutoff
    Xw = rng.standard_normal((200, 320))
    Yw = Xw @ (rng.standard_normal((320, 24)) * 0.05) + rng.standard_normal((200, 24))
    # tall: alpha = 0 is OLS
    Xs = rng.standard_normal((240, 16))
    Ys = Xs @ (rng.standard_normal((16, 24)) * 0.3) + rng.standard_normal((240, 24))
    al0 = np.array([0.0, 0.5, 5.0, 50.0])
    for tag, (Xc, Yc) in (("wide", (Xw, Yw)), ("tall", (Xs, Ys))):
        out.update({f"{tag}_X": Xc, f"{tag}_Y": Yc, f"{tag}_alphas": al0})
        n = len(Xc)
        tr_, va_ = np.arange(0, n - 40), np.arange(n - 40, n)
        Xt_, Yt_ = torch.tensor(Xc, dtype=torch.float32), torch.tensor(Yc, dtype=torch.float32)
        out[f"{tag}_tr"], out[f"{tag}_va"] = tr_, va_
        for na in (True, False):
            out[f"{tag}_scores_norm{int(na)}"] = ref.ridge_regression.ridge_corr_torch(
                Xt_[tr_], Xt_[va_], Yt_[tr_], Yt_[va_], al0, singcutoff=1e-10, use_corr=True, normalpha=na).numpy()
            out[f"{tag}_scores_r2_norm{int(na)}"] = ref.ridge_regression.ridge_corr_torch(
                Xt_[tr_], Xt_[va_], Yt_[tr_], Yt_[va_], al0, singcutoff=1e-10, use_corr=False, normalpha=na).numpy()
        out[f"{tag}_W_a0"] = ref.ridge_regression.ridge_torch(Xt_[tr_], Yt_[tr_], 0.0, singcutoff=1e-10, normalpha=True).numpy()
        random.seed(7)
        np.random.seed(7)
        m, W, best = quiet(model.fit_predict, Xc, Yc, alphas=al0, use_gpu=False, folding_type="kfold", n_outer_folds=3,
                           n_inner_folds=3, singcutoff=1e-10)
        out[f"{tag}_fit_W"], out[f"{tag}_fit_alphas"] = W, best
        out[f"{tag}_fit_correlations"] = np.asarray(m["correlations"])
        m, W, best = quiet(model.fit_predict, Xc[:160], Yc[:160], X_test=Xc[160:], y_test=Yc[160:], alphas=al0,
                           use_gpu=False, folding_type="kfold", n_inner_folds=3, singcutoff=1e-10)
        out[f"{tag}_tt_W"], out[f"{tag}_tt_alphas"] = W, best
        out[f"{tag}_tt_correlations"] = np.asarray(m["correlations"])
    save("spectral.npz", **out)


if __name__ == "__main__":
    gen_saver()
    gen_fir()
    gen_downsample()
    gen_folds()
    gen_ridge()
    gen_fits()
    gen_harness()
    gen_singcutoff()
    gen_spectral()
