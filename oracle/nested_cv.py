"""Oracle: nested-CV ridge fit, both modes (test infrastructure).

Follows ``encoding/models/nested_cv.py:18-616``.  fp32 torch-CPU for the fit,
float64 numpy/scipy for the statistics tail, exactly where the reference
switches (``.cpu().numpy()`` at :151/:251).
"""
import numpy as np
import torch

from . import folds as _folds
from . import ridge as _ridge
from . import stats as _stats


def select_alphas(X, Y, splits, alphas, single_alpha, normalpha, use_corr, singcutoff):
    """nested_cv.py:334-415: score every alpha on every inner fold, average over
    folds, then argmax per voxel (first maximum wins) or -- ``single_alpha`` --
    argmax of the across-voxel mean.  Returns (alpha tensor (V,), (A,V) means).

    dtype quirk kept: per-voxel alphas are an fp32 tensor (:409-411); the
    single-alpha tensor takes torch's default for the element type of
    ``alphas`` (fp64 for numpy floats, fp32 for Python floats) (:401-403)."""
    per_fold = []
    for tr, va in splits:
        per_fold.append(_ridge.alpha_sweep_scores(
            X[tr], X[va], Y[tr], Y[va], alphas,
            singcutoff=singcutoff, use_corr=use_corr, normalpha=normalpha))
    mean_scores = torch.stack(per_fold).mean(dim=0)
    if single_alpha:
        k = torch.argmax(mean_scores.mean(dim=1))
        chosen = torch.tensor([alphas[k]] * Y.shape[1])
    else:
        ks = torch.argmax(mean_scores, dim=0)
        chosen = torch.tensor([alphas[i] for i in ks], dtype=torch.float32)
    return chosen, mean_scores


def _score(Xte, Yte, W, pearson):
    pred = (Xte @ W).numpy()
    return pearson(Yte.numpy(), pred)


def _sig_block(metrics, scores, mask, n, tag):
    if n > 0:
        s = np.asarray(scores)[mask]
        metrics.update({f"median_{tag}_score": float(np.median(s)), f"mean_{tag}_score": float(np.mean(s)),
                        f"min_{tag}_score": float(np.min(s)), f"max_{tag}_score": float(np.max(s))})


def fit_predict(features, targets, X_test=None, y_test=None, groups=None, folding_type="chunked",
                n_outer_folds=5, n_inner_folds=5, chunk_length=20, alphas=None, alpha_fdr=0.05,
                single_alpha=False, normalpha=True, use_corr=True, normalize_features=False,
                normalize_targets=False, singcutoff=1e-10, pearson=None, fdr=None, detail=None):
    """nested_cv.py:18-331.  ``pearson``/``fdr`` are injectable (defaults: the
    scipy loop and ``stats.bh_fdr``); ``detail`` (a dict) receives per-fold
    intermediates for tests.  Returns (metrics, weights, best_alphas) with the
    reference's types (lists vs arrays, fp32 vs fp64) preserved."""
    pearson = pearson or _stats.pearson_per_voxel
    fdr = fdr or _stats.bh_fdr
    if alphas is None:
        alphas = np.logspace(-1, 8, 10)
    X = torch.tensor(features, dtype=torch.float32)
    Y = torch.tensor(targets, dtype=torch.float32)
    want_norm = normalize_features or normalize_targets

    if X_test is not None and y_test is not None:
        # ---- train/test mode, :105-171 ----
        Xte = torch.tensor(X_test, dtype=torch.float32)
        Yte = torch.tensor(y_test, dtype=torch.float32)
        if want_norm:
            nz = _ridge.TrainStatNormalizer(normalize_features, normalize_targets).fit(X, Y)
            X, Y = nz.apply(X, Y)
            Xte, Yte = nz.apply(Xte, Yte)
        # :130-132 -- ``groups`` is passed as the 5th POSITIONAL = trim_size
        splits = _folds.create_folds(len(features), folding_type, n_inner_folds, chunk_length, groups)
        chosen, mean_scores = select_alphas(X, Y, splits, alphas, single_alpha, normalpha, use_corr, singcutoff)
        W = _ridge.ridge_weights(X, Y, chosen, normalpha=normalpha, singcutoff=singcutoff)
        corrs, pvals = _score(Xte, Yte, W, pearson)
        sig, padj = fdr(pvals, alpha=alpha_fdr)
        nsig = np.sum(sig)
        s = _stats.summary(corrs)
        metrics = {
            "median_score": s["median"], "mean_score": s["mean"], "std_score": s["std"],
            "min_score": s["min"], "max_score": s["max"],
            "best_alphas": chosen.numpy().tolist(), "correlations": corrs, "p_values": pvals,
            "corrected_p_values": padj.tolist(), "significant_mask": sig.tolist(),
            "n_significant": int(nsig), "percent_significant": float(nsig / len(corrs) * 100),
        }
        _sig_block(metrics, corrs, sig, nsig, "significant")
        if detail is not None:
            detail.update(mean_scores=mean_scores.numpy(), splits=splits)
        return metrics, W.numpy(), chosen.numpy()

    # ---- full nested CV, :173-331 ----
    if groups is not None and folding_type == "group":
        outer = _folds.create_folds(len(features), "group", n_outer_folds, groups=groups)
    else:
        outer = _folds.create_folds(len(features), folding_type, n_outer_folds, chunk_length, groups)
    f_scores, f_p, f_alpha, f_sig, f_W, f_means, f_inner = [], [], [], [], [], [], []
    for tr, te in outer:
        Xtr, Xte, Ytr, Yte = X[tr], X[te], Y[tr], Y[te]
        if want_norm:
            nz = _ridge.TrainStatNormalizer(normalize_features, normalize_targets).fit(Xtr, Ytr)
            Xtr, Ytr = nz.apply(Xtr, Ytr)
            Xte, Yte = nz.apply(Xte, Yte)
        if groups is not None and folding_type == "group":
            inner = _folds.create_folds(len(tr), "group", n_inner_folds, groups=[groups[i] for i in tr])
        else:
            inner = _folds.create_folds(len(tr), folding_type, n_inner_folds, chunk_length)
        chosen, mean_scores = select_alphas(Xtr, Ytr, inner, alphas, single_alpha, normalpha, use_corr, singcutoff)
        f_alpha.append(chosen.numpy())
        W = _ridge.ridge_weights(Xtr, Ytr, chosen, normalpha=normalpha, singcutoff=singcutoff)
        f_W.append(W.numpy())
        corrs, pvals = _score(Xte, Yte, W, pearson)
        f_scores.append(corrs)
        f_p.append(pvals)
        f_sig.append(fdr(pvals, alpha=alpha_fdr)[0])
        f_means.append(mean_scores.numpy())
        f_inner.append(inner)
    scores = np.mean(f_scores, axis=0)
    pcomb = _stats.fisher_combine(f_p)
    sig, padj = fdr(pcomb, alpha=alpha_fdr)
    nsig = np.sum(sig)
    major = np.sum(f_sig, axis=0) >= (n_outer_folds // 2 + 1)
    nmajor = np.sum(major)
    mean_alpha = np.mean(f_alpha, axis=0)
    mean_W = np.mean(f_W, axis=0)
    s = _stats.summary(scores)
    metrics = {
        "median_score": s["median"], "mean_score": s["mean"], "std_score": s["std"],
        "min_score": s["min"], "max_score": s["max"],
        "best_alphas": mean_alpha.tolist(), "correlations": scores.tolist(), "p_values": pcomb.tolist(),
        "corrected_p_values": padj.tolist(), "significant_mask": sig.tolist(),
        "majority_significant_mask": major.tolist(),
        "n_significant": int(nsig), "n_majority_significant": int(nmajor),
        "percent_significant": float(nsig / len(scores) * 100),
        "percent_majority_significant": float(nmajor / len(scores) * 100),
    }
    _sig_block(metrics, scores, sig, nsig, "significant")
    _sig_block(metrics, scores, major, nmajor, "majority_significant")
    if detail is not None:
        detail.update(outer=outer, inner=f_inner, fold_scores=np.asarray(f_scores), fold_pvalues=np.asarray(f_p),
                      fold_alphas=np.asarray(f_alpha), fold_mean_scores=np.asarray(f_means))
    return metrics, mean_W, mean_alpha
