"""Oracle: per-voxel test statistics (test infrastructure).

Follows ``encoding/models/nested_cv.py:418-477`` and the three
``fdrcorrection`` call sites (``:158,:263,:282``).
"""
import numpy as np
from scipy.stats import combine_pvalues, pearsonr


def pearson_per_voxel(y_true, y_pred):
    """nested_cv.py:418-438: scipy ``pearsonr`` column by column (the reference's
    Python loop, kept as a loop so the CPU baseline pays what the reference
    pays); NaN r -> 0.0, NaN p -> 1.0.  Returns two Python lists."""
    import warnings
    rs, ps = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for v in range(y_true.shape[1]):
            r, p = pearsonr(y_true[:, v], y_pred[:, v])
            rs.append(0.0 if np.isnan(r) else r)
            ps.append(1.0 if np.isnan(p) else p)
    return rs, ps


def fisher_combine(fold_pvalues):
    """nested_cv.py:441-477: Fisher's method per voxel over the outer folds
    (``-2*sum(ln p) ~ chi2(2k)``); a voxel whose p-values are all exactly 1.0 is
    1.0 without calling scipy; on any scipy exception the max p is used."""
    out = []
    for v in range(len(fold_pvalues[0])):
        pv = [f[v] for f in fold_pvalues]
        if all(p == 1.0 for p in pv):
            out.append(1.0)
            continue
        try:
            out.append(combine_pvalues(pv, method="fisher")[1])
        except Exception:
            out.append(max(pv))
    return np.array(out)


def bh_fdr(pvals, alpha=0.05):
    """Benjamini-Hochberg as ``statsmodels.stats.multitest.fdrcorrection(pvals,
    alpha, method='indep')`` (statsmodels 0.14.4, not installed here -> this
    step is KAT-pinned only, "parity unpinned"): sort p; reject every rank up
    to the largest i with p_(i) <= i/n*alpha; adjusted p = running minimum from
    the right of p_(i)*n/i, clipped to 1; results returned in input order."""
    p = np.asarray(pvals, dtype=np.float64)
    n = p.size
    order = np.argsort(p)
    ps = p[order]
    frac = np.arange(1, n + 1) / float(n)
    rej = ps <= frac * alpha
    if rej.any():
        rej[: np.nonzero(rej)[0].max()] = True
    adj = np.minimum.accumulate((ps / frac)[::-1])[::-1]
    adj[adj > 1] = 1
    rej_out = np.empty(n, dtype=bool)
    adj_out = np.empty(n, dtype=np.float64)
    rej_out[order] = rej
    adj_out[order] = adj
    return rej_out, adj_out


def summary(scores):
    s = np.asarray(scores)
    return {"median": float(np.median(s)), "mean": float(np.mean(s)), "std": float(np.std(s)),
            "min": float(np.min(s)), "max": float(np.max(s))}
