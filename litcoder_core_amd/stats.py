"""Host-side statistics tail of the fit (O(V log V), vectorised numpy).

Replaces the reference's three per-voxel Python loops (``nested_cv.py:433-436`` scipy
``pearsonr`` p-values, ``:455-475`` Fisher combination) and the statsmodels
``fdrcorrection`` call sites (``:158,:263,:282``) with closed forms over whole vectors,
and builds the metrics dictionaries with the reference's keys and container types
(``:480-616``).  The correlations themselves come from the device.
"""
import numpy as np

try:                                     # scipy is a convenience, not a requirement
    from scipy.special import betainc as _betainc
except Exception:                        # pragma: no cover
    _betainc = None


def _betainc_cf(a, b, x, iters=300):
    """Regularised incomplete beta I_x(a, b) by Lentz's continued fraction (numpy only)."""
    from math import lgamma
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    flip = x > (a + 1.0) / (a + b + 2.0)
    xx = np.where(flip, 1.0 - x, x)
    aa = np.where(flip, b, a) * np.ones_like(x)
    bb = np.where(flip, a, b) * np.ones_like(x)
    tiny = 1e-300
    with np.errstate(all="ignore"):
        lbeta = np.vectorize(lambda p, q: lgamma(p) + lgamma(q) - lgamma(p + q))(aa, bb)
        front = np.exp(aa * np.log(xx) + bb * np.log1p(-xx) - lbeta) / aa
        c = np.ones_like(xx)
        d = 1.0 - (aa + bb) * xx / (aa + 1.0)
        d = np.where(np.abs(d) < tiny, tiny, d)
        d = 1.0 / d
        h = d.copy()
        for m in range(1, iters + 1):
            m2 = 2 * m
            num = m * (bb - m) * xx / ((aa + m2 - 1.0) * (aa + m2))
            d = 1.0 + num * d
            d = np.where(np.abs(d) < tiny, tiny, d)
            c = 1.0 + num / c
            c = np.where(np.abs(c) < tiny, tiny, c)
            d = 1.0 / d
            h = h * d * c
            num = -(aa + m) * (aa + bb + m) * xx / ((aa + m2) * (aa + m2 + 1.0))
            d = 1.0 + num * d
            d = np.where(np.abs(d) < tiny, tiny, d)
            c = 1.0 + num / c
            c = np.where(np.abs(c) < tiny, tiny, c)
            d = 1.0 / d
            delta = d * c
            h = h * delta
            if np.all(np.abs(delta - 1.0) < 1e-16):
                break
        val = front * h
    val = np.where(xx <= 0.0, 0.0, val)
    out = np.where(flip, 1.0 - val, val)
    return np.clip(out, 0.0, 1.0)


def pearson_pvalues(r, n):
    """Two-sided p-value of Pearson r for sample size n, as scipy.stats.pearsonr computes it:
    r ~ Beta(n/2-1, n/2-1) on [-1, 1] under H0, p = 2*sf(|r|) = 2*(1 - I_x(ab, ab)), x = (|r|+1)/2.
    ``x`` is formed in r's own dtype (scipy >= 1.14 keeps float32 statistics in float32 there and only
    then evaluates the incomplete beta in float64); NaN r -> p = 1 (nested_cv.py:436); n == 2 -> 1."""
    r = np.asarray(r)
    if r.dtype not in (np.float32, np.float64):
        r = r.astype(np.float64)
    p = np.ones(r.shape, dtype=np.float64)
    if n <= 2:
        return p
    ab = n / 2.0 - 1.0
    ok = ~np.isnan(r)
    one = r.dtype.type(1)
    x = ((np.abs(np.clip(r[ok], -one, one)) + one) / r.dtype.type(2)).astype(np.float64)
    # 1 - I_x(ab, ab) == I_{1-x}(ab, ab) by symmetry; 1 - x is exact in float64
    inc = _betainc(ab, ab, 1.0 - x) if _betainc is not None else _betainc_cf(ab, ab, 1.0 - x)
    p[ok] = np.minimum(2.0 * inc, 1.0)
    return p


def fisher_combine(fold_pvalues):
    """nested_cv.py:441-477: p = chi2.sf(-2*sum(ln p_k), 2k).  For even degrees of freedom the
    survival function is the closed form exp(-L) * sum_{i<k} L^i/i!, L = -sum(ln p_k); all-ones
    rows give exactly 1.0 (the reference's shortcut)."""
    P = np.asarray(fold_pvalues, dtype=np.float64)        # (k, V)
    k = P.shape[0]
    with np.errstate(divide="ignore"):
        L = -np.sum(np.log(P), axis=0)
    term = np.ones_like(L)
    acc = np.ones_like(L)
    with np.errstate(all="ignore"):
        for i in range(1, k):
            term = term * L / i
            acc = acc + term
        out = np.exp(-L) * acc
    out = np.where(np.isinf(L), 0.0, out)
    out = np.where(np.all(P == 1.0, axis=0), 1.0, out)
    return np.minimum(out, 1.0)


def fdrcorrection(pvals, alpha=0.05):
    """Benjamini-Hochberg, the ``method='indep'`` branch of statsmodels' ``fdrcorrection``
    (statsmodels is not a dependency here): returns (reject mask, adjusted p), input order."""
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
    rej_o = np.empty(n, dtype=bool)
    adj_o = np.empty(n, dtype=np.float64)
    rej_o[order] = rej
    adj_o[order] = adj
    return rej_o, adj_o


def _summary(scores):
    s = np.asarray(scores)
    return {"median_score": float(np.median(s)), "mean_score": float(np.mean(s)), "std_score": float(np.std(s)),
            "min_score": float(np.min(s)), "max_score": float(np.max(s))}


def _subset_values(scores, mask, n, tag):
    if n <= 0:
        return {}
    s = np.asarray(scores)[mask]
    return {f"median_{tag}_score": float(np.median(s)), f"mean_{tag}_score": float(np.mean(s)),
            f"min_{tag}_score": float(np.min(s)), f"max_{tag}_score": float(np.max(s))}


def _subset(metrics, scores, mask, n, tag):
    metrics.update(_subset_values(scores, mask, n, tag))


_POOL = None


def _pool():
    """Two worker threads for the scalar summaries (median = a partition of an 80 000-entry copy, ~0.6 ms each; numpy
    releases the interpreter lock inside it) while the calling thread builds the per-voxel Python lists, which cannot be
    shared out (every ``tolist`` holds the lock): the metrics dictionary is the last thing between the GPU's final
    download and the caller, ~5 ms at cfg2 when done one after the other."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=2, thread_name_prefix="lc-metrics")
    return _POOL


def train_test_metrics(correlations, pvalues, corrected, significant, best_alphas, n_significant, part=None,
                       all_scores=None):
    """nested_cv.py:480-530 (keys, order and container types kept).  ``part`` (a slice; voxel shards with local
    lists): the per-voxel containers passed in cover only that part of the voxels, while ``all_scores`` /
    ``significant`` / ``n_significant`` are those of all voxels and feed the scalar summaries."""
    scores_all = correlations if all_scores is None else all_scores
    sl = slice(None) if part is None else part
    big = len(scores_all) >= 20000 and isinstance(scores_all, np.ndarray)     # (as in full_cv_metrics below)
    if big:
        pool = _pool()
        f_all = pool.submit(_summary, scores_all)
        f_sig = pool.submit(_subset_values, scores_all, significant, n_significant, "significant")
    lists = {"best_alphas": best_alphas[sl].tolist(), "correlations": correlations, "p_values": pvalues,
             "corrected_p_values": corrected[sl].tolist(), "significant_mask": significant[sl].tolist(),
             "n_significant": int(n_significant),
             "percent_significant": float(n_significant / len(scores_all) * 100)}
    m = f_all.result() if big else _summary(scores_all)
    m.update(lists)
    m.update(f_sig.result() if big else _subset_values(scores_all, significant, n_significant, "significant"))
    return m


def full_cv_metrics(scores, pvalues, corrected, significant, majority, mean_alphas, n_significant, n_majority, part=None):
    """nested_cv.py:533-616.  ``part`` (a slice; voxel shards with local lists): the scalar summaries are those of all
    voxels, the per-voxel lists cover only ``part``."""
    sl = slice(None) if part is None else part
    big = len(scores) >= 20000                           # small fits: the thread hand-off costs more than it saves
    if big:
        pool = _pool()
        f_all = pool.submit(_summary, scores)
        f_sig = pool.submit(_subset_values, scores, significant, n_significant, "significant")
        f_maj = pool.submit(_subset_values, scores, majority, n_majority, "majority_significant")
    lists = {"best_alphas": mean_alphas[sl].tolist(), "correlations": scores[sl].tolist(),
             "p_values": pvalues[sl].tolist(), "corrected_p_values": corrected[sl].tolist(),
             "significant_mask": significant[sl].tolist(), "majority_significant_mask": majority[sl].tolist(),
             "n_significant": int(n_significant), "n_majority_significant": int(n_majority),
             "percent_significant": float(n_significant / len(scores) * 100),
             "percent_majority_significant": float(n_majority / len(scores) * 100)}
    m = f_all.result() if big else _summary(scores)      # key order of the reference's dictionary: summary, lists, subsets
    m.update(lists)
    m.update(f_sig.result() if big else _subset_values(scores, significant, n_significant, "significant"))
    m.update(f_maj.result() if big else _subset_values(scores, majority, n_majority, "majority_significant"))
    return m
