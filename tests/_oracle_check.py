"""Comparison of a fit of litcoder_core_amd with the CPU oracle's, with a PROOF for every alpha that differs.

The alpha of a voxel is an argmax over a score table whose neighbouring entries can be equal to within fp32 rounding
(plateaus at heavy shrinkage, noise voxels), so two correct implementations may pick different alphas for a few
voxels.  A comparison that simply drops such voxels would also pass a real selection bug.  Here every voxel whose
alpha differs from the oracle's in some outer fold must satisfy, in the ORACLE's own (fold-mean) score table of that
fold,  score[oracle's alpha] - score[our alpha] <= gap_tol  (a genuine near-tie: the oracle takes the first maximum,
so the gap is >= 0); all other voxels are compared in full; and for the flipped voxels the fold's correlation is
re-derived at the ORACLE's alpha through this package's own ridge solver and compared with the oracle's fold score.
"""
import numpy as np


def _record_flips(tag, n_flips, n_folds, n_voxels, n_voxels_flipped, min_same, gap_tol):
    """One line per comparison into gpurun_out/parity_flips.txt (LITCODER_PARITY_FLIPS_FILE overrides; the round's copy is
    profiles/rNN_parity_flips.txt): how many (fold, voxel) alpha choices differed from the oracle's / the reference's -- every
    one a proven near-tie, or the comparison fails -- and how far that is from the limit the test allows."""
    import os
    path = os.environ.get("LITCODER_PARITY_FLIPS_FILE")
    if path is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "gpurun_out", "parity_flips.txt")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(f"{tag}: {n_flips} of {n_folds * n_voxels} (fold, voxel) pairs flipped ({n_folds} folds x {n_voxels} voxels); "
                    f"{n_voxels_flipped} voxels flipped in some fold = {n_voxels_flipped / max(n_voxels, 1):.4f} "
                    f"(limit {1 - min_same:.2f}); every flip a near-tie <= {gap_tol:g} of the oracle's own score table\n")
    except OSError:
        pass


def _alpha_index(alphas, values):
    al = np.asarray(alphas, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    idx = np.abs(np.log(v[:, None]) - np.log(al[None, :])).argmin(axis=1)
    assert np.allclose(al[idx], v, rtol=1e-6), "alpha not on the grid"
    return idx


def assert_matches_oracle(lc, model, ours, oracle, detail, X, Y, kw, tag, corr_atol=3e-5, w_rtol=2e-4, w_atol=3e-6,
                          gap_tol=2e-6, min_same=0.9, X_test=None, y_test=None, cols=None, w_cols=None, r2_space=False):
    """``ours`` / ``oracle``: (metrics, W, alphas) of the two fits on the same inputs; ``detail``: the oracle's
    per-fold intermediates (oracle.nested_cv.fit_predict(detail=...)); ``cols``: columns of ``ours`` the oracle was run
    on (a voxel sample), default all; ``w_cols``: the oracle's weight matrix holds only its first ``w_cols`` columns
    (the reference-generated config fixtures keep 32).  Returns the number of flipped (fold, voxel) pairs."""
    from litcoder_core_amd import ridge
    (m, W, a), (m_o, W_o, a_o) = ours, oracle
    cols = np.arange(len(a_o)) if cols is None else np.asarray(cols)
    alphas = kw["alphas"]
    single = bool(kw.get("single_alpha", False))
    tt = X_test is not None
    fold_ours = [np.asarray(f)[cols] for f in model.last_fold_alphas]
    if tt:
        fold_orc, tables = [np.asarray(a_o)], [detail["mean_scores"]]
    else:
        fold_orc, tables = list(detail["fold_alphas"]), list(detail["fold_mean_scores"])
    assert len(fold_ours) == len(fold_orc), tag
    flipped = np.zeros(len(cols), dtype=bool)
    n_flips = 0
    for f, (ao, am, tab) in enumerate(zip(fold_orc, fold_ours, tables)):
        diff = ~np.isclose(am, ao, rtol=1e-6)
        if not diff.any():
            continue
        ko, km = _alpha_index(alphas, ao[diff]), _alpha_index(alphas, am[diff])
        if single:                                   # one alpha for all voxels: the tie is in the across-voxel mean
            gap = tab.mean(axis=1)[ko[0]] - tab.mean(axis=1)[km[0]]
            assert 0 <= gap <= gap_tol, f"{tag}: fold {f} single alpha differs, oracle mean-score gap {gap:.3g}"
        else:
            v = np.nonzero(diff)[0]
            gap = tab[ko, v] - tab[km, v]
            if r2_space:
                # R^2 scores are signed sqrt|R^2| (ridge_regression.py:126-130): near R^2 = 0 the square root amplifies the
                # fp32 rounding of 1 - resvar / var by 1 / (2 sqrt|R^2|) -- the near-tie is judged where the rounding
                # happens, on R^2 itself: gap = R^2[oracle's alpha] - R^2[ours], R^2 = s |s|
                gap = tab[ko, v] * np.abs(tab[ko, v]) - tab[km, v] * np.abs(tab[km, v])
            worst = int(np.argmax(np.abs(gap)))
            assert np.all(gap >= 0) and np.all(gap <= gap_tol), (
                f"{tag}: fold {f}: {diff.sum()} alphas differ; voxel {v[worst]} oracle alpha {ao[diff][worst]:g} vs "
                f"{am[diff][worst]:g} with an oracle score gap of {gap[worst]:.3g} > {gap_tol:g} (scores "
                f"{tab[ko, v][worst]:.6g} / {tab[km, v][worst]:.6g}): not a near-tie")
        flipped |= diff
        n_flips += int(diff.sum())
    clean = ~flipped
    _record_flips(tag, n_flips, len(fold_orc), len(cols), int(flipped.sum()), min_same, gap_tol)
    assert clean.mean() >= min_same, f"{tag}: only {clean.mean():.3f} of the voxels chose the oracle's alpha in every fold"
    r, r_o = np.asarray(m["correlations"], dtype=np.float64)[cols], np.asarray(m_o["correlations"], dtype=np.float64)
    np.testing.assert_allclose(r[clean], r_o[clean], rtol=0, atol=corr_atol, err_msg=tag)
    nw = len(cols) if w_cols is None else int(w_cols)
    Wc = np.asarray(W[:, cols[:nw]])
    np.testing.assert_allclose(Wc[:, clean[:nw]], W_o[:, :nw][:, clean[:nw]], rtol=w_rtol,
                               atol=w_atol * max(1.0, float(np.abs(W_o).max())), err_msg=tag)
    np.testing.assert_allclose(np.asarray(a)[cols][clean], np.asarray(a_o)[clean], rtol=1e-6, err_msg=tag)
    # flipped voxels: the fold correlation at the ORACLE's alpha, through this package's ridge solver
    plain = not (kw.get("normalize_features") or kw.get("normalize_targets")) and not single
    if flipped.any() and plain:
        Xa, Ya = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)[:, cols]
        normalpha = kw.get("normalpha", True)
        if tt:
            folds = [(np.arange(len(Xa)), None)]
        else:
            folds = [(np.asarray(tr), np.asarray(te)) for tr, te in detail["outer"]]
        for f, (tr, te) in enumerate(folds):
            diff = ~np.isclose(fold_ours[f], fold_orc[f], rtol=1e-6)
            if not diff.any():
                continue
            Wf = ridge.ridge(Xa[tr], Ya[tr][:, diff], np.asarray(fold_orc[f])[diff], normalpha=normalpha)
            Xte, Yte = (np.asarray(X_test, dtype=np.float64), np.asarray(y_test, dtype=np.float64)[:, cols][:, diff]) if tt \
                else (Xa[te], Ya[te][:, diff])
            pred = Xte.astype(np.float32) @ Wf
            pc, yc = pred - pred.mean(0), Yte.astype(np.float32) - Yte.astype(np.float32).mean(0)
            with np.errstate(all="ignore"):
                rr = np.nan_to_num((pc * yc).sum(0) / np.sqrt((pc ** 2).sum(0) * (yc ** 2).sum(0)))
            want = (np.asarray(m_o["correlations"], dtype=np.float64)[diff] if tt else detail["fold_scores"][f][diff])
            np.testing.assert_allclose(rr, want, rtol=0, atol=max(corr_atol, 5e-5),
                                       err_msg=f"{tag}: fold {f} r at the oracle's alpha (flipped voxels)")
    return n_flips
