"""Banded ridge on top of the nested-CV path.

The reference has no banded ridge (its nearest thing is the column concatenation of several feature extractors,
``trainer.py:146-150``), so the definition is this package's own and says so: feature band ``b`` gets its own
penalty ``(alpha * gamma_b)^2`` instead of ``alpha^2``,

    minimise  |y - sum_b X_b w_b|^2 + alpha^2 sum_b gamma_b^2 |w_b|^2 ,

which is ordinary ridge on the rescaled design ``X_b / gamma_b`` with ``w_b = w'_b / gamma_b``.  Everything else --
folds, alpha grid, scoring, statistics, the HIP kernels -- is the unchanged ``NestedCVModel`` path; parity is therefore
defined against the oracle run on the rescaled design (tests/test_gpu_parity.py).  ``normalize_features`` would undo
the scaling (train-statistics z-scoring happens inside the fit) and is rejected.

**Search over band scales** (round 3; ``band_scale_candidates``): with C candidate scale vectors gamma^(c) the model
selects, per voxel and outer fold, the pair (candidate, alpha) with the best inner-CV score,

    per outer fold:  table_c (A, V) = the inner-CV score table of the design X / gamma^(c)   (every candidate is a full
                     pass of the existing pipeline: Gram, Lanczos, batched Cholesky / series operators, fused sweeps);
                     (c*, alpha*) = first maximum of the stacked table, candidate-major;
                     refit of the voxels of candidate c on X_train / gamma^(c) at their alphas -- the grouped refit
                     contraction over that candidate's voxels only (lc_group_by_alpha leaves the others out) --
                     weights divided by gamma^(c), test r from X_test / gamma^(c);
    over the folds:  the reference's tail (mean r, Fisher, BH-FDR, majority vote, mean alpha, mean weights).

The candidates' engines share ONE resident copy of the targets; a candidate costs its own design (74 MB at cfg5's
shape) and operators.  Parity is defined against ``oracle/banded.py`` (the reference's score table / ridge_torch on each
rescaled design + the selection rule above).  Full cross-validation only (no X_test / y_test), per-voxel alpha,
resident one-GPU fits; the steps of a fold run one after the other (no cross-fold pipelining: this is the widened row,
not the headline path).
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import ops, stats
from .folding import create_folds
from .nested_cv import NestedCVModel, RidgeCVEngine, _DeviceShapes, _alpha_vector, check_penalties


def band_column_scales(n_features: int, bands: Sequence, band_scales: Sequence[float]) -> np.ndarray:
    """Per-column divisor gamma from ``bands`` -- either one band id per feature column or a list of
    ``(start, stop)`` column ranges covering every column exactly once -- and one ``gamma_b > 0`` per band."""
    gam = np.asarray(band_scales, dtype=np.float64)
    if gam.ndim != 1 or not np.all(np.isfinite(gam)) or np.any(gam <= 0):
        raise ValueError("band_scales must be a 1-D sequence of positive finite numbers")
    b = np.asarray(bands)
    if b.ndim == 1 and b.shape[0] == n_features and np.issubdtype(b.dtype, np.integer):
        if b.min() < 0 or b.max() >= gam.size:
            raise ValueError("band id out of range")
        return gam[b]
    ids = np.full(n_features, -1, dtype=np.int64)
    if len(bands) != gam.size:
        raise ValueError("need one band_scale per (start, stop) band")
    for k, (lo, hi) in enumerate(bands):
        if not (0 <= lo < hi <= n_features) or np.any(ids[lo:hi] >= 0):
            raise ValueError("bands must be disjoint (start, stop) ranges inside the feature axis")
        ids[lo:hi] = k
    if np.any(ids < 0):
        raise ValueError("bands must cover every feature column")
    return gam[ids]


class BandedNestedCVModel(NestedCVModel):
    """``NestedCVModel`` with a per-band penalty scale; ``fit_predict`` takes ``bands`` and ``band_scales``."""

    def fit_predict(self, features, targets, X_test: Optional[np.ndarray] = None, y_test: Optional[np.ndarray] = None,
                    bands=None, band_scales=None, **kwargs):
        if bands is None or band_scales is None:
            return super().fit_predict(features, targets, X_test=X_test, y_test=y_test, **kwargs)
        if kwargs.get("normalize_features", False):
            raise ValueError("normalize_features=True would undo the band scaling")
        X = np.asarray(features, dtype=np.float64)
        gamma = band_column_scales(X.shape[1], bands, band_scales)
        Xt = None if X_test is None else np.asarray(X_test, dtype=np.float64) / gamma
        metrics, W, alphas = super().fit_predict(X / gamma, targets, X_test=Xt, y_test=y_test, **kwargs)
        return metrics, (np.asarray(W) / gamma[:, None].astype(np.float32)).astype(np.float32), alphas

    # ------------------------------------------------------------------ search over band scales
    def fit_predict_search(self, features, targets, bands, band_scale_candidates, folding_type: str = "chunked",
                           n_outer_folds: int = 5, n_inner_folds: int = 5, chunk_length: int = 20, alphas=None,
                           alpha_fdr: float = 0.05, normalpha: bool = True, use_corr: bool = True,
                           singcutoff: float = 1e-10):
        """Banded ridge with the band scales chosen per voxel among ``band_scale_candidates`` (C sequences of one
        positive scale per band) by the inner-CV score; see the module docstring.  Returns (metrics, weights with respect
        to the original features (p, V) float32, mean alphas) like ``fit_predict``; ``last_fold_candidates`` holds the
        (n_outer_folds, V) candidate indices."""
        alphas = np.logspace(-1, 8, 10) if alphas is None else alphas
        if check_penalties(alphas, singcutoff, normalpha, n_inner_folds):
            raise ValueError("the search over band scales needs a penalty grid the Cholesky route serves "
                             "(alphas > 0, negligible singcutoff)")
        X = np.asarray(features, dtype=np.float64)
        Y = np.asarray(targets)
        T, p = X.shape
        V = Y.shape[1]
        cands = [np.asarray(c, dtype=np.float64) for c in band_scale_candidates]
        if not cands:
            raise ValueError("band_scale_candidates is empty")
        gcols = [band_column_scales(p, bands, c) for c in cands]
        C, A = len(cands), len(alphas)
        if C * A > 64 * 64:
            raise ValueError("too many (candidate, alpha) pairs")
        dev = ops.device()
        outer = []
        for tr, te in create_folds(T, folding_type, n_outer_folds, chunk_length):
            outer.append((tr, te, create_folds(len(tr), folding_type, n_inner_folds, chunk_length)))
        n = len(outer)
        min_train = min(len(tr_i) for _, _, inner in outer for tr_i, _ in inner)
        Vp = ops.pad_to(max(V, 1), 128)
        dY = ops.upload_f32(Y, Vp, dev)                                   # ONE resident copy of the targets
        engs, lmax = [], []
        for g in gcols:
            dX = ops.upload_f32(X / g, ops.pad_to(p, 32), dev)
            eng = RidgeCVEngine(_DeviceShapes(dX, p), _DeviceShapes(dY, V), alphas, normalpha, use_corr, False, False,
                                precision=self.precision, singcutoff=singcutoff, V_total=V, min_train_rows=min_train,
                                form="dual", options=self.options)
            eng.begin_fit(n)
            engs.append(eng)
            lmax.append(eng.precompute_lmax(outer))
        main = torch.cuda.current_stream()
        W_acc = ops.zeros((p, Vp), torch.float32, dev)
        inv_g = [torch.as_tensor((1.0 / g).astype(np.float32), device=dev) for g in gcols]
        fold_r, fold_p, fold_alpha, fold_cand, fold_nan = [], [], [], [], []
        for f, (tr, te, inner) in enumerate(outer):
            n_t = len(te)
            sts = [eng.fold_begin(tr, te, inner, lmax_pre=lmax[c][f], step=(f, None)) for c, eng in enumerate(engs)]
            table = torch.empty((C * A, Vp), dtype=torch.float32, device=dev)
            for c, st in enumerate(sts):
                table[c * A:(c + 1) * A].copy_(st["scores"])
            best = ops.select_alpha(table, C * A, Vp)[0][:V].cpu().numpy()         # first maximum, candidate-major
            cand, aidx = best // A, best % A
            r_fold = np.full(V, np.nan, dtype=np.float64)
            for c, (eng, st) in enumerate(zip(engs, sts)):
                if not np.any(cand == c):
                    continue
                eng._enter(st)
                mine = np.full(Vp, -1, dtype=np.int32)
                mine[:V] = np.where(cand == c, aidx, -1)
                best_c = ops.upload(mine, dev)
                perm, used, tiles, Vs, used_all = eng._refit_groups(best_c, st["split"])
                for s_ in (eng.aux, eng.aux2):                                    # the fold's operators are complete
                    main.wait_stream(s_)
                Malpha, info_o = eng._refit_systems(st["X"], st["K"], st["tr"], used, st.get("tr_o"), st.get("lmax_o"),
                                                    st["te"], used_all=used_all, cache={})
                o = eng._refit_operands(st["Y"], st["tr"], st["te"], perm, tiles, Vs, Malpha, st["split"], st["cs"],
                                        image=st["hat"].get("image"))
                pred = eng._refit_product(o, eng.p_pad, Malpha[0].shape[0], n_t)[:n_t]
                if o.get("te_src") is not None:
                    r_s = ops.pearson_cols_gather(*o["te_src"], pred, n_t, Vs)
                else:
                    r_s = ops.pearson_cols(o["Ys_te"], pred, n_t, Vs)
                Ws = eng._refit_product(o, 0, eng.p_pad, p)
                Ws[:p].mul_(inv_g[c][:, None])                                    # weights of the ORIGINAL features
                ops.scatter_axpy(Ws, p, perm, Vs, 1.0 / n, W_acc)
                flags = torch.cat([st["info"].reshape(-1), info_o.reshape(-1)]).cpu().numpy()
                if np.any(flags != 0):
                    raise RuntimeError("Cholesky failed: Gram matrix + alpha^2 I is not positive definite")
                perm_h, r_h = perm[:Vs].cpu().numpy(), r_s.cpu().numpy()
                live = perm_h >= 0
                r_fold[perm_h[live]] = r_h[live]
            r32 = r_fold.astype(np.float32)
            p_f = stats.pearson_pvalues(r32, n_t)
            p_f = np.where(np.isnan(r32), 1.0, p_f)
            fold_r.append(np.nan_to_num(r32, nan=0.0))
            fold_nan.append(bool(np.isnan(r32).any()))
            fold_p.append(p_f)
            fold_alpha.append(_alpha_vector(alphas, aidx, False))
            fold_cand.append(cand.astype(np.int32))
        any_nan = any(fold_nan)
        scores = np.mean(np.stack(fold_r).astype(np.float64 if any_nan else np.float32), axis=0)
        pcomb = stats.fisher_combine(fold_p)
        sig, padj = stats.fdrcorrection(pcomb, alpha=alpha_fdr)
        fold_sig = [stats.fdrcorrection(pf, alpha=alpha_fdr)[0] for pf in fold_p]
        majority = np.sum(fold_sig, axis=0) >= (n_outer_folds // 2 + 1)
        mean_alphas = np.mean(fold_alpha, axis=0)
        metrics = stats.full_cv_metrics(scores, pcomb, padj, sig, majority, mean_alphas, np.sum(sig), np.sum(majority))
        self.last_fold_candidates = np.stack(fold_cand)
        self.last_fold_alphas = [np.asarray(a) for a in fold_alpha]
        return metrics, W_acc[:, :V].cpu().numpy(), mean_alphas
