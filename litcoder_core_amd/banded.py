"""Banded ridge on top of the nested-CV path.

The reference has no banded ridge (its nearest thing is the column concatenation of several feature extractors,
``trainer.py:146-150``), so the definition is this package's own and says so: feature band ``b`` gets its own
penalty ``(alpha * gamma_b)^2`` instead of ``alpha^2``,

    minimise  |y - sum_b X_b w_b|^2 + alpha^2 sum_b gamma_b^2 |w_b|^2 ,

which is ordinary ridge on the rescaled design ``X_b / gamma_b`` with ``w_b = w'_b / gamma_b``.  Everything else --
folds, alpha grid, scoring, statistics, the HIP kernels -- is the unchanged ``NestedCVModel`` path; parity is therefore
defined against the oracle run on the rescaled design (tests/test_gpu_parity.py).  ``normalize_features`` would undo
the scaling (train-statistics z-scoring happens inside the fit) and is rejected.
"""
from typing import Optional, Sequence

import numpy as np

from .nested_cv import NestedCVModel


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
