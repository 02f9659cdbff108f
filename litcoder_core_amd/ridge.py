"""Function-level mirrors of the reference's two ridge solvers, on the MI355X.

``ridge_corr`` = ``ridge_corr_torch`` (``encoding/models/ridge_regression.py:66-141``): validation
score of every alpha for every voxel.  ``ridge`` = ``ridge_torch`` (``:9-63``): weights with a
per-voxel (or scalar) alpha.  Both take and return host arrays; the arithmetic is the same HIP
pipeline ``NestedCVModel`` uses (Gram + batched Cholesky instead of the SVD, fused MFMA sweep).
``singcutoff``: a value that is negligible against the smallest penalty truncates nothing visible in fp32 (a direction
with singular value <= cutoff enters a prediction with weight <= (cutoff / a)^2); alpha = 0 or a cutoff that bites take
the spectral route (``nested_cv.check_penalties``), which reproduces ``svd_wrapper``'s truncation.  Any number of
distinct alphas, negative values included (the penalty is alpha^2).
"""
from typing import Sequence, Union

import numpy as np
import torch

from . import ops
from .nested_cv import RidgeCVEngine


def ridge_corr(Rstim, Pstim, Rresp, Presp, alphas: Sequence[float], singcutoff: float = 1e-30,
               use_corr: bool = True, normalpha: bool = False) -> np.ndarray:
    """(A, V) float32 scores: mean(z(Presp) * z(pred_alpha)) with unbiased std + 1e-8, or signed
    sqrt|R^2|; NaN -> 0."""
    Rstim, Pstim = np.asarray(Rstim), np.asarray(Pstim)
    n_tr, n_va = len(Rstim), len(Pstim)
    if n_va == 0:
        # an empty validation block: z_score of nothing is NaN, nan_to_num makes every score 0 (ridge_regression.py:124-133)
        ops.device()                                   # (still no CPU path: raises without a gfx950 device)
        return np.zeros((len(list(alphas)), np.shape(Rresp)[1]), dtype=np.float32)
    eng = RidgeCVEngine(np.concatenate([Rstim, Pstim]), np.concatenate([np.asarray(Rresp), np.asarray(Presp)]),
                        alphas, normalpha, use_corr, False, False, singcutoff=singcutoff)
    scores, info = eng._alpha_scores(eng.K, eng.dY, [(np.arange(n_tr), n_tr + np.arange(n_va))])
    if int(info.cpu().numpy().any()):
        raise RuntimeError("Cholesky failed: Gram matrix + alpha^2 I is not positive definite")
    return scores[:, : eng.V].cpu().numpy()


def ridge(Rstim, Rresp, alphas: Union[float, Sequence[float]], singcutoff: float = 1e-30,
          normalpha: bool = False) -> np.ndarray:
    """(p, V) float32 weights; ``alphas`` is a scalar or one value per voxel."""
    Rstim, Rresp = np.asarray(Rstim), np.asarray(Rresp)
    V = Rresp.shape[1]
    per_voxel = np.full(V, float(alphas)) if np.isscalar(alphas) else np.asarray(alphas, dtype=np.float64)
    grid, idx = np.unique(per_voxel, return_inverse=True)
    eng = RidgeCVEngine(Rstim, Rresp, grid, normalpha, True, False, False, singcutoff=singcutoff)
    best = torch.zeros(eng.Vp, dtype=torch.int32, device=eng.dev)
    best[:V] = torch.from_numpy(idx.astype(np.int32)).to(eng.dev)
    Ws, _, perm, _, info = eng.refit(eng.dX, eng.dY, eng.K, np.arange(len(Rstim)), best)
    if int(info.cpu().numpy().any()):
        raise RuntimeError("Cholesky failed in the refit: Gram matrix + alpha^2 I is not positive definite")
    Wh = Ws[: eng.p].cpu().numpy()
    perm_h = perm[: Ws.shape[1]].cpu().numpy()
    live = perm_h >= 0
    out = np.empty((eng.p, V), dtype=np.float32)
    out[:, perm_h[live]] = Wh[:, live]
    return out


# the reference's own names, so ``from ... import ridge_torch, ridge_corr_torch`` call sites port 1:1
ridge_corr_torch = ridge_corr
ridge_torch = ridge
