"""Oracle: SVD-route ridge solvers in fp32 torch-CPU (test infrastructure).

Follows ``encoding/models/ridge_regression.py:9-141`` and
``encoding/models/ridge_utils.py:6-180``.  The ridge penalty is the SQUARE of
the (optionally ``S[0]``-scaled) alpha: weights = Vh' diag(S/(S^2+a^2)) U' Y.
"""
import numpy as np
import torch

EPS = 1e-8


def zscore_cols(x, eps=EPS):
    """ridge_utils.py:6-15, torch branch: column mean, UNBIASED std, ``+eps``
    added to the std (not to the variance)."""
    return (x - x.mean(dim=0, keepdim=True)) / (x.std(dim=0, keepdim=True) + eps)


def thin_svd(X, singcutoff):
    """ridge_utils.py:34-67: thin SVD, singular values ``<= singcutoff`` dropped."""
    U, S, Vh = torch.linalg.svd(X, full_matrices=False)
    keep = int((S > singcutoff).sum().item())
    return U[:, :keep], S[:keep], Vh[:keep]


def alpha_sweep_scores(Xtr, Xva, Ytr, Yva, alphas, singcutoff=1e-30, use_corr=True, normalpha=False):
    """ridge_regression.py:66-141 (``ridge_corr_torch``): validation score of
    every alpha for every voxel, shape (A, V).

    corr mode: mean over validation rows of z(Yva)*z(pred) -- i.e. (n-1)/n times
    Pearson r, with the 1e-8 in both stds.  R2 mode: sign(R2)*sqrt|R2| with
    unbiased variances.  NaN -> 0.
    """
    U, S, Vh = thin_svd(Xtr, singcutoff)
    s0 = S[0].item()
    scaled = [a * s0 for a in alphas] if normalpha else list(alphas)
    UtY = U.T @ Ytr                       # (r, V)
    XvaV = Xva @ Vh.T                     # (n_v, r)
    zYva = zscore_cols(Yva)
    var_va = Yva.var(dim=0)
    rows = []
    for a in scaled:
        shrink = S / (S ** 2 + a ** 2)
        pred = (XvaV * shrink.unsqueeze(0)) @ UtY
        if use_corr:
            score = (zYva * zscore_cols(pred)).mean(dim=0)
        else:
            r2 = 1 - (Yva - pred).var(dim=0) / var_va
            score = torch.sqrt(torch.abs(r2)) * torch.sign(r2)
        rows.append(torch.nan_to_num(score))
    return torch.stack(rows)


def ridge_weights(X, Y, alphas, singcutoff=1e-30, normalpha=False):
    """ridge_regression.py:9-63 (``ridge_torch``): (p, V) weights where voxel v
    uses ``alphas[v]``.  With ``normalpha`` the per-voxel alpha tensor is
    multiplied by ``S[0]`` IN THE TENSOR'S OWN DTYPE before squaring (:41,:56),
    which is fp32 for per-voxel alphas and fp64 for the single-alpha path."""
    U, S, Vh = thin_svd(X, singcutoff)
    UtY = U.T @ Y
    if isinstance(alphas, (int, float)):
        alphas = torch.ones(Y.shape[1]) * alphas
    s0 = S[0].item()
    scaled = alphas * s0 if normalpha else alphas
    W = torch.zeros((X.shape[1], Y.shape[1]))
    for a in torch.unique(scaled):
        cols = torch.nonzero(scaled == a).reshape(-1)
        shrink = S / (S ** 2 + a ** 2)
        W[:, cols] = (Vh.T * shrink.unsqueeze(0)) @ UtY[:, cols]
    return W


class TrainStatNormalizer:
    """ridge_utils.py:70-180 (``DataNormalizer``): z-score X and/or Y columns
    with the TRAIN block's mean and unbiased std (+1e-8)."""

    def __init__(self, do_x, do_y, eps=EPS):
        self.do_x, self.do_y, self.eps = do_x, do_y, eps
        self.xm = self.xs = self.ym = self.ys = None

    def fit(self, X, Y):
        if self.do_x:
            self.xm, self.xs = X.mean(dim=0, keepdim=True), X.std(dim=0, keepdim=True)
        if self.do_y:
            self.ym, self.ys = Y.mean(dim=0, keepdim=True), Y.std(dim=0, keepdim=True)
        return self

    def apply(self, X, Y):
        if self.do_x:
            X = (X - self.xm) / (self.xs + self.eps)
        if self.do_y:
            Y = (Y - self.ym) / (self.ys + self.eps)
        return X, Y
