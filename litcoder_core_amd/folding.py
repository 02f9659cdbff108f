"""Cross-validation index generation (host side), semantics of ``encoding/models/folding.py``.

Same fold types, same argument order (so the reference's positional call
``create_folds(n, type, k, chunk_length, groups)`` lands ``groups`` in ``trim_size`` here
too), same use of Python's global ``random`` for chunk shuffles.  The contiguous K-fold,
time-series and group splitters are restated in numpy (the published scikit-learn
algorithms of ``KFold``, ``TimeSeriesSplit``, ``GroupKFold``) so the package does not need
scikit-learn at run time.
"""
import logging
import random
from typing import List, Optional, Tuple

import numpy as np

Split = Tuple[List[int], List[int]]


def _kfold(n_samples: int, n_folds: int, shuffle: bool = False):
    """sklearn ``KFold``: the first ``n % k`` folds get one extra sample; ``shuffle`` permutes
    the indices with numpy's global RandomState (``random_state=None``)."""
    if n_folds < 2:
        raise ValueError(f"k-fold cross-validation requires at least one train/test split by setting "
                         f"n_splits=2 or more, got n_splits={n_folds}.")
    if n_folds > n_samples:
        raise ValueError(f"Cannot have number of splits n_splits={n_folds} greater than the number of "
                         f"samples: n_samples={n_samples}.")
    idx = np.arange(n_samples)
    if shuffle:
        np.random.shuffle(idx)
    sizes = np.full(n_folds, n_samples // n_folds, dtype=int)
    sizes[: n_samples % n_folds] += 1
    out, start = [], 0
    for sz in sizes:
        mask = np.zeros(n_samples, dtype=bool)
        mask[idx[start:start + sz]] = True
        out.append((np.nonzero(~mask)[0], np.nonzero(mask)[0]))
        start += sz
    return out


def _timeseries(n_samples: int, n_folds: int):
    """sklearn ``TimeSeriesSplit`` defaults: test blocks of ``n // (k+1)``, growing train prefix."""
    test = n_samples // (n_folds + 1)
    if n_folds + 1 > n_samples:
        raise ValueError(f"Cannot have number of folds={n_folds + 1} greater than the number of samples={n_samples}.")
    idx = np.arange(n_samples)
    return [(idx[:s], idx[s:s + test]) for s in range(n_samples - n_folds * test, n_samples, test)]


def _group_kfold(n_samples: int, n_folds: int, groups):
    """sklearn ``GroupKFold``: groups sorted by size (descending), each assigned to the currently
    lightest fold."""
    groups = np.asarray(groups)
    uniq, inv = np.unique(groups, return_inverse=True)
    if n_folds > len(uniq):
        raise ValueError(f"Cannot have number of splits n_splits={n_folds} greater than the number of groups: "
                         f"{len(uniq)}.")
    per_group = np.bincount(inv)
    order = np.argsort(per_group)[::-1]
    load = np.zeros(n_folds)
    fold_of = np.zeros(len(uniq))
    for rank, w in enumerate(per_group[order]):
        light = np.argmin(load)
        load[light] += w
        fold_of[order[rank]] = light
    fold = fold_of[inv]
    idx = np.arange(n_samples)
    return [(idx[fold != f], idx[fold == f]) for f in range(n_folds)]


def _rows_of(chunks, chunk_length, n_samples, trim=0):
    rows: List[int] = []
    for c in chunks:
        lo = c * chunk_length
        hi = min(lo + chunk_length, n_samples)
        if lo + trim < hi - trim:
            rows.extend(range(lo + trim, hi - trim))
    return rows


def _chunked(n_samples, n_folds, chunk_length, shuffle, trim=None) -> List[Split]:
    """folding.py:67-199.  Whole chunks only; fold i tests ``n_chunks // n_folds`` chunks (the
    last fold also the remainder); rows after the last whole chunk belong to no fold; the
    trimmed variant drops ``trim`` rows at both ends of every TEST chunk."""
    n_chunks = n_samples // chunk_length
    order = list(range(n_chunks))
    if shuffle:
        random.shuffle(order)
    per_fold = n_chunks // n_folds
    if per_fold == 0:
        logging.warning("Not enough chunks for the requested folds, falling back to regular KFold")
        return _kfold(n_samples, n_folds, shuffle=(shuffle if trim is None else False))
    splits = []
    for i in range(n_folds):
        stop = (i + 1) * per_fold if i < n_folds - 1 else n_chunks
        held = order[i * per_fold:stop]
        held_set = set(held)
        kept = [c for c in order if c not in held_set]
        splits.append((_rows_of(kept, chunk_length, n_samples), _rows_of(held, chunk_length, n_samples, trim or 0)))
    return splits


def _kfold_trimmed(n_samples, n_folds, trim) -> List[Split]:
    """folding.py:202-255."""
    out = []
    for tr, te in _kfold(n_samples, n_folds):
        te = list(te)
        if len(te) > 2 * trim:
            te = te[trim:-trim]
        else:
            logging.warning(f"Test fold too small ({len(te)} samples) to trim {trim} from each end, "
                            f"keeping original test set")
        out.append((list(tr), te))
    return out


def create_folds(n_samples: int, fold_type: str, n_folds: int, chunk_length: Optional[int] = None,
                 trim_size: Optional[int] = None, groups: Optional[np.ndarray] = None) -> List[Split]:
    """folding.py:8-64."""
    if fold_type == "chunked":
        return _chunked(n_samples, n_folds, chunk_length, shuffle=True)
    if fold_type == "chunked_trimmed":
        return _chunked(n_samples, n_folds, chunk_length, shuffle=True, trim=5 if trim_size is None else trim_size)
    if fold_type == "chunked_contiguous":
        return _chunked(n_samples, n_folds, chunk_length, shuffle=False)
    if fold_type == "kfold":
        return _kfold(n_samples, n_folds)
    if fold_type == "kfold_trimmed":
        return _kfold_trimmed(n_samples, n_folds, 5 if trim_size is None else trim_size)
    if fold_type == "timeseries":
        return _timeseries(n_samples, n_folds)
    if fold_type == "group":
        if groups is None:
            raise ValueError("Groups must be provided for group folding")
        return _group_kfold(n_samples, n_folds, groups)
    raise ValueError(f"Unknown folding type: {fold_type}")
