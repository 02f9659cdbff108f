"""Oracle: cross-validation index generation (test infrastructure).

Follows ``encoding/models/folding.py:8-255``.  Uses sklearn's splitters and
Python's global ``random`` exactly where the reference does, so that seeding
``random`` reproduces the reference's ``chunked`` shuffles.
"""
import random

import numpy as np
from sklearn.model_selection import GroupKFold, KFold, TimeSeriesSplit


def _chunk_rows(chunks, chunk_length, n_samples, trim=0):
    rows = []
    for c in chunks:
        lo = c * chunk_length
        hi = min(lo + chunk_length, n_samples)
        if lo + trim < hi - trim:
            rows.extend(range(lo + trim, hi - trim))
    return rows


def chunked(n_samples, n_folds, chunk_length, shuffle, trim=None):
    """folding.py:67-124 (trim None) / :127-199 (trim given).  Only complete
    chunks take part; fold i tests ``n_chunks // n_folds`` chunks, the last fold
    also takes the remainder; rows past the last complete chunk are in no fold.
    Too few chunks -> plain KFold (shuffled iff this variant shuffles and is
    untrimmed: folding.py:95 vs :160)."""
    n_chunks = n_samples // chunk_length
    order = list(range(n_chunks))
    if shuffle:
        random.shuffle(order)
    per_fold = n_chunks // n_folds
    if per_fold == 0:
        kf = KFold(n_splits=n_folds, shuffle=(shuffle if trim is None else False))
        return list(kf.split(range(n_samples)))
    splits = []
    for i in range(n_folds):
        stop = (i + 1) * per_fold if i < n_folds - 1 else n_chunks
        test_chunks = order[i * per_fold:stop]
        held = set(test_chunks)
        train_chunks = [c for c in order if c not in held]
        splits.append((_chunk_rows(train_chunks, chunk_length, n_samples),
                       _chunk_rows(test_chunks, chunk_length, n_samples, trim or 0)))
    return splits


def kfold_trimmed(n_samples, n_folds, trim):
    """folding.py:202-255: contiguous KFold, test folds lose ``trim`` rows at
    each end when they are longer than ``2*trim``."""
    out = []
    for tr, te in KFold(n_splits=n_folds, shuffle=False).split(range(n_samples)):
        te = list(te)
        if len(te) > 2 * trim:
            te = te[trim:-trim]
        out.append((list(tr), te))
    return out


def create_folds(n_samples, fold_type, n_folds, chunk_length=None, trim_size=None, groups=None):
    """folding.py:8-64 dispatcher (positional order kept: the reference's
    callers pass ``groups`` as the 5th positional, i.e. into ``trim_size``)."""
    if fold_type == "chunked":
        return chunked(n_samples, n_folds, chunk_length, shuffle=True)
    if fold_type == "chunked_trimmed":
        return chunked(n_samples, n_folds, chunk_length, shuffle=True,
                       trim=5 if trim_size is None else trim_size)
    if fold_type == "chunked_contiguous":
        return chunked(n_samples, n_folds, chunk_length, shuffle=False)
    if fold_type == "kfold":
        return list(KFold(n_splits=n_folds, shuffle=False).split(range(n_samples)))
    if fold_type == "kfold_trimmed":
        return kfold_trimmed(n_samples, n_folds, 5 if trim_size is None else trim_size)
    if fold_type == "timeseries":
        return list(TimeSeriesSplit(n_splits=n_folds).split(range(n_samples)))
    if fold_type == "group":
        if groups is None:
            raise ValueError("Groups must be provided for group folding")
        return list(GroupKFold(n_splits=n_folds).split(range(n_samples), groups=groups))
    raise ValueError(f"Unknown folding type: {fold_type}")
