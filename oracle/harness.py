"""Oracle: the trainer-side data structuring that feeds the fit (test infra).

Follows ``encoding/utils.py:23-34`` (``zs``) and
``encoding/trainer.py:203-282`` (FIR per story, train/test split, concatenated
mode).
"""
import numpy as np

from . import fir as _fir


def zs(v):
    """utils.py:23-29: column z-score with POPULATION std; a column whose std is
    exactly 0 is only de-meaned (left un-divided)."""
    v = np.asarray(v)
    s = v.std(0)
    m = v - v.mean(0)
    nz = s != 0.0
    m[:, nz] /= s[nz]
    return m


def train_test_matrices(features, brain, trimming):
    """trainer.py:223-262: all stories but the last train, the last one tests;
    per story ``zs(x[start:end])``; vstack; ``nan_to_num`` on X only."""
    stories = list(features.keys())
    tr, te = stories[:-1], stories[-1:]
    g = trimming.get

    def stack(src, names, a, b):
        return np.vstack([zs(src[s][g(a, 0):g(b, None)]) for s in names])

    return {
        "Rstim": np.nan_to_num(stack(features, tr, "train_features_start", "train_features_end")),
        "Rresp": stack(brain, tr, "train_targets_start", "train_targets_end"),
        "Pstim": np.nan_to_num(stack(features, te, "test_features_start", "test_features_end")),
        "Presp": stack(brain, te, "test_targets_start", "test_targets_end"),
    }


def concatenated_matrices(features, brain, order, trimming):
    """trainer.py:264-282: concatenate stories, then trim; no z-scoring."""
    g = trimming.get
    X = np.concatenate([features[s] for s in order], axis=0)
    Y = np.concatenate([brain[s] for s in order], axis=0)
    return {"X": X[g("features_start", 0):g("features_end", None)],
            "Y": Y[g("targets_start", 0):g("targets_end", None)]}


def delay_all(features, delays):
    """trainer.py:203-209."""
    return {k: _fir.make_delayed(v, delays) for k, v in features.items()}
