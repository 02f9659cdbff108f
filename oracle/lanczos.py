"""Oracle: temporal downsamplers (test infrastructure, see oracle/__init__.py).

Lanczos follows ``encoding/downsample/interpdata.py:45-63`` (``lanczosfun``) and
``:87-126`` (``lanczosinterp2D``); the simple per-TR reducers follow
``encoding/downsample/downsampling.py:24-319``; sinc follows
``interpdata.py:29-84``.  All float64 numpy, like the reference.
"""
import numpy as np


def lanczos_kernel(cutoff, t, window=3):
    """interpdata.py:59-63: ``window*sin(pi t)*sin(pi t/window)/(pi^2 t^2)`` on
    ``t*cutoff``; exactly 1 at t==0 and exactly 0 beyond ``window`` lobes."""
    x = np.asarray(t, dtype=np.float64) * cutoff
    with np.errstate(divide="ignore", invalid="ignore"):
        w = window * np.sin(np.pi * x) * np.sin(np.pi * x / window) / (np.pi ** 2 * x ** 2)
    w = np.where(x == 0, 1.0, w)
    w = np.where(np.abs(x) > window, 0.0, w)
    return w


def lanczos_weights(oldtime, newtime, window=3, cutoff_mult=1.0):
    """Dense (n_new, n_old) weight matrix, interpdata.py:107-113.  The cutoff is
    the output sampling rate ``1/mean(diff(newtime))`` times ``cutoff_mult``."""
    oldtime = np.asarray(oldtime, dtype=np.float64)
    newtime = np.asarray(newtime, dtype=np.float64)
    cutoff = 1 / np.mean(np.diff(newtime)) * cutoff_mult
    return lanczos_kernel(cutoff, newtime[:, None] - oldtime[None, :], window)


def lanczos_interp(data, oldtime, newtime, window=3, cutoff_mult=1.0, rectify=False):
    """interpdata.py:115-124: ``W @ data``; with ``rectify`` the negative and the
    positive parts are filtered separately and stacked side by side."""
    w = lanczos_weights(oldtime, newtime, window, cutoff_mult)
    data = np.asarray(data)
    if rectify:
        return np.hstack([w @ np.clip(data, -np.inf, 0), w @ np.clip(data, 0, np.inf)])
    return w @ data


def sinc_kernel(B, t, window=np.inf, causal=False, renorm=True):
    """interpdata.py:29-42 (array branch)."""
    t = np.asarray(t, dtype=np.float64)
    v = 2 * B * np.sin(2 * np.pi * B * t) / (2 * np.pi * B * t + 1e-20)
    v[np.abs(t) > window / (2 * B)] = 0
    if causal:
        v[t < 0] = 0
    if not np.sum(v) == 0.0 and renorm:
        v = v / np.sum(v)
    return v


def sinc_interp(data, oldtime, newtime, cutoff_mult=1.0, window=1, causal=False, renorm=True):
    """interpdata.py:66-84."""
    oldtime = np.asarray(oldtime, dtype=np.float64)
    newtime = np.asarray(newtime, dtype=np.float64)
    cutoff = 1 / np.mean(np.diff(newtime)) * cutoff_mult
    w = np.stack([sinc_kernel(cutoff, tn - oldtime, window, causal, renorm) for tn in newtime])
    return w @ np.asarray(data)


def rect(data, data_times, tr_times):
    """downsampling.py:31-39: mean of the samples in [t - TR/2, t + TR/2)."""
    data = np.asarray(data)
    out = np.zeros((len(tr_times), data.shape[1]))
    tr = np.mean(np.diff(tr_times))
    for i, t in enumerate(tr_times):
        m = (data_times >= t - tr / 2) & (data_times < t + tr / 2)
        if m.any():
            out[i] = data[m].mean(axis=0)
    return out


def by_label(data, split_indices, how):
    """downsampling.py:45-136,239-284: ``split_indices[w]`` is the TR of sample
    w; reduce each TR's samples by mean / sum / last (highest index)."""
    data = np.asarray(data)
    lab = np.asarray(split_indices)
    out = np.zeros((int(lab.max()) + 1, data.shape[1]))
    for tr in range(out.shape[0]):
        idx = np.nonzero(lab == tr)[0]
        if idx.size:
            out[tr] = {"average": lambda a: a.mean(0), "sum": lambda a: a.sum(0),
                       "last": lambda a: a[-1]}[how](data[idx])
    return out


def by_chunks(data, split_indices, how):
    """downsampling.py:180-236,287-319 (legacy_*): ``np.split`` boundaries."""
    data = np.asarray(data)
    out = np.zeros((len(split_indices) + 1, data.shape[1]))
    for ci, ch in enumerate(np.split(data, split_indices)):
        if len(ch):
            out[ci] = {"average": lambda a: a.mean(0), "sum": lambda a: a.sum(0),
                       "last": lambda a: a[-1]}[how](ch)
    return out
