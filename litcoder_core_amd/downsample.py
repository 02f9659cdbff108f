"""Downsampler front-end with the reference's interface (``downsampling.py:322-445``).

``lanczos`` -- the method every shipped LITcoder config uses -- runs on the MI355X
(``lc_lanczos_interp``: weights evaluated on the fly in fp64, banded weighted row sums).
The per-TR reducers (rect/average/sum/last/legacy_*) and sinc/gabor are not on the hot path
(SURVEY.md section 2, rows 6-7): they are short numpy routines on the host here, kept so the
class is a complete drop-in for ``AbstractTrainer(downsampler=...)``.
"""
from typing import List

import numpy as np
import torch

from . import ops


def _lanczos(data, data_times, tr_times, window=3, cutoff_mult=1.0, rectify=False):
    """interpdata.py:87-126.  cutoff = 1/mean(diff(newtime))*cutoff_mult (:107)."""
    data = np.asarray(data)
    newtime = np.ascontiguousarray(tr_times, dtype=np.float64)
    oldtime = np.ascontiguousarray(data_times, dtype=np.float64)
    if len(oldtime) != data.shape[0]:
        raise ValueError(f"shapes {(len(newtime), len(oldtime))} and {data.shape} not aligned")
    cutoff = 1 / np.mean(np.diff(newtime)) * cutoff_mult
    dev = ops.device()
    if data.dtype == np.float32:
        d = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    else:
        d = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(dev)
    out = ops.lanczos_interp(d, torch.from_numpy(oldtime).to(dev), torch.from_numpy(newtime).to(dev),
                             cutoff, window, rectify)
    return out.cpu().numpy()


def _rect(data, data_times, tr_times):
    out = np.zeros((len(tr_times), data.shape[1]))
    half = np.mean(np.diff(tr_times)) / 2
    for i, t in enumerate(tr_times):
        sel = (data_times >= t - half) & (data_times < t + half)
        if np.any(sel):
            out[i] = np.mean(data[sel], axis=0)
    return out


_REDUCERS = {"average": lambda a: np.mean(a, axis=0), "sum": lambda a: np.sum(a, axis=0), "last": lambda a: a[-1]}


def _by_label(how, what):
    def run(data, data_times=None, tr_times=None, split_indices=None):
        if split_indices is None:
            raise ValueError(f"split_indices must be provided for {what} downsampling")
        lab = np.asarray(split_indices)
        out = np.zeros((int(max(split_indices)) + 1, data.shape[1]))
        for tr in range(out.shape[0]):
            idx = np.nonzero(lab == tr)[0]
            if idx.size:
                out[tr] = _REDUCERS[how](data[idx])
        return out
    return run


def _by_chunks(how):
    def run(data, data_times=None, tr_times=None, split_indices=None):
        if split_indices is None:
            raise ValueError("split_indices must be provided for Legacy downsampling")
        out = np.zeros((len(split_indices) + 1, data.shape[1]))
        for ci, chunk in enumerate(np.split(data, split_indices)):
            if len(chunk):
                out[ci] = _REDUCERS[how](chunk)
        return out
    return run


def _sinc(data, data_times, tr_times, window=1, cutoff_mult=1.0, causal=False, renorm=True):
    """interpdata.py:29-42,66-84."""
    B = 1 / np.mean(np.diff(tr_times)) * cutoff_mult
    rows = []
    for tn in tr_times:
        t = tn - np.asarray(data_times, dtype=np.float64)
        v = 2 * B * np.sin(2 * np.pi * B * t) / (2 * np.pi * B * t + 1e-20)
        v[np.abs(t) > window / (2 * B)] = 0
        if causal:
            v[t < 0] = 0
        if not np.sum(v) == 0.0 and renorm:
            v = v / np.sum(v)
        rows.append(v)
    return np.dot(np.stack(rows), data)


def _gabor(data, data_times, tr_times, freqs, sigma):
    """interpdata.py:129-145 through downsampling.py:160-167 (|Gabor transform| of each column)."""
    s = np.vstack([np.sin(data_times * f * 2 * np.pi) for f in freqs])
    c = np.vstack([np.cos(data_times * f * 2 * np.pi) for f in freqs])
    cols = []
    for d in data.T:
        o = np.zeros((len(tr_times), len(freqs)), dtype=np.complex128)
        for ti, t in enumerate(tr_times):
            g = np.exp(-0.5 * (data_times - t) ** 2 / (2 * sigma ** 2)) * d
            o[ti] = np.dot(c, g) + 1j * np.dot(s, g)
        cols.append(o.T)
    return np.abs(np.vstack(cols)).T


class Downsampler:
    METHOD_PARAMS = {
        "lanczos": {"required": ["window", "cutoff_mult"], "optional": ["rectify"]},
        "sinc": {"required": ["window", "cutoff_mult"], "optional": ["causal", "renorm"]},
        "average": {"required": ["split_indices"], "optional": []},
        "sum": {"required": ["split_indices"], "optional": []},
        "last": {"required": ["split_indices"], "optional": []},
        "legacy_average": {"required": ["split_indices"], "optional": []},
        "legacy_sum": {"required": ["split_indices"], "optional": []},
        "legacy_last": {"required": ["split_indices"], "optional": []},
        "rect": {"required": [], "optional": []},
        "gabor": {"required": ["freqs", "sigma"], "optional": []},
    }

    def __init__(self):
        self._methods = {
            "rect": _rect,
            "average": _by_label("average", "average"),
            "sinc": _sinc,
            "lanczos": _lanczos,
            "last": _by_label("last", "last point"),
            "gabor": _gabor,
            "legacy_average": _by_chunks("average"),
            "legacy_last": _by_chunks("last"),
            "sum": _by_label("sum", "sum"),
            "legacy_sum": _by_chunks("sum"),
        }

    def _validate_method_params(self, method: str, **kwargs) -> dict:
        """downsampling.py:361-393: unknown method / missing required kwarg -> ValueError with
        the reference's wording; kwargs the method does not list are dropped silently."""
        if method not in self._methods:
            raise ValueError(f"Unsupported downsampling method: {method}")
        spec = self.METHOD_PARAMS.get(method, {"required": [], "optional": []})
        kept = {}
        for name in spec["required"]:
            if name not in kwargs:
                raise ValueError(f"Required parameter '{name}' missing for method '{method}'")
            kept[name] = kwargs[name]
        for name in spec["optional"]:
            if name in kwargs:
                kept[name] = kwargs[name]
        return kept

    def downsample(self, data: np.ndarray, data_times: np.ndarray, tr_times: np.ndarray, method: str = "rect",
                   **kwargs) -> np.ndarray:
        kept = self._validate_method_params(method, **kwargs)
        return self._methods[method](data, data_times, tr_times, **kept)

    @property
    def available_methods(self) -> List[str]:
        return list(self._methods.keys())

    def get_method_params(self, method: str) -> dict:
        if method not in self._methods:
            raise ValueError(f"Unsupported downsampling method: {method}")
        return self.METHOD_PARAMS.get(method, {"required": [], "optional": []})
