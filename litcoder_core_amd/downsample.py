"""Downsampler front-end with the reference's interface (``downsampling.py:322-445``).

``lanczos`` -- the method every shipped LITcoder config uses -- and ``sinc`` run on the MI355X as banded
weighted row sums with the weights evaluated on the fly in fp64 (``lc_lanczos_interp`` / ``lc_sinc_interp``);
the per-TR reducers (rect / average / sum / last / legacy_*) are one segment-reduction kernel
(``lc_segment_reduce``) fed with row-index lists the host derives from time windows, TR labels or split points.
Only ``gabor`` (never used by a shipped config) stays a short HOST numpy routine (the one
method of this class that does not touch the device).

Where the banded kernels differ from the reference's dense ``sincmat @ data`` (interpdata.py:118-126), by construction:
  * a NON-FINITE sample (NaN / Inf word feature) makes the reference's WHOLE output column of that story NaN (every
    sample is multiplied by a weight, zeros included: 0 * NaN); here only the output rows whose sample batches (8
    samples) cover it are NaN -- the rows whose window holds it plus at most 7 samples either side
    (tests/test_gpu_parity.py::test_lanczos_non_finite_sample_is_confined).  Downstream the trainer's ``np.nan_to_num``
    on the design (trainer.py:257) zeroes such entries either way;
  * the window weights use sin(pi t) evaluated as sinpi (exactly 0 at integer t; np.sin(np.pi * t) leaves ~1e-16 there):
    golden outputs agree to 1e-12.
"""
from typing import List

import numpy as np
import torch

from . import ops


def _to_device(data, dev):
    """Sample matrix on the device: float32 stays float32, every other real dtype is widened to float64 (the
    reference's numpy arithmetic promotes to float64 as well)."""
    data = np.asarray(data)
    if data.dtype == np.float32:
        return torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    return torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(dev)


def _times(data, data_times, tr_times):
    newtime = np.ascontiguousarray(tr_times, dtype=np.float64)
    oldtime = np.ascontiguousarray(data_times, dtype=np.float64)
    if len(oldtime) != np.shape(data)[0]:
        raise ValueError(f"shapes {(len(newtime), len(oldtime))} and {np.shape(data)} not aligned")
    return oldtime, newtime


def _lanczos(data, data_times, tr_times, window=3, cutoff_mult=1.0, rectify=False):
    """interpdata.py:87-126.  cutoff = 1/mean(diff(newtime))*cutoff_mult (:107)."""
    oldtime, newtime = _times(data, data_times, tr_times)
    dev = ops.device()
    if len(newtime) == 0 or np.shape(data)[0] == 0:
        cutoff = 1 / np.mean(np.diff(newtime)) * cutoff_mult
        out = ops.lanczos_interp(_to_device(data, dev), torch.from_numpy(oldtime).to(dev), torch.from_numpy(newtime).to(dev),
                                 cutoff, window, rectify)
        return out.cpu().numpy()
    # one story through the batched kernel (round 5: several output rows per workgroup, the window found by bisection when
    # the sample times are sorted; same cutoff expression, same weights, same order of accumulation: the same bits)
    out, _ = ops.lanczos_interp_stories(_to_device(data, dev), [oldtime], [newtime], window, cutoff_mult, rectify)
    return out.cpu().numpy()


def _sinc(data, data_times, tr_times, window=1, cutoff_mult=1.0, causal=False, renorm=True):
    """interpdata.py:29-42,66-84."""
    oldtime, newtime = _times(data, data_times, tr_times)
    cutoff = 1 / np.mean(np.diff(newtime)) * cutoff_mult
    dev = ops.device()
    out = ops.sinc_interp(_to_device(data, dev), torch.from_numpy(oldtime).to(dev), torch.from_numpy(newtime).to(dev),
                          cutoff, window, causal, renorm)
    return out.cpu().numpy()


_HOW = {"average": 0, "sum": 1, "last": 2}


def _segments(data, groups, how):
    """``groups`` = list of row-index arrays, one per output row -> device segment reduction."""
    dev = ops.device()
    sizes = np.array([len(g) for g in groups], dtype=np.int64)
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    idx = (np.concatenate(groups) if len(groups) and seg[-1] else np.zeros(0)).astype(np.int32)
    if idx.size == 0:
        idx = np.zeros(1, dtype=np.int32)
    out = ops.segment_reduce(_to_device(data, dev), torch.from_numpy(seg).to(dev), torch.from_numpy(idx).to(dev), how)
    return out.cpu().numpy()


def _rect(data, data_times, tr_times):
    """downsampling.py:31-39: mean of the samples in [t - TR/2, t + TR/2)."""
    data_times = np.asarray(data_times)
    half = np.mean(np.diff(tr_times)) / 2
    groups = [np.nonzero((data_times >= t - half) & (data_times < t + half))[0] for t in tr_times]
    return _segments(data, groups, _HOW["average"])


def _by_label(how, what):
    def run(data, data_times=None, tr_times=None, split_indices=None):
        if split_indices is None:
            raise ValueError(f"split_indices must be provided for {what} downsampling")
        lab = np.asarray(split_indices)
        n_trs = int(max(split_indices)) + 1
        order = np.argsort(lab, kind="stable")
        bounds = np.searchsorted(lab[order], np.arange(n_trs + 1))
        groups = [order[bounds[i]:bounds[i + 1]] for i in range(n_trs)]
        return _segments(data, groups, _HOW[how])
    return run


def _by_chunks(how):
    def run(data, data_times=None, tr_times=None, split_indices=None):
        if split_indices is None:
            raise ValueError("split_indices must be provided for Legacy downsampling")
        groups = np.split(np.arange(np.shape(data)[0]), split_indices)
        return _segments(data, groups, _HOW[how])
    return run


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
