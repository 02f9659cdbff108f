"""Trainer-side data structuring on the device: what ``AbstractTrainer`` does between feature extraction and
``model.fit_predict`` (``encoding/trainer.py:203-282`` with ``encoding/utils.py:23-29``), so that a fit can start from
per-story arrays without the design / target matrices ever returning to the host.

    per story:  FIR.make_delayed (lc_fir_delay)  ->  trim  ->  zs (lc_zscore_story_f64; nan_to_num on features)
    stack stories  ->  Rstim / Rresp / Pstim / Presp   (train = all stories but the last, test = the last)
    or concatenate + trim without z-scoring (LPP / Narratives style)

``structure_*`` return host float64 arrays with the reference's values (tests compare them with captures of the
reference's own trainer); ``StoryPipeline.fit`` keeps everything resident and calls ``fit_predict_device``.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .nested_cv import NestedCVModel


def _dev_f64(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)


def zs(v: np.ndarray) -> np.ndarray:
    """``encoding.utils.zs``: column z-score, population std, zero-std columns only de-meaned."""
    v = np.asarray(v)
    dev = ops.device()
    x = _dev_f64(v, dev)
    out = torch.empty_like(x)
    ops.zscore_story(x, x.shape[0], x.shape[1], False, out)
    return out.cpu().numpy()


def apply_fir_delays(features: Dict[str, np.ndarray], delays: Sequence[int]) -> Dict[str, torch.Tensor]:
    """trainer.py:203-209, result kept on the device (float64, like the reference's arrays)."""
    dev = ops.device()
    out = {}
    for story, feat in features.items():
        feat = np.asarray(feat)
        src = (torch.from_numpy(np.ascontiguousarray(feat)).to(dev) if feat.dtype == np.float32 else _dev_f64(feat, dev))
        out[story] = ops.fir_delay(src, [int(d) for d in delays], False)
    return out


def _zs_stack(parts: List[torch.Tensor], nan_to_num: bool) -> torch.Tensor:
    """vstack([zs(part) for part in parts]) on the device."""
    rows = sum(int(p.shape[0]) for p in parts)
    cols = int(parts[0].shape[1])
    out = torch.empty((rows, cols), dtype=torch.float64, device=parts[0].device)
    r0 = 0
    for p in parts:
        n = int(p.shape[0])
        if n < 1:
            raise ValueError("a story is empty after trimming")
        ops.zscore_story(p, n, cols, nan_to_num, out[r0:r0 + n])
        r0 += n
    return out


def _on_device(d, dev):
    return {k: (v if torch.is_tensor(v) else _dev_f64(v, dev)) for k, v in d.items()}


def structure_train_test_device(features, brain, trimming: dict):
    """trainer.py:223-262 on the device: dict of float64 device tensors Rstim, Rresp, Pstim, Presp."""
    dev = ops.device()
    features, brain = _on_device(features, dev), _on_device(brain, dev)
    stories = list(features.keys())
    train, test = stories[:-1], stories[-1:]
    g = trimming.get

    def cut(src, names, a, b):
        return [src[s][g(a, 0):g(b, None)] for s in names]

    return {
        "Rstim": _zs_stack(cut(features, train, "train_features_start", "train_features_end"), True),
        "Rresp": _zs_stack(cut(brain, train, "train_targets_start", "train_targets_end"), False),
        "Pstim": _zs_stack(cut(features, test, "test_features_start", "test_features_end"), True),
        "Presp": _zs_stack(cut(brain, test, "test_targets_start", "test_targets_end"), False),
    }


def structure_train_test(features, brain, trimming: dict) -> Dict[str, np.ndarray]:
    return {k: v.cpu().numpy() for k, v in structure_train_test_device(features, brain, trimming).items()}


def structure_concatenated(features, brain, order: Sequence[str], trimming: dict) -> Dict[str, np.ndarray]:
    """trainer.py:264-282: concatenate the stories, then trim; no z-scoring (pure indexing, host)."""
    g = trimming.get
    X = np.concatenate([np.asarray(features[s]) for s in order], axis=0)
    Y = np.concatenate([np.asarray(brain[s]) for s in order], axis=0)
    return {"X": X[g("features_start", 0):g("features_end", None)], "Y": Y[g("targets_start", 0):g("targets_end", None)]}


class StoryPipeline:
    """Stories in, metrics out, with the matrices resident on the GPU from the FIR kernel to the fit
    (the Lebel-style train/test paradigm of ``AbstractTrainer.train``, trainer.py:284-320)."""

    def __init__(self, fir_delays: Sequence[int], trimming: dict, model: Optional[NestedCVModel] = None):
        self.fir_delays = [int(d) for d in fir_delays]
        self.trimming = dict(trimming)
        self.model = model or NestedCVModel("ridge_regression")

    def fit(self, features: Dict[str, np.ndarray], brain: Dict[str, np.ndarray], **model_kwargs):
        delayed = apply_fir_delays(features, self.fir_delays)
        d = structure_train_test_device(delayed, brain, self.trimming)
        dev = d["Rstim"].device
        T, Tt = d["Rstim"].shape[0], d["Pstim"].shape[0]
        p, V = d["Rstim"].shape[1], d["Rresp"].shape[1]
        if d["Rresp"].shape[0] != T or d["Presp"].shape[0] != Tt:
            raise RuntimeError("features and targets have different numbers of rows after trimming")
        X = torch.zeros((T + Tt, ops.pad_to(p, 32)), dtype=torch.float32, device=dev)
        Y = torch.zeros((T + Tt, ops.pad_to(V, 128)), dtype=torch.float32, device=dev)
        for dst, top, bottom, n in ((X, d["Rstim"], d["Pstim"], p), (Y, d["Rresp"], d["Presp"], V)):
            ops.cast_f64_f32(top, dst[:T], T, n)
            ops.cast_f64_f32(bottom, dst[T:], Tt, n)
        return self.model.fit_predict_device(X, Y, p, V, n_test_rows=Tt, weights_on_host=True, **model_kwargs)
