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
    """Stories in, metrics out: the Lebel-style train/test paradigm of ``AbstractTrainer.train`` (trainer.py:284-320 with
    :125-262) as ONE pipelined pass, built around the host link (round 4):

    * features: all stories' Lanczos resampling in one launch (``fit_words``; lc_lanczos_interp_stories), then FIR delays
      + trim + per-story ``zs`` + ``nan_to_num`` + float32 cast of ALL stories in one launch straight into the design
      matrix (lc_story_design_f32) -- sums in numpy's order, so the design has the reference's bits;
    * brain data: never copied or z-scored by Python.  The stories' trimmed row ranges go to the fit's own native
      uploader as z-scored jobs (ops.HostRows(zscore=True)): its staging threads compute ``zs`` of each (story, column
      chunk) in the data's own precision -- bit-identical to ``utils.zs`` + the float32 cast of nested_cv.py:99-100 -- and
      4 bytes per value cross the link, story by story into their rows of the (T, V) target matrix, voxel panel by voxel
      panel, while the fit already works on the panels that have landed.

    The result equals ``NestedCVModel.fit_predict`` on the matrices ``structure_train_test`` / the reference's trainer
    builds, bit for bit given the same design (tests/test_gpu_configs.py).

    Voxel shards (round 5): with ``model = NestedCVModel(..., shard=ShardContext(...))`` every rank runs the same
    pipeline on ITS block of voxel columns -- the features (V-independent, ~1 ms of kernels) are resampled and stacked on
    every rank, each rank's uploader stages and z-scores only its own columns of every story (a rank's share of the link
    and of the host's z-scoring is V / G), the fit's collectives are those of DESIGN.md 6 (``single_alpha``: the per-alpha
    sums all-reduced, nested_cv.py:396-400) -- and returns the metrics of ALL voxels with its own block of the weights,
    bit-identical to the one-GPU pipeline (tests/test_gpu_shards.py)."""

    def __init__(self, fir_delays: Sequence[int], trimming: dict, model: Optional[NestedCVModel] = None):
        self.fir_delays = [int(d) for d in fir_delays]
        self.trimming = dict(trimming)
        self.model = model or NestedCVModel("ridge_regression")
        self.last_design = None                        # (dX, T, Tt, p) of the most recent fit (tests)
        self._V_total = None                           # voxels of the whole job (all shards) of the fit being set up

    # ---------------------------------------------------------------- features
    def _feature_rows(self, features):
        """The stories' (downsampled) features as one float64 device matrix, stories concatenated by rows:
        (matrix, per-story row offsets, per-story row counts)."""
        dev = ops.device()
        names = list(features.keys())
        n_in = [int(features[s].shape[0]) for s in names]
        off = np.concatenate([[0], np.cumsum(n_in)]).astype(np.int64)
        ndim = int(features[names[0]].shape[1])
        feat = torch.empty((int(off[-1]), ndim), dtype=torch.float64, device=dev)
        for i, s in enumerate(names):
            f = features[s]
            if f.shape[1] != ndim:
                raise RuntimeError("stories with different numbers of feature columns")
            feat[off[i]:off[i + 1]].copy_(f if torch.is_tensor(f) else _dev_f64(f, dev), non_blocking=True)
        return feat, off, n_in

    def _trimmed_rows(self, n_in, names):
        """Per story the (first, one-past-last) feature row that survives trimming (trainer.py:231-249)."""
        g = self.trimming.get
        a, b = [], []
        for i in range(len(names)):
            kind = "train" if i < len(names) - 1 else "test"
            lo, hi, _ = slice(g(f"{kind}_features_start", 0), g(f"{kind}_features_end", None)).indices(n_in[i])
            if hi - lo < 1:
                raise ValueError("a story is empty after trimming")
            a.append(lo)
            b.append(hi)
        return a, b

    def design(self, feat, off, n_in, names):
        """The float32 design matrix [Rstim ; Pstim] (zero-padded to 32 columns) from the concatenated features: one
        launch (lc_story_design_f32).  Returns (dX, T, Tt, p, per-story number of rows after trimming)."""
        nd, ndim = len(self.fir_delays), feat.shape[1]
        p = nd * ndim
        train = names[:-1]
        a, b = self._trimmed_rows(n_in, names)
        rows = np.asarray(b) - np.asarray(a)
        row0 = np.concatenate([[0], np.cumsum(rows)])[:-1]
        T, Tt = int(rows[:len(train)].sum()), int(rows[len(train):].sum())
        dX = ops.zeros((T + Tt, ops.pad_to(p, 32)), torch.float32, feat.device)
        ops.story_design(feat, off[:-1], n_in, a, b, row0, self.fir_delays, dX)
        return dX, T, Tt, p, rows

    def _voxel_block(self, brain, names):
        """(lo, hi, V_total): this rank's block of voxel columns of the stories' brain arrays and the voxel count of the
        whole job.  One GPU: everything.  Voxel shards (``NestedCVModel(shard=...)``): ``brain`` holds all voxels and the
        rank takes columns [lo, hi) of every story as views -- z-scoring is per voxel (utils.zs, trainer.py:235-257), so a
        rank's columns are the reference's whatever the other ranks hold -- or, with ``local_targets``, ``brain`` holds the
        rank's own block already and the total is the sum over the ranks (one all-reduce before the fit, as
        NestedCVModel.fit_predict does)."""
        shard = self.model.shard
        V = int(np.shape(brain[names[0]])[1])
        if shard is None or shard.world == 1:
            return 0, V, V
        if self.model.local_targets:
            return 0, V, shard.total_of_local_blocks(V)        # (checked on every rank alike: the blocks are bounds()'s)
        lo, hi = shard.bounds(V)
        return lo, hi, V

    def _targets(self, brain, names, rows, lo=0, hi=None):
        """The brain data's trimmed story blocks (columns [lo, hi): this rank's voxels) as the fit's host targets
        (z-scored in the upload threads)."""
        g = self.trimming.get
        blocks = []
        for i, s in enumerate(names):
            kind = "train" if i < len(names) - 1 else "test"
            blk = np.asarray(brain[s])[g(f"{kind}_targets_start", 0):g(f"{kind}_targets_end", None), lo:hi]
            if blk.shape[0] < 1:
                raise ValueError("a story is empty after trimming")
            if blk.shape[0] != rows[i]:
                raise RuntimeError("features and targets have different numbers of rows after trimming")
            blocks.append(blk)
        return ops.HostRows(blocks, zscore=True)

    def fit(self, features: Dict[str, np.ndarray], brain: Dict[str, np.ndarray], **model_kwargs):
        """``features``: per story the (downsampled) feature matrix, host or device; ``brain``: per story the (TRs,
        voxels) host array.  kwargs as ``NestedCVModel.fit_predict``.  Returns (metrics, float32 host weights, alphas)."""
        names = list(features.keys())
        feat, off, n_in = self._feature_rows(features)      # (small, and first on the link: everything waits for the design)
        flying = self._start_targets(brain, names, n_in)
        return self._fit_rows(feat, off, n_in, names, flying, model_kwargs)

    def fit_words(self, words, word_times, tr_times, brain, window=3, cutoff_mult=1.0, **model_kwargs):
        """From word-level features: per story ``words`` (n_words, D) float32 / float64 host arrays at ``word_times``,
        resampled to ``tr_times`` by the Lanczos filter (Downsampler(method="lanczos"), trainer.py:174-201) for all
        stories in one launch, then as ``fit``."""
        dev = ops.device()
        names = list(words.keys())
        blocks = [np.asarray(words[s]) for s in names]
        n_in = [len(tr_times[s]) for s in names]
        if all(b.dtype == np.float32 for b in blocks):
            # the word features cross the link first, the brain data right behind them through the same staging ring --
            # the resampling, the design and the fit's set-up run while the first voxel panel is already on its way
            rows = ops.HostRows(blocks)
            dW = ops.zeros(rows.shape, torch.float32, dev)
            flying = self._start_targets(brain, names, n_in, lead=[(rows, dW, 0, rows.shape[1])])
            flying.wait_lead()
        else:
            flying = self._start_targets(brain, names, n_in)
            dW = _dev_f64(np.concatenate(blocks, axis=0), dev)     # (joined on the host: one upload, no framework cat kernel)
        feat, off = ops.lanczos_interp_stories(dW, [word_times[s] for s in names], [tr_times[s] for s in names],
                                               window, cutoff_mult, False)
        return self._fit_rows(feat, off, n_in, names, flying, model_kwargs)

    def _start_targets(self, brain, names, n_in, lead=()):
        """The brain data's trimmed, z-scored story blocks on their way to the device (NestedCVModel.start_targets)."""
        if len(names) < 2:
            raise ValueError("the train/test paradigm needs at least two stories")
        a, b = self._trimmed_rows(n_in, names)
        lo, hi, self._V_total = self._voxel_block(brain, names)
        return self.model.start_targets(self._targets(brain, names, np.asarray(b) - np.asarray(a), lo, hi),
                                        n_voxels_total=self._V_total, lead=lead)

    def _fit_rows(self, feat, off, n_in, names, flying, model_kwargs):
        dX, T, Tt, p, _ = self.design(feat, off, n_in, names)
        self.last_design = (dX, T, Tt, p)
        return self.model.fit_predict_device(dX, flying, p, flying.shape[1], n_voxels_total=self._V_total, n_test_rows=Tt,
                                             weights_on_host=True, **model_kwargs)
