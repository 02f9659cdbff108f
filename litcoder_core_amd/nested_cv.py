"""Nested-CV ridge fit on the MI355X behind the reference's ``NestedCVModel`` interface.

Interface: ``encoding/models/nested_cv.py:14-42`` (``NestedCVModel.fit_predict``) and the README's
``fit_nested_cv(features=, targets=, ...)`` (``README.md:212-226``; absent from the reference's
code, provided here as the thin alias the README describes).

How the work is laid out (DESIGN.md has the derivation):

* The reference takes a thin SVD of every inner/outer training block and forms, per alpha,
  ``pred = Pstim Vh' diag(S/(S^2+a^2)) U' Rresp``.  With ``K = X X'`` this is exactly
  ``pred = K[va, tr] (K[tr, tr] + a^2 I)^-1 Rresp``: ONE fp64 Gram matrix of all rows serves every
  fold by row/column selection, ``S[0]^2`` is the top eigenvalue of ``K[tr, tr]`` (Lanczos), and the
  per-(fold, alpha) hat matrices come from a batched blocked Cholesky.  No SVD is computed.
* The V-wide work -- hat matrices times the training targets for all alphas of an inner fold,
  z-scoring and correlating against the validation targets -- is one fused f32-MFMA kernel whose
  epilogue reduces each 32-row block to three moments; predictions never reach HBM.
* Refit: voxels are grouped by their selected alpha (counting sort), weights come from a grouped
  MFMA GEMM ``W[:, group] = X_tr'(K + a_g^2 I)^-1 Y[:, group]``, test predictions from a second
  GEMM, and Pearson r from a column-reduction kernel.  Mean weights accumulate on the device.
* The statistics tail (p-values, Fisher, BH-FDR, metrics dict) is vectorised numpy on the host.

Every device operation is a call into liblitcoder_hip.so (``ops.py``); there is no CPU fallback.
"""
import dataclasses
import logging
from typing import Any, Dict, List, Optional, Tuple, Union

import os

import numpy as np
import torch

from . import ops, series, stats
from ._lib import COL_TILE, K_TILE, LC_MB, LC_NB, LC_SCORE_CORR, LC_SCORE_R2
from .dist import ShardContext, job_share
from .folding import create_folds

logger = logging.getLogger(__name__)

SERIES_TERMS = 4                    # terms of the polynomial form of the hat matrices of large alphas (series.py): the
                                    # moments epilogue of the sweep kernel is laid out for exactly four
SINGCUTOFF_REL = 1e-3               # a direction with singular value S <= singcutoff enters a prediction with weight
                                    # S^2 / (S^2 + a^2) <= (singcutoff / a)^2: below 1e-6 it is invisible in fp32
GROUPS_PER_LAUNCH = ops.GROUP_RANGE  # lc_group_by_alpha / the grouped GEMMs carry 64 alpha groups per launch: a larger
                                    # grid (the reference takes any number, ridge_regression.py:46-50,115) goes range by range
MAX_INNER_FOLDS = 64                # inner folds per grouped launch of the series chain / per batch of outer folds prepared together
                                    # (more inner folds than this are taken in chunks: no limit on n_inner_folds)


@dataclasses.dataclass
class FitOptions:
    """Policy switches and tuning values of ONE fit.  Every engine carries its own copy (``NestedCVModel(options=...)``,
    ``RidgeCVEngine(options=...)``): two fits in one process with different settings do not see each other's (until round
    3 these were module-level constants that tests and tools assigned).  The defaults are the measured choices."""
    lanczos_steps: int = 64                 # Lanczos iterations for S[0]^2: <= 1e-11 relative on the cfg2 Grams (profiles/);
                                            # the reference's own S[0] is an fp32 SVD value (~1e-7)
    aug_budget_bytes: int = 24 << 30        # cap on the batched (fold, alpha) fp64 systems resident at once
    series_tol: float = 2e-9                # an alpha takes the polynomial form when its worst relative error over the
                                            # spectrum, 1 / T_d(1 + 2 alpha^2), is <= this: 30x below the fp32 epsilon
    primal_max_scale_ratio: float = 64.0    # primal V-wide route: feature column norms within this factor (fp16x3)
    refit_by_inverse: bool = True           # refit operators through the explicit inverse + one fp16x3 product
    refit_inverse_min_alpha: float = 0.05   # ... for alphas (in units of S[0]) from here on
    refit_inverse_max_world: int = 4        # ... and up to this many voxel-shard ranks
    folds_in_one_launch: bool = False       # one launch per pass for all inner folds of an outer fold (_sweeps): measured
                                            # neutral at 10 000 voxels per rank, 2 ms slower at 80 000 -- off; tested
    series_fused_moments: bool = True       # series terms reduced to moments in the contraction's epilogue (never stored)
    primal_moments_max_p: int = 16          # up to this many features the tall form scores from block products X'Y alone
    primal_max_p: int = 4096                # the primal (p x p) form is taken for tall designs up to this many features
                                            # (round 4: 512 before; LeBel-style train/test fits have 9000 rows x 3072 features)
    primal_series_min_p: int = 256          # from this many (padded) features on the primal form shares the large alphas'
                                            # polynomial terms and takes Gram matrices / block products of the inner training
                                            # sets as sums over the OTHER folds' validation blocks (_prepare_primal)
    speculate_first_fold: bool = True       # the first fold's refit systems for every factorised alpha, beside its chain
    speculate_max_rows: int = 4608          # ... and any refit system ahead of its alpha choice only up to this many rows
    refit_from_image: bool = True           # the refit's alpha-sorted fp16 operand gathered out of the inner CV's image
    panel_cols: int = 36864                 # voxel columns per panel of a host-to-host fit (_column_panels): 12 288 / 24 576 /
                                            # 30 720 / 12 416 at cfg2 (measured 144.1 ms against 145.2 for 24 576-wide panels,
                                            # 145.2 for 73 728, 151.6 without panels)
    panel_min_cols: int = 16384             # below twice this many voxels a fit is not cut into panels
    tail_panels_geometric: bool = True      # the end of a host-to-host fit in few panels of falling width (_download_panels)
    tail_last_frac: float = 0.3             # ... the last panel's share of the voxels (0: the geometric plan's own last panel):
                                            # its weights + transfer (~7 ms) run while the host builds the metrics dictionary
                                            # (measured 140.0 -> 138.1 / 137.8 ms at 0.27 / 0.35)
    tail_folds: int = 2                     # ... spread over this many folds, voxel-major (plan_steps)
    refit_ahead_behind_hat_batch: bool = False     # host inputs: the later folds' refit inverses behind their hat-matrix batch
    series_lookahead: bool = True           # the next step's first sweep part queued before a step's fused sweeps (driver)
    resident_refit_batch: bool = True       # resident inputs: refit inverses of folds 1.. as one batch after fold 0's choice
    shard_first_sweeps_before_batch: bool = False  # voxel shards: fold 0's sweeps queued before the other folds' batch
                                            # (measured: 38.7 vs 38.5 ms per rank of 8 -- no gain): off
    second_fold_own_batch: bool = False     # voxel shards: fold 1's hat matrices as a batch of their own -- measured SLOWER
                                            # (rank 0 of 8: 39.5 vs 37.7 ms, of 4: 54.6 vs 52.6: one more chain latency): off
    refit_ahead_after_first_choice: bool = False  # host inputs: the later folds' refit inverses only for the alphas the first
                                            # panel chose -- measured 2.4 ms SLOWER at cfg2 (148.4 vs 150.8 ms: the batch then
                                            # starts at 18 ms beside full-width sweeps instead of in the upload window): off
    alpha_progress_log: bool = dataclasses.field(       # per-alpha progress lines (ridge_regression.py:136-139): a device
        default_factory=lambda: os.environ.get("LITCODER_AMD_ALPHA_LOG", "0") == "1")   # round trip per fold, opt-in
    chol_outer_block: int = 512             # lc_batch_chol_solve: columns per outer block of the two-level blocking
    chol_big_kernel: int = 2                # ... deep updates: 2 = 4x4x4 fp64 MFMA, 1 = vector ALU, 0 = 16x16x4 MFMA
    chol_fused_steps: bool = True           # ... fused left-looking 64-column steps
    chol_left_deep: bool = False            # ... deep updates left-looking too (measured: no gain)
    chol_persistent: int = 1                # ... bit 0: the back substitution's steps of an outer block in one launch
    lanczos_mfma: bool = True               # lc_lambda_max_masked: the matvec on the fp64 MFMA


def check_penalties(alphas, singcutoff, normalpha, n_inner_folds=None):
    """Host-side validation of the penalty grid, before anything touches the device.  Returns True when the fit has to
    take the SPECTRAL route (csrc/lc_eig.hip) instead of the Cholesky one.

    The reference takes a thin SVD, DROPS singular values <= ``singcutoff`` (ridge_utils.py:44-63) and shrinks the
    rest by S / (S^2 + a^2) (ridge_regression.py:56,117), which is defined for alpha = 0 (pseudo-inverse).  The fast
    route factors (K + a^2 I) by Cholesky: it needs a^2 > 0, and it truncates nothing -- a direction the reference
    would drop contributes at most (singcutoff / a)^2 to a prediction, < 1e-6 (invisible in fp32) whenever
    singcutoff <= 1e-3 a_min, which holds for every shipped caller (singcutoff 1e-10 / 1e-30, alphas >= 0.1).
    Outside that range -- alpha = 0 in the grid, or a singcutoff that is not negligible against the smallest penalty
    (with ``normalpha`` a = alpha S[0] and S[0] is not known yet: not negligible against alpha_min itself) -- the
    operators come from the eigendecomposition of K[tr, tr] with exactly the reference's truncation, in fp64, slower."""
    al = np.asarray(list(alphas), dtype=np.float64).reshape(-1)
    if al.size == 0:
        raise ValueError("alphas is empty")
    if n_inner_folds is not None and int(n_inner_folds) < 1:
        raise ValueError("n_inner_folds must be >= 1")
    if not np.all(np.isfinite(al)):
        raise ValueError("alphas must be finite (the penalty is alpha^2: ridge_regression.py:56,117)")
    al = np.abs(al)                                    # ... so a negative alpha is the penalty of |alpha|
    sc = float(singcutoff)
    if not (sc >= 0) or not np.isfinite(sc):
        raise ValueError("singcutoff must be a finite number >= 0")
    pos = al[al > 0]
    if pos.size < al.size:
        return True                                    # alpha = 0: the pseudo-inverse of the kept directions
    return bool(sc > (1e-6 if normalpha else SINGCUTOFF_REL) * float(pos.min()))


class _PrimalUnsuitable(Exception):
    """Raised while preparing a fit in the primal form when the data rule it out; the driver falls back to the dual."""


class BasePredictivityModel:
    """``encoding/models/base.py:7-41``: the interface AbstractTrainer calls."""

    def __init__(self, model_name: str):
        self.model_name = model_name

    def fit_predict(self, features, targets, groups=None, **kwargs):  # pragma: no cover - interface
        raise NotImplementedError


class _FoldResult:
    __slots__ = ("r", "p", "best_idx", "n_test", "sig")

    def __init__(self, r, p, best_idx, n_test, sig=None):
        self.r, self.p, self.best_idx, self.n_test = r, p, best_idx, n_test
        self.sig = sig             # (reject mask, adjusted p) of the fold when the device made them (one GPU), else None


_AUX_STREAMS: Dict[Any, Any] = {}


def _aux_stream(dev, which=0):
    """The auxiliary streams of a device, created once for the life of the process.  A fresh ``torch.cuda.Stream()`` per
    engine walks through torch's pool of 32 streams, and the FIRST cross-stream wait on a stream that has never run
    anything blocks the host for ~6 ms (its hardware queue is created there): every fit of a series paid that before
    its first fold was queued."""
    key = (dev.type, dev.index, which)
    if key not in _AUX_STREAMS:
        _AUX_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _AUX_STREAMS[key]


_MAIN_STREAMS: Dict[Any, Any] = {}


def _main_stream():
    """The stream the V-wide (MFMA) work of a fit runs on: restricted to CUs 0-223 of an MI355X, so that 32 CUs stay
    free for the auxiliary stream's short fp64 kernels (diagonal tiles, panels: ~70 launches per Cholesky batch), which
    otherwise each wait for a whole MFMA-sweep workgroup to retire before they find a slot.  Measured on the cfg2 fit
    (profiles/experiments/README.md): 167-169 ms unrestricted, 162 ms with this mask; holding back 16 or 8 CUs 163 ms,
    48-64 CUs no gain, and CU masks that cut through a group of 8 can be pathological (250 ms) -- hence one fixed,
    measured mask, only on a device with 256 CUs.  OPT-IN (LITCODER_AMD_CU_MASK=1): the sweep's launches get 256/224
    longer on 224 CUs, i.e. the fit trades 3 % of wall time against 0.06 of the dominant kernel's whole-chip roofline
    fraction (0.44 -> 0.38); the default keeps the kernel on the whole chip."""
    import os
    if os.environ.get("LITCODER_AMD_CU_MASK", "0") != "1":
        return None
    dev = ops.device()
    key = (dev.type, dev.index)
    if key not in _MAIN_STREAMS:
        stream = None
        try:
            if torch.cuda.get_device_properties(dev).multi_processor_count == 256:
                stream = ops.masked_stream([0xFFFFFFFF] * 7 + [0])
        except Exception as exc:  # noqa: BLE001 -- an optimisation only: fall back to the caller's stream
            logger.info("no CU-masked main stream (%s): the fit runs on the caller's stream", exc)
        _MAIN_STREAMS[key] = stream
    return _MAIN_STREAMS[key]


class _WideTargets(Exception):
    """precision="auto" met a target column whose dynamic range the fp16 hi/lo split cannot carry AFTER the fit was set up
    for it (host inputs arrive panel by panel, so the decision cannot be taken up front): the driver repeats the fit on
    the f32 MFMA path with the targets that are resident by then."""


class _Range:
    """A contiguous range [c0, c0 + V) of this rank's voxel columns: the unit a V-wide phase of a fold works on.  The
    targets and the mean weights of the rank live in ONE (T, Vp) / (p, Vp) buffer each; a range sees column views of
    them (row stride = the buffer's), so a fold can be processed full width or panel by panel -- panels while the
    targets are still arriving from the host (first fold) and while the finished weights leave for it (last fold).
    Interior boundaries are multiples of 256 columns (the widest column tile), so only the last range carries padding."""
    __slots__ = ("c0", "V", "Vp", "Y", "W", "scales", "natural", "key")

    def __init__(self, c0, V, Vp, Y, W):
        self.c0, self.V, self.Vp, self.Y, self.W = int(c0), int(V), int(Vp), Y, W
        self.scales = None             # (cs, split) of the un-normalised targets of the range (_target_scales)
        self.natural = None            # 0 .. V-1 on the device (moments form)
        self.key = (self.c0, self.V)


def _column_panels(V, cols=None, min_cols=None, v_ref=None):
    """[c0, c1) panels of V voxel columns for a host-to-host fit, boundaries on multiples of 256: ``cols`` wide in the
    middle, ramping up from cols / 3 at the front (the first fold starts on the first panel while the others still
    cross PCIe: a narrow one is there early) and down to <= cols / 3 at the end (the last panel's weights are the only
    download nothing overlaps).  With the default width the panels of the sweeps' 8 M-tiles are whole rounds of
    workgroups on 256 CUs (8192 columns = one round).  ``v_ref``: the column count the PLAN is derived from (voxel
    shards: the narrowest rank's, so that every rank cuts its block into the same number of panels -- the ranks'
    collectives pair up range by range); the last panel absorbs the difference."""
    cols = FitOptions.panel_cols if cols is None else int(cols)
    min_cols = FitOptions.panel_min_cols if min_cols is None else int(min_cols)
    v_ref = int(V) if v_ref is None else min(int(v_ref), int(V))
    if cols % 256:
        raise ValueError("panel width must be a multiple of 256 columns")
    if v_ref < 2 * min_cols or v_ref <= cols:
        return [(0, int(V))]
    third = max(256, (cols // 3) // 256 * 256)
    widths, left = [], v_ref
    for w in (third, 2 * third):                           # ramp up
        if left > w + third:
            widths.append(w)
            left -= w
    while left > cols + third:                             # full panels
        widths.append(cols)
        left -= cols
    if left > 2 * 256:                                     # ramp down: what is left, minus a narrow last panel
        tail = min(third, (left // 2) // 256 * 256)
        body = (left - tail) // 256 * 256
        widths += [body, left - body]
    else:
        widths.append(left)
    edges = np.concatenate([[0], np.cumsum(widths)]).astype(np.int64)
    edges[-1] = int(V)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(len(widths))]


def _download_panels(V, first=8.0 / 15.0, ratio=0.5, min_cols=None, v_ref=None, last_min=4096, last_frac=0.0):
    """[c0, c1) panels the END of a host-to-host fit works in (the last two folds voxel-major, plan_steps): a panel's
    finished weights cross PCIe while the next panel is computed, so what is not hidden is the LAST panel's transfer --
    and wide panels run the V-wide kernels more efficiently than narrow ones.  Widths fall geometrically: two folds of
    V-wide work on a panel take ~2.2x its transfer time (cfg2: 40 ms of work, 18 ms of PCIe for all voxels), so with
    ``ratio`` = 1/2 every transfer ends before the next panel's work does, and the tail is the transfer of 1/15 of the
    voxels (~1.2 ms) with four panels instead of five equal ones.  Boundaries on multiples of 256; ``v_ref`` as in
    _column_panels (the same number of panels on every rank of a sharded fit)."""
    min_cols = FitOptions.panel_min_cols if min_cols is None else int(min_cols)
    v_ref = int(V) if v_ref is None else min(int(v_ref), int(V))
    if v_ref < 2 * min_cols:
        return [(0, int(V))]
    last_min = max(256, min(int(last_min), min_cols // 4))
    if last_frac > 0.0:
        # ... unless the caller has host work of its own after the last fold's results (the metrics dictionary: ~7 ms at
        # cfg2, during which the GPU would idle): then the LAST panel is sized so that its weights, their mean and its
        # transfer take about that long -- what runs after the last results are out is hidden behind the host, and the
        # results themselves are out that much earlier
        tail = max(last_min, int(round(v_ref * last_frac / 256.0)) * 256)
        head = max(256, (v_ref - tail) // 256 * 256)
        h1 = max(256, int(round(head * 0.6 / 256.0)) * 256)
        edges = [0, h1, head, int(V)] if head - h1 >= last_min else [0, head, int(V)]
        return [(int(edges[i]), int(edges[i + 1])) for i in range(len(edges) - 1)]
    widths, left, w = [], v_ref, v_ref * first
    while left > 0:
        wi = max(256, int(round(w / 256.0)) * 256)
        if left - wi < last_min or wi < last_min:
            widths.append(left)
            break
        widths.append(wi)
        left -= wi
        w *= ratio
    edges = np.concatenate([[0], np.cumsum(widths)]).astype(np.int64)
    edges[-1] = int(V)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(len(widths))]


class RidgeCVEngine:
    """Device-resident state of one fit: fp32 copies of X / Y (zero padded), the Gram matrix, and the
    per-fold pipeline.  ``Y`` holds only this rank's voxel block."""

    def __init__(self, X_all, Y_all, alphas, normalpha, use_corr, normalize_features, normalize_targets,
                 shard: Optional[ShardContext] = None, lanczos_steps: Optional[int] = None, precision: str = "auto",
                 singcutoff: float = 0.0, V_total: Optional[int] = None, min_train_rows: Optional[int] = None,
                 form: str = "dual", panels=None, options: Optional[FitOptions] = None, down_panels=None):
        """``form``: "dual" (n x n Gram / hat matrices: every shape), "primal" (p x p systems, see _prepare_primal) or
        "auto" = primal when the design is tall, 2 p <= ``min_train_rows`` (the smallest inner training set) and
        p <= FitOptions.primal_max_p.  ``Y_all``: a host array / ops.HostRows (uploaded in the column ``panels`` [(c0, c1), ...] on a
        background thread while the fit is being set up) or resident targets (_DeviceShapes)."""
        self.opt = dataclasses.replace(options) if options is not None else FitOptions()    # this engine's own copy
        self._chol_opt = ops.chol_options(self.opt.chol_outer_block, self.opt.chol_big_kernel, self.opt.chol_fused_steps,
                                          self.opt.chol_left_deep, self.opt.chol_persistent)
        self.spectral = check_penalties(alphas, singcutoff, normalpha)
        self.singcutoff = float(singcutoff)
        self.dev = ops.device()
        self.shard = shard or ShardContext.single()
        if not isinstance(X_all, _DeviceShapes):
            X_all = np.asarray(X_all)
        if not isinstance(Y_all, (_DeviceShapes, ops.HostRows)):
            Y_all = ops.HostRows([Y_all])
        self.Ttot, self.p = X_all.shape
        self.V_rank = Y_all.shape[1]                   # voxel columns of this rank (all its ranges together)
        if Y_all.shape[0] != self.Ttot:
            raise RuntimeError(f"shape mismatch: features have {self.Ttot} rows, targets {Y_all.shape[0]}")
        self.p_pad = ops.pad_to(self.p, K_TILE)
        self.Vp_rank = ops.pad_to(max(self.V_rank, 1), COL_TILE)
        # the penalty is alpha^2 (ridge_regression.py:56,117): a negative grid value IS |alpha| for every operator; the
        # caller's own values (sign included) come back in best_alphas (_alpha_vector works on the caller's grid)
        self.alphas = [abs(float(a)) for a in alphas]
        self.A = len(self.alphas)
        self.normalpha = bool(normalpha)
        self.mode = LC_SCORE_CORR if use_corr else LC_SCORE_R2
        self.norm_x, self.norm_y = bool(normalize_features), bool(normalize_targets)
        self.steps = int(lanczos_steps if lanczos_steps is not None else self.opt.lanczos_steps)
        if precision not in ("auto", "f32", "f16x3"):
            raise ValueError(f"precision must be 'auto', 'f32' or 'f16x3', got {precision!r}")
        self.precision = precision
        if form not in ("dual", "primal", "auto"):
            raise ValueError(f"form must be 'dual', 'primal' or 'auto', got {form!r}")
        self.primal = form == "primal" or (form == "auto" and min_train_rows is not None
                                           and 2 * self.p <= int(min_train_rows) and self.p <= self.opt.primal_max_p)
        if self.spectral:
            # alpha = 0 / a biting singcutoff: the reference's truncated SVD, reproduced from the eigendecomposition of
            # the n x n Gram blocks (dual form for every shape; see check_penalties and _spectral_operators)
            logger.info("penalty grid outside the Cholesky route (alpha = 0 or singcutoff not negligible): spectral route")
            self.primal = False
        # primal: padded system size (whole 128-column tiles from 256 features on: the polynomial chain's f32 / fp16x3 GEMMs)
        self.PP = ops.pad_to(self.p, COL_TILE if self.p >= self.opt.primal_series_min_p else LC_NB)
        # a handful of features + correlation scoring: the whole nested CV from block products X'Y (_prepare_moments)
        self.moments = self.primal and self.p <= self.opt.primal_moments_max_p and bool(use_corr)
        # ---- the targets: resident already, or arriving from the host panel by panel on a background thread (started
        # FIRST: everything below -- the design, its Gram matrix, the first fold's operators -- runs beside it)
        self.uploader = None
        self.upload_panels = [(0, self.V_rank)]
        self.download_panels = [(0, self.V_rank)]      # ranges the end of the fit works in when the weights go to the host
        jobs = []
        if isinstance(X_all, _DeviceShapes):
            self.dX = self._resident(X_all, self.p_pad)
        else:
            self.dX = ops.zeros((self.Ttot, self.p_pad), torch.float32, self.dev)
            if self.Ttot and self.p:
                jobs.append((X_all, self.dX, 0, self.p))
        self._x_job = 0 if jobs else None
        if isinstance(Y_all, _DeviceShapes):
            self.dY_full = self._resident(Y_all, self.Vp_rank)
        else:
            self.dY_full = torch.empty((self.Ttot, self.Vp_rank), dtype=torch.float32, device=self.dev)
            ops.zero_cols(self.dY_full, self.V_rank, self.Vp_rank)
            if self.V_rank and self.Ttot:
                self.upload_panels = [(int(a), int(b)) for a, b in (panels or [(0, self.V_rank)])]
                self.download_panels = ([(int(a), int(b)) for a, b in down_panels] if down_panels
                                        else list(self.upload_panels))
                self._y_job0 = len(jobs)
                jobs += [(Y_all, self.dY_full, a, b) for a, b in self.upload_panels]
        if jobs:
            zeroed = torch.cuda.Event()
            zeroed.record()
            self.uploader = ops.PanelUploader(jobs, self.dev, after=zeroed)
        self.W_full = ops.zeros((self.p, self.Vp_rank), torch.float32, self.dev)
        self.full = _Range(0, self.V_rank, self.Vp_rank, self.dY_full, self.W_full)
        self.cur = self.full                           # the range the V-wide phase being queued works on (_enter)
        self._ranges = {self.full.key: self.full}
        # (the host-side set-up below -- polynomial coefficients, index tables -- runs while the design is crossing PCIe)
        self.d_alphas = ops.upload(np.asarray(self.alphas, dtype=np.float64), self.dev)
        # alphas whose penalty dwarfs the spectrum take the polynomial form of the inverse (shared matrix powers,
        # minimax coefficients: series.py), the rest the batched Cholesky.  Needs normalpha (a^2 = alpha^2 lambda_max
        # makes the coefficients a function of alpha alone).
        # (primal form: a handful of features -> every alpha is a tiny p x p factorisation; from primal_series_min_p
        # features on the polynomial in G / lambda_max shares its terms exactly as the one in K / lambda_max does)
        self.primal_series = (self.primal and not self.moments and self.PP % COL_TILE == 0
                              and self.PP >= self.opt.primal_series_min_p)
        self.ser = [a for a in range(self.A) if (not self.primal or self.primal_series)
                    and not self.spectral                             # spectral: every alpha from the eigenpairs
                    and self.normalpha and series.residual_bound(self.alphas[a], SERIES_TERMS) <= self.opt.series_tol]
        self.cho = [a for a in range(self.A) if a not in self.ser]
        self.d_ser = ops.upload(np.asarray(self.ser, dtype=np.int32), self.dev) if self.ser else None
        self.coef_host = (np.stack([series.minimax_inverse_coefficients(self.alphas[a], SERIES_TERMS)
                                    for a in self.ser]) if self.ser else None)
        self.d_coef = ops.upload(np.asarray(self.coef_host, dtype=np.float64), self.dev) if self.ser else None
        self.d_cho = ops.upload(np.asarray(self.cho, dtype=np.int32), self.dev)
        self.aux = _aux_stream(self.dev)
        self.aux2 = _aux_stream(self.dev, 1)            # refit systems (see _refit_stream)
        self.comm = _aux_stream(self.dev, 2)            # per-fold result exchange + global statistics
        self.aux3 = _aux_stream(self.dev, 3)            # voxel shards: what a fold's refit still needs after refit_ahead
        self.dl = _aux_stream(self.dev, 4)              # finished weight panels on their way to the host
        self.scales_stream = _aux_stream(self.dev, 5)   # column scales of target panels as they arrive (_target_scales)
        # voxel shards: this rank's block is columns [lo[rank], lo[rank + 1]) of V_total; the statistics tail (BH-FDR
        # ranks ALL p-values) runs on the gathered vectors, on the device, on every rank; the driver sets alpha_fdr
        self.V_total = int(V_total) if V_total is not None else self.V_rank
        lo = self.shard.all_bounds(self.V_total)
        if int(lo[self.shard.rank + 1] - lo[self.shard.rank]) != self.V_rank:
            raise ValueError(f"rank {self.shard.rank} of {self.shard.world} holds {self.V_rank} voxel columns, its block of "
                             f"{self.V_total} has {int(lo[self.shard.rank + 1] - lo[self.shard.rank])}")
        self.w_max = int(np.max(np.diff(lo)))
        self.d_lo = ops.upload(lo, self.dev)
        self.alpha_fdr = 0.05
        self.p_folds = None                            # (n_folds, V_total) NaN-free p-values of all voxels, device
        self.n_folds = 1
        self._fold_blk = {}                            # fold -> the rank's packed (4, ld) result block being filled
        self.sweeps_done = None                        # end of the sweeps queued last (chain_gate)
        self._host_weights = None                      # future of the page-locked result buffer (reserve_host_weights)
        self._host_w = None                            # ... the buffer itself once panels are leaving for it
        self._sent = 0                                 # voxel columns of the weights already on their way to the host
        self._cs_all, self._cs_known = None, None      # column scales of the target panels that have arrived (_target_scales)
        self._ws = {}                                  # fold -> its alpha-sorted weight matrix + where each voxel went (_ws_slot)
        self._combined = 0                             # voxel columns whose mean weights are final (_combine_weights)
        self._assume_split = None                      # the arithmetic the operators are prepared for (_split_assumed)
        self._decided = False                          # ... decided from ALL resident target columns (begin_fit)
        # constants of the fit that every stream reads: made here, before ``ready``
        self._d_one = ops.upload(np.ones(1, dtype=np.float64), self.dev)
        self._eye, self._eye_key = None, None
        self._eig_cache, self._n_real = {}, {}         # spectral route: eigenpairs of a fold's outer block; list lengths
        self._scale_checks = []                        # primal form: pending looks at the features' column norms
        # what this fit ran, for the caller (NestedCVModel.last_fit; bench.py prices the roofline with it): arithmetic
        # of the sweeps, alphas scored inside the fused launch, algorithmic flops of the plain fp16x3 GEMMs.  Per
        # engine: two fits in one process do not share it.
        self.info = {"precision": None, "fused_alphas": self.A, "series_terms": 0, "plain_flops": 0.0,
                     "plain_launches": 0, "used_all": None, "fused_flops": 0.0, "fused_launches": 0}
        if self.uploader is not None:
            if self._x_job is not None:
                self.uploader.wait(self._x_job)        # the design is needed now (Gram matrix)
            if len(jobs) == (1 if self._x_job is not None else 0):
                self.uploader.join()                   # resident targets: nothing arrives later
                self.uploader = None
            elif panels is None:
                self.finish_uploads()                  # no panel plan: the caller (tests, ridge.py) uses the targets at once
        self.K = None if (self.norm_x or self.primal) else ops.gram(self.dX, self.Ttot, self.p)
        self.ready = torch.cuda.Event()               # X, K resident: the only thing the aux stream waits for
        self.ready.record()

    # the V-wide phases read the voxel range they work on through these (see _enter)
    V = property(lambda self: self.cur.V)
    Vp = property(lambda self: self.cur.Vp)
    dY = property(lambda self: self.cur.Y)
    W_acc = property(lambda self: self.cur.W)

    def _resident(self, arr, ld):
        if isinstance(arr, _DeviceShapes):  # already resident: fp32, contiguous, zero-padded to the tile width
            t = arr.tensor
            if t.dtype != torch.float32 or not t.is_cuda or t.shape[1] != ld or not t.is_contiguous():
                raise ValueError(f"device inputs must be contiguous fp32 tensors with {ld} (zero-padded) columns")
            return t
        return ops.upload_f32(arr, ld, self.dev)

    # -------------------------------------------------------------- voxel ranges
    def range_of(self, c0, c1):
        """The _Range of columns [c0, c1) of this rank's block (cached: its column scales are computed once)."""
        c0, c1 = int(c0), int(c1)
        key = (c0, c1 - c0)
        if key not in self._ranges:
            if not (0 <= c0 < c1 <= self.V_rank) or c0 % 256 or (c1 % 256 and c1 != self.V_rank):
                raise ValueError("voxel ranges must start and end on multiples of 256 columns (the last one at V)")
            vp = (c1 - c0) if c1 != self.V_rank else self.Vp_rank - c0
            self._ranges[key] = _Range(c0, c1 - c0, vp, self.dY_full[:, c0:c0 + vp], self.W_full[:, c0:c0 + vp])
        return self._ranges[key]

    def _enter(self, st):
        """Make the range of a fold state the one the engine's V-wide methods see (V, Vp, dY, W_acc)."""
        self.cur = st["rg"]
        return st

    def _wait_targets(self, rg, stream=None):
        """Host inputs: the upload panels that cover the range have been issued (host) and the given (default: current)
        stream waits for their copies (device)."""
        if self.uploader is None:
            return
        for b, (c0, c1) in enumerate(self.upload_panels):
            if c0 < rg.c0 + rg.V and rg.c0 < c1:
                self.uploader.wait(self._y_job0 + b, stream)

    def plan_steps(self, n_folds, single_alpha=False, ahead=False):
        """The (fold, range) steps of the fit in execution order.  Folds are processed full width, except:
          * while the targets arrive from the host the first fold works panel by panel (a panel's sweeps start when ITS
            columns are resident);
          * when the weights go back to the host (0.98 GB at cfg2: ~18 ms of PCIe) the END of the fit runs panel by
            panel, so that a panel's finished weights leave while the next panel is computed: the last TWO folds
            voxel-major -- (n-2, panel), (n-1, panel), next panel -- when every fold's operators exist ahead of the
            choices (``ahead``: the panels then finish spread over two folds of work, which hides the transfer behind
            a few wide panels), else the last fold alone.
        ``single_alpha`` needs the scores of all voxels before any refit: full width throughout."""
        full = [(0, self.V_rank)]
        paneled = len(self.upload_panels) > 1 and not single_alpha
        up = self.upload_panels if paneled else full
        down = self.download_panels if (len(self.download_panels) > 1 and not single_alpha
                                        and self._host_weights is not None) else full
        tail = max(1, min(int(self.opt.tail_folds), 2)) if (ahead and n_folds >= 3 and len(down) > 1) else 1
        plan = []
        for f in range(n_folds - (tail if len(down) > 1 else 0)):
            for c in (up if (f == 0 and self.uploader is not None) else full):
                plan.append((f, c))
        if len(down) > 1:
            first_tail = n_folds - tail
            for c in down:
                for f in range(first_tail, n_folds):
                    plan.append((f, c))
        return plan

    # -------------------------------------------------------------- per-outer-fold data
    def _fold_design(self, tr_rows):
        """Train-statistics z-scoring of X for this outer fold (DataNormalizer, ridge_utils.py:70-180;
        nested_cv.py:111-124,204-213) and the matching Gram matrix: the V-independent half of the fold's data."""
        X, K = self.dX, self.K
        if self.norm_x:
            rows = ops.idx_tensor(tr_rows, len(tr_rows), self.dev)
            mean, std = ops.col_mean_std(self.dX, rows, len(tr_rows), self.p)
            X = self.dX.clone()
            ops.col_normalize_(X, self.Ttot, self.p, mean, std)
            K = None if self.primal else ops.gram(X, self.Ttot, self.p)
        return X, K

    def _fold_targets(self, rg, tr_rows):
        """(Y, cs, split) of a voxel range for one outer fold: the resident targets, or -- normalize_targets -- their
        train-statistics z-scored copy with column scales of its own (per-fold state: folds are pipelined)."""
        Y = rg.Y
        if self.norm_y:
            rows = ops.idx_tensor(tr_rows, len(tr_rows), self.dev)
            mean, std = ops.col_mean_std(rg.Y, rows, len(tr_rows), rg.V)
            Y = rg.Y.clone()
            ops.col_normalize_(Y, self.Ttot, rg.V, mean, std)
        if self.moments:
            return Y, None, False            # fp64 block products: no fp16 operands, no column scales
        cs, split = self._target_scales(Y, rg)
        return Y, cs, split

    def _split_assumed(self):
        """The arithmetic the V-independent operators are prepared for before any target value has been looked at:
        f16x3 unless the caller asked for f32 (a range of "auto" that turns out too wide raises _WideTargets)."""
        if self._assume_split is None:
            self._assume_split = self.precision != "f32"
        return self._assume_split

    def _target_scales(self, Y, rg=None):
        """(cs, split) for one target matrix: ``split`` = the V-wide contractions run as "f16x3" -- fp16 hi + lo
        operands after an exact power-of-two scale per H row / Y column, three fp16 MFMAs per product, fp32
        accumulate (22-bit operands: fp32-level scores, ~3x faster than the f32-input MFMA) -- and ``cs`` the
        (2 Vp,) column scales that go with it (2^-e, then 2^e).  "auto" takes the split unless a target column is
        non-finite or dominated by outliers (most entries > 2^9 below the column maximum).  The scales belong to
        the VALUES of ``Y``: with normalize_targets every outer fold has its own (fold state, never engine state:
        folds are pipelined over streams); only those of the resident, un-normalised targets are cached (per range).
        The flag is agreed over the voxel shards (MAX all-reduce on the device) BEFORE the host looks at it: the
        arithmetic decides which collectives _hat_matrices issues, and every rank must issue the same ones."""
        rg = rg or self.cur
        if self.precision == "f32":
            return None, False
        if Y is rg.Y and rg.scales is not None:
            return rg.scales
        if Y is rg.Y and self._cs_known is not None and bool(self._cs_known[rg.c0 // 256:(rg.c0 + rg.Vp + 255) // 256].all()):
            # every column of the range belongs to a range whose scales exist (the end of a host-to-host fit works in
            # other panels than its beginning): the values are per column -- two slices of the engine-wide table
            cs = torch.empty(2 * rg.Vp, dtype=torch.float32, device=self.dev)
            cs[:rg.Vp].copy_(self._cs_all[0, rg.c0:rg.c0 + rg.Vp])
            cs[rg.Vp:].copy_(self._cs_all[1, rg.c0:rg.c0 + rg.Vp])
            rg.scales = (cs, True)
            return rg.scales
        check = self.precision == "auto" and not (self._decided and Y is rg.Y)
        if check and self.uploader is not None and Y is rg.Y:
            # targets still arriving from the host: the scales and the flag of a range on a stream of their own, which
            # waits for the range's upload panels only -- looking at the flag on the main stream would make the host
            # wait for everything queued there (the previous panel's sweeps), once per panel of the first fold
            main = torch.cuda.current_stream()
            side = self.scales_stream
            self._wait_targets(rg, side)
            with torch.cuda.stream(side):
                cs, flag = ops.col_scales_f16(Y, self.Ttot, rg.Vp)
                self.shard.all_reduce_(flag, "max")
                flag_h = torch.empty(1, dtype=torch.int32, pin_memory=True)
                flag_h.copy_(flag, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            cs.record_stream(main)
            ev.synchronize()
            main.wait_event(ev)
            wide = bool(int(flag_h[0]))
        else:
            cs, flag = ops.col_scales_f16(Y, self.Ttot, rg.Vp)
            wide = False
            if check:
                self.shard.all_reduce_(flag, "max")
                wide = bool(int(flag.cpu()[0]))
        if wide:
            if self._assume_split:
                raise _WideTargets("target dynamic range too wide for the fp16x3 sweep")
            logger.info("target dynamic range too wide for the fp16x3 sweep: using the f32 MFMA path")
        out = (cs, not wide)
        if Y is rg.Y:
            rg.scales = out
            if not wide and self.uploader is not None:       # (current stream = the one every V-wide phase is queued on)
                if self._cs_all is None:
                    self._cs_all = torch.empty((2, self.Vp_rank), dtype=torch.float32, device=self.dev)
                    self._cs_known = np.zeros((self.Vp_rank + 255) // 256, dtype=bool)
                self._cs_all[0, rg.c0:rg.c0 + rg.Vp].copy_(cs[:rg.Vp])
                self._cs_all[1, rg.c0:rg.c0 + rg.Vp].copy_(cs[rg.Vp:])
                self._cs_known[rg.c0 // 256:(rg.c0 + rg.Vp + 255) // 256] = True
        return out

    # -------------------------------------------------------------- S[0]^2 of every train set (Lanczos)
    def lmax_systems(self, K, row_sets):
        """lambda_max(K[I, I]) for every row set I: all sets are principal submatrices of the one Gram matrix, so
        they share a single pass over K per Lanczos iteration, 32 systems per launch chain."""
        res = torch.empty(len(row_sets), dtype=torch.float64, device=self.dev)
        for c0 in range(0, len(row_sets), 32):
            chunk = row_sets[c0:c0 + 32]
            bits = np.zeros(self.Ttot, dtype=np.uint32)
            for f, rows in enumerate(chunk):
                bits[np.asarray(rows, dtype=np.int64)] |= np.uint32(1 << f)
            member = ops.upload(bits.view(np.int32), self.dev)
            ops.lambda_max_masked(K, self.Ttot, member, len(chunk), self.steps, out=res[c0:c0 + len(chunk)],
                                  use_mfma=self.opt.lanczos_mfma)
        return res

    def _check_singcutoff(self, lmax):
        """(Round 3: a singcutoff that could bite takes the spectral route from the start -- check_penalties -- so there
        is nothing left to verify against the measured S[0]; kept as the hook the callers have.)"""
        return

    def begin_fit(self, n_folds=1):
        """Decide the arithmetic of the V-wide contractions now (column scales of the targets + the one flag that
        comes to the host), so that the first fold's set-up is enqueued without waiting on the device."""
        if not self.norm_y and not self.moments and self.uploader is None and self.precision == "auto":
            # resident targets: one look at all columns decides the arithmetic of the whole fit, ranges included
            _, split = self._target_scales(self.dY_full, self.full)
            self._assume_split, self._decided = split, True
            if not split:
                self.precision = "f32"
        self.p_folds = torch.empty((int(n_folds), self.V_total), dtype=torch.float64, device=self.dev)
        self.n_folds = int(n_folds)
        self._fold_blk = {}

    def _join_flags(self, parts):
        """One int32 vector from the pivot-flag vectors of several batches (D2D copies, no framework kernel)."""
        if not parts:
            return ops.zeros(1, torch.int32, self.dev)
        if len(parts) == 1:
            return parts[0]
        out = torch.empty(sum(int(p.numel()) for p in parts), dtype=torch.int32, device=self.dev)
        o = 0
        for p in parts:
            out[o:o + p.numel()].copy_(p)
            o += p.numel()
        return out

    def _cs_inv_padded(self, cs, Vt, V=None):
        """The 2^e column scales padded to the plain GEMM's 256-column tiles (padding columns are never read back)."""
        V = self.Vp if V is None else V
        out = ops.zeros(Vt, torch.float32, self.dev)
        out[:V].copy_(cs[V:])
        return out

    # -------------------------------------------------------------- V-independent fp64 systems, dealt out over ranks
    def _sharded_solve(self, n_jobs, N, M, assemble, out=None, slot=None, lane="hat", inverse=False):
        """``n_jobs`` independent augmented systems (same list, same order on every rank): rank r factors jobs
        [r n_per, (r + 1) n_per), n_per = ceil(n_jobs / world), and the f32 results are all-gathered -- on return
        ``H`` (>= n_jobs, M, N) is complete on every rank, job j in slot j.  ``assemble(jobs)`` builds the
        (len(jobs), N + M, N) fp64 batch of the listed jobs.  Returns (H, pivot flags of THIS rank's jobs).
        One rank: the whole batch, no copy, no collective -- and with ``out`` / ``slot`` (int32 device vector) job j
        is written straight to out[slot[j]] (the f32 / R2 paths keep the series alphas' hat matrices in the same
        buffer); with several ranks the caller places the gathered blocks itself."""
        G = self.shard.world
        n_per, mine = job_share(n_jobs, G, self.shard.rank)
        mine = list(mine)
        direct = not self.shard.active and out is not None
        H = out if direct else torch.empty((n_per, M, N), dtype=torch.float32, device=self.dev)
        if mine:
            aug = assemble(mine)
            if inverse:                                 # bottom block = identity, M == N: the explicit inverse
                info = ops.batch_chol_inverse(aug, len(mine), N, H, slot if direct else None, options=self._chol_opt)
            else:
                info = ops.batch_chol_solve(aug, len(mine), N, M, H, slot if direct else None, options=self._chol_opt)
            del aug
        else:
            info = ops.zeros(1, torch.int32, self.dev)
        if self.shard.active:
            H = self.shard.all_gather(H, lane=lane).view(G * n_per, M, N)
        return H, info

    def precompute_lmax(self, outer):
        """(inner-fold lmax (F,), outer-train lmax (1,)) per outer fold from ONE Lanczos run over the shared Gram
        matrix; [None, ...] when there is nothing to share (no normalpha, or normalize_features gives every
        outer fold its own Gram matrix -- fold_prepare then runs the fold's systems by itself).  The inner-fold
        values of consecutive outer folds are neighbours in one vector (prepare_folds takes slices spanning folds)."""
        if not self.normalpha or self.norm_x or self.primal:
            return [None] * len(outer)
        inner_sets, outer_sets, spans = [], [], []
        for tr_rows, _, inner_rel in outer:
            tr_rows = np.asarray(tr_rows, dtype=np.int64)
            spans.append((len(inner_sets), len(inner_rel)))
            inner_sets += [tr_rows[np.asarray(a, dtype=np.int64)] for a, _ in inner_rel]
            outer_sets.append(tr_rows)
        # on the AUXILIARY stream, where every consumer of these values runs
        self.ready.record()                           # X, Y, K resident
        self.aux.wait_event(self.ready)
        with torch.cuda.stream(self.aux):
            lm = self.lmax_systems(self.K, inner_sets + outer_sets)
            self._check_singcutoff(lm)
        n_in = len(inner_sets)
        self._lm_inner = lm[:n_in]
        return [(lm[s:s + n], lm[n_in + i:n_in + i + 1]) for i, (s, n) in enumerate(spans)]

    # -------------------------------------------------------------- inner CV: hat matrices, then the sweeps
    def _series_layout(self, M):
        """Row layout of the stacked series terms for the plain fp16x3 GEMM: every term padded to whole 128-row
        slabs, heavy slabs (terms 0 and 1: full three-MFMA products) paired with light ones (terms >= 2, which
        enter a prediction scaled by rho^2 <= 2.7e-4 relative to term 0 and only need fp16 operands) inside the
        256-row tiles, so that the two waves of a SIMD together issue 32 instead of 48 MFMAs per K-tile.
        Returns (rows, rowmap (terms*M,) int32 device, slab_light uint8 device)."""
        key = ("series_layout", M, self.opt.series_fused_moments)
        if getattr(self, "_layout_key", None) != key and self.opt.series_fused_moments and SERIES_TERMS == 4:
            # the moments epilogue (lc_series_sweep_scores_f16x3): every 256-row tile holds all four terms of two
            # 32-row validation blocks -- wave row 0: [T0 b0, T0 b1, T1 b0, T1 b1], wave row 1: the same of T2, T3
            nblk = M // LC_MB
            rows = 256 * ((nblk + 1) // 2)
            rowmap = np.empty(SERIES_TERMS * M, dtype=np.int32)
            i = np.arange(M)
            b = i // LC_MB
            for j in range(SERIES_TERMS):
                rowmap[j * M:(j + 1) * M] = 256 * (b // 2) + 128 * (j >> 1) + 32 * (2 * (j & 1) + (b & 1)) + i % LC_MB
            self._layout = (rows, ops.upload(rowmap, self.dev), None)
            self._rowmap_host = rowmap
            self._layout_key = key
        if getattr(self, "_layout_key", None) != key:
            per = (M + 127) // 128
            heavy = [(j, s) for j in range(min(2, SERIES_TERMS)) for s in range(per)]
            light = [(j, s) for j in range(2, SERIES_TERMS) for s in range(per)]
            order, cls = [], []
            while heavy or light:                        # one 256-row tile per round: (wm = 0 slab, wm = 1 slab)
                for _ in range(2):
                    if heavy and (not cls or len(cls) % 2 == 0 or not light):
                        order.append(heavy.pop(0)); cls.append(0)
                    elif light:
                        order.append(light.pop(0)); cls.append(1)
                    else:
                        order.append(None); cls.append(1)
            rows = 128 * len(order)
            rowmap = np.full(SERIES_TERMS * M, -1, dtype=np.int32)
            for slab, js in enumerate(order):
                if js is None:
                    continue
                j, s = js
                lo, hi = s * 128, min(M, (s + 1) * 128)
                rowmap[j * M + lo:j * M + hi] = slab * 128 + np.arange(hi - lo)
            self._layout = (rows, ops.upload(rowmap, self.dev), ops.upload(np.asarray(cls, dtype=np.uint8), self.dev))
            self._rowmap_host = rowmap
            self._layout_key = key
        return self._layout

    def _shared_image(self, inner_abs, N):
        """One tiled fp16 image of the targets for all inner folds of an outer fold: possible when every inner
        training set is the same row sequence minus one block whose position and length are multiples of 16 (the
        K-tile of the MFMA kernels) and no padding rows are needed -- contiguous K-folds of a multiple-of-16 fold
        length.  Returns (union rows, [(gap_begin, gap_rows) per fold]) or None (one split per inner fold)."""
        sets = [np.asarray(t_, dtype=np.int64) for t_, _ in inner_abs]
        if len(sets) < 2 or any(len(s_) != N for s_ in sets):
            return None
        union = np.unique(np.concatenate(sets))
        if len(union) % 16 or (len(union) - N) % 16:
            return None
        gaps = []
        for s_ in sets:
            idx = np.searchsorted(union, s_)                 # position of every row in the (sorted) union
            if np.any(np.diff(idx) <= 0):
                return None                                  # not in the union's order
            missing = np.setdiff1d(np.arange(len(union)), idx)
            if len(missing) != len(union) - N or (len(missing) and
                                                  (missing[-1] - missing[0] + 1 != len(missing) or missing[0] % 16)):
                return None
            gaps.append((int(missing[0]) if len(missing) else N, int(len(missing))))
        return union, gaps

    def _series_by_moments(self, split):
        """Score the series alphas from the moments of the shared terms T_j = P'_j Y (one contraction for all of
        them, lc_series_scores) instead of one hat matrix per alpha: correlation scoring on the fp16x3 path only
        (the R2 score needs the elementwise fl32 residual, see lc_epilogue.h)."""
        return bool(self.normalpha and self.mode == LC_SCORE_CORR and split)

    def _hat_matrices(self, K, inner_abs, lmax=None, moments=False, chol_after=None):
        """V-independent part of the inner CV of one outer fold: row lists, S[0]^2 (Lanczos), penalties and the
        hat matrices H_alpha of every (inner fold, alpha) -- batched Cholesky for the small alphas, the shared
        polynomial series for the large ones (as hat matrices, or with ``moments`` as the scaled matrix powers
        themselves).  Returns a dict the sweeps consume."""
        F, A = len(inner_abs), self.A
        n_i = [len(t) for t, _ in inner_abs]
        n_v = [len(v) for _, v in inner_abs]
        if min(n_i) < 1 or min(n_v) < 1:
            raise ValueError("every inner fold needs at least one training and one validation row")
        N = ops.pad_to(max(n_i), LC_NB)
        M = ops.pad_to(max(n_v), LC_MB)
        tr = ops.idx_matrix([t for t, _ in inner_abs], N, self.dev)          # (F, N) / (F, M) int32, -1 padded
        va = ops.idx_matrix([v for _, v in inner_abs], M, self.dev)
        if self.normalpha and lmax is None:
            lmax = self.lmax_systems(K, [t for t, _ in inner_abs])
            self._check_singcutoff(lmax)
        a2 = ops.penalties(lmax, F, self.d_alphas, self.normalpha)
        ser, cho, d_ser = self.ser, self.cho, self.d_ser
        moments = bool(moments and ser and min(n_v) > 1)
        Ac = len(cho)
        per_sys = (N + M) * N * 8
        chunk = max(1, min(F, MAX_INNER_FOLDS, self.opt.aug_budget_bytes // max(1, per_sys * max(Ac, 1))))
        infos, Hs, imgs = [], [], []
        for f0 in range(0, F, chunk):
            fc = min(chunk, F - f0)
            H = torch.empty((fc * A, M, N), dtype=torch.float32, device=self.dev) if not moments else None
            P = None
            if ser and moments:
                rows_p, rowmap, _ = self._series_layout(M)
                # voxel shards: the chains of the folds are independent and V-independent -- dealt out like the Cholesky
                # systems (contiguous shares, all-gathered into fold order); one rank: all of them
                n_per, mine = job_share(fc, self.shard.world, self.shard.rank) if self.shard.active else (fc, range(fc))
                m0, fcl = (mine[0], len(mine)) if len(mine) else (0, 0)
                P = ops.zeros((n_per, rows_p, N), torch.float32, self.dev)
                g0 = f0 + m0
                if fcl and N % COL_TILE == 0:
                    # the chain P'_j = P'_(j-1) (K[tr,tr] / lambda) on the f32 MFMA: its terms enter a prediction
                    # scaled by rho^j, fp32 products with fp32 accumulation keep them at full fp32 accuracy.
                    # Run transposed, Q_j = Kn Q_(j-1) with the folds as column groups of one grouped launch.
                    Mq = ops.pad_to(M, COL_TILE)
                    Kn = torch.empty((fcl, N, N), dtype=torch.float32, device=self.dev)
                    ops.gather_sub_f32(K, tr[g0:g0 + fcl], tr[g0:g0 + fcl], fcl, N, N, lmax[g0:g0 + fcl], Kn)
                    # Q_0[n][f][i] = K[tr_f[n], va_f[i]] / lambda_f  (K symmetric), zero in the padding columns
                    Q = ops.zeros((N, fcl, Mq), torch.float32, self.dev)
                    ops.gather_sub_f32_strided(K, tr[g0:g0 + fcl], va[g0:g0 + fcl], fcl, N, M, lmax[g0:g0 + fcl], Q, Mq,
                                               fcl * Mq, 1)
                    tiles = [f * (Mq // COL_TILE) for f in range(fcl + 1)]
                    for j in range(SERIES_TERMS):
                        if j:
                            Qn = torch.empty_like(Q)
                            ops.gemm_grouped(Kn, N, N * N, Q, fcl * Mq, None, Qn, fcl * Mq, N, fcl * Mq, N, tiles)
                            Q = Qn
                        ops.series_place(Q, N, fcl, Mq, M, rowmap[j * M:(j + 1) * M], P, rows_p)
                elif fcl:
                    ops.batch_series_terms(K, tr[g0:g0 + fcl], va[g0:g0 + fcl], fcl, N, M, lmax[g0:g0 + fcl], SERIES_TERMS,
                                           P, rowmap)
                if self.shard.active:
                    P = self.shard.all_gather(P, lane="hat").view(self.shard.world * n_per, rows_p, N)
            elif ser:
                ops.batch_series_hat(K, tr[f0:f0 + fc], va[f0:f0 + fc], fc, N, M, lmax[f0:f0 + fc], self.d_coef, d_ser, A,
                                     SERIES_TERMS, H)
            # the fp16 hi/lo images of the operators (the A operands of the V-wide contractions) are V-independent too:
            # made HERE, once per outer fold, and shared by every voxel range of the fold -- a host-to-host fit works
            # through the first and the last folds panel by panel, and each panel used to split the same matrices again
            img = None
            if moments and P is not None and self._split_assumed():
                tp = ops.pad_to(P.shape[1], 256)
                img = dict(tp=tp, Pt=torch.empty(fc * tp * N * 2, dtype=torch.float16, device=self.dev),
                           rs_p=torch.empty(fc * tp, dtype=torch.float32, device=self.dev), Ht=None, rs_h=None, hp=0)
                ops.split_rows_f16_groups(P.view(-1, N), fc, P.shape[1], N, img["Pt"], img["rs_p"])
            series_ready = torch.cuda.Event() if self.dev.type == "cuda" else None
            if series_ready is not None:
                series_ready.record()           # the series operands of this chunk are complete; Cholesky follows
            if Ac and self.spectral:
                if chol_after is not None:
                    torch.cuda.current_stream().wait_event(chol_after)
                # every alpha of every inner fold of the chunk from ONE eigendecomposition per fold (replicated on every
                # rank of a sharded fit: no collective)
                rows_f = tr[f0:f0 + fc]
                Hs_ = self._spectral_operators(K, rows_f, va[f0:f0 + fc], None, fc, N, M, a2[f0 * A:(f0 + fc) * A], A,
                                               [min(n_i[f0 + j], self.p) for j in range(fc)], out=H)
                infos.append(ops.zeros(fc * A, torch.int32, self.dev))
                assert Hs_ is H
            elif Ac:
                if chol_after is not None:
                    torch.cuda.current_stream().wait_event(chol_after)
                # job j = (inner fold f0 + j // Ac, Cholesky alpha j % Ac) = system (f0 + j // Ac) * A + cho[j % Ac] of
                # the (fold, alpha) grid that tr / va / a2 are laid out on
                grid_id = [(f0 + j // Ac) * A + cho[j % Ac] for j in range(fc * Ac)]

                def assemble(jobs, grid_id=grid_id):
                    aug = torch.empty((len(jobs), N + M, N), dtype=torch.float64, device=self.dev)
                    sysv = ops.upload(np.asarray([grid_id[j] for j in jobs], dtype=np.int32), self.dev)
                    ops.batch_assemble_sel(K, tr, va, None, a2, sysv, len(jobs), A, N, M, aug)
                    return aug

                slot = None if moments else ops.upload(np.asarray(
                    [(j // Ac) * A + cho[j % Ac] for j in range(fc * Ac)], dtype=np.int32), self.dev)
                Hc, info_c = self._sharded_solve(fc * Ac, N, M, assemble, out=H, slot=slot)
                infos.append(info_c)
                if moments:
                    H = Hc                                   # (>= fc * Ac, M, N): fold j's alphas at [j * Ac, (j + 1) * Ac)
                elif Hc is not H:
                    for j in range(fc * Ac):                 # voxel shards: beside the series alphas' hat matrices
                        H[(j // Ac) * A + cho[j % Ac]].copy_(Hc[j])          # (D2D copies)
            if img is not None and Ac and H is not None:
                hp = ops.pad_to(Ac * M, 256)
                img.update(hp=hp, Ht=torch.empty(fc * hp * N * 2, dtype=torch.float16, device=self.dev),
                           rs_h=torch.empty(fc * hp, dtype=torch.float32, device=self.dev))
                ops.split_rows_f16_groups(H.view(-1, N), fc, Ac * M, N, img["Ht"], img["rs_h"])
            Hs.append((f0, fc, H, P))
            imgs.append(img)
        info = self._join_flags(infos)
        return dict(F=F, N=N, M=M, n_v=n_v, n_i=n_i, tr=tr, va=va, shared=self._shared_image(inner_abs, N), Hs=Hs, info=info, lmax=lmax, a2=a2, cho=cho, ser=ser,
                    d_ser=d_ser, moments=moments, series_ready=series_ready, imgs=imgs)

    def _sweeps(self, hat, Y, done=None, split_phase=False):
        """Sum over inner folds of the (A, Vp) validation scores (ridge_corr_torch for every fold,
        nested_cv.py:366-393): the V-wide fused MFMA sweeps, plus -- with ``hat["moments"]`` -- one plain
        contraction of the shared series terms and the moment kernel for the alphas on the series.  ``done``: event
        after which the hat matrices are complete; the series part only waits for ``hat["series_ready"]`` and
        runs first, so the main stream has work while the auxiliary stream is still in the Cholesky chains.
        ``split_phase``: queue only that first part now and return a callable that queues the rest (the fused sweeps
        behind ``done``) and returns the scores -- the driver puts the NEXT step's first part in between, so that the
        main stream has V-wide work while it waits for a fold's Cholesky chains (the range the phases work on is
        captured here: the engine's current range may have moved on when the callable runs)."""
        Vp_, V_ = self.Vp, self.V
        if hat.get("no_inner"):
            # no inner fold of this outer fold has validation rows: the reference scores every alpha 0 for every voxel
            # (z_score of an empty block -> NaN -> nan_to_num, ridge_regression.py:124-133) and its first-maximum
            # argmax takes alphas[0]
            scores = ops.zeros((self.A, Vp_), torch.float32, self.dev)
            self.info.update(precision="f16x3" if hat["split"] else "f32", fused_alphas=0, series_terms=0)
            self.sweeps_done = torch.cuda.Event()
            self.sweeps_done.record()
            return (lambda: scores) if split_phase else scores
        if self.primal:
            out = self._sweeps_primal(hat, Y, done)
            return (lambda: out) if split_phase else out
        A, N, M, tr, va, n_v = self.A, hat["N"], hat["M"], hat["tr"], hat["va"], hat["n_v"]
        F = hat["F"]
        moments, cho = hat["moments"], hat["cho"]
        Ad = len(cho) if moments else A                   # alphas that go through the fused sweep
        main = torch.cuda.current_stream()
        scores = torch.empty((A, Vp_), dtype=torch.float32, device=self.dev)
        cho_first = list(cho) == list(range(len(cho)))     # ascending grids: the factorised alphas are rows 0 .. Ad-1
        scores_d = scores
        if moments and Ad:
            scores_d = scores[:Ad] if cho_first else torch.empty((Ad, Vp_), dtype=torch.float32, device=self.dev)
        part = torch.empty((max(Ad, 1) * M // LC_MB, 4, Vp_), dtype=torch.float32, device=self.dev)
        split, cs = hat["split"], hat["cs"]
        if hat.get("data_ready") is not None:
            main.wait_event(hat["data_ready"])            # this fold's (normalised) targets and their column scales
        self.info.update(precision="f16x3" if split else "f32", fused_alphas=Ad,
                          series_terms=SERIES_TERMS if moments else 0, folds_per_launch=1)
        nbuf = F if moments else 1                        # two passes over the folds keep every fold's operands
        ystat = torch.empty((nbuf, 3, Vp_), dtype=torch.float32, device=self.dev)
        yblk = torch.empty((nbuf, M // LC_MB, Vp_), dtype=torch.float32, device=self.dev)
        yv = torch.empty((nbuf, M, Vp_), dtype=torch.float32, device=self.dev)
        shared = hat.get("shared") if split else None
        if split:
            rows_pad = ops.pad_to(max(Ad, 1) * M, 256)
            Ht = torch.empty(rows_pad * N * 2, dtype=torch.float16, device=self.dev)
            rs_inv = torch.empty(rows_pad, dtype=torch.float32, device=self.dev)
            Vt = ops.pad_to(Vp_, 256)
            if shared is not None:
                # the targets of the whole outer training set split once; every inner fold contracts it minus one
                # aligned block (B view): saves F - 1 passes over Y per outer fold
                union, gaps = shared
                Yu = torch.empty(Vt * len(union) * 2, dtype=torch.float16, device=self.dev)
                ops.split_cols_f16(Y, Vp_, ops.idx_tensor(union, len(union), self.dev), len(union), cs, Yu)
                Yt = [Yu] * nbuf
                views = [(len(union), g0, gl) for g0, gl in gaps]
                hat["image"] = (Yu, union)                # the refit permutes its operand out of it (_refit_operands)
            else:
                Yt = [torch.empty(Vt * N * 2, dtype=torch.float16, device=self.dev) for _ in range(nbuf)]
                views = [(0, 0, 0)] * F
        folds = [(f0 + j, j, H, P) for f0, fc, H, P in hat["Hs"] for j in range(fc)]
        # the operators' fp16 images made with the hat matrices (_hat_matrices), per chunk: fold f0 + j is group j
        imgs = hat.get("imgs") or [None] * len(hat["Hs"])
        img_of = {f0 + j: (im, j) for (f0, fc, _, _), im in zip(hat["Hs"], imgs) if im is not None for j in range(fc)}
        merged, fused = False, False
        Pt = rs_p = part_s = Tbuf = cs_inv = rowmap = slab_light = Tm = None

        def series_part():
            nonlocal merged, fused, Pt, rs_p, part_s, Tbuf, cs_inv, rowmap, slab_light, Tm
            if moments:
                # ---- pass 1: validation statistics, operand split, series contraction + moment kernel
                Tm, rowmap, slab_light = self._series_layout(M)
                fused = slab_light is None                   # layout of the moments epilogue: the terms are never stored
                # all inner folds in ONE launch per pass (stacked A images, one shared target image with a gap per fold):
                # the folds are independent, and one launch fills the chip where F small ones each end in a partial round
                # of workgroups -- at 10 000 voxels per rank (8 GPUs) a fold's launch is 1.25 rounds
                merged = fused and shared is not None and F <= 64 and self.opt.folds_in_one_launch
                nst = F if merged else 1
                tp = ops.pad_to(Tm, 256)
                Pt = torch.empty(nst * tp * N * 2, dtype=torch.float16, device=self.dev)
                rs_p = torch.empty(nst * tp, dtype=torch.float32, device=self.dev)
                if fused:
                    part_s = torch.empty((nst, M // LC_MB, 18, Vp_), dtype=torch.float32, device=self.dev)
                else:
                    Tbuf = torch.empty((Tm, Vt), dtype=torch.float32, device=self.dev)
                cs_inv = self._cs_inv_padded(cs, Vt)                               # padded to the plain GEMM's tiles
                if hat.get("series_ready") is not None:
                    main.wait_event(hat["series_ready"])
                # validation statistics of all inner folds in one launch (the blocks are independent)
                for f0 in range(0, F, 64):
                    f1 = min(F, f0 + 64)
                    ops.val_stats_folds(Y, Vp_, va[f0:f1], f1 - f0, M, n_v[f0:f1], ystat[f0:f1], yblk[f0:f1], yv[f0:f1])
                if merged:
                    for f0, fc, H, P in hat["Hs"]:           # the folds' terms follow one another in P: one split per chunk
                        ops.split_rows_f16_groups(P.view(-1, N), fc, Tm, N, Pt[f0 * tp * N * 2:], rs_p[f0 * tp:])
                    for f in range(F):
                        self.info["plain_flops"] += 2.0 * SERIES_TERMS * n_v[f] * hat["n_i"][f] * V_
                    self.info["plain_launches"] += 1
                    ops.series_sweep_scores_f16x3_folds(Pt, rs_p, M, n_v, N, Yu, cs_inv, Vt, yv, Vp_, ystat, yblk,
                                                        self.d_coef, hat["d_ser"], part_s, scores, False, views)
                for f, j, H, P in (() if merged else folds):
                    if shared is None:
                        ops.split_cols_f16(Y, Vp_, tr[f], N, cs, Yt[f])
                    Pt_f, rs_p_f = Pt, rs_p
                    if f in img_of:
                        im, g = img_of[f]
                        Pt_f, rs_p_f = im["Pt"][g * im["tp"] * N * 2:], im["rs_p"][g * im["tp"]:]
                    else:
                        ops.split_rows_f16(P[j], Tm, N, Pt, rs_p)
                    self.info["plain_flops"] += 2.0 * SERIES_TERMS * n_v[f] * hat["n_i"][f] * V_
                    self.info["plain_launches"] += 1
                    if fused:
                        ops.series_sweep_scores_f16x3(Pt_f, rs_p_f, M, n_v[f], N, Yt[f], cs_inv, Vt, yv[f], Vp_, ystat[f], yblk[f],
                                                      self.d_coef, hat["d_ser"], part_s, scores, accumulate=f > 0,
                                                      bview=views[f])
                        continue
                    ops.gemm_grouped_f16x3(Pt_f, rs_p_f, Tm, Yt[f], cs_inv, Tbuf, Vt, Vt, N, [0, Vt // 256], slab_light,
                                           bview=views[f])
                    ops.series_scores(Tbuf, Vt, SERIES_TERMS, M, n_v[f], Vp_, yv[f], ystat[f], self.d_coef, hat["d_ser"],
                                      scores, accumulate=f > 0, rowmap=rowmap)

        def fused_part():
            nonlocal Ht, rs_inv, part
            if done is not None:
                main.wait_event(done)
            # ---- pass 2 (the only one without the moment path): fused sweeps of the alphas that have hat matrices
            if moments and merged and Ad:
                Ht = torch.empty(F * rows_pad * N * 2, dtype=torch.float16, device=self.dev)
                rs_inv = torch.empty(F * rows_pad, dtype=torch.float32, device=self.dev)
                part = torch.empty((F, Ad * M // LC_MB, 4, Vp_), dtype=torch.float32, device=self.dev)
                for f0, fc, H, P in hat["Hs"]:
                    ops.split_rows_f16_groups(H.view(-1, N), fc, Ad * M, N, Ht[f0 * rows_pad * N * 2:], rs_inv[f0 * rows_pad:])
                ops.alpha_sweep_scores_f16x3_folds(Ht, rs_inv, Ad, M, N, Yu, cs[Vp_:], yv, Vp_, n_v, ystat, yblk, self.mode,
                                                   part, scores_d, False, views)
                self.info["fused_flops"] += sum(2.0 * Ad * n_v[f] * hat["n_i"][f] * V_ for f in range(F))
                self.info["fused_launches"] += 1
                self.info["folds_per_launch"] = F
            for f, j, H, P in (() if (moments and merged) else folds):
                b = f if moments else 0
                if not moments:
                    ops.val_stats(Y, Vp_, va[f], M, n_v[f], ystat[b], yblk[b], yv[b])
                if split:
                    if not moments and shared is None:
                        ops.split_cols_f16(Y, Vp_, tr[f], N, cs, Yt[b])
                    if Ad:
                        self.info["fused_flops"] += 2.0 * Ad * n_v[f] * hat["n_i"][f] * V_
                        self.info["fused_launches"] += 1
                        Ht_f, rs_h_f = Ht, rs_inv
                        if moments and f in img_of and img_of[f][0]["Ht"] is not None:
                            im, g = img_of[f]
                            Ht_f, rs_h_f = im["Ht"][g * im["hp"] * N * 2:], im["rs_h"][g * im["hp"]:]
                        else:
                            ops.split_rows_f16(H[j * Ad:(j + 1) * Ad].reshape(Ad * M, N), Ad * M, N, Ht, rs_inv)
                        ops.alpha_sweep_scores_f16x3(Ht_f, rs_h_f, Ad, M, N, Yt[b], cs[Vp_:], yv[b], Vp_, n_v[f], ystat[b],
                                                     yblk[b], self.mode, part, scores_d, accumulate=f > 0, bview=views[f])
                else:
                    self.info["fused_flops"] += 2.0 * A * n_v[f] * hat["n_i"][f] * V_
                    self.info["fused_launches"] += 1
                    ops.alpha_sweep_scores(H[j * A:(j + 1) * A], A, M, N, Y, Vp_, tr[f], yv[b], n_v[f], ystat[b], yblk[b],
                                           self.mode, part, scores, accumulate=f > 0)
            if moments and Ad and not cho_first:
                for i, a in enumerate(cho):
                    scores[a].copy_(scores_d[i])
            self.sweeps_done = torch.cuda.Event()
            self.sweeps_done.record()
            return scores


        if not moments:                                   # one pass only: nothing to put another step's work behind
            out = fused_part()
            return (lambda: out) if split_phase else out
        series_part()
        return fused_part if split_phase else fused_part()

    def _alpha_scores(self, K, Y, inner_abs):
        cs, split = self._target_scales(Y)
        hat = self._hat_matrices(K, inner_abs, moments=self._series_by_moments(split))
        hat.update(cs=cs, split=split)
        return self._sweeps(hat, Y), hat["info"]

    # -------------------------------------------------------------- alpha selection
    def choose(self, scores, single_alpha):
        """(Vp,) int32 device vector of alpha indices: per-voxel first argmax (nested_cv.py:405-411)
        or, for ``single_alpha``, the argmax of the across-voxel mean (:396-400; the per-alpha sums
        are all-reduced over the voxel shards)."""
        if single_alpha:
            _, rowsum = ops.select_alpha(scores, self.A, self.Vp, want_best=False, want_rowsum=True)
            self.shard.all_reduce_(rowsum, "sum")          # A doubles, on the device: the choice never visits the host
            best = torch.empty(self.Vp, dtype=torch.int32, device=self.dev)
            return ops.fill_argmax(rowsum, self.A, best, self.Vp)     # first maximum, like torch.argmax
        return ops.select_alpha(scores, self.A, self.Vp)[0]

    # -------------------------------------------------------------- refit (ridge_torch)
    # three steps, so that the driver can put the fp64 systems on the auxiliary stream beside the next fold's
    # sweeps: groups (argmax histogram -> host), systems (M_alpha of the alphas in use), apply (V-wide GEMM)
    def _refit_groups(self, best, split, pending=None):
        """Voxels sorted by chosen alpha: (perm, used alphas, column-tile offsets per group, Vs).  The one
        host synchronisation of a fold: the histogram decides how many systems the refit solves.  ``pending``
        (from _group_async) holds a grouping whose histogram is already on its way to pinned memory."""
        tile = 256 if split else COL_TILE                 # column-tile width of the GEMM that follows
        if pending is None:
            pending = self._group_async(best, split)
        perm, count_h, ev = pending
        ev.synchronize()
        count_h = count_h.numpy()
        if isinstance(perm, list):                       # more than 64 alphas: grouped range by range, joined now
            perm = ops.join_group_ranges(perm, count_h[0], tile)
        used = [a for a in range(self.A) if count_h[0, a] > 0]
        used_all = [a for a in range(self.A) if count_h[1, a] > 0]       # over all voxel shards
        if self.shard.simulate:     # one rank run alone for timing: its peers' choices are unknown -- assume they
            used_all = sorted(set(used_all) | set(self.cho))             # need every factorised alpha (worst case)
        self.info["used_all"] = list(used_all)
        tiles = [0]
        for a in used:
            tiles.append(tiles[-1] + (int(count_h[0, a]) + tile - 1) // tile)
        return perm, used, tiles, tiles[-1] * tile, used_all

    def _group_async(self, best, split):
        """Grouping kernel + asynchronous copy of the alpha histogram to pinned memory: (perm, host counts, event)."""
        tile = 256 if split else COL_TILE
        perm, count2 = ops.group_by_alpha(best, self.V, self.A, tile)    # count2: (2, A), both rows = this rank's counts
        self.shard.all_reduce_(count2[1], "sum")                         # row 1 -> the histogram over all voxel shards
        count_h = torch.empty((2, self.A), dtype=torch.int32, pin_memory=True)
        count_h.copy_(count2, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return perm, count_h, ev

    def _refit_row_granule(self):
        """The augmented rows of a refit system can be cut into up to ``world`` slices (a power of two) that different
        ranks transform; every slice must be a multiple of LC_MB rows."""
        g = 1
        while 2 * g <= self.shard.world:
            g *= 2
        return LC_MB * g

    def _refit_rhs(self, X, K, tr_rows, tr_o, te_rows):
        """The augmented rows of the refit systems, fp64 (rows, N_o):  Xtr' above K[te,tr]  (primal form: the identity
        above X_te -- the weights ARE (G + a^2 I)^-1 B)."""
        n_t = len(te_rows)
        N_o = tr_o.shape[-1]
        if self.primal:
            rows = ops.pad_to(self.PP + ops.pad_to(n_t, LC_MB), self._refit_row_granule())
            idx = np.full(rows, -1, dtype=np.int64)
            idx[: self.p] = -(2 + np.arange(self.p))                      # unit rows e_c
            idx[self.PP:self.PP + n_t] = np.asarray(te_rows, dtype=np.int64)
            return ops.gather_rows_f64(X, ops.idx_tensor(idx, rows, self.dev), 1, rows, self.p, self.PP)[0]
        rows = ops.pad_to(self.p_pad + ops.pad_to(n_t, LC_MB), self._refit_row_granule())
        rhs = ops.zeros((rows, N_o), torch.float64, self.dev)
        ops.transpose_rows(X, tr_o, N_o, self.p, rhs)
        if n_t:                                        # K[te, tr] below X', padded columns (index -1) zero
            ops.gather_sub_f64(K, ops.idx_tensor(te_rows, n_t, self.dev), tr_o, 1, n_t, N_o, rhs[self.p_pad:self.p_pad + n_t])
        return rhs

    def _spectral_operators(self, K, rows, rows_r, rhs, F, N, M, a2, A, caps, out=None, cache_key=None):
        """The reference's operators where the Cholesky route cannot follow it (alpha = 0, biting singcutoff):
            out[f A + a] (M, N) f32 = R_f U_k diag(1 / (lambda_k + a2[f A + a])) U_k',
        U, lambda the eigenpairs of K[rows_f, rows_f] (fp64 cyclic Jacobi, lc_batch_eigh_jacobi), kept when
        sqrt(lambda) > singcutoff and among the ``caps[f]`` = min(n, p) largest -- exactly svd_wrapper's truncation
        (ridge_utils.py:44-63) followed by D = S / (S^2 + a^2) (ridge_regression.py:56,117).  R_f = K[rows_r[f], rows_f]
        (hat matrices) or the given ``rhs`` (F, M, N) f64 (refit rows).  ``cache_key``: keep the eigenpairs of the
        (single) system for later calls of the same fold."""
        eig = self._eig_cache.get(cache_key) if cache_key is not None else None
        if eig is None:
            Ksub = torch.empty((F, N, N), dtype=torch.float64, device=self.dev)
            ops.gather_sub_f64(K, rows, rows, F, N, N, Ksub)
            lam, vt, _, sweeps = ops.batch_eigh(Ksub)
            logger.info("spectral route: %d system(s) of %d rows diagonalised in %d Jacobi sweeps", F, N, sweeps)
            eig = (lam, vt)
            if cache_key is not None:
                self._eig_cache[cache_key] = eig
        lam, vt = eig
        if rhs is None:
            rhs = torch.empty((F, M, N), dtype=torch.float64, device=self.dev)
            ops.gather_sub_f64(K, rows_r, rows, F, M, N, rhs)
        if out is None:
            out = torch.empty((F * A, M, N), dtype=torch.float32, device=self.dev)
        cap = ops.upload(np.asarray(caps, dtype=np.int32), self.dev)
        ops.batch_spectral_apply(lam, vt, rhs, a2, A, self.singcutoff, cap, out)
        return out

    def _refit_chol(self, K, tr_o, lmax_o, rhs, alphas_idx):
        """rhs (K[tr,tr] + a^2 I)^-1 for the listed alphas by the augmented batched Cholesky in fp64:
        ((len(alphas_idx), rows, N_o) f32, pivot flags of this rank's share).  Voxel shards: the batch is dealt out
        over the ranks as (alpha, row slice) jobs -- with fewer alphas than ranks every system's augmented rows are
        cut into S slices (each job then factors K + a^2 I again, N^3/3 of the system's N^3/3 + 2 N^2 rows flops) --
        and all-gathered; every rank must be called with the same ``alphas_idx``."""
        Gc, (rows, N_o) = len(alphas_idx), rhs.shape
        if self.spectral:
            d_al = ops.upload(np.asarray([self.alphas[a] for a in alphas_idx], dtype=np.float64), self.dev)
            a2_sel = ops.penalties(lmax_o, 1, d_al, self.normalpha)
            H = self._spectral_operators(K, tr_o, None, rhs.reshape(1, rows, N_o), 1, N_o, rows, a2_sel, Gc,
                                         [min(self._real_rows(tr_o), self.p)], cache_key=("refit", tr_o.data_ptr()))
            return H.view(Gc, rows, N_o), ops.zeros(max(Gc, 1), torch.int32, self.dev)
        a2_o = ops.penalties(lmax_o, 1, self.d_alphas, self.normalpha)              # (A,): grid F = 1
        if self._refit_by_inverse(alphas_idx):
            eye = self._identity_rows(N_o)

            def assemble_inv(jobs):
                aug = torch.empty((len(jobs), 2 * N_o, N_o), dtype=torch.float64, device=self.dev)
                sysv = ops.upload(np.asarray([alphas_idx[j] for j in jobs], dtype=np.int32), self.dev)
                ops.batch_assemble_sel(K, tr_o, None, eye, a2_o, sysv, len(jobs), self.A, N_o, N_o, aug)
                return aug

            Pj, info = self._sharded_solve(Gc, N_o, N_o, assemble_inv, lane="refit", inverse=True)
            return self._apply_inverses(rhs, Pj[:Gc]), info
        S = 1
        while 2 * S * Gc <= self.shard.world and rows % (2 * S * LC_MB) == 0:
            S *= 2
        rs = rows // S                                                              # rows per job

        def assemble(jobs):
            # job j = (alpha j // S, row slice j % S): slices of one system are neighbours, so the gathered blocks are
            # already the (Gc, rows, N_o) result; one assemble launch per run of jobs that share a row slice
            aug = torch.empty((len(jobs), N_o + rs, N_o), dtype=torch.float64, device=self.dev)
            for k, j in enumerate(jobs):
                q = j % S
                sysv = ops.upload(np.asarray([alphas_idx[j // S]], dtype=np.int32), self.dev)
                ops.batch_assemble_sel(K, tr_o, None, rhs[q * rs:(q + 1) * rs], a2_o, sysv, 1, self.A, N_o, rs, aug[k:k + 1])
            return aug

        def assemble_whole(jobs):
            aug = torch.empty((len(jobs), N_o + rows, N_o), dtype=torch.float64, device=self.dev)
            sysv = ops.upload(np.asarray([alphas_idx[j] for j in jobs], dtype=np.int32), self.dev)
            ops.batch_assemble_sel(K, tr_o, None, rhs, a2_o, sysv, len(jobs), self.A, N_o, rows, aug)
            return aug

        Hj, info = self._sharded_solve(Gc * S, N_o, rs, assemble if S > 1 else assemble_whole, lane="refit")
        return Hj[: Gc * S].view(Gc, rows, N_o), info

    def _real_rows(self, idx):
        """Number of real (non-padding) entries of an int32 device index list (one small D2H: spectral route only)."""
        key = idx.data_ptr()
        if key not in self._n_real:
            self._n_real[key] = int((idx.cpu() >= 0).sum())
        return self._n_real[key]

    def _refit_by_inverse(self, alphas_idx):
        """The refit operator  R (K + a^2 I)^-1,  R = [Xtr' ; K[te,tr]]  (3072 + 600 rows at cfg2), through the explicit
        inverse (N^3 fp64 flops, lc_batch_chol_inverse) and ONE product R P on the fp16x3 MFMA instead of triangular
        solves with every row of R (N^3/3 + 2 N^2 rows: 3.4x the fp64 work).  The operator goes through 22-bit fp16
        triples afterwards anyway (the V-wide contraction), but in the product R P the entries of P ~ 1/a^2 cancel
        down to ~ 1/(2 a S0): the relative error is ~ 2^-22 x 2 S0 / a = 2^-21 / alpha for alpha S[0] scaling -- taken
        for alpha >= 0.05 (< 1e-5), on the fp16x3 path, with normalpha (S[0] known); the solves otherwise."""
        # (voxel shards: every rank applies every inverse it needs itself -- the same products on every rank -- while
        # the row-sliced solves shrink with the ranks: measured per simulated rank 81.8 vs 84.2 ms at 2, 53.6 vs 53.8
        # at 4, 40.2 vs 39.0 ms at 8 ranks; so the solves from 8 ranks on)
        return (self.opt.refit_by_inverse and self.normalpha and not self.primal and not self.spectral and self.precision != "f32"
                and self.shard.world <= self.opt.refit_inverse_max_world
                and len(alphas_idx) > 0 and min(self.alphas[a] for a in alphas_idx) >= self.opt.refit_inverse_min_alpha)

    def _identity_rows(self, N_o):
        """(N_o, N_o) f64 identity, cached.  It is made on whichever stream asks first and read from others later: the
        event recorded behind its creation is waited for at every later use."""
        if self._eye_key != N_o:
            idx = ops.upload((-(2 + np.arange(N_o))).astype(np.int32).reshape(1, N_o), self.dev)
            self._eye = ops.gather_rows_f64(self.dX, idx, 1, N_o, 1, N_o)[0]       # unit rows only
            self._eye_key = N_o
            self._eye_ev = torch.cuda.Event()
            self._eye_ev.record()
        else:
            torch.cuda.current_stream().wait_event(self._eye_ev)
            self._eye.record_stream(torch.cuda.current_stream())
        return self._eye

    def _apply_inverses(self, rhs, P):
        """(G, rows, N_o) f32 = rhs . P[g] for the (G, N_o, N_o) f32 inverses, on the fp16x3 MFMA."""
        R = ops.scale_cast_f64_f32(rhs, self._one(), torch.empty(rhs.shape, dtype=torch.float32, device=self.dev))
        return self._times_symmetric(R, P)

    def _times_symmetric(self, R, mats):
        """(G, rows, N) f32 = R . mats[g] for SYMMETRIC (N, N) f32 matrices, as plain fp16x3 contractions: A = the rows
        of R as fp16 triples; column n of the B operand is row n of the matrix, and the tiled images of the two
        operands have the same layout -- so the row split (coalesced reads, per-row power-of-two scales) of the matrix
        IS its column split."""
        G, (rows, N) = mats.shape[0], R.shape
        rows_pad, Nc = ops.pad_to(rows, 256), ops.pad_to(N, 256)
        At = torch.empty(rows_pad * N * 2, dtype=torch.float16, device=self.dev)
        rs_inv = torch.empty(rows_pad, dtype=torch.float32, device=self.dev)
        ops.split_rows_f16(R, rows, N, At, rs_inv)
        Bt = torch.empty(Nc * N * 2, dtype=torch.float16, device=self.dev)
        cs_inv = torch.empty(Nc, dtype=torch.float32, device=self.dev)
        out = torch.empty((G, rows, N), dtype=torch.float32, device=self.dev)
        for g in range(G):
            ops.split_rows_f16(mats[g], N, N, Bt, cs_inv)
            ops.gemm_grouped_f16x3(At, rs_inv, rows, Bt, cs_inv, out[g], N, Nc, N, [0, Nc // 256])
        return out

    def _one(self):
        return self._d_one                             # made in __init__, before ``ready``: every stream may read it

    def _refit_systems(self, X, K, tr_rows, used, tr_o=None, lmax_o=None, te_rows=(), spec=None, used_all=None, cache=None):
        """Per alpha in use, the rows  [Xtr' ; K[te,tr]] (K[tr,tr] + a^2 I)^-1  as f32 (G, p_pad + pad32(n_te), N_o):
        M_alpha, whose product with the targets is the weight matrix (the V-independent half of
        ridge_regression.py:46-61), and below it the hat matrix of the test rows, whose product with the same
        targets is the test prediction X_te W (nested_cv.py:151,251) -- one V-wide contraction gives both.
        Augmented batched Cholesky in fp64; ``spec`` (fold_speculate) holds systems solved ahead of the alpha
        choice, which are taken from there.  ``used`` = the alphas THIS rank's voxels chose (the groups of its refit
        contraction), ``used_all`` = those of all ranks: the Cholesky systems are solved collectively
        (_sharded_solve), so every rank must ask for the same ones.  ``cache``: the fold's dict of operators already
        built (alpha -> (rows, N_o) f32), shared by the voxel ranges of the fold -- a later range only solves what an
        alpha nobody chose before needs."""
        G = len(used)
        used_all = list(used) if used_all is None else list(used_all)
        cache = {} if cache is None else cache
        done_M = cache.setdefault("M", {})               # alpha -> operator rows
        flag_parts = cache.setdefault("flags", [])       # pivot flags of every system solved for this fold
        n_o = len(tr_rows)
        N_o = ops.pad_to(n_o, LC_NB) if tr_o is None else tr_o.shape[-1]
        if tr_o is None:
            tr_o = ops.idx_tensor(tr_rows, N_o, self.dev).reshape(1, N_o)
            lmax_o = ops.lambda_max(K, tr_o, 1, N_o, self.steps) if self.normalpha else None
            self._check_singcutoff(lmax_o)
        if "rhs" not in cache:
            cache["rhs"] = spec["rhs"] if spec is not None else self._refit_rhs(X, K, tr_rows, tr_o, te_rows)
        rhs = cache["rhs"]
        rows = rhs.shape[0]
        # alphas on the polynomial series (large penalties: what real recordings usually select) need no
        # factorisation:  [Xtr' ; K_te] (K + a^2 I)^-1 = sum_j c_j(alpha) R_j,  R_j = [Xtr' ; K_te] K^j / lambda^(j+1),
        # with the chain R_j = R_(j-1) (K / lambda) on the f32 MFMA, shared by all such alphas (cf. _hat_matrices)
        on_series = set(self.ser) if (N_o % COL_TILE == 0 and lmax_o is not None) else set()
        poly = [a for a in used if a in on_series]
        have = list(spec["alphas"]) if spec is not None else []
        if have and not cache.get("spec_flags"):
            # the flags of the systems solved ahead: those of the alphas somebody chose only (a failed pivot in a system
            # nobody uses must not fail the fit: the one-GPU path never solves it -- ADVICE r2)
            cache["spec_flags"] = True
            by_alpha = spec.get("info_by_alpha")
            if by_alpha is None:
                flag_parts.append(spec["info"])
            else:
                cache["spec_by_alpha"] = by_alpha
        if cache.get("spec_by_alpha"):
            for a in list(cache["spec_by_alpha"]):
                if a in used_all:
                    flag_parts += cache["spec_by_alpha"].pop(a)
        # systems solved ahead of the choice enter the fold's cache when an alpha that has one is first used
        ahead = [a for a in used_all if a in have and a not in done_M and a not in on_series]
        if ahead and spec.get("P") is not None:          # refit_ahead left the inverses: apply them to the rows now
            Pa = spec["P"] if ahead == have else torch.stack([spec["P"][have.index(a)] for a in ahead])
            Ma = self._apply_inverses(rhs, Pa)
            for i, a in enumerate(ahead):
                done_M[a] = Ma[i]
        else:
            for a in ahead:
                done_M[a] = spec["M"][have.index(a)]
        need = [a for a in used_all if a not in on_series and a not in done_M]
        if need:
            Mc, info_n = self._refit_chol(K, tr_o, lmax_o, rhs, need)
            for i, a in enumerate(need):
                done_M[a] = Mc[i]
            flag_parts.append(info_n)
        info = self._join_flags(flag_parts)
        new_poly = [a for a in poly if a not in done_M]
        if new_poly:
            if "terms" not in cache:
                Kn = torch.empty((1, N_o, N_o), dtype=torch.float32, device=self.dev)
                ops.gather_sub_f32(K, tr_o, tr_o, 1, N_o, N_o, lmax_o, Kn)
                R = ops.scale_cast_f64_f32(rhs, lmax_o, torch.empty(rhs.shape, dtype=torch.float32, device=self.dev))
                terms = [R]
                for _ in range(1, SERIES_TERMS):
                    if self.precision != "f32":          # 43 GFLOP per step: 0.13 ms on the fp16x3 MFMA, 0.5 ms in f32
                        terms.append(self._times_symmetric(terms[-1], Kn)[0])
                        continue
                    Rn = torch.empty_like(R)
                    ops.gemm_grouped(terms[-1], N_o, 0, Kn[0], N_o, None, Rn, N_o, rows, N_o, N_o, [0, N_o // COL_TILE])
                    terms.append(Rn)
                cache["terms"] = terms
            for a in new_poly:
                done_M[a] = ops.combine_terms(cache["terms"], self.coef_host[self.ser.index(a)],
                                              torch.empty((rows, N_o), dtype=torch.float32, device=self.dev))
        return [done_M[a] for a in used], info

    def _refit_operands(self, Y, tr_rows, extra_rows, perm, tiles, Vs, Malpha, split, cs, image=None):
        """Operands of the V-wide refit contraction: Ys (N_o + len(extra_rows), Vs), the targets gathered in
        alpha-sorted voxel order (``extra_rows``, the test targets, below the training rows), and on the fp16x3 path
        their tiled fp16 image with the column scales carried through the permutation.  ``image`` = (tiled fp16 image,
        its rows) the inner CV made of the same targets in natural voxel order: when its rows ARE the training rows the
        sorted image is a 16-byte-unit column gather out of it (lc_permute_cols_f16) and only the test rows are gathered
        from the fp32 targets -- no sorted fp32 copy of the training rows, no second split pass."""
        n_o = len(tr_rows)
        N_o = ops.pad_to(n_o, LC_NB)
        n_x = len(extra_rows)
        from_image = (split and image is not None and self.opt.refit_from_image and n_o % K_TILE == 0
                      and len(image[1]) == n_o and np.array_equal(np.asarray(image[1]), np.asarray(tr_rows)))
        if from_image:
            rows_x = ops.idx_tensor(np.asarray(extra_rows, dtype=np.int64), n_x, self.dev)
            Ys_te, te_src = None, None
            if 0 < n_x <= 640:                           # Pearson r reads the test rows through (rows, perm) in place
                te_src = (Y, rows_x, perm)
            else:
                Ys_te = torch.empty((n_x, Vs), dtype=torch.float32, device=self.dev)
                ops.gather(Y, Y.stride(0), rows_x, n_x, perm, Vs, Ys_te)
            cs_s = torch.empty((2, Vs), dtype=torch.float32, device=self.dev)
            ops.gather(cs.reshape(2, self.Vp), self.Vp, None, 2, perm, Vs, cs_s)
            Yt = torch.empty(Vs * n_o * 2, dtype=torch.float16, device=self.dev)
            ops.permute_cols_f16(image[0], perm, Vs, n_o, Yt)
            return dict(Ys=None, Ys_te=Ys_te, te_src=te_src, N_o=N_o, K=n_o, n_o=n_o, Vs=Vs, tiles=tiles, Malpha=Malpha,
                        split=split, cs_s=cs_s, Yt=Yt)
        rows_s = ops.idx_tensor(np.concatenate([tr_rows, np.full(N_o - n_o, -1), np.asarray(extra_rows, dtype=np.int64)]),
                                N_o + n_x, self.dev)
        Ys = torch.empty((N_o + n_x, Vs), dtype=torch.float32, device=self.dev)
        ops.gather(Y, Y.stride(0), rows_s, N_o + n_x, perm, Vs, Ys)
        o = dict(Ys=Ys, Ys_te=Ys[N_o:], N_o=N_o, K=N_o, n_o=n_o, Vs=Vs, tiles=tiles, Malpha=Malpha, split=split)
        if split:
            cs_s = torch.empty((2, Vs), dtype=torch.float32, device=self.dev)
            ops.gather(cs.reshape(2, self.Vp), self.Vp, None, 2, perm, Vs, cs_s)
            Yt = torch.empty(Vs * N_o * 2, dtype=torch.float16, device=self.dev)
            ops.split_cols_f16(Ys, Vs, ops.idx_tensor(np.arange(N_o), N_o, self.dev), N_o, cs_s[0], Yt)
            o.update(cs_s=cs_s, Yt=Yt)
        return o

    def _refit_product(self, o, r0, r1, useful_rows, out=None):
        """Rows [r0, r1) of  C = [M_alpha ; H_te,alpha](group) . Ys  as an (r1 - r0, Vs) f32 matrix (fp16x3 path: the
        rows of every group split to fp16 triples, one grouped launch).  The caller takes the test predictions first
        -- what the host statistics wait for -- and the weight rows afterwards.  ``out``: a (r1 - r0, Vs) view (any row
        stride) the product is written to."""
        Malpha, Vs, N_o = o["Malpha"], o["Vs"], o["N_o"]          # one (rows, N_o) operator per alpha group
        G, rows = len(Malpha), r1 - r0
        C = out if out is not None else torch.empty((rows, Vs), dtype=torch.float32, device=self.dev)
        if o["split"]:
            rows_pad = ops.pad_to(rows, 256)
            Kc = o["K"]                                  # contraction depth: the training rows (the operators' padding
            # columns beyond them are zero).  The operators' images are the same for every voxel range of the fold
            # whose voxels chose the same alphas: kept in the fold's cache
            key = (o.get("used"), r0, r1, Kc)
            cache = o.get("img_cache")
            if cache is not None and key in cache:
                At, rs_inv = cache[key]
            else:
                At = torch.empty(G * rows_pad * N_o * 2, dtype=torch.float16, device=self.dev)
                rs_inv = torch.empty(G * rows_pad, dtype=torch.float32, device=self.dev)
                for g in range(G):
                    ops.split_rows_f16(Malpha[g][r0:r1], rows, Kc, At[g * rows_pad * Kc * 2:], rs_inv[g * rows_pad:])
                if cache is not None and o.get("used") is not None:
                    cache[key] = (At, rs_inv)
            ops.gemm_grouped_f16x3(At, rs_inv, rows, o["Yt"], o["cs_s"][1], C, C.stride(0), Vs, Kc, o["tiles"])
            self.info["plain_flops"] += 2.0 * useful_rows * o["n_o"] * self.V
            self.info["plain_launches"] += 1
        else:
            Ms = torch.empty((G, rows, N_o), dtype=torch.float32, device=self.dev)
            for g in range(G):
                Ms[g].copy_(Malpha[g][r0:r1])                       # (D2D copies: the groups' operators side by side)
            ops.gemm_grouped(Ms, N_o, Ms.stride(0), o["Ys"], Vs, None, C, C.stride(0), rows, Vs, N_o, o["tiles"])
        return C

    def _refit_apply(self, Y, tr_rows, extra_rows, perm, tiles, Vs, Malpha, split, cs):
        """C (rows, Vs) = [M_alpha ; H_te,alpha](group) . Ys -- the weights in its first p_pad rows, the test
        predictions below -- together with Ys and N_o (see _refit_operands)."""
        o = self._refit_operands(Y, tr_rows, extra_rows, perm, tiles, Vs, Malpha, split, cs)
        C = self._refit_product(o, 0, Malpha[0].shape[0], self.p + len(extra_rows))
        return C, o["Ys"], o["N_o"]

    def refit(self, X, Y, K, tr_rows, best, extra_rows=(), tr_o=None, lmax_o=None):
        """Weights of every voxel at its chosen alpha, in alpha-sorted voxel order
        (ridge_regression.py:9-63), all on the current stream.  Returns (Ws (p_pad, Vs), Ys, perm, N_o, info):
        column j of Ws / Ys is voxel perm[j] (-1 = padding)."""
        cs, split = self._target_scales(Y)
        perm, used, tiles, Vs, used_all = self._refit_groups(best, split)
        Malpha, info = self._refit_systems(X, K, tr_rows, used, tr_o, lmax_o, used_all=used_all)
        Ws, Ys, N_o = self._refit_apply(Y, tr_rows, extra_rows, perm, tiles, Vs, Malpha, split, cs)
        return Ws[: self.p_pad], Ys, perm, N_o, info

    def unsort(self, vec_sorted, perm, Vs):
        """Sorted-voxel-order host vector -> natural voxel order."""
        perm_h = perm[:Vs].cpu().numpy()
        live = perm_h >= 0
        out = np.empty(self.V, dtype=vec_sorted.dtype)
        out[perm_h[live]] = vec_sorted[live]
        return out

    # -------------------------------------------------------------- one outer fold, in phases
    # prepare (aux stream: fp64, V-independent) -> begin (main: inner-CV sweeps) -> select (one sync on the alpha
    # histogram; refit systems on aux) -> finish (main: V-wide refit, prediction, Pearson, D2H) -> collect (wait
    # for the fold's results).  The caller interleaves the phases of consecutive folds so that the main stream
    # always has MFMA work, the auxiliary stream the fp64 work, and the host statistics of fold f run meanwhile.
    def fold_prepare(self, tr_rows, te_rows, inner_rel, lmax_pre=None, chol_after=None):
        """prepare_folds for a single outer fold."""
        return self.prepare_folds([(tr_rows, te_rows, inner_rel)], [lmax_pre], chol_after)[0]

    def _hat_slice(self, hat, s, Fo, inner_abs):
        """The hat-matrix set of the inner folds [s, s + Fo) of a batch prepared together (one chunk): views."""
        (f0, fc, H, P), = hat["Hs"]
        per = len(hat["cho"]) if hat["moments"] else self.A          # hat matrices kept per inner fold
        sub = dict(hat)
        sub.update(F=Fo, n_v=hat["n_v"][s:s + Fo], n_i=hat["n_i"][s:s + Fo],
                   tr=None if hat["tr"] is None else hat["tr"][s:s + Fo], va=hat["va"][s:s + Fo],
                   lmax=None if hat["lmax"] is None else hat["lmax"][s:s + Fo], a2=hat["a2"][s * self.A:(s + Fo) * self.A],
                   shared=self._shared_image(inner_abs, hat["N"]), xt_off=hat.get("xt_off", 0) + s,
                   Hs=[(0, Fo, None if H is None else H[s * per:(s + Fo) * per], None if P is None else P[s:s + Fo])])
        pim = hat.get("img")
        if pim is not None:                            # primal form: images of the block-product / series / hat operands
            Mv, PPn, ap, tp, hp = hat["M"], hat["N"], pim["ap"], pim["tp"], pim["hp"]
            sub["img"] = dict(ap=ap, tp=tp, hp=hp, At=pim["At"][s * ap * Mv * 2:(s + Fo) * ap * Mv * 2],
                              rs_a=pim["rs_a"][s * ap:(s + Fo) * ap],
                              **({"Pt": pim["Pt"][s * tp * PPn * 2:(s + Fo) * tp * PPn * 2], "rs_p": pim["rs_p"][s * tp:(s + Fo) * tp]}
                                 if "Pt" in pim else {}),
                              **({"Ht": pim["Ht"][s * hp * PPn * 2:(s + Fo) * hp * PPn * 2], "rs_h": pim["rs_h"][s * hp:(s + Fo) * hp]}
                                 if "Ht" in pim else {}))
        img = (hat.get("imgs") or [None])[0]
        if img is not None:
            N, tp, hp = hat["N"], img["tp"], img["hp"]
            sub["imgs"] = [dict(tp=tp, hp=hp, Pt=img["Pt"][s * tp * N * 2:(s + Fo) * tp * N * 2], rs_p=img["rs_p"][s * tp:(s + Fo) * tp],
                                Ht=None if img["Ht"] is None else img["Ht"][s * hp * N * 2:(s + Fo) * hp * N * 2],
                                rs_h=None if img["rs_h"] is None else img["rs_h"][s * hp:(s + Fo) * hp])]
        return sub

    def prepare_folds(self, folds, lmax_pre, chol_after=None):
        """Everything of the given outer folds that does not touch the voxel axis beyond O(V) copies -- train-statistics
        normalisation, Lanczos, the batched Cholesky / series hat matrices -- enqueued on the engine's AUXILIARY
        stream, so that it overlaps the V-wide MFMA sweeps running on the main stream (these fp64 kernels are latency
        chains with small grids; on their own they leave most CUs idle).  ``folds``: [(tr_rows, te_rows, inner_rel)];
        ``lmax_pre``: precompute_lmax's entries for them.  Folds that share the Gram matrix and the padded system
        size go through ONE batch (a chain of ~N/64 dependent steps costs the same for 3 systems as for 30; with
        voxel shards the batch is what gets dealt out over the ranks).  Returns one state dict per fold."""
        main = torch.cuda.current_stream()
        metas = []
        for tr_rows, te_rows, inner_rel in folds:
            tr_rows = np.asarray(tr_rows, dtype=np.int64)
            te_rows = np.asarray(te_rows, dtype=np.int64)
            if len(te_rows) < 2:
                raise ValueError("x and y must have length at least 2.")      # scipy.stats.pearsonr's message
            inner_abs = [(tr_rows[np.asarray(a, dtype=np.int64)], tr_rows[np.asarray(b, dtype=np.int64)])
                         for a, b in inner_rel]
            # an inner fold WITHOUT validation rows scores NaN -> 0 for every alpha in the reference (z_score of an
            # empty block, nan_to_num: ridge_regression.py:124-133) and so adds nothing to the sum the alpha is chosen
            # from: dropped here, same result.  With no validation rows in ANY inner fold every alpha scores 0 for every
            # voxel and the reference's first-maximum argmax takes alphas[0] (the trimmed fold types in train/test mode,
            # where nested_cv.py:130-132 passes ``groups`` as the trim size, can do that): the fold then has no inner
            # CV at all -- zero scores, same choice (_sweeps).
            if not inner_abs or min(len(t) for t, _ in inner_abs) < 1:
                raise ValueError("every inner fold needs at least one training row")
            if any(len(v) == 0 for _, v in inner_abs):
                logger.warning("inner folds without validation rows contribute nothing to the alpha choice: skipped")
                inner_abs = [(t, v) for t, v in inner_abs if len(v) > 0]
            if not inner_abs:
                if self.primal:
                    raise _PrimalUnsuitable("an outer fold without validation rows in any inner fold")
                metas.append(dict(tr=tr_rows, te=te_rows, inner_abs=[], N=0, M=0, no_inner=True))
                continue
            N = ops.pad_to(max(len(t) for t, _ in inner_abs), LC_NB)
            M = ops.pad_to(max(len(v) for _, v in inner_abs), LC_MB)
            metas.append(dict(tr=tr_rows, te=te_rows, inner_abs=inner_abs, N=N, M=M))
        # groups of consecutive folds prepared as one batch: shared data (no per-fold normalisation), equal padded
        # sizes, neighbouring precomputed lmax, and the whole group's fp64 systems within the memory budget
        groups = []
        batchable = not self.norm_x and (self.primal or not self.normalpha
                                                           or all(l is not None for l in lmax_pre))
        for i, m in enumerate(metas):
            g = groups[-1] if groups else None
            per_fold = (m["N"] + m["M"]) * m["N"] * 8 * max(len(self.cho), 1) * len(m["inner_abs"])
            if self.moments:
                per_fold = 0                           # p x p systems only
            if (g is not None and batchable and not m.get("no_inner") and not metas[g[0]].get("no_inner")
                    and (metas[g[0]]["N"], metas[g[0]]["M"]) == (m["N"], m["M"])
                    and per_fold * (len(g) + 1) <= self.opt.aug_budget_bytes and self._lmax_adjacent(lmax_pre, g[-1], i)
                    and sum(len(metas[k]["inner_abs"]) for k in g) + len(m["inner_abs"]) <= MAX_INNER_FOLDS):
                g.append(i)
            else:
                groups.append([i])
        self.aux.wait_event(self.ready)                # inputs (X, Y, K) were produced on the main stream
        out = [None] * len(folds)
        with torch.cuda.stream(self.aux):
            for g in groups:
                X, K = self._fold_design(metas[g[0]]["tr"])                  # per-fold design only when len(g) == 1
                split = False if self.moments else self._split_assumed()     # the targets' side belongs to the ranges
                data_ready = torch.cuda.Event()
                data_ready.record()
                # S[0]^2 of the inner train sets and of the whole outer-train block (refit penalty scale,
                # independent of the alpha choice): precomputed for the whole fit, or one run for this fold
                lmax_i, lmax_os = None, [None] * len(g)
                if self.primal:
                    self._prepare_primal(g, metas, X, split, data_ready, out, main)
                    continue
                if self.normalpha:
                    if lmax_pre[g[0]] is None and metas[g[0]].get("no_inner"):
                        lmax_i, lmax_os = None, [None]
                    elif lmax_pre[g[0]] is None:
                        m = metas[g[0]]
                        lm = self.lmax_systems(K, [t for t, _ in m["inner_abs"]] + [m["tr"]])
                        self._check_singcutoff(lm)
                        lmax_i, lmax_os = lm[:len(m["inner_abs"])], [lm[len(m["inner_abs"]):]]
                    else:
                        lmax_os = [lmax_pre[i][1] for i in g]
                        lmax_i = lmax_pre[g[0]][0] if len(g) == 1 else self._lmax_span(lmax_pre, g)
                inner_all = [ia for i in g for ia in metas[i]["inner_abs"]]
                tr_os = [ops.idx_tensor(metas[i]["tr"], ops.pad_to(len(metas[i]["tr"]), LC_NB), self.dev).reshape(1, -1)
                         for i in g]
                if self.normalpha and lmax_os[0] is None:      # (a fold without inner CV and no precomputed values)
                    lmax_os = [ops.lambda_max(K, tr_os[0], 1, tr_os[0].shape[-1], self.steps)]
                ids_ready = torch.cuda.Event()         # what the refit systems need (row lists, lmax) exists from here on
                ids_ready.record()
                if metas[g[0]].get("no_inner"):
                    i, m = g[0], metas[g[0]]
                    done = torch.cuda.Event()
                    hat = dict(no_inner=True, info=ops.zeros(1, torch.int32, self.dev), split=split, data_ready=data_ready)
                    out[i] = dict(tr=m["tr"], te=m["te"], X=X, K=K, split=split, hat=hat, done=done, tr_o=tr_os[0],
                                  lmax_o=lmax_os[0], ids_ready=ids_ready)
                    done.record()
                    for t in (X, K, tr_os[0], lmax_os[0], hat["info"]):
                        if t is not None and t.is_cuda:
                            t.record_stream(main)
                    continue
                hat = self._hat_matrices(K, inner_all, lmax_i, self._series_by_moments(split), chol_after=chol_after)
                hat.update(split=split, data_ready=data_ready)
                done = torch.cuda.Event()
                s = 0
                for k, i in enumerate(g):
                    m = metas[i]
                    Fo = len(m["inner_abs"])
                    sub = hat if len(g) == 1 else self._hat_slice(hat, s, Fo, m["inner_abs"])
                    s += Fo
                    out[i] = dict(tr=m["tr"], te=m["te"], X=X, K=K, split=split, hat=sub, done=done,
                                  tr_o=tr_os[k], lmax_o=lmax_os[k], ids_ready=ids_ready)
                done.record()
                for t in ([X, K, hat["tr"], hat["va"], hat["info"], hat["a2"], hat["lmax"], hat["d_ser"]]
                          + [out[i]["tr_o"] for i in g] + lmax_os
                          + [h for _, _, h, _ in hat["Hs"]] + [q for _, _, _, q in hat["Hs"]]
                          + [im[k] for im in (hat.get("imgs") or []) if im is not None for k in ("Pt", "rs_p", "Ht", "rs_h")]):
                    if t is not None and t.is_cuda:
                        t.record_stream(main)              # allocated on aux, consumed on main
        return out

    # -------------------------------------------------------------- primal form (tall designs, p << n)
    def _prepare_primal(self, g, metas, X, split, data_ready, out, main):
        """prepare_folds for a group of outer folds in the PRIMAL form: with G = Rstim'Rstim (p x p),
            pred_alpha = Pstim (G + a^2 I)^-1 Rstim'Rresp  =:  A_alpha B ,   B = Rstim'Rresp  (p x V),
        the same quantity the reference forms through its thin SVD of a tall Rstim (rank p, ridge_utils.py:52;
        ridge_regression.py:104-120) and the dual route forms through n x n systems.  Per training set one p x p Gram
        matrix (lc_gram_blocks_f64 on the gathered, transposed design), S[0]^2 by Lanczos on it, and per (fold, alpha)
        an augmented p x p Cholesky system whose augmented rows are Pstim -- the same batched solver, the same sharding.
        The V-wide part (B by one contraction over the training rows, then the fused sweep of depth p) is
        _sweeps_primal."""
        if self.moments:
            return self._prepare_moments(g, metas, X, data_ready, out, main)
        PP, p, A = self.PP, self.p, self.A
        inner_all = [ia for i in g for ia in metas[i]["inner_abs"]]
        F = len(inner_all)
        n_i = [len(t) for t, _ in inner_all]
        n_v = [len(v) for _, v in inner_all]
        M = ops.pad_to(max(n_v), LC_MB)
        va = ops.idx_matrix([v for _, v in inner_all], M, self.dev)
        ident = ops.idx_matrix([np.arange(p)] * (F + len(g)), PP, self.dev)  # rows / columns of a system: 0..p-1
        # ---- round 4, designs of hundreds to thousands of features (LeBel-style train/test fits: 9000 rows x 3072):
        # (a) the alphas on the polynomial series share their terms  P'_j = Pstim G^j / lambda^(j+1)  (scored from moments
        #     in the contraction's epilogue, like the dual form's); (b) when every inner training set is its outer block
        #     minus its validation block (every fold type but the trimmed ones), the Gram matrix and the block product
        #     B = Rstim'Rresp of a training set are the SUMS over the other folds' validation blocks: one pass over the
        #     rows of the outer block instead of one per inner fold, and no transposed copy of the training rows
        use_series = bool(self.ser) and self._series_by_moments(split) and min(n_v) > 1
        cho = list(self.cho) if use_series else list(range(A))
        ser = list(self.ser) if use_series else []
        by_blocks = self.PP >= self.opt.primal_series_min_p and all(self._inner_partition(metas[i]) for i in g)
        Xt = Xt_val = None
        Nmax = ops.pad_to(max(len(t) for t in [t for t, _ in inner_all] + [metas[i]["tr"] for i in g]), LC_NB)
        rows_all = None
        blocks_ready, img = None, None
        if by_blocks:
            Xt_val = ops.gather_transpose_f32(X, va, F, M, p, PP)            # (F * PP, M): Pstim' of every inner fold
            if split:
                # its fp16 hi/lo image (the A side of the block products), once per fold, shared by every voxel range
                ap = ops.pad_to(PP, 256)
                img = dict(ap=ap, tp=0, hp=0, At=torch.empty(F * ap * M * 2, dtype=torch.float16, device=self.dev),
                           rs_a=torch.empty(F * ap, dtype=torch.float32, device=self.dev))
                ops.split_rows_f16_groups(Xt_val, F, PP, M, img["At"], img["rs_a"])
            blocks_ready = torch.cuda.Event()                                # what the block products X_v'Y_v need: they
            blocks_ready.record()                                            # run while the p x p side is still at work
            G_val = ops.gram_blocks(Xt_val, F, PP, M)                        # (F, PP, PP) f64
            G = torch.empty((F + len(g), PP, PP), dtype=torch.float64, device=self.dev)
            s0 = 0
            for k, i in enumerate(g):
                Fo = len(metas[i]["inner_abs"])
                ops.combine_many([G_val[s0 + j] for j in range(Fo)], [1.0] * Fo, G[F + k])          # the outer block's
                for j in range(Fo):
                    ops.combine_many([G_val[s0 + q] for q in range(Fo) if q != j], [1.0] * (Fo - 1), G[s0 + j])
                s0 += Fo
        else:
            sets = [t for t, _ in inner_all] + [metas[i]["tr"] for i in g]   # inner training sets, then the outer ones
            rows_all = ops.idx_matrix(sets, Nmax, self.dev)                  # (S, Nmax)
            Xt = ops.gather_transpose_f32(X, rows_all, len(sets), Nmax, p, PP)   # (S * PP, Nmax): Rstim' of every set
            G = ops.gram_blocks(Xt, len(sets), PP, Nmax)                     # (S, PP, PP) f64
        S = F + len(g)
        lmax = ops.lambda_max_strided(G, PP, PP * PP, ident, S, PP, self.steps) if self.normalpha else None
        self._check_singcutoff(lmax)
        for k in range(len(g)):
            self._check_feature_scales(G[F + k])
        a2 = ops.penalties(None if lmax is None else lmax[:F], F, self.d_alphas, self.normalpha)
        rhs = ops.gather_rows_f64(X, va, F, M, p, PP)                        # (F, M, PP): Pstim of every inner fold
        Ac = len(cho)
        grid_id = [(j // Ac) * A + cho[j % Ac] for j in range(F * Ac)] if Ac else []

        def assemble(jobs):                                                  # job -> system fold * A + alpha of the grid
            aug = torch.empty((len(jobs), PP + M, PP), dtype=torch.float64, device=self.dev)
            sysv = ops.upload(np.asarray([grid_id[j] for j in jobs], dtype=np.int32), self.dev)
            ops.batch_assemble_sel(G, ident, None, rhs, a2, sysv, len(jobs), A, PP, M, aug, k_fold_stride=PP * PP)
            return aug

        if Ac:
            H, info = self._sharded_solve(F * Ac, PP, M, assemble)           # (>= F * Ac, M, PP) f32: A_alpha
        else:
            H, info = None, ops.zeros(1, torch.int32, self.dev)
        P = None
        if use_series:
            # the shared terms of the large alphas:  P'_0 = Pstim / lambda,  P'_j = P'_(j-1) (G / lambda)  -- term j enters a
            # prediction scaled by rho^j, so fp16x3 products (22-bit operands, fp32 accumulation) keep fp32 accuracy, as
            # in the refit's chain (_refit_systems) -- placed into the slab layout of the moments epilogue
            rows_p, rowmap, _ = self._series_layout(M)
            inv = np.full(rows_p, -1, dtype=np.int32)
            live = self._rowmap_host >= 0
            inv[self._rowmap_host[live]] = np.arange(SERIES_TERMS * M, dtype=np.int32)[live]
            inv = ops.upload(inv, self.dev)
            P = torch.empty((F, rows_p, PP), dtype=torch.float32, device=self.dev)
            stack = torch.empty((SERIES_TERMS * M, PP), dtype=torch.float32, device=self.dev)
            Gn = torch.empty((1, PP, PP), dtype=torch.float32, device=self.dev)
            for f in range(F):
                ops.gather_sub_f32(G[f], ident[:1], ident[:1], 1, PP, PP, lmax[f:f + 1], Gn)
                ops.scale_cast_f64_f32(rhs[f], lmax[f:f + 1], stack[:M])
                for j in range(1, SERIES_TERMS):
                    stack[j * M:(j + 1) * M].copy_(self._times_symmetric(stack[(j - 1) * M:j * M], Gn)[0])
                ops.gather(stack, PP, inv, rows_p, None, PP, P[f])
        # the fp16 hi/lo images of the V-independent operands (the A sides of the V-wide contractions), once per fold on
        # this stream, shared by every voxel range of the fold (a host-to-host fit works through the targets panel by panel)
        if img is not None:
            tp = hp = 0
            if use_series:
                tp = ops.pad_to(P.shape[1], 256)
                img.update(Pt=torch.empty(F * tp * PP * 2, dtype=torch.float16, device=self.dev),
                           rs_p=torch.empty(F * tp, dtype=torch.float32, device=self.dev))
                ops.split_rows_f16_groups(P.view(-1, PP), F, P.shape[1], PP, img["Pt"], img["rs_p"])
            if Ac:
                hp = ops.pad_to(Ac * M, 256)
                img.update(Ht=torch.empty(F * hp * PP * 2, dtype=torch.float16, device=self.dev),
                           rs_h=torch.empty(F * hp, dtype=torch.float32, device=self.dev))
                ops.split_rows_f16_groups(H.view(-1, PP), F, Ac * M, PP, img["Ht"], img["rs_h"])
            img.update(tp=tp, hp=hp)
        hat = dict(F=F, N=PP, M=M, n_v=n_v, n_i=n_i, tr=None if rows_all is None else rows_all[:F], va=va, shared=None,
                   img=img, blocks_ready=blocks_ready,
                   Hs=[(0, F, H, P)], info=info, lmax=None if lmax is None else lmax[:F], a2=a2, cho=cho, ser=ser,
                   d_ser=self.d_ser if use_series else None, moments=use_series, series_ready=None, split=split,
                   data_ready=data_ready, Xt=Xt, Xt_val=Xt_val, Nmax=Nmax, xt_off=0)
        done = torch.cuda.Event()
        s = 0
        for k, i in enumerate(g):
            m = metas[i]
            Fo = len(m["inner_abs"])
            sub = dict(hat) if len(g) == 1 else self._hat_slice(hat, s, Fo, m["inner_abs"])
            # one tiled image of the outer training targets for all inner folds (see _shared_image): possible when the
            # inner training sets need no padding rows (only the route that contracts over the training rows uses it)
            n_in = len(m["inner_abs"][0][0])
            sub["shared"] = (self._shared_image(m["inner_abs"], n_in) if (n_in % (2 * K_TILE) == 0 and not by_blocks)
                             else None)
            s += Fo
            out[i] = dict(tr=m["tr"], te=m["te"], X=X, K=G[F + k], split=split, hat=sub, done=done,
                          tr_o=ident[:1], lmax_o=None if lmax is None else lmax[F + k:F + k + 1],
                          Xt_o=None if Xt is None else Xt[(F + k) * PP:(F + k + 1) * PP],
                          tr_o_rows=None if rows_all is None else rows_all[F + k], Nmax=Nmax)
        done.record()
        for t in (X, Xt, Xt_val, G, ident, lmax, a2, va, rows_all, rhs, H, P, info) + (
                tuple(v for v in img.values() if torch.is_tensor(v)) if img else ()):
            if t is not None and t.is_cuda:
                t.record_stream(main)                      # allocated on aux, consumed on main

    @staticmethod
    def _inner_partition(meta):
        """Every inner training set of the outer fold is its training block minus the fold's validation block, and the
        validation blocks partition the training block (K-folds, chunked folds; not the trimmed fold types, not
        time-series splits): Gram matrices and block products of the training sets are then sums over validation blocks."""
        tr = np.sort(np.asarray(meta["tr"], dtype=np.int64))
        vals = [np.asarray(v, dtype=np.int64) for _, v in meta["inner_abs"]]
        if len(vals) < 2 or sum(len(v) for v in vals) != len(tr) or not np.array_equal(np.sort(np.concatenate(vals)), tr):
            return False
        for t, v in meta["inner_abs"]:
            if len(t) + len(v) != len(tr) or not np.array_equal(np.sort(np.concatenate([np.asarray(t, dtype=np.int64), v])), tr):
                return False
        return True

    def _prepare_moments(self, g, metas, X, data_ready, out, main):
        """prepare_folds for a group of outer folds when p <= FitOptions.primal_moments_max_p and the scores are correlations
        (csrc/lc_primal.hip): every statistic of a prediction X w is a p-dimensional form in w = (G + a^2 I)^-1 Rstim'y,
        so all the V-wide work of a fold is ONE pass over the targets that forms X'y per row set (_sweeps_moments) --
        here, on the auxiliary stream, only the p x p side: per row set the column sums / second moments of the
        features, per training set its Gram matrix (an inner training set that is the outer block minus its validation
        block is taken as that difference, of the block products too), S[0]^2, and (G + a^2 I)^-1 for every alpha.
        Row sets of a fold: 0 = outer training rows, 1 = test rows, then the validation sets (and the inner training
        sets that are not such differences).  Systems of a fold: its inner folds, then the outer training set."""
        p, A = self.p, self.A
        PT = ops.primal_pad(p)
        sets, shrow, sysdef, per_fold = [], [], [], []
        for i in g:
            m = metas[i]
            s0, y0 = len(sets), len(sysdef)
            sets += [m["tr"], m["te"]]
            tr_sorted = np.sort(m["tr"])
            src = []
            for t, v in m["inner_abs"]:
                va = len(sets) - s0
                sets.append(v)
                if len(t) + len(v) == len(tr_sorted) and np.array_equal(np.sort(np.concatenate([t, v])), tr_sorted):
                    src.append((0, va, va))                         # training rows = outer block minus validation rows
                else:
                    src.append((len(sets) - s0, -1, va))
                    sets.append(t)
                sysdef.append((s0 + src[-1][0], -1 if src[-1][1] < 0 else s0 + src[-1][1]))
            sysdef.append((s0, -1))
            shrow += [int(m["tr"][0])] * (len(sets) - s0)
            per_fold.append((s0, len(sets) - s0, y0, len(src), np.asarray(src, dtype=np.int32)))
        S, n_sys = len(sets), len(sysdef)
        Nmax = ops.pad_to(max(len(r) for r in sets), 4)
        rows = ops.idx_matrix(sets, Nmax, self.dev)
        meta = ops.upload(np.concatenate([np.asarray([len(r) for r in sets], dtype=np.int32),
                                          np.asarray(shrow, dtype=np.int32),
                                          np.asarray(sysdef, dtype=np.int32).reshape(-1)]
                                         + [pf[4].reshape(-1) for pf in per_fold]), self.dev)
        nrows, shr, sysd = meta[:S], meta[S:2 * S], meta[2 * S:2 * S + 2 * n_sys]
        xstat = ops.primal_set_stats(X, p, rows, nrows, S)
        gsys = ops.primal_gsys(xstat, sysd, n_sys, p)
        lmax = None
        if self.normalpha:
            ident = ops.idx_matrix([np.arange(p)] * n_sys, PT, self.dev)
            lmax = ops.lambda_max_strided(gsys, PT, PT * PT, ident, n_sys, PT, self.steps)
            self._check_singcutoff(lmax)
        a2 = ops.penalties(lmax, n_sys, self.d_alphas, self.normalpha)
        pinv, info = ops.primal_inverse(gsys, a2, n_sys, A, p)
        done = torch.cuda.Event()
        off = 2 * S + 2 * n_sys
        for k, i in enumerate(g):
            m = metas[i]
            s0, ns, y0, F, src = per_fold[k]
            hat = dict(moments_p=True, F=F, X=X, rows=rows[s0:s0 + ns], nrows=nrows[s0:s0 + ns], shrow=shr[s0:s0 + ns],
                       n_sets=ns, src=meta[off:off + 3 * F], xstat=xstat[s0:s0 + ns], pinv=pinv[y0 * A:(y0 + F) * A],
                       pinv_o=pinv[(y0 + F) * A:(y0 + F + 1) * A], info=info[y0 * A:(y0 + F) * A],
                       info_o=info[(y0 + F) * A:(y0 + F + 1) * A], data_ready=data_ready, cs=None, split=False)
            off += 3 * F
            out[i] = dict(tr=m["tr"], te=m["te"], X=X, K=None, split=False, hat=hat, done=done)
        done.record()
        for t in (X, rows, meta, xstat, gsys, lmax, a2, pinv, info):
            if t is not None and t.is_cuda:
                t.record_stream(main)                      # allocated on aux, consumed on main

    def _sweeps_moments(self, hat, Y, done=None):
        """_sweeps of the moments form: one pass over the fold's targets (block products of every row set), then the
        per-voxel scores of all inner folds and alphas from them."""
        main = torch.cuda.current_stream()
        if hat.get("data_ready") is not None:
            main.wait_event(hat["data_ready"])
        if done is not None:
            main.wait_event(done)
        self.info.update(precision="f64 block products", fused_alphas=self.A, series_terms=0)
        hat["part"] = ops.xty(hat["X"], self.p, Y, self.V, hat["rows"], hat["nrows"], hat["shrow"], hat["n_sets"])
        scores = torch.empty((self.A, self.Vp), dtype=torch.float32, device=self.dev)
        ops.primal_scores(hat["part"], hat["nrows"], hat["shrow"], Y, self.V, hat["src"], hat["xstat"], hat["pinv"],
                          hat["F"], self.A, self.p, scores)
        self.sweeps_done = torch.cuda.Event()
        self.sweeps_done.record()
        return scores

    def _check_feature_scales(self, G_o):
        """The primal V-wide contraction sums over FEATURES: with the fp16 hi/lo operands (22 bits relative to a
        row's / column's largest entry) a feature whose scale is orders of magnitude below another's would lose
        its digits.  One look at the column norms (the Gram diagonal; a p-long copy to page-locked memory, looked at
        by _verify_feature_scales when the fold's first V-wide phase is queued -- the host does not wait here, where it
        would wait for everything queued on this stream before: round 4)."""
        if self.precision == "f32":
            return
        h = torch.empty(self.p, dtype=torch.float64, pin_memory=True)
        h.copy_(G_o.diagonal()[: self.p], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._scale_checks.append((ev, h))

    def _verify_feature_scales(self):
        while self._scale_checks:
            ev, h = self._scale_checks.pop(0)
            ev.synchronize()
            d = np.sqrt(h.numpy())
            d = d[d > 0]
            if d.size and float(d.max() / d.min()) > self.opt.primal_max_scale_ratio:
                self._scale_checks.clear()
                raise _PrimalUnsuitable(f"feature column norms span a factor {float(d.max() / d.min()):.3g}")

    def _sweeps_primal(self, hat, Y, done=None):
        """_sweeps in the primal form: per inner fold  B = Rstim'Rresp  (one plain contraction over the training rows,
        p_pad x V), then the fused sweep of all alphas at depth p_pad:  pred_alpha = A_alpha B, scored in the epilogue
        exactly as in the dual form (same kernel, same validation statistics)."""
        if hat.get("moments_p"):
            return self._sweeps_moments(hat, Y, done)
        A, PP, M, tr, va, n_v, n_i = self.A, hat["N"], hat["M"], hat["tr"], hat["va"], hat["n_v"], hat["n_i"]
        F, Xt, Xt_val, Nmax, off = hat["F"], hat.get("Xt"), hat.get("Xt_val"), hat["Nmax"], hat["xt_off"]
        (_, _, H, P), = hat["Hs"]
        moments, cho = hat["moments"], hat["cho"]
        Ad = len(cho)                                      # alphas with hat matrices (all of them without the series)
        cho_first = list(cho) == list(range(Ad))
        main = torch.cuda.current_stream()
        split, cs = hat["split"], hat["cs"]
        if hat.get("data_ready") is not None:
            main.wait_event(hat["data_ready"])
        by_blocks = Xt_val is not None
        img = hat.get("img") if split else None
        # the block products need the transposed validation rows only: they run BEFORE the wait for the fold's p x p side
        # (Lanczos run, Cholesky chains, series terms: ~40 ms at the LeBel shape, while the first target panels land)
        if by_blocks and hat.get("blocks_ready") is not None:
            main.wait_event(hat["blocks_ready"])
        elif done is not None:
            main.wait_event(done)
        self.info.update(precision="f16x3" if split else "f32", fused_alphas=Ad, series_terms=SERIES_TERMS if moments else 0)
        Vp_, V_ = self.Vp, self.V
        scores = torch.empty((A, Vp_), dtype=torch.float32, device=self.dev)
        scores_d = scores if not moments else (scores[:Ad] if cho_first else
                                               torch.empty((max(Ad, 1), Vp_), dtype=torch.float32, device=self.dev))
        part = torch.empty((max(Ad, 1) * M // LC_MB, 4, Vp_), dtype=torch.float32, device=self.dev)
        ystat = torch.empty((3, Vp_), dtype=torch.float32, device=self.dev)
        yblk = torch.empty((M // LC_MB, Vp_), dtype=torch.float32, device=self.dev)
        yv = torch.empty((M, Vp_), dtype=torch.float32, device=self.dev)
        Vt = ops.pad_to(Vp_, 256)
        B = ops.zeros((PP, Vt), torch.float32, self.dev)
        ident = ops.idx_tensor(np.arange(self.p), PP, self.dev)
        shared = hat.get("shared") if split else None
        views = [(0, 0, 0)] * F
        if split:
            depth = M if by_blocks else Nmax
            At = torch.empty(ops.pad_to(PP, 256) * depth * 2, dtype=torch.float16, device=self.dev)
            rs_a = torch.empty(ops.pad_to(PP, 256), dtype=torch.float32, device=self.dev)
            if shared is not None:
                union, gaps = shared
                Yt = torch.empty(Vt * len(union) * 2, dtype=torch.float16, device=self.dev)
                ops.split_cols_f16(Y, Vp_, ops.idx_tensor(union, len(union), self.dev), len(union), cs, Yt)
                views = [(len(union), g0, gl) for g0, gl in gaps]
            else:
                Yt = torch.empty(Vt * depth * 2, dtype=torch.float16, device=self.dev)
            cs_inv = self._cs_inv_padded(cs, Vt)
            Bt = torch.empty(Vt * PP * 2, dtype=torch.float16, device=self.dev)
            if Ad:
                rows_pad = ops.pad_to(Ad * M, 256)
                Ht = torch.empty(rows_pad * PP * 2, dtype=torch.float16, device=self.dev)
                rs_inv = torch.empty(rows_pad, dtype=torch.float32, device=self.dev)
            if moments:
                Tm, rowmap, _ = self._series_layout(M)
                tp = ops.pad_to(Tm, 256)
                Pt = torch.empty(tp * PP * 2, dtype=torch.float16, device=self.dev)
                rs_p = torch.empty(tp, dtype=torch.float32, device=self.dev)
                part_s = torch.empty((1, M // LC_MB, 18, Vp_), dtype=torch.float32, device=self.dev)
        Bv = None
        if by_blocks:
            # the block products of the validation blocks, X_v' Y_v (one pass over the rows of the outer training block);
            # an inner training set's  B = Rstim'Rresp  is the sum over the OTHER folds' (fp32 adds, fold order)
            Bv = torch.empty((F, PP, Vt), dtype=torch.float32, device=self.dev)
            for f in range(F):
                Xv = Xt_val[(off + f) * PP:(off + f + 1) * PP]
                if split:
                    ops.split_cols_f16(Y, Vp_, va[f], M, cs, Yt)
                    At_f, rs_a_f = At, rs_a
                    if img is not None:                    # (the images of a fold group are sliced per outer fold: index f)
                        At_f, rs_a_f = img["At"][f * img["ap"] * M * 2:], img["rs_a"][f * img["ap"]:]
                    else:
                        ops.split_rows_f16(Xv, PP, M, At, rs_a)
                    ops.gemm_grouped_f16x3(At_f, rs_a_f, PP, Yt, cs_inv, Bv[f], Vt, Vt, M, [0, Vt // 256])
                    self.info["plain_flops"] += 2.0 * self.p * n_v[f] * V_
                    self.info["plain_launches"] += 1
                else:
                    ops.gemm_grouped(Xv, M, 0, Y, Y.stride(0), va[f], Bv[f], Vt, PP, Vp_, M, [0, Vp_ // COL_TILE])
        if by_blocks and done is not None:
            main.wait_event(done)                          # from here on: the hat matrices / series terms of the fold
        for f in range(F):
            ops.val_stats(Y, Vp_, va[f], M, n_v[f], ystat, yblk, yv)
            if by_blocks:
                ops.combine_many([Bv[q] for q in range(F) if q != f], [1.0] * (F - 1), B)
            else:
                Ni = ops.pad_to(n_i[f], 2 * K_TILE)                               # contraction depth, padded rows are -1
                Xt_f = Xt[(off + f) * PP:(off + f + 1) * PP]
            if split:
                if not by_blocks:
                    if shared is None:
                        ops.split_cols_f16(Y, Vp_, tr[f], Ni, cs, Yt)
                    ops.split_rows_f16(Xt_f, PP, Ni, At, rs_a)
                    ops.gemm_grouped_f16x3(At, rs_a, PP, Yt, cs_inv, B, Vt, Vt, Ni, [0, Vt // 256], bview=views[f])
                    self.info["plain_flops"] += 2.0 * self.p * n_i[f] * V_
                    self.info["plain_launches"] += 1
                csB, _ = ops.col_scales_f16(B, self.p, Vp_, want_flag=False)
                ops.split_cols_f16(B, Vp_, ident, PP, csB, Bt)
                if moments:
                    Pt_f, rs_p_f = Pt, rs_p
                    if img is not None and "Pt" in img:
                        Pt_f, rs_p_f = img["Pt"][f * img["tp"] * PP * 2:], img["rs_p"][f * img["tp"]:]
                    else:
                        ops.split_rows_f16(P[f], Tm, PP, Pt, rs_p)
                    self.info["plain_flops"] += 2.0 * SERIES_TERMS * n_v[f] * self.p * V_
                    self.info["plain_launches"] += 1
                    ops.series_sweep_scores_f16x3(Pt_f, rs_p_f, M, n_v[f], PP, Bt, self._cs_inv_padded(csB, Vt), Vt, yv, Vp_,
                                                  ystat, yblk, self.d_coef, hat["d_ser"], part_s, scores, accumulate=f > 0)
                if Ad:
                    Ht_f, rs_h_f = Ht, rs_inv
                    if img is not None and "Ht" in img:
                        Ht_f, rs_h_f = img["Ht"][f * img["hp"] * PP * 2:], img["rs_h"][f * img["hp"]:]
                    else:
                        ops.split_rows_f16(H[f * Ad:(f + 1) * Ad].reshape(Ad * M, PP), Ad * M, PP, Ht, rs_inv)
                    self.info["fused_flops"] += 2.0 * Ad * n_v[f] * self.p * V_
                    self.info["fused_launches"] += 1
                    ops.alpha_sweep_scores_f16x3(Ht_f, rs_h_f, Ad, M, PP, Bt, csB[Vp_:], yv, Vp_, n_v[f], ystat, yblk,
                                                 self.mode, part, scores_d, accumulate=f > 0)
            else:
                if not by_blocks:
                    ops.gemm_grouped(Xt_f, Nmax, 0, Y, Y.stride(0), tr[f], B, Vt, PP, Vp_, Ni, [0, Vp_ // COL_TILE])
                ops.alpha_sweep_scores(H[f * A:(f + 1) * A], A, M, PP, B, Vp_, ident, yv, n_v[f], ystat, yblk,
                                       self.mode, part, scores, accumulate=f > 0)
        if moments and Ad and not cho_first:
            for i, a in enumerate(cho):
                scores[a].copy_(scores_d[i])
        if by_blocks:
            # the outer block's product = the sum of all its validation blocks': the refit's operand (_primal_refit_inputs)
            hat["B_all"] = ops.combine_many([Bv[q] for q in range(F)], [1.0] * F, B)
        self.sweeps_done = torch.cuda.Event()
        self.sweeps_done.record()
        return scores

    def _primal_refit_inputs(self, st):
        """The primal refit contracts over features:  [ (G + a^2 I)^-1 ; X_te (G + a^2 I)^-1 ] . B_o  with
        B_o = Rstim'Rresp of the outer training block.  Returns the stand-ins for (Y, training rows, test rows, column
        scales) that _refit_operands takes in the dual form: the (p_pad + n_t) x V matrix [B_o ; Y_te]."""
        PP, Y, te = self.PP, st["Y"], st["te"]
        n_t = len(te)
        Vt = ops.pad_to(self.Vp, 256)
        No = ops.pad_to(len(st["tr"]), 2 * K_TILE)
        ext = ops.zeros((PP + n_t, Vt), torch.float32, self.dev)
        B_all = st["hat"].get("B_all")
        if B_all is not None:
            # the inner CV of this step left  B_o = Rstim'Rresp  of the outer block (the sum of its validation blocks')
            ext[:PP].copy_(B_all)
            csB = ops.col_scales_f16(ext, self.p, self.Vp, want_flag=False)[0] if st["split"] else None
        elif st["split"]:
            At = torch.empty(ops.pad_to(PP, 256) * st["Nmax"] * 2, dtype=torch.float16, device=self.dev)
            rs_a = torch.empty(ops.pad_to(PP, 256), dtype=torch.float32, device=self.dev)
            Yt = torch.empty(Vt * No * 2, dtype=torch.float16, device=self.dev)
            cs_inv = self._cs_inv_padded(st["cs"], Vt)
            ops.split_cols_f16(Y, self.Vp, st["tr_o_rows"], No, st["cs"], Yt)
            ops.split_rows_f16(st["Xt_o"], PP, No, At, rs_a)
            ops.gemm_grouped_f16x3(At, rs_a, PP, Yt, cs_inv, ext, Vt, Vt, No, [0, Vt // 256])
            csB, _ = ops.col_scales_f16(ext, self.p, self.Vp, want_flag=False)
        else:
            ops.gemm_grouped(st["Xt_o"], st["Nmax"], 0, Y, Y.stride(0), st["tr_o_rows"], ext, Vt, PP, self.Vp, No,
                             [0, self.Vp // COL_TILE])
            csB = None
        ops.gather(Y, Y.stride(0), ops.idx_tensor(te, n_t, self.dev), n_t, None, self.Vp, ext[PP:])
        return ext, np.arange(PP), PP + np.arange(n_t), csB

    @staticmethod
    def _lmax_adjacent(lmax_pre, i, j):
        """The precomputed inner-fold lmax of folds i and j are neighbouring slices of one vector."""
        if lmax_pre[i] is None or lmax_pre[j] is None:
            return lmax_pre[i] is None and lmax_pre[j] is None
        a, b = lmax_pre[i][0], lmax_pre[j][0]
        return (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                and a.storage_offset() + a.numel() == b.storage_offset())

    @staticmethod
    def _lmax_span(lmax_pre, g):
        """One view over the neighbouring inner-fold lmax slices of the folds in ``g``."""
        first, last = lmax_pre[g[0]][0], lmax_pre[g[-1]][0]
        n = last.storage_offset() + last.numel() - first.storage_offset()
        return torch.as_strided(first, (n,), (1,), first.storage_offset())

    def fold_begin(self, tr_rows, te_rows, inner_rel, prepared=None, lmax_pre=None, step=None, split_phase=False):
        """The V-wide inner CV of one (fold, voxel range) step.  ``prepared``: the fold's V-independent state
        (prepare_folds), shared by all ranges of the fold; ``step`` = (fold number, (c0, c1)) from plan_steps, default:
        fold 0, all columns.  Returns the step's own state: the fold's entries plus the range's targets, column scales
        and scores."""
        base = prepared if prepared is not None else self.fold_prepare(tr_rows, te_rows, inner_rel, lmax_pre)
        if self._scale_checks:
            self._verify_feature_scales()              # primal form: may send the driver to the dual form
        fold_no, cols = step if step is not None else (0, None)
        rg = self.full if cols is None else self.range_of(*cols)
        st = dict(base)
        st.update(base=base, rg=rg, fold=int(fold_no))
        self._enter(st)
        self._wait_targets(rg)
        Y, cs, split = self._fold_targets(rg, base["tr"])
        if bool(split) != bool(base["split"]) and not self.moments:
            raise _WideTargets("the fold's operators were prepared for the other arithmetic")
        hat = dict(base["hat"])
        hat.update(cs=cs, split=split)
        st.update(Y=Y, cs=cs, split=split, hat=hat)
        st["info"] = hat["info"]
        if split_phase:
            # only the part of the sweeps that does not wait for the fold's Cholesky chains; fold_sweeps_finish queues the
            # rest (the driver puts the next step's first part in between)
            st["sweeps_rest"] = self._sweeps(hat, Y, st["done"], split_phase=True)
            st["scores"] = None
            return st
        st["scores"] = self._sweeps(hat, Y, st["done"])
        return st

    def fold_sweeps_finish(self, st):
        """Second part of a step begun with ``split_phase``: the fused sweeps behind the fold's hat matrices."""
        if st.get("sweeps_rest") is not None:
            self._enter(st)
            st["scores"] = st.pop("sweeps_rest")()
        return st

    def _refit_stream(self, st):
        """The refit systems run on a SECOND auxiliary stream, ordered behind the fold's prepare: their chain of short
        fp64 launches then interleaves with the inner-fold chain of the fold after next on the first one -- each
        launch waits for workgroup slots the MFMA sweeps own, and two chains wait in parallel (cfg2 fit 166.7 -> 161.2
        ms, three interleaved pairs on one box)."""
        s2 = self.aux2
        if st.get("done") is not None:
            s2.wait_event(st["done"])
        for t in (st.get("X"), st.get("K"), st.get("tr_o"), st.get("lmax_o")):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(s2)                    # made on the first auxiliary stream (or at set-up), read here
        return s2

    def chain_gate(self):
        """Event the NEXT fold_prepare's Cholesky chain waits for (see the driver loop): the end of the sweeps queued
        last -- on one GPU, where the fp64 chains would otherwise take CUs from the dominant MFMA kernel at no gain
        in fit time.  With voxel shards the V-wide work per rank is a fraction and the chains are the critical path:
        no gate."""
        return self.sweeps_done if self.shard.world == 1 else None

    def refit_ahead_pays(self):
        """Forming the refit operators of EVERY factorised alpha of every fold before any alpha is chosen is cheap enough
        on one GPU when they come from explicit inverses (N^3 flops each) and the grid has only a few such alphas."""
        return (bool(self.cho) and len(self.cho) <= 8 and not self.primal and self._refit_by_inverse(self.cho)
                and self.speculation_pays())

    def speculation_pays(self):
        """Refit systems solved BEFORE the alpha choice cost N^3 fp64 flops each whether or not their alpha is chosen: at
        cfg2's 2400 training rows that is 14 GFLOP (0.4 ms), hidden beside the sweeps; at 9000 rows (LeBel-style
        train/test fits) 730 GFLOP -- ~20 ms of the fp64 pipe per alpha nobody may choose.  Ahead only while cheap."""
        return self.primal or self.Ttot <= self.opt.speculate_max_rows        # (primal: p x p systems, always cheap)

    def refit_ahead(self, states, alphas=None, after_hat=False):
        """Voxel shards: the refit systems of ALL the given (prepared) folds for ALL factorised alphas in one
        collective batch, before any alpha is chosen.  With W ranks a rank's share of a fold's handful of systems is a
        chain of ~N/64 dependent steps either way (latency, not flops), and solving them fold by fold after each
        choice puts that chain -- and its all-gather -- on the critical path of every fold; one batch over the folds
        costs one chain for the whole fit, hidden behind the first folds' sweeps.  (On one GPU the systems of alphas
        nobody chooses would be wasted fp64 work, so there the driver keeps fold_speculate.)  Folds whose systems
        differ in size fall back to fold_speculate / fold_select."""
        cho = [a for a in self.cho if alphas is None or a in alphas]     # ``alphas``: only these (a first choice is known)
        sts = [st for st in states if st.get("tr_o") is not None and "spec" not in st]
        if not cho or not sts or self.primal or self.spectral:
            return
        N_o = sts[0]["tr_o"].shape[-1]
        rs_stream = self.aux2
        for st in sts:                                 # not behind the folds' hat-matrix batches: beside them (default)
            rs_stream.wait_event(st["done"] if (after_hat and st.get("done") is not None)
                                 else (st.get("ids_ready") or st["done"]))
        with torch.cuda.stream(rs_stream):
            rhss = [self._refit_rhs(st["X"], st["K"], st["tr"], st["tr_o"], st["te"]) for st in sts]
            rows = rhss[0].shape[0]
            if any(st["tr_o"].shape[-1] != N_o for st in sts) or any(r.shape[0] != rows for r in rhss):
                return
            Gc, nF = len(cho), len(sts)
            a2s = [ops.penalties(st["lmax_o"], 1, self.d_alphas, self.normalpha) for st in sts]
            if self._refit_by_inverse(cho):
                # one explicit inverse per (fold, alpha) -- N^3 flops each, no row slices -- all-gathered; a fold applies
                # the inverses of the alphas its voxels chose to its rows on the MFMA (_refit_systems)
                eye = self._identity_rows(N_o)

                def assemble_inv(jobs):                # job = fold * Gc + alpha
                    aug = torch.empty((len(jobs), 2 * N_o, N_o), dtype=torch.float64, device=self.dev)
                    for k, j in enumerate(jobs):
                        sysv = ops.upload(np.asarray([cho[j % Gc]], dtype=np.int32), self.dev)
                        ops.batch_assemble_sel(sts[j // Gc]["K"], sts[j // Gc]["tr_o"], None, eye, a2s[j // Gc], sysv, 1,
                                               self.A, N_o, N_o, aug[k:k + 1])
                    return aug

                Pj, info = self._sharded_solve(nF * Gc, N_o, N_o, assemble_inv, lane="refit", inverse=True)
                Pall = Pj[: nF * Gc].view(nF, Gc, N_o, N_o)
                ready = torch.cuda.Event()
                ready.record()
                by_alpha = self._flags_by_alpha(info, nF * Gc, lambda j: (j // Gc, cho[j % Gc]), nF)
                for fo, st in enumerate(sts):
                    st["spec"] = dict(alphas=list(cho), M=None, P=Pall[fo], info=info, rhs=rhss[fo], ready=ready,
                                      info_by_alpha=by_alpha[fo])
                    for t in (st.get("X"), st.get("K"), st.get("tr_o"), st.get("lmax_o")):
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(rs_stream)
                return
            S = 1
            while S * Gc * nF < self.shard.world and rows % (2 * S * LC_MB) == 0:
                S *= 2
            rsz = rows // S

            def assemble(jobs):                        # job = (fold * Gc + alpha) * S + row slice
                aug = torch.empty((len(jobs), N_o + rsz, N_o), dtype=torch.float64, device=self.dev)
                for k, j in enumerate(jobs):
                    fo, a, q = j // (Gc * S), (j // S) % Gc, j % S
                    sysv = ops.upload(np.asarray([cho[a]], dtype=np.int32), self.dev)
                    ops.batch_assemble_sel(sts[fo]["K"], sts[fo]["tr_o"], None, rhss[fo][q * rsz:(q + 1) * rsz], a2s[fo], sysv,
                                           1, self.A, N_o, rsz, aug[k:k + 1])
                return aug

            Hj, info = self._sharded_solve(nF * Gc * S, N_o, rsz, assemble, lane="refit")
            Mall = Hj[: nF * Gc * S].view(nF, Gc, rows, N_o)
            ready = torch.cuda.Event()
            ready.record()
            by_alpha = self._flags_by_alpha(info, nF * Gc * S, lambda j: (j // (Gc * S), cho[(j // S) % Gc]), nF)
        for fo, st in enumerate(sts):
            st["spec"] = dict(alphas=list(cho), M=Mall[fo], info=info, rhs=rhss[fo], ready=ready, info_by_alpha=by_alpha[fo])
            for t in (st.get("X"), st.get("K"), st.get("tr_o"), st.get("lmax_o")):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(rs_stream)

    def _flags_by_alpha(self, info, n_jobs, job_fold_alpha, n_folds):
        """Pivot flags of THIS rank's share of a batch of jobs, sorted per (fold, alpha): [{alpha: [one-entry views]}] --
        a fold later joins only those of alphas somebody chose."""
        _, mine = job_share(n_jobs, self.shard.world, self.shard.rank)
        out = [dict() for _ in range(n_folds)]
        for k, j in enumerate(mine):
            fo, a = job_fold_alpha(j)
            out[fo].setdefault(a, []).append(info[k:k + 1])
        return out

    def fold_speculate(self, st, alphas_idx, early=False):
        """Solve the refit systems of a prepared fold for the listed alphas BEFORE its alpha choice is known, on the
        auxiliary stream (the driver passes the alphas the previous fold used: the histogram of the chosen alphas
        hardly moves between outer folds).  fold_select then only solves what is missing; without this the last
        fold's systems are a serial 8 ms at the end of the fit, with nothing left to run beside them.  ``early`` (the
        FIRST fold, whose systems nothing can predict: all factorised alphas): beside the fold's own hat-matrix chain
        instead of behind it -- the chip is idle then, and the fold's refit otherwise waits for a chain that can only
        start once its first histogram is on the host."""
        st = st.get("base", st)                        # the fold's V-independent state (shared by its voxel ranges)
        todo = [a for a in alphas_idx if a in self.cho] if st.get("tr_o") is not None else []
        if not todo or "spec" in st:                   # nothing to factor, or refit_ahead has covered the fold
            return
        if early and st.get("ids_ready") is not None:
            rs = self.aux2
            rs.wait_event(st["ids_ready"])
            for t in (st.get("X"), st.get("K"), st.get("tr_o"), st.get("lmax_o")):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(rs)
        else:
            rs = self._refit_stream(st)
            if self.chain_gate() is not None:
                rs.wait_event(self.chain_gate())       # like the inner-fold chain: not beside the sweeps just queued
        with torch.cuda.stream(rs):
            rhs = self._refit_rhs(st["X"], st["K"], st["tr"], st["tr_o"], st["te"])
            Mc, info = self._refit_chol(st["K"], st["tr_o"], st["lmax_o"], rhs, todo)
        # flags per alpha (one rank: job k = alpha todo[k]): a fold joins only those of alphas somebody chose
        by_alpha = {a: [info[k:k + 1]] for k, a in enumerate(todo)} if self.shard.world == 1 else None
        st["spec"] = dict(alphas=todo, M=Mc, info=info, rhs=rhs, info_by_alpha=by_alpha)

    def fold_choose(self, st, single_alpha):
        """Alpha choice of the fold and the grouping of the voxels by it, enqueued behind the fold's sweeps; the
        histogram travels to pinned memory asynchronously, so the caller can queue the next fold's sweeps on the
        main stream BEFORE waiting for it in fold_select (the stream then never idles through the host round trip)."""
        self._enter(st)
        st["best"] = self.choose(st["scores"], single_alpha)
        if self.opt.alpha_progress_log and logger.isEnabledFor(logging.INFO):
            # ridge_regression.py:136-139 logs "Alpha=..., mean corr=..." per alpha and inner fold; here the scores exist
            # as the sum over the inner folds, so one line per alpha and outer fold (a device round trip: opt-in)
            _, rowsum = ops.select_alpha(st["scores"], self.A, self.Vp, want_best=False, want_rowsum=True)
            nf = max(1, int(st["hat"].get("F", 1)))
            for a_, tot in zip(self.alphas, rowsum.cpu().tolist()):
                logger.info("Alpha=%.3f, mean corr=%.5f (mean over %d inner folds and %d voxels)", a_,
                            tot / (nf * max(self.V, 1)), nf, self.V)
        if not self.moments:                           # the moments form refits voxel by voxel: no grouping by alpha
            st["grouping"] = self._group_async(st["best"], st["split"])
        return st

    def fold_choose_joint(self, sts):
        """``single_alpha`` when a fold is worked through in several voxel ranges (host inputs arriving panel by panel):
        the ONE alpha is the argmax of the across-voxel mean of the scores (nested_cv.py:396-400), so the per-alpha sums
        of all ranges -- and of all voxel shards -- are added up on the device before any range is grouped.  Every
        range's state gets its ``best`` vector and its grouping, as fold_choose would give it."""
        total = None
        for st in sts:
            self._enter(st)
            _, rowsum = ops.select_alpha(st["scores"], self.A, self.Vp, want_best=False, want_rowsum=True)
            total = rowsum if total is None else ops.accumulate_f64(rowsum, total)
        self.shard.all_reduce_(total, "sum")
        for st in sts:
            self._enter(st)
            best = torch.empty(self.Vp, dtype=torch.int32, device=self.dev)
            st["best"] = ops.fill_argmax(total, self.A, best, self.Vp)
            if not self.moments:
                st["grouping"] = self._group_async(st["best"], st["split"])
        return sts

    def fold_select(self, st, single_alpha):
        """Waits for the fold's alpha histogram (fold_choose; the one host synchronisation of a fold) and puts the
        fp64 systems of the refit on the auxiliary stream -- they run beside whatever the main stream does next."""
        self._enter(st)
        if self.moments:                               # nothing to factor after the choice, and no host sync
            if "best" not in st:
                self.fold_choose(st, single_alpha)
            st.update(used=[], used_all=[])
            return st
        if "grouping" not in st:
            self.fold_choose(st, single_alpha)
        best, split = st["best"], st["split"]
        perm, used, tiles, Vs, used_all = self._refit_groups(best, split, st.pop("grouping"))
        main = torch.cuda.current_stream()
        base = st.get("base", st)
        spec = base.get("spec")
        cache = base.setdefault("refit_cache", {})
        if spec is not None and spec.get("ready") is not None:
            # voxel shards: the fold's factorised systems came from refit_ahead; what is left (the shared powers of the
            # polynomial alphas, copies) must not queue behind the later folds' batches on the refit stream
            rs = self.aux3
            rs.wait_event(spec["ready"])
            rs.wait_event(st["done"])
            for t in (st.get("X"), st.get("K"), st.get("tr_o"), st.get("lmax_o"), spec["M"], spec.get("P"), spec["rhs"],
                      spec["info"]):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(rs)
        else:
            rs = self._refit_stream(st)                    # inputs: X, K, tr_o, lmax_o -- all made on aux or at start
        with torch.cuda.stream(rs):
            Malpha, info_o = self._refit_systems(st["X"], st["K"], st["tr"], used, st.get("tr_o"), st.get("lmax_o"),
                                                 st["te"], spec=spec, used_all=used_all, cache=cache)
            ready = torch.cuda.Event()
            ready.record()
        for x in list(Malpha) + [info_o]:
            x.record_stream(main)
        st.update(best=best, perm=perm, used=used, used_all=used_all, tiles=tiles, Vs=Vs, split=split, Malpha=Malpha,
                  info_o=info_o, systems_ready=ready)
        return st

    def fold_finish(self, st, weight_scale):
        """V-wide half of the refit of one (fold, voxel range) step, test predictions, Pearson r / p-values.  Returns the
        pending results of the FOLD (see _publish) when this was the last range of the fold to finish, else None."""
        self._enter(st)
        rg = st["rg"]
        tr_rows, te_rows, Y = st["tr"], st["te"], st["Y"]
        n_t = len(te_rows)
        if self.moments:
            # per voxel: weights at its alpha from the outer block product, accumulated into W; Pearson r of the test
            # rows from the test block product (lc_primal_refit) -- natural voxel order, no sorted copy, no scatter
            hat, best = st["hat"], st["best"]
            r_d = torch.empty(max(self.V, 1), dtype=torch.float64, device=self.dev)
            ops.primal_refit(hat["part"], hat["nrows"], hat["shrow"], Y, self.V, 0, 1, hat["xstat"], hat["pinv_o"], best,
                             self.p, weight_scale, self.W_acc, r_d)
            p_d = ops.pearson_pvalues(r_d, self.V, n_t)
            if rg.natural is None:
                rg.natural = ops.upload(np.arange(max(self.V, 1), dtype=np.int32), self.dev)
            pend = self._publish(st, r_d, p_d, rg.natural, self.V, best, st["info"], hat["info_o"], n_t)
            self._range_finished(st)
            return pend
        best, perm, Vs = st["best"], st["perm"], st["Vs"]
        torch.cuda.current_stream().wait_event(st["systems_ready"])
        row0 = self.p_pad                              # first row of the test-row hat matrix inside M_alpha
        if self.primal:
            row0 = self.PP
            ext, rows_b, rows_t, csB = self._primal_refit_inputs(st)
            o = self._refit_operands(ext, rows_b, rows_t, perm, st["tiles"], Vs, st["Malpha"], st["split"], csB)
        else:
            o = self._refit_operands(Y, tr_rows, te_rows, perm, st["tiles"], Vs, st["Malpha"], st["split"], st["cs"],
                                     image=st["hat"].get("image"))
        # ---- test predictions first (nested_cv.py:151,251: X_te W, here as the hat matrix of the test rows applied
        # to the same targets) and per-voxel Pearson r (:152-155, 252-257); the weight rows of the same contraction
        # follow once the fold's results are on their way to the host
        o.update(used=tuple(st["used"]), img_cache=st.get("base", st).setdefault("refit_cache", {}).setdefault("imgs", {}))
        pred = self._refit_product(o, row0, st["Malpha"][0].shape[0], n_t)[:n_t]
        if o.get("te_src") is not None:
            r_s = ops.pearson_cols_gather(*o["te_src"], pred, n_t, Vs)
        else:
            r_s = ops.pearson_cols(o["Ys_te"], pred, n_t, Vs)
        p_s = ops.pearson_pvalues(r_s, Vs, n_t)
        pend = self._publish(st, r_s, p_s, perm, Vs, best, st["info"], st["info_o"], n_t)
        # the weights last: nothing the host waits for depends on them (for the last fold the host statistics then
        # run beside this part of the contraction)
        # the weight rows stay in alpha-sorted order where the contraction writes them; the mean over the folds is taken
        # in one pass per voxel range once its last fold is in (_combine_weights), not accumulated fold by fold
        ent, off = self._ws_slot(st["fold"], rg, Vs, weight_scale)
        self._refit_product(o, 0, self.p_pad, self.p, out=ent["buf"][:, off:off + Vs])
        ops.invert_perm(perm, Vs, off, ent["pos"][rg.c0:])
        self._range_finished(st)
        return pend

    def _ws_slot(self, fold, rg, Vs, scale):
        """Where the alpha-sorted weight columns of a (fold, voxel range) step go: one (p_pad, cap) matrix per fold, the
        ranges of the fold side by side (cap covers every range's padding to whole column tiles per alpha group), plus
        the fold's position list  pos[voxel] = its column  (lc_invert_perm)."""
        ent = self._ws.get(fold)
        if ent is None or ent["cols"] >= self.V_rank:          # (a fold number coming round again: a new fit of the engine)
            cap = ops.pad_to(max(self.V_rank, 1), 256) + 256 * self.A * max(1, len(self.upload_panels), len(self.download_panels))
            ent = self._ws[fold] = dict(buf=torch.empty((self.p_pad, cap), dtype=torch.float32, device=self.dev),
                                        pos=ops.filled((max(self.V_rank, 1),), torch.int32, self.dev, 0xFF),
                                        used=0, cols=0, scale=float(scale))
        off = ent["used"]
        if off + Vs > ent["buf"].shape[1]:
            raise RuntimeError("alpha-sorted weight buffer of the fold is full (more voxel ranges than planned)")
        ent["used"] += Vs
        ent["cols"] += rg.V
        return ent, off

    def _combine_weights(self, rg):
        """The mean weights of a voxel range, once its last fold is in:  W[:, v] = sum_f scale_f Ws_f[:, pos_f[v]]  in
        fold order (lc_combine_folds_f32: one gather per fold and element, one write -- the accumulate it replaces
        read and re-wrote the whole accumulator once per fold; same expression per term, same bits)."""
        parts = [(self._ws[f]["buf"], self._ws[f]["pos"][rg.c0:], self._ws[f]["scale"]) for f in sorted(self._ws)]
        ops.combine_folds(parts, self.p, rg.V, rg.W)
        self._combined += rg.V
        if self._combined >= self.V_rank:
            self._ws = {}
            self._combined = 0

    def _range_finished(self, st):
        """After the last fold's refit of a voxel range its block of the mean weights is final: with the weights wanted
        on the host (reserve_host_weights) it leaves NOW, on the download stream, beside the next range's refit."""
        if st["fold"] != self.n_folds - 1:
            return
        if not self.moments:                           # (the moments form accumulates voxel by voxel: lc_primal_refit)
            self._combine_weights(st["rg"])
        if self._host_weights is None:
            return
        if self._host_w is None:
            self._host_w = self._host_weights.result()
        rg = st["rg"]
        final = torch.cuda.Event()
        final.record()
        self.dl.wait_event(final)
        ops.download_cols(rg.W, self._host_w, rg.c0, rg.V, self.dl)
        self._sent += rg.V

    def _publish(self, st, r_s, p_s, perm, Vs, best, info, info_o, n_t):
        """The per-voxel results of one (fold, range) step go into the rank's packed block of the fold, natural voxel
        order (r, p, alpha index, pivot flags).  Once every range of the fold is in, the block is all-gathered over the
        voxel shards, unpacked to V_total-long vectors, and the fold's BH-FDR runs on ALL p-values -- on the
        communication stream, so that neither the collective nor the sort hold up the main stream.  Returns the pending
        host copies of the fold then, None before."""
        fold_no, rg = st["fold"], st["rg"]
        ent = self._fold_blk.get(fold_no)
        if ent is None:
            ent = self._fold_blk[fold_no] = dict(
                blk=torch.empty((4, max(self.w_max, 2)), dtype=torch.float64, device=self.dev), cols=0, keep=[])
        ops.fold_pack(r_s, p_s, perm, Vs, best, rg.V, info, info_o, ent["blk"], col0=rg.c0, clear=ent["cols"] == 0)
        ent["cols"] += rg.V
        ent["keep"] += [r_s, p_s, perm, best, info, info_o]
        if ent["cols"] < self.V_rank:
            return None
        blk = ent["blk"]
        packed = torch.cuda.Event()
        packed.record()
        self.comm.wait_event(packed)
        Vt = self.V_total
        with torch.cuda.stream(self.comm):
            gathered = self.shard.all_gather(blk)                                  # (world, 4, ld)
            dres = torch.empty((2, Vt), dtype=torch.float64, device=self.dev)      # r, p of all voxels
            didx = torch.empty(Vt, dtype=torch.int32, device=self.dev)
            dbad = torch.empty(2, dtype=torch.int32, device=self.dev)
            ops.fold_unpack(gathered, self.shard.world, blk.shape[1], self.d_lo, self.w_max, dres[0], dres[1], didx,
                            self.p_folds[fold_no], dbad)
            # the fold's BH-FDR: a cross-validated fit only takes the rejection MASKS of its folds (their majority vote,
            # nested_cv.py:283-290) -- no sort, no adjusted p-values (lc_bh_reject); a train/test fit returns both
            stat_d = None
            if self.n_folds > 1:
                (rej_d, stat_d), padj_d = ops.bh_reject(self.p_folds[fold_no], self.alpha_fdr, want_status=True), None
            else:
                rej_d, padj_d = ops.bh_fdr(self.p_folds[fold_no], self.alpha_fdr)
            # results leave through pinned buffers so the copies do not stall the host
            h_res = torch.empty((2, Vt), dtype=torch.float64, pin_memory=True)
            h_idx = torch.empty(Vt, dtype=torch.int32, pin_memory=True)
            h_rej = torch.empty(Vt, dtype=torch.uint8, pin_memory=True)
            h_padj = torch.empty(Vt, dtype=torch.float64, pin_memory=True) if padj_d is not None else None
            h_bad = torch.empty(2, dtype=torch.int32, pin_memory=True)
            h_stat = torch.empty(1, dtype=torch.int32, pin_memory=True) if stat_d is not None else None
            for h, d in ((h_res, dres), (h_idx, didx), (h_rej, rej_d), (h_padj, padj_d), (h_bad, dbad), (h_stat, stat_d)):
                if h is not None:
                    h.copy_(d, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        for t in [blk] + ent["keep"]:
            if t is not None:
                t.record_stream(self.comm)
        del self._fold_blk[fold_no]
        self.results_ready = done
        return dict(done=done, res=h_res, idx=h_idx, n_t=n_t, bad=h_bad, rej=h_rej, padj=h_padj, stat=h_stat, fold=fold_no,
                    keep=(dres, didx, dbad, rej_d, padj_d, gathered, stat_d))

    def fold_refit(self, st, single_alpha, weight_scale):
        return self.fold_finish(self.fold_select(st, single_alpha), weight_scale)

    def fold_collect(self, pend) -> _FoldResult:
        """Waits for a fold's results: r / p / alpha index of ALL voxels (every shard), and the fold's BH-FDR.  The
        pivot flags are OR-ed over the ranks, so a failed factorisation raises on every rank together."""
        pend["done"].synchronize()
        if int(pend["bad"][0]):
            raise RuntimeError("Cholesky failed: Gram matrix + alpha^2 I is not positive definite")
        if int(pend["bad"][1]):
            raise RuntimeError("Cholesky failed in the refit: Gram matrix + alpha^2 I is not positive definite")
        res = pend["res"].numpy()
        rej = pend["rej"].numpy().astype(bool)
        if pend.get("stat") is not None and int(pend["stat"][0]):
            # the counting iteration of lc_bh_reject hit its cap (p-values hugging the BH line): the sort-based routine, now
            with torch.cuda.stream(self.comm):
                rej = ops.bh_fdr(self.p_folds[pend["fold"]], self.alpha_fdr)[0].cpu().numpy().astype(bool)
        sig = (rej, None if pend["padj"] is None else pend["padj"].numpy().copy())
        return _FoldResult(res[0].copy(), res[1].copy(), pend["idx"].numpy().copy(), pend["n_t"], sig)

    def combined_significance(self):
        """Fisher's combination of the folds' p-values and its BH-FDR on the device, over the voxels of all shards
        (every rank, redundantly): (p_comb, reject, adjusted p) as host arrays."""
        # on the communication stream, behind the last fold's results: the main stream is still busy with the weight
        # rows of that fold's refit, which nothing here depends on
        return self.combined_significance_end(self.combined_significance_begin())

    def combined_significance_begin(self):
        """Queues Fisher + BH-FDR + the copies to page-locked memory on the communication stream and returns at once."""
        Vt = self.V_total
        with torch.cuda.stream(self.comm):
            pcomb = ops.fisher_combine(self.p_folds[: self.n_folds])
            rej, padj = ops.bh_fdr(pcomb, self.alpha_fdr)
            h_pc = torch.empty(Vt, dtype=torch.float64, pin_memory=True)
            h_rej = torch.empty(Vt, dtype=torch.uint8, pin_memory=True)
            h_padj = torch.empty(Vt, dtype=torch.float64, pin_memory=True)
            for h, d in ((h_pc, pcomb), (h_rej, rej), (h_padj, padj)):
                h.copy_(d, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        return dict(done=done, host=(h_pc, h_rej, h_padj), keep=(pcomb, rej, padj))

    def combined_significance_end(self, pend):
        pend["done"].synchronize()
        h_pc, h_rej, h_padj = pend["host"]
        return h_pc.numpy().copy(), h_rej.numpy().astype(bool), h_padj.numpy().copy()

    def run_fold(self, tr_rows, te_rows, inner_rel, single_alpha, weight_scale) -> _FoldResult:
        st = self.fold_begin(tr_rows, te_rows, inner_rel)
        return self.fold_collect(self.fold_refit(st, single_alpha, weight_scale))

    def weights(self) -> np.ndarray:
        """The (p, V) float32 weights as a host array.  The array lives in page-locked memory (the D2H copy is then one
        DMA at link rate instead of a staged copy through the driver: 0.98 GB at cfg2); it is an ordinary numpy array
        that owns its buffer through torch's caching host allocator.  Voxel ranges whose last fold finished earlier are
        already there or on their way (_range_finished)."""
        if self._host_w is None:
            self._host_w = self._host_weights.result() if self._host_weights is not None else \
                torch.empty((self.p, self.V_rank), dtype=torch.float32, pin_memory=True)
        h = self._host_w
        if self._sent < self.V_rank:                   # nothing left early (weights() without reserve_host_weights)
            done = torch.cuda.Event()
            done.record()
            self.dl.wait_event(done)
            ops.download_cols(self.W_full, h, 0, self.V_rank, self.dl)
            self._sent = self.V_rank
        self.dl.synchronize()
        self._host_weights = self._host_w = None
        return h.numpy()

    def reserve_host_weights(self):
        """Page-lock the result buffer NOW, on a worker thread: when the caller still holds the previous fit's weights
        the caching host allocator has no free block of that size and hipHostMalloc of 0.98 GB takes ~50 ms -- beside
        the fit's GPU work instead of after it."""
        if self._host_weights is None and self.dev.type == "cuda":
            shape = (self.p, self.V_rank)
            self._host_weights = ops.misc_pool().submit(lambda: torch.empty(shape, dtype=torch.float32, pin_memory=True))

    def abandon(self):
        """The fit is given up half-way (an exception in the driver): wait for everything that still writes into host
        memory this engine owns -- weight panels on the download stream, staging threads of the upload."""
        try:
            self.dl.synchronize()
            self.comm.synchronize()
        finally:
            self._host_weights = self._host_w = None
            try:
                self.finish_uploads()
            except Exception:  # noqa: BLE001 -- the original error is the one to report
                pass

    def finish_uploads(self):
        """Host inputs: wait until every panel of the targets is resident (the fit is being abandoned or repeated)."""
        if self.uploader is not None:
            for b in range(len(self.upload_panels)):
                self.uploader.wait(self._y_job0 + b)
            self.uploader.join()
            self.uploader = None


def _alpha_vector(alphas, idx, single_alpha):
    """The array the reference returns for the chosen alphas, dtype quirks included
    (nested_cv.py:401-403: ``torch.tensor([alpha] * V)`` takes torch's default for the element type --
    float64 for numpy scalars, float32 for Python floats; :409-411: explicit float32)."""
    if single_alpha:
        dtype = torch.tensor([alphas[int(idx[0])]]).numpy().dtype
        return np.full(len(idx), alphas[int(idx[0])], dtype=dtype)
    return np.asarray(alphas, dtype=np.float64)[np.asarray(idx, dtype=np.int64)].astype(np.float32)


def _fold_lists(r32: np.ndarray, p: np.ndarray):
    """What ``_calculate_correlations_pvalues`` (nested_cv.py:418-438) returns for one fold:
    list of np.float32 r (NaN -> Python 0.0) and list of float64 p (NaN -> 1.0).  ``p`` comes from the
    device (``lc_pearson_pvalues``; ``stats.pearson_pvalues`` is the same formula on the host)."""
    corrs, pvals = list(r32), list(p)
    for i in np.nonzero(np.isnan(r32))[0]:
        corrs[i] = 0.0
        pvals[i] = 1.0
    return corrs, pvals


class NestedCVModel(BasePredictivityModel):
    """Drop-in for ``encoding.models.nested_cv.NestedCVModel``; same ``fit_predict`` signature,
    defaults, return triple and metrics keys.  ``shard`` (optional) makes the instance fit only its
    rank's block of voxel columns and gather the per-voxel results across ranks."""

    def __init__(self, model_name: str, shard: Optional[ShardContext] = None, precision: str = "auto",
                 form: str = "auto", panel_cols: Optional[int] = None, local_targets: bool = False,
                 options: Optional[FitOptions] = None):
        """``precision``: arithmetic of the V-wide alpha sweep -- "f32" (f32-input MFMA), "f16x3" (fp16
        hi/lo operands, three fp16 MFMAs per product, fp32 accumulate; fp32-level accuracy, ~3x faster) or
        "auto" (f16x3 unless the targets' dynamic range is too wide for it; see RidgeCVEngine._target_scales).
        ``form``: "dual" (n x n systems), "primal" (p x p systems) or "auto" (primal for tall designs, 2 p <= the
        smallest inner training set; see RidgeCVEngine).
        ``panel_cols``: width (a multiple of 256 voxel columns) of the panels a host-to-host fit moves its targets
        and weights in -- the first fold starts on a panel while the others are still crossing PCIe, the last fold's
        weights leave panel by panel (None: FitOptions.panel_cols for fits of >= 2 x panel_min_cols voxels; 0: no panels).  The
        results do not depend on it, bit for bit.
        ``local_targets`` (voxel shards): ``targets`` / ``y_test`` hold only this rank's block of voxel columns (the
        blocks of ShardContext.bounds, in rank order) instead of all of them -- a rank then never touches the other
        ranks' 1.7 GB of host memory."""
        super().__init__(model_name)
        self.local_targets = bool(local_targets)
        self.options = options                         # FitOptions of this model's fits (None: the defaults)
        self.shard = shard
        self.precision = precision
        self.form = form
        self.panel_cols = panel_cols
        self.last_form = None
        self.last_fit = {}                             # RidgeCVEngine.info of the most recent fit (+ "form")
        self.last_fold_alphas = None

    def fit_predict(
        self,
        features: np.ndarray,
        targets: np.ndarray,
        X_test: Optional[np.ndarray] = None,
        y_test: Optional[np.ndarray] = None,
        groups: Optional[np.ndarray] = None,
        folding_type: str = "chunked",
        n_outer_folds: int = 5,
        n_inner_folds: int = 5,
        chunk_length: int = 20,
        alphas: Optional[List[float]] = None,
        alpha_fdr: float = 0.05,
        use_gpu: bool = True,
        single_alpha: bool = False,
        normalpha: bool = True,
        use_corr: bool = True,
        normalize_features: bool = False,
        normalize_targets: bool = False,
        singcutoff: float = 1e-10,
    ) -> Tuple[Dict[str, Union[float, List[float], List[bool]]], np.ndarray, np.ndarray]:
        if alphas is None:
            alphas = np.logspace(-1, 8, 10)
        check_penalties(alphas, singcutoff, normalpha, n_inner_folds)
        if not use_gpu:
            logger.warning("use_gpu=False ignored: this implementation has no CPU path, the fit runs on the MI355X")
        features, targets = np.asarray(features), np.asarray(targets)   # lists / nested lists, like torch.tensor(...)
        if X_test is not None and y_test is not None:
            X_test, y_test = np.asarray(X_test), np.asarray(y_test)
        shard = self.shard or ShardContext.single()
        train_test = X_test is not None and y_test is not None
        V_total = np.shape(targets)[1]
        if self.local_targets and shard.world > 1:
            mine = np.zeros(shard.world)
            mine[shard.rank] = V_total
            V_total = int(round(shard.allreduce_sum(mine).sum()))
        lo, hi = shard.bounds(V_total)

        def cols(y):                       # this rank's voxel block (the whole matrix on one GPU): a view, no copy
            return y if (shard.world == 1 or self.local_targets) else np.asarray(y)[:, lo:hi]

        if train_test:                      # row blocks side by side: no host-side concatenation of 2 GB of targets
            X_all = np.concatenate([np.asarray(features), np.asarray(X_test)], axis=0)
            Y_all = ops.HostRows([cols(targets), cols(y_test)])
        else:
            X_all, Y_all = features, ops.HostRows([cols(targets)])
        return self._run(X_all, Y_all, len(features), len(X_test) if train_test else 0, V_total, groups, folding_type,
                         n_outer_folds, n_inner_folds, chunk_length, alphas, alpha_fdr, single_alpha, normalpha,
                         use_corr, normalize_features, normalize_targets, weights_on_host=True, singcutoff=singcutoff)

    def fit_predict_device(self, features_dev: torch.Tensor, targets_dev: torch.Tensor, n_features: int,
                           n_voxels_local: int, n_voxels_total: Optional[int] = None, n_test_rows: int = 0,
                           weights_on_host: bool = False, **kwargs):
        """Same fit with the inputs already resident in HBM (extension, not in the reference):
        ``features_dev`` (T[+T_test], pad32(p)) and ``targets_dev`` (T[+T_test], pad128(V_local)) are
        zero-padded contiguous fp32 device tensors; ``targets_dev`` holds this rank's voxel block.
        kwargs as ``fit_predict`` (folding / alphas / flags).  With ``weights_on_host=False`` the
        (p, V_local) weights come back as a device tensor."""
        opt = dict(groups=None, folding_type="chunked", n_outer_folds=5, n_inner_folds=5, chunk_length=20, alphas=None,
                   alpha_fdr=0.05, single_alpha=False, normalpha=True, use_corr=True, normalize_features=False,
                   normalize_targets=False, singcutoff=1e-10)
        unknown = set(kwargs) - set(opt) - {"use_gpu"}
        if unknown:
            raise TypeError(f"unexpected keyword arguments: {sorted(unknown)}")
        opt.update({k: v for k, v in kwargs.items() if k in opt})
        if opt["alphas"] is None:
            opt["alphas"] = np.logspace(-1, 8, 10)
        check_penalties(opt["alphas"], opt["singcutoff"], opt["normalpha"], opt["n_inner_folds"])
        T = features_dev.shape[0] - n_test_rows
        # (the targets may also be host row blocks -- ops.HostRows, e.g. the stories of harness.StoryPipeline, z-scored in
        # the upload threads -- beside a resident design: they then arrive panel by panel like fit_predict's)
        shapes = (_DeviceShapes(features_dev, n_features),
                  targets_dev if isinstance(targets_dev, ops.HostRows) else _DeviceShapes(targets_dev, n_voxels_local))
        return self._run(shapes[0], shapes[1], T, n_test_rows, n_voxels_total or n_voxels_local, opt["groups"],
                         opt["folding_type"], opt["n_outer_folds"], opt["n_inner_folds"], opt["chunk_length"],
                         opt["alphas"], opt["alpha_fdr"], opt["single_alpha"], opt["normalpha"], opt["use_corr"],
                         opt["normalize_features"], opt["normalize_targets"], weights_on_host=weights_on_host,
                         singcutoff=opt["singcutoff"])

    def _run(self, *args, **kwargs):
        """The fit on the process's MAIN stream of the device (see _main_stream), ordered after the caller's stream
        at entry and before it at exit."""
        ms = _main_stream() if torch.cuda.is_available() else None
        if ms is None:
            return self._run_on_current_stream(*args, **kwargs)
        caller = torch.cuda.current_stream()
        ms.wait_stream(caller)
        with torch.cuda.stream(ms):
            out = self._run_on_current_stream(*args, **kwargs)
        caller.wait_stream(ms)
        for x in out:
            if isinstance(x, torch.Tensor) and x.is_cuda:
                x.record_stream(caller)
        return out

    def _run_on_current_stream(self, X_all, Y_all, T, n_test_rows, V_total, groups, folding_type, n_outer_folds,
                               n_inner_folds, chunk_length, alphas, alpha_fdr, single_alpha, normalpha, use_corr,
                               normalize_features, normalize_targets, weights_on_host, singcutoff=0.0):
        shard = self.shard or ShardContext.single()
        train_test = n_test_rows > 0
        if train_test:
            # nested_cv.py:130-132 passes ``groups`` positionally into ``trim_size``
            inner = create_folds(T, folding_type, n_inner_folds, chunk_length, groups)
            outer = [(np.arange(T), T + np.arange(n_test_rows), inner)]
        else:
            if groups is not None and folding_type == "group":
                splits = create_folds(T, "group", n_outer_folds, groups=groups)
            else:
                splits = create_folds(T, folding_type, n_outer_folds, chunk_length, groups)
            outer = []
            for tr, te in splits:
                if groups is not None and folding_type == "group":
                    inner = create_folds(len(tr), "group", n_inner_folds, groups=[groups[i] for i in tr])
                else:
                    inner = create_folds(len(tr), folding_type, n_inner_folds, chunk_length)
                outer.append((tr, te, inner))

        min_train = min(len(tr_i) for _, _, inner in outer for tr_i, _ in inner)

        V_rank = Y_all.shape[1]
        panels = down_panels = None
        if not isinstance(Y_all, _DeviceShapes):
            # the same NUMBER of panels on every rank of a sharded fit (narrowest rank decides)
            o = self.options or FitOptions()
            panels = [(0, V_rank)] if self.panel_cols == 0 else _column_panels(
                V_rank, o.panel_cols if self.panel_cols is None else self.panel_cols,
                o.panel_min_cols if self.panel_cols is None else 256, v_ref=V_total // max(shard.world, 1))
            # the end of the fit: fewer, wider panels of geometrically falling width (explicit panel_cols: the same panels
            # at both ends, what the tests of the panel logic ask for)
            if self.panel_cols is None and o.tail_panels_geometric:
                down_panels = _download_panels(V_rank, min_cols=o.panel_min_cols, v_ref=V_total // max(shard.world, 1),
                                               last_frac=o.tail_last_frac)

        def attempt(form, precision, X_in, Y_in):
            eng = RidgeCVEngine(X_in, Y_in, alphas, normalpha, use_corr, normalize_features, normalize_targets, shard,
                                precision=precision, singcutoff=singcutoff, V_total=V_total,
                                min_train_rows=min_train, form=form, panels=panels, options=self.options,
                                down_panels=down_panels)
            self._engine = eng
            drv_opt = getattr(eng, "opt", None) or FitOptions()     # (the tests' oracle-backed engine has none)
            scale = 1.0 if train_test else 1.0 / len(outer)
            fold_scores, fold_p, fold_alpha, fold_sig = [], [], [], []
            score_rows, any_nan = [], []

            def tail(pend):
                """Host statistics of one finished fold; runs while the GPU works on the next fold.  The engine hands
                over the vectors of ALL voxels: the one exchange of per-voxel results over the voxel shards (and the
                fold's BH-FDR on them) happened on the device (RidgeCVEngine.fold_finish)."""
                f = eng.fold_collect(pend)
                r32 = f.r.astype(np.float32)
                if train_test:                  # the per-fold Python lists are only returned by the train/test metrics;
                    own = slice(None) if (shard.world == 1 or shard.global_lists) else slice(*shard.bounds(V_total))
                    corrs, pvals = _fold_lists(r32[own], f.p[own])      # the CV summary works on the arrays below
                    fold_scores.append(corrs)
                    fold_p.append(pvals)
                fold_alpha.append(_alpha_vector(alphas, f.best_idx, single_alpha))
                fold_sig.append(f.sig)
                score_rows.append(np.nan_to_num(r32, nan=0.0))
                any_nan.append(bool(np.isnan(r32).any()))

            pending = None
            if weights_on_host:
                eng.reserve_host_weights()
            eng.alpha_fdr = alpha_fdr
            n = len(outer)
            eng.begin_fit(n)                                    # resident targets: the one host sync of the set-up
            lmax_pre = eng.precompute_lmax(outer)               # one Lanczos run for every train set of the fit
            # the (fold, voxel range) steps in execution order: folds full width, the first / last one panel by panel
            # while the targets arrive from / the weights leave for the host
            # host inputs: the targets need ~30 ms to cross PCIe, and until they are there the chip has little V-wide
            # work -- so EVERYTHING that does not depend on a voxel is queued now and runs in that window, nothing gated:
            # the hat matrices of all folds (aux) and the refit operators of every fold for every factorised alpha (aux2,
            # refit_ahead: explicit inverses, cheap enough to form for alphas nobody will choose); the V-wide phases then
            # find the chip to themselves (fp64 chains beside the MFMA sweeps cost a resident fit ~20 of 137 ms)
            hosted = getattr(eng, "uploader", None) is not None and shard.world == 1
            ahead = shard.world > 1 or (hosted and eng.refit_ahead_pays())
            if (single_alpha and hosted and len(eng.upload_panels) > 1 and hasattr(eng, "fold_choose_joint")):
                # ---- single_alpha with host inputs (example.py:104-117, the LeBel-style train/test call): the choice needs
                # the scores of ALL voxels, but not their sweeps at once -- every fold's sweeps run range by range (the first
                # fold's as the upload panels land, instead of after the last one: ~55 ms of PCIe at cfg3's 2.9 GB), the
                # per-alpha sums of the ranges are added on the device (fold_choose_joint), and each range is then refitted
                # with the one alpha; a range's weights leave for the host as soon as its last fold is in
                self._plan = list(eng.upload_panels)
                first = eng.prepare_folds(outer[:1], lmax_pre[:1])[0]
                if ahead:
                    eng.refit_ahead([first])
                sts = [eng.fold_begin(*outer[0], prepared=first, step=(0, eng.upload_panels[0]))]
                prepared = [first] + (eng.prepare_folds(outer[1:], lmax_pre[1:]) if n > 1 else [])
                if ahead:
                    eng.refit_ahead(prepared[1:])
                elif getattr(eng, "cho", None) and eng.opt.speculate_first_fold and eng.speculation_pays():
                    eng.fold_speculate(first, list(eng.cho), early=True)
                sts += [eng.fold_begin(*outer[0], prepared=first, step=(0, c)) for c in eng.upload_panels[1:]]
                for f in range(n):
                    eng.fold_choose_joint(sts)
                    nxt = None
                    if f + 1 < n:                              # the next fold's sweeps behind this fold's choice
                        cols = eng.download_panels if (f + 1 == n - 1 and weights_on_host) else [(0, eng.V_rank)]
                        nxt = [eng.fold_begin(*outer[f + 1], prepared=prepared[f + 1], step=(f + 1, c)) for c in cols]
                    for i, st in enumerate(sts):
                        st = eng.fold_select(st, True)
                        if i == 0 and f + 1 < n:
                            eng.fold_speculate(prepared[f + 1], st["used_all"])
                        pend = eng.fold_finish(st, scale)
                        if pend is not None:
                            if pending is not None:
                                tail(pending)
                            pending = pend
                    sts = nxt
                tail(pending)
                return eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan
            plan = eng.plan_steps(n, single_alpha, ahead=hosted and ahead)
            self._plan = sorted({c for _, c in plan})
            # V-independent part of every fold, ahead of everything on the auxiliary stream: fold 0 by itself (its sweeps
            # start as soon as its own systems are done), then ALL other folds as one batch
            first = eng.prepare_folds(outer[:1], lmax_pre[:1])[0]
            if shard.world > 1 or hosted:
                # voxel shards: a rank's V-wide work per fold is a few ms, the same order as one latency chain of its
                # share of a fold's systems -- so the folds are prepared one by one (all queued now, nothing gated), each
                # ready when its sweeps come up; and every fold's refit systems for every factorised alpha are solved
                # collectively ahead of the choices (aux2), fold 0's first
                # (one batch over folds 1..n-1 for the hat matrices; fold 0's refit systems by themselves, then the other
                # folds' in one batch: per-fold batches and other mixtures measured the same, 54-57 ms per simulated rank
                # of 8 -- the rank is bound by its total work, not by the batching)
                if ahead:
                    eng.refit_ahead([first])
                early_begin = hosted or (shard.world > 1 and drv_opt.shard_first_sweeps_before_batch)
                if early_begin:  # the first panel is there within a few ms: its sweeps are queued before the big batch is
                    st = eng.fold_begin(*outer[0], prepared=first, step=plan[0])     # (shards: ~3 ms of host enqueue time)
                if shard.world > 1 and n > 2 and drv_opt.second_fold_own_batch:
                    # voxel shards: fold 1's systems as a batch of their own, so that its sweeps start one (short) chain
                    # after fold 0's instead of behind the chain of all the other folds' systems
                    prepared = ([first] + eng.prepare_folds(outer[1:2], lmax_pre[1:2])
                                + eng.prepare_folds(outer[2:], lmax_pre[2:]))
                else:
                    prepared = [first] + (eng.prepare_folds(outer[1:], lmax_pre[1:]) if n > 1 else [])
                # one GPU, host inputs: the other folds' refit operators wait for the FIRST choice (fold 0's first panel, a
                # few ms from now) and are formed for the alphas it used only -- an alpha nobody chooses (the smallest
                # of a grid, typically) costs an N^3 inverse per fold; one that turns up later is solved then
                defer_ahead = ahead and hosted and shard.world == 1 and n > 1 and drv_opt.refit_ahead_after_first_choice
                if ahead and not defer_ahead:
                    eng.refit_ahead(prepared[1:], **({"after_hat": True} if (hosted and drv_opt.refit_ahead_behind_hat_batch)
                                                      else {}))
                if not early_begin:
                    st = eng.fold_begin(*outer[0], prepared=first, step=plan[0])
            else:
                # resident targets: the later folds' refit inverses as ONE batch once fold 0 has chosen (for the alphas it
                # used), instead of one short chain per fold beside that fold's sweeps
                defer_ahead = bool(n > 1 and drv_opt.resident_refit_batch and getattr(eng, "refit_ahead_pays", lambda: False)())
                # one GPU, resident targets: the batch's series operands now, its Cholesky chains once fold 0's sweeps
                # (just queued) are done -- same fit time, and fold 0's fused launches, the dominant kernel, run without 80
                # systems of fp64 work beside them (1.66 -> 1.45 ms per launch over the fit)
                # (fold 1 first / one batch per fold instead of one batch measured the same within the box-to-box spread,
                # 145.8-147.6 ms: the fit is bound by the total work of the streams, not by which batch the main stream
                # waits for; all refit inverses ahead in one batch, as with voxel shards, costs 4 ms here: work for
                # alphas nobody chooses, beside the fused launches)
                if getattr(eng, "cho", None) and eng.opt.speculate_first_fold and eng.speculation_pays():
                    eng.fold_speculate(first, list(eng.cho), early=True)      # aux2: fold 0's refit systems, all of them
                st = eng.fold_begin(*outer[0], prepared=first, step=plan[0])
                prepared = [first] + (eng.prepare_folds(outer[1:], lmax_pre[1:], chol_after=eng.chain_gate())
                                      if n > 1 else [])
            # the weights leave panel by panel during the last fold (0.98 GB at cfg2: ~18 ms of PCIe): its first panel is
            # taken through refit BEFORE the next panel's sweeps are queued (no look-ahead at that step and at the one
            # before it), so that the link starts at the head of the fold and the rest of the fold hides the transfer
            first_last = next((k for k, (f_, _) in enumerate(plan) if f_ == n - 1), None)
            interleaved = any(f_ < n - 1 for f_, _ in plan[first_last:])           # last two folds voxel-major: spread anyway
            early_out = (hosted and weights_on_host and n > 1 and not interleaved
                         and sum(1 for f_, _ in plan if f_ == n - 1) > 1)
            # look-ahead of the sweeps' FIRST part (validation statistics, operand split, series contraction: it needs the
            # series operands only): step k + 2's first part is queued before step k + 1's fused sweeps, so that the main
            # stream has V-wide work while those wait for a fold's Cholesky chains (fold 0's, then the big batch's)
            # (measured: resident 129.4 -> 128.6 ms, host to host 141.3 -> 141.9: kept for resident inputs only)
            ahead_ok = bool(drv_opt.series_lookahead) and not hosted and hasattr(eng, "fold_sweeps_finish")
            begun = {}

            def begin_series(j):
                if ahead_ok and j < len(plan) and j not in begun:
                    fj = plan[j][0]
                    begun[j] = eng.fold_begin(*outer[fj], prepared=prepared[fj], step=plan[j], split_phase=True)

            def begin_step(j):
                """Step j with all its sweeps queued (its first part may be there already)."""
                fj = plan[j][0]
                if not ahead_ok:
                    return eng.fold_begin(*outer[fj], prepared=prepared[fj], step=plan[j])
                begin_series(j)
                begin_series(j + 1)
                return eng.fold_sweeps_finish(begun.pop(j))

            for k, (f, _) in enumerate(plan):
                eng.fold_choose(st, single_alpha)               # main: argmax + grouping; the histogram leaves asynchronously
                look = k + 1 < len(plan) and not (early_out and k in (first_last - 1, first_last))
                st_next = None
                if look:                                        # main: sweeps of the next (fold, range)
                    st_next = begin_step(k + 1)
                st = eng.fold_select(st, single_alpha)          # host waits for the histogram of this step here
                if k == 0 and defer_ahead:
                    eng.refit_ahead(prepared[1:], alphas=st["used_all"])
                if f + 1 < n and (k == 0 or plan[k - 1][0] != f):
                    eng.fold_speculate(prepared[f + 1], st["used_all"])         # aux: refit systems of the next fold
                if pending is not None:
                    tail(pending)
                    pending = None
                pending = eng.fold_finish(st, scale)            # main: V-wide refit of this step behind those sweeps
                if not look and k + 1 < len(plan):
                    st_next = begin_step(k + 1)
                st = st_next
            tail(pending)
            return eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan

        def run_(form):
            try:
                return attempt(form, self.precision, X_all, Y_all)
            except (_WideTargets, _PrimalUnsuitable):
                raise                                   # handled below / by the caller: the engine lives on
            except BaseException:
                # the fit is being abandoned (e.g. "Cholesky failed" from fold_collect in the last folds): finished weight
                # panels may still be crossing PCIe into the page-locked result buffer through the library's own copies,
                # which torch's caching host allocator knows nothing about -- drain them (and the uploads) before the
                # buffer can go back to the allocator
                eng = getattr(self, "_engine", None)
                if eng is not None and hasattr(eng, "abandon"):
                    eng.abandon()
                self._engine = None
                raise

        def run(form):
            try:
                return run_(form)
            except _WideTargets as why:
                # host inputs + precision "auto": a panel that arrived later is too wide for the fp16 split -- once, on
                # the f32 MFMA path, with everything that is resident by now
                eng = self._engine
                logger.info("%s: the fit is repeated on the f32 MFMA path", why)
                eng.finish_uploads()
                torch.cuda.synchronize()
                return attempt(form, "f32", _DeviceShapes(eng.dX, eng.p), _DeviceShapes(eng.dY_full, eng.V_rank))

        try:
            eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan = run(self.form)
        except _PrimalUnsuitable as why:
            if self.form == "primal":
                raise ValueError(f"form='primal' is not usable for these features: {why}") from None
            logger.info("primal form not used (%s): dual form", why)
            prev = self._engine
            prev.finish_uploads()
            eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan = attempt(
                "dual", self.precision, _DeviceShapes(prev.dX, prev.p), _DeviceShapes(prev.dY_full, prev.V_rank))
        self._engine = None
        self.last_form = "primal" if eng.primal else "dual"
        self.last_fit = dict(getattr(eng, "info", {}), form=self.last_form, panels=self._plan)
        # the weights last: their final panels are still crossing PCIe while the host statistics below are computed
        def weights_now():
            return eng.weights() if weights_on_host else eng.W_full[:, : eng.V_rank]
        # diagnostics (not in the reference's return value): the alpha vector of every outer fold, all voxels --
        # the returned best_alphas is their mean (nested_cv.py:293-296)
        self.last_fold_alphas = [np.asarray(a) for a in fold_alpha]

        # voxel shards with local lists: the V-long containers cover this rank's block only (ShardContext.global_lists)
        part = None
        if shard.world > 1 and not shard.global_lists:
            part = slice(*shard.bounds(V_total))
        if train_test:
            sig, padj = fold_sig[0]
            metrics = stats.train_test_metrics(fold_scores[0], fold_p[0], padj, sig, fold_alpha[0], np.sum(sig), part=part,
                                               all_scores=None if part is None else
                                               score_rows[0].astype(np.float64 if any_nan[0] else np.float32))
            return metrics, weights_now(), fold_alpha[0] if part is None else fold_alpha[0][part]

        # np.mean(fold_scores, axis=0) of the reference (nested_cv.py:276): the nested lists hold np.float32
        # scalars, plus Python 0.0 where r was NaN -- numpy then builds a float64 array, else a float32 one
        # (the device part of the combined significance is queued first: it runs while the host forms the means below)
        pend_sig = eng.combined_significance_begin() if hasattr(eng, "combined_significance_begin") else None
        scores = np.mean(np.stack(score_rows).astype(np.float64 if any(any_nan) else np.float32), axis=0)
        majority = np.sum([s for s, _ in fold_sig], axis=0) >= (n_outer_folds // 2 + 1)
        mean_alphas = np.mean(fold_alpha, axis=0)
        pcomb, sig, padj = eng.combined_significance_end(pend_sig) if pend_sig is not None else eng.combined_significance()
        metrics = stats.full_cv_metrics(scores, pcomb, padj, sig, majority, mean_alphas, np.sum(sig), np.sum(majority),
                                        part=part)
        return metrics, weights_now(), mean_alphas if part is None else mean_alphas[part]


class _DeviceShapes:
    """A resident, zero-padded device matrix together with its logical column count."""

    def __init__(self, tensor: torch.Tensor, n_cols: int):
        self.tensor, self.shape = tensor, (tensor.shape[0], int(n_cols))


def fit_nested_cv(features: np.ndarray, targets: np.ndarray, **kwargs: Any):
    """README.md:128,137,212-226: functional entry point, same kwargs as ``fit_predict``."""
    return NestedCVModel("ridge_regression").fit_predict(features=features, targets=targets, **kwargs)
