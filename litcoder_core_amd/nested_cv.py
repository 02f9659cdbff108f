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

This module holds the reference-facing model and the driver loop; the engine's parts live in ``engine/`` (round 4: one
file of 2 700 lines before): ``common`` (options, ranges, panel plans), ``core`` (set-up, scales, Lanczos, sharded
systems), ``dual`` / ``primal`` (inner CV in either form), ``refit``, ``folds`` (the phase interface).
"""
import dataclasses
import logging
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops, series, stats  # noqa: F401
from ._lib import COL_TILE, K_TILE, LC_MB, LC_NB, LC_SCORE_CORR, LC_SCORE_R2  # noqa: F401  (names tests / tools reach through here)
from .dist import ShardContext
from .engine.common import (_ScreenMissed, SERIES_TERMS, SINGCUTOFF_REL, GROUPS_PER_LAUNCH, MAX_INNER_FOLDS, FitOptions,  # noqa: F401
                            check_penalties, _PrimalUnsuitable, _WideTargets, _GuessMissed, _FoldResult, _aux_stream, _Range,
                            _column_panels, _download_panels, _DeviceShapes)
from .engine.core import EngineCore
from .engine.dual import DualSweeps
from .engine.folds import FoldPhases
from .engine.mean_refit import MeanOperatorRefit
from .engine.primal import PrimalForm
from .engine.refit import Refit
from .folding import create_folds

logger = logging.getLogger(__name__)


class BasePredictivityModel:
    """``encoding/models/base.py:7-41``: the interface AbstractTrainer calls."""

    def __init__(self, model_name: str):
        self.model_name = model_name

    def fit_predict(self, features, targets, groups=None, **kwargs):  # pragma: no cover - interface
        raise NotImplementedError


class RidgeCVEngine(EngineCore, DualSweeps, PrimalForm, Refit, MeanOperatorRefit, FoldPhases):
    """Device-resident state of one fit: fp32 copies of X / Y (zero padded), the Gram matrix, and the
    per-fold pipeline.  ``Y`` holds only this rank's voxel block."""


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
        self.debug_scores = None                       # a list: receives (fold, first column, score table) per step

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
            V_total = shard.total_of_local_blocks(V_total)       # (checked on every rank alike: the blocks are bounds()'s)
        lo, hi = shard.bounds(V_total)

        def cols(y):                       # this rank's voxel block (the whole matrix on one GPU): a view, no copy
            return y if (shard.world == 1 or self.local_targets) else np.asarray(y)[:, lo:hi]

        if train_test:                      # row blocks side by side: no host-side concatenation of 2 GB of targets
            X_all = np.concatenate([np.asarray(features), np.asarray(X_test)], axis=0)
            Y_all = ops.HostRows([cols(targets), cols(y_test)])
        else:
            X_all, Y_all = features, ops.HostRows([cols(targets)])
        return self._run_on_current_stream(X_all, Y_all, len(features), len(X_test) if train_test else 0, V_total, groups, folding_type,
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
        # the upload threads -- beside a resident design: they then arrive panel by panel like fit_predict's; or such
        # blocks already on the link, start_targets)
        shapes = (_DeviceShapes(features_dev, n_features),
                  targets_dev if isinstance(targets_dev, (ops.HostRows, ops.TargetsInFlight))
                  else _DeviceShapes(targets_dev, n_voxels_local))
        return self._run_on_current_stream(shapes[0], shapes[1], T, n_test_rows, n_voxels_total or n_voxels_local, opt["groups"],
                         opt["folding_type"], opt["n_outer_folds"], opt["n_inner_folds"], opt["chunk_length"],
                         opt["alphas"], opt["alpha_fdr"], opt["single_alpha"], opt["normalpha"], opt["use_corr"],
                         opt["normalize_features"], opt["normalize_targets"], weights_on_host=weights_on_host,
                         singcutoff=opt["singcutoff"])

    def _panel_plan(self, V_rank, V_total):
        """(upload panels, download panels) of host targets of ``V_rank`` columns: the column ranges they cross the link
        in at the start of a fit, and the ranges the end of the fit works in when the weights go back to the host."""
        shard = self.shard or ShardContext.single()
        o = self.options or FitOptions()
        # the same NUMBER of panels on every rank of a sharded fit (narrowest rank decides)
        panels = [(0, V_rank)] if self.panel_cols == 0 else _column_panels(
            V_rank, o.panel_cols if self.panel_cols is None else self.panel_cols,
            o.panel_min_cols if self.panel_cols is None else 256, v_ref=V_total // max(shard.world, 1))
        # the end of the fit: fewer, wider panels of geometrically falling width (explicit panel_cols: the same panels
        # at both ends, what the tests of the panel logic ask for)
        down_panels = None
        if self.panel_cols is None and o.tail_panels_geometric:
            down_panels = _download_panels(V_rank, min_cols=o.panel_min_cols, v_ref=V_total // max(shard.world, 1),
                                           last_frac=o.tail_last_frac)
        return panels, down_panels

    def start_targets(self, targets, n_voxels_total: Optional[int] = None, lead=()):
        """Puts host targets (a host matrix or ops.HostRows of this rank's voxel block) on the link NOW, in the panels a
        fit of this model will work through, and returns the ops.TargetsInFlight that ``fit_predict_device`` takes in
        place of the targets: for callers that still have the design to build (harness.StoryPipeline).  ``lead``: upload
        jobs that go first through the same staging ring."""
        targets = targets if isinstance(targets, ops.HostRows) else ops.HostRows([np.asarray(targets)])
        V_rank = targets.shape[1]
        panels, _ = self._panel_plan(V_rank, n_voxels_total or V_rank)
        return ops.TargetsInFlight(targets, ops.device(), panels, lead=lead)

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
            panels, down_panels = self._panel_plan(V_rank, V_total)
            if isinstance(Y_all, ops.TargetsInFlight):
                panels = list(Y_all.panels)             # (already crossing the link in these)

        options_now = [None]                            # (a repeated fit's own options: _ScreenMissed)

        def prepare_rest(eng, drv_opt, chol_after=None):
            """The V-independent state of the outer folds 1.. : one batch, or batches of FitOptions.prepare_batch_folds folds."""
            k = int(getattr(drv_opt, "prepare_batch_folds", 0) or 0)
            if len(outer) <= 1:
                return []
            if k <= 0:
                return eng.prepare_folds(outer[1:], lmax_pre_all[0][1:], chol_after=chol_after) if chol_after is not None \
                    else eng.prepare_folds(outer[1:], lmax_pre_all[0][1:])
            out = []
            for i in range(1, len(outer), k):
                kw = dict(chol_after=chol_after) if (chol_after is not None and i == 1) else {}
                out += eng.prepare_folds(outer[i:i + k], lmax_pre_all[0][i:i + k], **kw)
            return out

        lmax_pre_all = [None]

        def attempt(form, precision, X_in, Y_in):
            eng = RidgeCVEngine(X_in, Y_in, alphas, normalpha, use_corr, normalize_features, normalize_targets, shard,
                                precision=precision, singcutoff=singcutoff, V_total=V_total,
                                min_train_rows=min_train, form=form, panels=panels,
                                options=options_now[0] if options_now[0] is not None else self.options, down_panels=down_panels)
            self._engine = eng
            # the inner CV's score tables are consumed by a per-voxel argmax and nothing else (nested_cv.py:405-411), or --
            # ONE alpha for all voxels -- by the argmax of their voxel mean (:396-400): the engine may screen them
            # (FitOptions.screen_inner)
            eng.argmax_only = not bool(single_alpha)
            eng.mean_only = bool(single_alpha)              # ... its voxel mean is, with a check of the winner's lead (_mean_check)
            if self.debug_scores is not None:
                eng.debug_scores = self.debug_scores
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
            lmax_pre_all[0] = lmax_pre
            # the (fold, voxel range) steps in execution order: folds full width, the first / last one panel by panel
            # while the targets arrive from / the weights leave for the host
            # host inputs: the targets need ~30 ms to cross PCIe, and until they are there the chip has little V-wide
            # work -- so EVERYTHING that does not depend on a voxel is queued now and runs in that window, nothing gated:
            # the hat matrices of all folds (aux) and the refit operators of every fold for every factorised alpha (aux2,
            # refit_ahead: explicit inverses, cheap enough to form for alphas nobody will choose); the V-wide phases then
            # find the chip to themselves (fp64 chains beside the MFMA sweeps cost a resident fit ~20 of 137 ms)
            arriving = getattr(eng, "uploader", None) is not None        # host targets, crossing the link panel by panel
            hosted = arriving and shard.world == 1
            ahead = shard.world > 1 or (hosted and eng.refit_ahead_pays())
            # (voxel shards, round 5: the panelled single-alpha path below runs on every rank alike -- each rank's block is cut
            # into the SAME number of panels (_panel_plan: v_ref), the per-alpha sums are all-reduced inside fold_choose_joint,
            # and everything the host decides from them it decides from the all-reduced values)
            if (single_alpha and arriving and len(eng.upload_panels) > 1 and hasattr(eng, "fold_choose_joint")):
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
                guessed = None
                if (n == 1 and weights_on_host and len(eng.upload_panels) >= 3 and drv_opt.single_alpha_guess
                        and not getattr(eng, "moments", False)):
                    # ---- the tail of such a fit was: last sweep -> choice -> refit of every range -> 0.98 GB of weights over
                    # the link (17 ms at the LeBel shape), all behind the last panel's sweeps.  Once every panel BUT the last
                    # has been swept their alpha is almost always THE alpha: refit them with it now, let their weights leave
                    # while the last panel is swept, and check the choice over all voxels afterwards (_GuessMissed: once more,
                    # without the guess)
                    sts += [eng.fold_begin(*outer[0], prepared=first, step=(0, c)) for c in eng.upload_panels[1:-1]]
                    early_sums = eng.fold_choose_joint(sts).cpu().numpy()        # (host: these panels' sweeps are done)
                    order = np.argsort(-early_sums, kind="stable")
                    # the sums run over the voxels of the early panels of ALL ranks (every rank's panels but the last have
                    # the plan's widths: _column_panels, v_ref) and over the inner folds that were scored (the scores of a
                    # step are accumulated over them) -- the lead is quoted per voxel AND per inner fold
                    v_early = sum(c1 - c0 for c0, c1 in eng.upload_panels[:-1]) * max(1, shard.world)
                    n_scored = max(1, int(first["hat"].get("F", 1)))
                    lead = (early_sums[order[0]] - early_sums[order[1]]) if len(order) > 1 else np.inf
                    eng.info["single_alpha_lead"] = float(lead / (max(v_early, 1) * n_scored))   # mean score, best - second
                    if np.isfinite(early_sums).all() and lead >= drv_opt.single_alpha_guess_margin * v_early * n_scored:
                        guessed = int(order[0])
                        for st in sts:
                            st = eng.fold_select(st, True)
                            eng.fold_finish(st, scale)                           # (not the fold's last range: nothing pending)
                        last = eng.fold_begin(*outer[0], prepared=first, step=(0, eng.upload_panels[-1]))
                        sums = eng.fold_choose_joint(sts + [last], assign=[last]).cpu().numpy()
                        if not np.isfinite(sums).all():
                            # (the host's argsort and the device's first-maximum rule part ways on a NaN: no shortcut then)
                            raise _GuessMissed("non-finite score sums over all voxels")
                        if int(np.argsort(-sums, kind="stable")[0]) != guessed:
                            raise _GuessMissed(f"alpha {alphas[guessed]:g} of the first {len(sts)} voxel panels, "
                                               f"{alphas[int(np.argsort(-sums, kind='stable')[0])]:g} over all voxels")
                        last = eng.fold_select(last, True)
                        pending = eng.fold_finish(last, scale)
                        eng.info["single_alpha_guess"] = "held"
                        tail(pending)
                        return eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan
                    eng.info["single_alpha_guess"] = "not decisive"
                    sts.append(eng.fold_begin(*outer[0], prepared=first, step=(0, eng.upload_panels[-1])))
                else:
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
                early_begin = hosted
                if early_begin:  # the first panel is there within a few ms: its sweeps are queued before the big batch is
                    st = eng.fold_begin(*outer[0], prepared=first, step=plan[0])
                # (voxel shards: fold 0's sweeps queued before the batch, or fold 1's systems as a batch of their own, were
                # measured and gave nothing -- profiles/experiments/README.md)
                prepared = [first] + prepare_rest(eng, drv_opt)
                # (one GPU, host inputs: forming the later folds' inverses only for the alphas the first panel chose was
                # measured 2.4 ms slower than forming all of them in the upload window -- profiles/experiments/README.md)
                defer_ahead = False
                if ahead:
                    eng.refit_ahead(prepared[1:])
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
                prepared = [first] + prepare_rest(eng, drv_opt, chol_after=eng.chain_gate())
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

        def run_(form, precision=None, X_in=None, Y_in=None):
            try:
                return attempt(form, self.precision if precision is None else precision, X_all if X_in is None else X_in,
                               Y_all if Y_in is None else Y_in)
            except (_WideTargets, _PrimalUnsuitable, _GuessMissed, _ScreenMissed):
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

        def run(form, X_in=None, Y_in=None):
            """The fit, repeated on what is resident when a shortcut of the first attempt did not hold -- as a LOOP, so that
            a repeated fit that meets the other condition is handled like a first one (an exception raised inside
            one ``except`` clause is not caught by its sibling, and a missed single-alpha guess whose repeat met a wide
            target column escaped with an internal exception)."""
            precision, missed, wide_tries, rescored = None, False, 0, False
            while True:
                try:
                    out = run_(form, precision, X_in, Y_in)
                    if missed:
                        out[0].info["single_alpha_guess"] = "missed"
                    if rescored:
                        out[0].info["screen_mean_repeated"] = True
                    return out
                except _GuessMissed as why:
                    # single_alpha, host inputs: the early panels' alpha was not the alpha of all voxels -- once more,
                    # resident (no panels: no guess the second time)
                    eng = self._engine
                    logger.info("single-alpha guess missed (%s): the fit is repeated without it", why)
                    eng.abandon()                       # the early panels' weights may still be on their way to the host
                    torch.cuda.synchronize()
                    missed = True
                    precision = self.precision if precision is None else precision
                except _ScreenMissed as why:
                    # single_alpha: the screening pass cannot vouch for the winner of the voxel-mean scores -- once more on
                    # three MFMAs throughout, with what is resident
                    eng = self._engine
                    logger.info("screening pass not decisive (%s): the fit is repeated on three MFMAs", why)
                    eng.abandon()                       # (early panels' weights may be on their way to the host)
                    torch.cuda.synchronize()
                    options_now[0] = dataclasses.replace(self.options or FitOptions(), screen_inner=False)
                    rescored = True
                    precision = self.precision if precision is None else precision
                except _WideTargets as why:
                    # host inputs + precision "auto": a panel that arrived later holds a column too wide for the fp16 split
                    # -- once more with everything that is resident by now: the fit then knows ALL its columns up front and
                    # moves only the wide ones to the f32 side path (round 5; the whole fit to the f32 MFMA path when there
                    # are too many of them or the form has no side path -- decided in begin_fit).  Targets normalised fold by
                    # fold (normalize_targets) are looked at fold by fold as well: no decision up front, no side panel --
                    # their repeat takes the f32 path as a whole; and should a repeated "auto" fit meet a wide column after
                    # all, the f32 path is what is left (it never raises this)
                    eng = self._engine
                    wide_tries += 1
                    if wide_tries > 2:
                        raise RuntimeError(f"the f32 path reported a wide target column ({why})") from None
                    logger.info("%s: the fit is repeated %s", why,
                                "with the targets resident" if wide_tries == 1 else "on the f32 path")
                    eng.finish_uploads()
                    torch.cuda.synchronize()
                    precision = "f32" if (normalize_targets or wide_tries == 2) else self.precision
                X_in, Y_in = _DeviceShapes(eng.dX, eng.p), _DeviceShapes(eng.dY_full, eng.V_rank)

        try:
            eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan = run(self.form)
        except _PrimalUnsuitable as why:
            if self.form == "primal":
                raise ValueError(f"form='primal' is not usable for these features: {why}") from None
            logger.info("primal form not used (%s): dual form", why)
            prev = self._engine
            prev.finish_uploads()
            eng, fold_scores, fold_p, fold_alpha, fold_sig, score_rows, any_nan = run(
                "dual", _DeviceShapes(prev.dX, prev.p), _DeviceShapes(prev.dY_full, prev.V_rank))
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
            # (the scalar summaries from the ARRAY the list was made of -- np.asarray of a list of 80 000 np.float32 scalars is
            # a millisecond, twice, on the tail of the fit; same dtype rule: float64 when a Python 0.0 stands for a NaN)
            metrics = stats.train_test_metrics(fold_scores[0], fold_p[0], padj, sig, fold_alpha[0], np.sum(sig), part=part,
                                               all_scores=score_rows[0].astype(np.float64 if any_nan[0] else np.float32))
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


def fit_nested_cv(features: np.ndarray, targets: np.ndarray, **kwargs: Any):
    """README.md:128,137,212-226: functional entry point, same kwargs as ``fit_predict``."""
    return NestedCVModel("ridge_regression").fit_predict(features=features, targets=targets, **kwargs)
