"""What every part of the fit engine shares: constants, the per-fit options, the penalty-grid check, voxel ranges and
the panel plans of a host-to-host fit, auxiliary streams, small containers."""
import dataclasses
import logging
import os
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import ops, series, stats
from .._lib import COL_TILE, K_TILE, LC_MB, LC_NB, LC_SCORE_CORR, LC_SCORE_R2

logger = logging.getLogger("litcoder_core_amd.nested_cv")


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
    lanczos_tol: float = 0.0                # > 0: a system stops before that once its top Ritz value has moved by <= this
                                            # (relative) over 8 steps.  OFF by default: a Lanczos run whose start vector is
                                            # short of the top eigenvector sits on the SECOND eigenvalue for a while before
                                            # it finds the first, and a stop on the Ritz value's movement (or on its
                                            # residual) takes that plateau for convergence -- tools/fuzz_vs_oracle.py, seed
                                            # 1234 / case 4: S[0] 60.10 instead of 60.27 for one training set at 1e-6, the
                                            # weights 9e-4 off (profiles/r05_lanczos_stop_misfire.txt); 1.3 ms of cfg3's 136
    lanczos_dense: bool = True              # primal form: the streaming matvec for the p x p Gram blocks (lc_lambda_max_dense)
    aug_budget_bytes: int = 24 << 30        # cap on the batched (fold, alpha) fp64 systems resident at once
    series_tol: float = 2e-9                # an alpha takes the polynomial form when its worst relative error over the
                                            # spectrum, 1 / T_d(1 + 2 alpha^2), is <= this: 30x below the fp32 epsilon
    primal_max_scale_ratio: float = 64.0    # primal V-wide route: feature column norms within this factor (fp16x3)
    single_alpha_guess: bool = True         # train/test fits with ONE alpha and host inputs / weights: once all voxel panels
                                            # but the last have been swept, refit them with the alpha they choose and send
                                            # their weights home while the last panel is swept (0.98 GB at the LeBel shape:
                                            # 17 ms of PCIe that followed the last sweep); the choice over ALL voxels is then
                                            # checked against it -- another alpha: the fit is repeated without the guess
    single_alpha_guess_margin: float = 2e-5  # ... only when the early panels' best alpha leads the second best by this
                                            # much in mean score per voxel and inner fold (the LeBel-shaped bench data:
                                            # 4.2e-5; the last panel, 15 % of the voxels, would have to average six times
                                            # that the other way.  Until round 4 the value read 1e-4 and was compared with
                                            # the lead SUMMED over the five inner folds: the same gate, in its real unit)
    side_panel_max_cols: int = 512          # precision "auto": up to this many target columns whose dynamic range the fp16
                                            # split cannot carry (a spike > 512 x the typical entry) are recomputed on an
                                            # exact-f32 side path and written over the main path's results -- only more of them
                                            # (or a form the side path does not cover) move the WHOLE fit to f32 (round 5)
    refit_fused_pearson: bool = True        # test predictions reduced to Pearson r in the contraction's epilogue (fp16x3
                                            # path): never stored, lc_pearson_cols never reads them back (SURVEY K8 + K9)
    refit_by_inverse: bool = True           # refit operators through the explicit inverse + one fp16x3 product
    refit_inverse_min_alpha: float = 0.1    # ... for alphas (in units of S[0]) from here on, decided alpha by alpha (0.05 until
                                            # round 4: a fuzz case at alpha = 0.066 came out 5e-5 of max|W| off a float64
                                            # solve; the error is ~2.5e-6 / alpha: 2.5e-5 at 0.1003, 1.7e-7 at 0.68, round 5.
                                            # Raising it was tried: with many more operator rows than training samples --
                                            # 7680 features -- the solves cost 7x the inverse, and the cfg5 shape, whose voxels
                                            # do choose alpha = 0.1, went from 207 to 221 ms)
    refit_ahead_min_alpha: float = 0.15     # ... and AHEAD of the alpha choice (first fold's speculation, the batch over all
                                            # folds of host inputs / voxel shards) only from here on: smaller alphas are
                                            # rarely chosen, and an operator nobody uses is wasted fp64 work (cfg2's grid
                                            # starts at 0.1, which no voxel of the bench takes: 140.0 -> 137.6 ms host to host)
    refit_inverse_max_world: int = 4        # ... and up to this many voxel-shard ranks
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
    mean_operator_refit: bool = True        # (round 6) the mean weights of a cross-validated fit from the MEAN of the folds'
                                            # refit operators, one contraction of depth T per group of voxels with the same alpha
                                            # in every fold, instead of one of depth n_train per fold (engine/mean_refit.py)
    mean_operator_min_cols: int = 16384     # ... for fits of at least this many voxels on this rank (small fits and narrow voxel
                                            # shards keep the folds' own products: an operator image per alpha tuple costs what
                                            # ~3 column tiles do, and a rank of 10 000 voxels has no tuple that pays)
    mean_operator_max_tuples: int = 256     # ... and while a voxel range has at most this many distinct alpha tuples
    mean_operator_min_share: float = 0.25   # ... and, judged from the first two folds' choices, at least this share of the voxels is
                                            # expected in tuples that pay (weak-signal data: every fold another alpha -- the folds'
                                            # own products then stay where they were, beside the next fold's sweeps; deferring them
                                            # to the range's end cost 2-7 % of such a fit).  The estimate is pessimistic -- measured
                                            # expected / actual share and fit with / without: cfg5's shape 0.76 / 0.83, 154 / 166 ms;
                                            # half the voxels pure noise 0.32 / 0.55, 133.3 / 135.1; a tenth of the signal 0.13 /
                                            # 0.06, 178.7 / 174.0 (0.8 until the round's last day: cfg5's shape lost 7 % to it)
    mean_operator_cost_ratio: float = 0.85  # ... and the grouped contraction + the operator images it needs are estimated at
                                            # no more than this share of the folds' own products (tests force the path: 1e9)
    panel_cols: int = 36864                 # voxel columns per panel of a host-to-host fit (_column_panels): 12 288 / 24 576 /
                                            # 30 720 / 12 416 at cfg2 (measured 144.1 ms against 145.2 for 24 576-wide panels,
                                            # 145.2 for 73 728, 151.6 without panels)
    panel_min_cols: int = 16384             # below twice this many voxels a fit is not cut into panels
    tail_panels_geometric: bool = True      # the end of a host-to-host fit in few panels of falling width (_download_panels)
    tail_last_frac: float = 0.3             # ... the last panel's share of the voxels (0: the geometric plan's own last panel):
                                            # its weights + transfer (~7 ms) run while the host builds the metrics dictionary
                                            # (measured 140.0 -> 138.1 / 137.8 ms at 0.27 / 0.35)
    tail_folds: int = 2                     # ... spread over this many folds, voxel-major (plan_steps)
    series_lookahead: bool = True           # the next step's first sweep part queued before a step's fused sweeps (driver)
    resident_refit_batch: bool = True       # resident inputs: refit inverses of folds 1.. as one batch after fold 0's choice
    prepare_batch_folds: int = 0            # outer folds 1.. are prepared (Lanczos slices, series chains, Cholesky chains) in
                                            # batches of this many folds (0: all of them as ONE batch -- a chain of N / 64
                                            # dependent steps costs the same for 20 systems as for 80; a fold's sweeps can only
                                            # start when ITS batch is complete, though)
    folds_in_one_launch_tiles: int = 512    # all inner folds of a step in ONE launch per pass while a fold's launch has fewer
                                            # 256 x 256 tiles than this (narrow voxel ranges: partial rounds of workgroups; measured -2.8 %
                                            # at 10 000 voxels, -0.5 % at 20 000 on one GPU, +1.7 % for a rank of 4 at 20 000); 0: never
    screen_inner: bool = True               # (round 6) the inner CV in two precisions: a SCREENING pass with one fp16 MFMA per
                                            # product (hi planes: 11-bit operands) decides the alpha of every voxel whose two
                                            # best alphas lie further apart than the screening error can bridge; the other
                                            # voxels (~1 % at cfg2) are scored again with the three-MFMA products
    screen_tau: float = 5e-3                # ... a voxel is undecided when the gap between its two best fold-MEAN scores is
                                            # below screen_tau / sqrt(validation rows scored over all inner folds) (x rms / std
                                            # of the column): the screening error of a fold-mean score is that of ~2e-4-relative
                                            # prediction errors averaged over those rows -- measured rms 3.7e-6, max 4.1e-5 over
                                            # 8e6 scores at cfg2 (2400 rows: gap 1.0e-4 = 27 rms; the largest gap of a voxel
                                            # whose screening argmax was wrong: 1.4e-5; profiles/r06_screen_probe_cfg2.txt)
    series_chain_f16x3: bool = True         # the V-independent chain Q_j = (K[tr,tr] / lambda) Q_(j-1) of the shared series terms on
                                            # the fp16x3 MFMA kernel (three fp16 MFMAs per product, 22-bit operands: as accurate
                                            # as the f32-input MFMA, ~4x its rate) instead of k_gemm_f32
    finalize_folds_at_once: bool = True     # the partial moments of an outer fold's inner folds turned into scores in ONE pass per
                                            # sweep kind instead of one small launch per inner fold (50 -> 10 launches per fit)
    refine_on_side_stream: bool = True      # ... the undecided voxels' panel (a few column tiles: a quarter of the chip for ~1 ms
                                            # per step) and the alpha choice behind it on a stream of their own, beside the NEXT
                                            # step's full-width sweeps instead of in front of them (one GPU, no side panel)
    refine_on_refit_stream: bool = False    # ... that stream = the refit systems' (aux2) instead of one of its own: level on cfg2,
                                            # 8-12 % slower on cfg4 / cfg5 (the choice waits behind heavier refit chains)
    results_on_refit_stream: bool = False   # one GPU: a fold's result unpack / BH-FDR / host copies on aux2 instead of the
                                            # communication stream (experiments with the runtime's hardware queues:
                                            # profiles/experiments/README.md, round 6; slower at cfg4's and cfg5's shapes)
    screen_series_tol: float = 1e-5         # ... and the LARGEST factorised alphas whose 4-term series is accurate to this (relative:
                                            # 1 / T_4(1 + 2 alpha^2); 2.5e-6 at alpha = 2.64, far below the screening arithmetic's
                                            # own ~2e-4) are screened from the shared series terms, not from their hat matrices:
                                            # the fused screening launch carries fewer alphas (3 of cfg2's 4; 0: never).  The
                                            # undecided voxels' three-MFMA scores use the hat matrices of all of them, as before
    screen_two_workgroups: bool = False     # ... True: the screening sweeps in 4-wave workgroups on 256 x 128 tiles, two per CU
                                            # (k_sweep_hi2: one's prologue / epilogue / barrier under the other's MFMAs; VERDICT
                                            # r5 #1a).  Built, bit-identical scores, MEASURED SLOWER and not adopted: 0.574 ms
                                            # against 0.533 per full-width score launch, matrix pipe busy 0.54 against 0.58 at the
                                            # same 1.83 GHz, 1.94 GB leaving L2 against 1.31 (half tiles fetch the operator
                                            # slabs 1.5 x as often): profiles/experiments/r06_hi2_kernel_forms_pmc.txt; a cfg2
                                            # fit 103.4 against 100.9 ms interleaved
    screen_max_undecided: float = 0.35      # ... once a step reports a larger share of undecided voxels the rest of the fit is
                                            # scored on three MFMAs throughout (flat score curves -- pure-noise voxels on the
                                            # plateau of the large alphas, where neighbouring alphas agree to fp32 rounding --
                                            # cannot be decided by screening; scoring most voxels twice costs more than it saves)
    screen_panel_first: float = 0.5         # ... share of a range the refinement's panel can hold while no step of the fit has
                                            # reported its undecided voxels (from then on: screen_panel_margin).  Generous,
                                            # since every pass over the panel reads the number of voxels it holds on the device:
                                            # 1/16 until then, and two steps of a weak-signal fit overflowed before the host knew
                                            # better (~8 ms each)
    screen_panel_margin: float = 1.25       # ... from the first report on: the largest share seen x this + 1024 columns (0: the first
                                            # rule throughout)
    screen_second_panel_max: float = 0.6    # ... a panel that cannot hold a step's undecided voxels is followed by ONE that can (the
                                            # count is known by then) while they are at most this share of the range; beyond it
                                            # the range is scored again on three MFMAs (what every overflow cost before: ~15 ms
                                            # per step at cfg2's shape with a fifth of the voxels undecided)
    screen_mean_coherent: float = 0.02      # ... single_alpha: the ONE alpha is the argmax of the voxel mean of the scores; the
                                            # screening error of that mean is taken as screen_tau / sqrt(rows) x max(0.4 sqrt(sum
                                            # kappa^2), this x sum kappa) / V -- independent errors, or this share of them coherent
                                            # across voxels -- and a smaller lead of the best alpha repeats the fit on three MFMAs
    screen_panel_cols: int = 0              # ... columns of the refinement's panel (0: adaptive, _refine_capacity; tests force
                                            # the overflow path with a small value)
    alpha_progress_log: bool = dataclasses.field(       # per-alpha progress lines (ridge_regression.py:136-139): a device
        default_factory=lambda: os.environ.get("LITCODER_AMD_ALPHA_LOG", "0") == "1")   # round trip per fold, opt-in
    chol_outer_block: int = 512             # lc_batch_chol_solve: columns per outer block of the two-level blocking
    chol_big_kernel: int = 2                # ... deep updates: 2 = 4x4x4 fp64 MFMA, 1 = vector ALU, 0 = 16x16x4 MFMA
    chol_fused_steps: bool = True           # ... fused left-looking 64-column steps
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


class _FoldResult:
    __slots__ = ("r", "p", "best_idx", "n_test", "sig")

    def __init__(self, r, p, best_idx, n_test, sig=None):
        self.r, self.p, self.best_idx, self.n_test = r, p, best_idx, n_test
        self.sig = sig             # (reject mask, adjusted p) of the fold when the device made them (one GPU), else None


_AUX_STREAMS: Dict[Any, Any] = {}


def _aux_stream(dev, which=0, priority=0):
    """The auxiliary streams of a device, created once for the life of the process.  A fresh ``torch.cuda.Stream()`` per
    engine walks through torch's pool of 32 streams, and the FIRST cross-stream wait on a stream that has never run
    anything blocks the host for ~6 ms (its hardware queue is created there): every fit of a series paid that before
    its first fold was queued."""
    which = _STREAM_ALIAS.get(which, which)
    if which == "m":
        return torch.cuda.default_stream(dev)
    key = (dev.type, dev.index, which)
    if key not in _AUX_STREAMS:
        _AUX_STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return _AUX_STREAMS[key]


def _parse_stream_alias(text):
    """LITCODER_AMD_STREAM_ALIAS="7=1,2=m": auxiliary stream 7 is stream 1, stream 2 the default stream (experiments with
    which of the engine's streams share a queue: tools/stream_alias_ab.sh)."""
    out = {}
    for item in filter(None, (t.strip() for t in (text or "").split(","))):
        a, b = item.split("=")
        out[int(a)] = "m" if b.strip() == "m" else int(b)
    return out


_STREAM_ALIAS = _parse_stream_alias(os.environ.get("LITCODER_AMD_STREAM_ALIAS"))
_STREAMS_TOUCHED = set()


def _touch_streams(dev, order):
    """Give the engine's streams their hardware queues in a FIXED order: the HIP runtime creates a stream's queue at its first
    use and deals the process's four hardware queues out in that order, so which of the engine's streams share a queue --
    and a resident cfg2 fit's 81 or 88 ms -- depended on whether the process's first fit had host or resident inputs
    (tools/resident_after_host_ab.py).  ``order``: auxiliary stream numbers, "u" = the upload stream; once per device."""
    key = (dev.type, dev.index)
    if key in _STREAMS_TOUCHED or dev.type != "cuda":
        return
    _STREAMS_TOUCHED.add(key)
    for item in order:
        s = ops.upload_stream(dev) if item == "u" else _aux_stream(dev, int(item))
        ev = torch.cuda.Event()
        with torch.cuda.stream(s):
            ops.zeros((64,), torch.float32, dev)           # (a first command: the stream's queue exists from here on)
            ev.record()
        ev.synchronize()


class _WideTargets(Exception):
    """precision="auto" met a target column whose dynamic range the fp16 hi/lo split cannot carry AFTER the fit was set up
    for it (host inputs arrive panel by panel, so the decision cannot be taken up front): the driver repeats the fit on
    the f32 MFMA path with the targets that are resident by then."""


class _GuessMissed(Exception):
    """single_alpha with host inputs: the early voxel panels were refitted with the alpha THEY chose while the last panel was
    still being swept, and the choice over all voxels turned out to be another one -- the driver repeats the fit with
    what is resident (same result as without the guess; FitOptions.single_alpha_guess)."""


class _ScreenMissed(Exception):
    """single_alpha under the screening pass (FitOptions.screen_inner): the voxel MEANS of the screening scores of the two best
    alphas lie closer than the screening error can vouch for -- the driver repeats the fit with every score on three MFMAs
    (what is resident stays resident)."""


class _Range:
    """A contiguous range [c0, c0 + V) of this rank's voxel columns: the unit a V-wide phase of a fold works on.  The
    targets and the mean weights of the rank live in ONE (T, Vp) / (p, Vp) buffer each; a range sees column views of
    them (row stride = the buffer's), so a fold can be processed full width or panel by panel -- panels while the
    targets are still arriving from the host (first fold) and while the finished weights leave for it (last fold).
    Interior boundaries are multiples of 256 columns (the widest column tile), so only the last range carries padding."""
    __slots__ = ("c0", "V", "Vp", "Y", "W", "scales", "natural", "key", "flag_check")

    def __init__(self, c0, V, Vp, Y, W):
        self.c0, self.V, self.Vp, self.Y, self.W = int(c0), int(V), int(Vp), Y, W
        self.scales = None             # (cs, split) of the un-normalised targets of the range (_target_scales)
        self.natural = None            # 0 .. V-1 on the device (moments form)
        self.flag_check = None         # (event, pinned flag) of the range's dynamic-range check, not looked at yet
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


class _DeviceShapes:
    """A resident, zero-padded device matrix together with its logical column count."""

    def __init__(self, tensor: torch.Tensor, n_cols: int):
        self.tensor, self.shape = tensor, (tensor.shape[0], int(n_cols))
