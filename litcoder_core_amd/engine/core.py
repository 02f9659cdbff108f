"""Device-resident state of one fit: the inputs (resident, or arriving from the host panel by panel), the Gram matrix,
voxel ranges, column scales and the arithmetic decision, S[0]^2 of every training set, the batched fp64 systems dealt out
over voxel-shard ranks.
"""
import dataclasses
import logging
import os
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import ops, series, stats
from .._lib import COL_TILE, K_TILE, LC_MB, LC_NB, LC_SCORE_CORR, LC_SCORE_R2
from ..dist import ShardContext, job_share
from .common import (_touch_streams, SERIES_TERMS, SINGCUTOFF_REL, GROUPS_PER_LAUNCH, MAX_INNER_FOLDS, FitOptions, check_penalties, _PrimalUnsuitable, _WideTargets, _FoldResult, _aux_stream, _Range, _column_panels, _download_panels, _DeviceShapes, logger)


class EngineCore:
    """Set-up and shared helpers of RidgeCVEngine (nested_cv.py assembles the engine from its parts)."""

    def __init__(self, X_all, Y_all, alphas, normalpha, use_corr, normalize_features, normalize_targets,
                 shard: Optional[ShardContext] = None, lanczos_steps: Optional[int] = None, precision: str = "auto",
                 singcutoff: float = 0.0, V_total: Optional[int] = None, min_train_rows: Optional[int] = None,
                 form: str = "dual", panels=None, options: Optional[FitOptions] = None, down_panels=None):
        """``form``: "dual" (n x n Gram / hat matrices: every shape), "primal" (p x p systems, see _prepare_primal) or
        "auto" = primal when the design is tall, 2 p <= ``min_train_rows`` (the smallest inner training set) and
        p <= FitOptions.primal_max_p.  ``Y_all``: a host array / ops.HostRows (uploaded in the column ``panels`` [(c0, c1), ...] on a
        background thread while the fit is being set up) or resident targets (_DeviceShapes)."""
        self.opt = dataclasses.replace(options) if options is not None else FitOptions()    # this engine's own copy
        if os.environ.get("LITCODER_AMD_FIT_OPTS"):
            # debugging / fuzzing aid: FitOptions fields over whatever the caller passed, e.g.
            # LITCODER_AMD_FIT_OPTS="mean_operator_min_cols=0,mean_operator_cost_ratio=1e9" (tools/fuzz_*.py on small problems)
            for item in os.environ["LITCODER_AMD_FIT_OPTS"].split(","):
                k, v = item.split("=")
                setattr(self.opt, k.strip(), type(getattr(self.opt, k.strip()))(float(v)))
        self._chol_opt = ops.chol_options(self.opt.chol_outer_block, self.opt.chol_big_kernel, self.opt.chol_fused_steps,
                                          False, self.opt.chol_persistent)
        self.spectral = check_penalties(alphas, singcutoff, normalpha)
        self.singcutoff = float(singcutoff)
        self.dev = ops.device()
        self.shard = shard or ShardContext.single()
        if not isinstance(X_all, _DeviceShapes):
            X_all = np.asarray(X_all)
        if not isinstance(Y_all, (_DeviceShapes, ops.HostRows, ops.TargetsInFlight)):
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
        self._y_job0 = 0
        in_flight = isinstance(Y_all, ops.TargetsInFlight)
        if isinstance(Y_all, _DeviceShapes):
            self.dY_full = self._resident(Y_all, self.Vp_rank)
        elif in_flight:
            # the caller put the targets on the link before this engine existed (ops.TargetsInFlight): its buffer, its
            # panels, its uploader -- the design is then resident already (no job of this engine's own)
            if jobs or Y_all.Vp != self.Vp_rank:
                raise ValueError("targets in flight go with a resident design and this engine's column padding")
            self.dY_full = Y_all.buffer
            self.upload_panels = list(Y_all.panels)
            self.download_panels = ([(int(a), int(b)) for a, b in down_panels] if down_panels else list(self.upload_panels))
            self._y_job0 = Y_all.n_lead
            self.uploader = Y_all.uploader
            panels = self.upload_panels
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
        # the screening pass (FitOptions.screen_inner) takes the trailing factorised alphas whose series residual is far below
        # its own arithmetic error from the series terms as well (screen_series_tol): scr_drop of them, in the order of cho
        self.scr_drop = 0
        if self.ser is not None and self.normalpha and not self.primal and self.opt.screen_series_tol > 0:
            while (self.scr_drop < len(self.cho)
                   and series.residual_bound(self.alphas[self.cho[len(self.cho) - 1 - self.scr_drop]], SERIES_TERMS)
                   <= self.opt.screen_series_tol):
                self.scr_drop += 1
        self.d_ser_scr = self.d_coef_scr = None
        if self.scr_drop:
            ser_scr = list(self.ser) + list(self.cho[len(self.cho) - self.scr_drop:])
            self.d_ser_scr = ops.upload(np.asarray(ser_scr, dtype=np.int32), self.dev)
            self.d_coef_scr = ops.upload(np.stack([series.minimax_inverse_coefficients(self.alphas[a], SERIES_TERMS)
                                                   for a in ser_scr]).astype(np.float64), self.dev)
        if os.environ.get("LITCODER_AMD_STREAM_ORDER"):
            _touch_streams(self.dev, [t.strip() for t in os.environ["LITCODER_AMD_STREAM_ORDER"].split(",") if t.strip()])
        self.aux = _aux_stream(self.dev)
        self.aux2 = _aux_stream(self.dev, 1)            # refit systems (see _refit_stream)
        # per-fold result exchange + global statistics (FitOptions.results_on_refit_stream: an experiment, see there)
        self.comm = (_aux_stream(self.dev, 2) if (self.shard.world > 1 or not self.opt.results_on_refit_stream)
                     else self.aux2)
        self.aux3 = _aux_stream(self.dev, 3)            # voxel shards: what a fold's refit still needs after refit_ahead
        self.dl = _aux_stream(self.dev, 4)              # finished weight panels on their way to the host
        self.scales_stream = _aux_stream(self.dev, 5)   # column scales of target panels as they arrive (_target_scales)
        # the f32 side path of too-wide target columns (_side_sweeps_begin): a few dozen workgroups per launch, beside the
        # sweeps' thousands (a HIGH-priority stream for it was measured: one spiked column then costs a cfg2 fit 12.6 %
        # instead of 5.1 % -- as with the fp64 chains, priorities only move the waiting around)
        self.side_stream = _aux_stream(self.dev, 6)
        # the screening pass' undecided voxels + the alpha choice behind them (_after_screening): beside the main stream's next
        # sweeps, on a stream of their own (on the refit systems' stream the choice waits behind the refit's fp64 chains:
        # level on cfg2, 8-12 % slower at cfg4's and cfg5's shapes)
        self.refine_stream = self.aux2 if self.opt.refine_on_refit_stream else _aux_stream(self.dev, 7)
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
        self._mo = None                                # the mean-operator refit's state (engine/mean_refit.py): decided at the first refit
        self._assume_split = None                      # the arithmetic the operators are prepared for (_split_assumed)
        self._decided = False                          # ... decided from ALL resident target columns (begin_fit)
        # constants of the fit that every stream reads: made here, before ``ready``
        self._d_one = ops.upload(np.ones(1, dtype=np.float64), self.dev)
        self._eye, self._eye_key = None, None
        self._eig_cache, self._n_real = {}, {}         # spectral route: eigenpairs of a fold's outer block; list lengths
        self._scale_checks = []                        # primal form: pending looks at the features' column norms
        self.side = None                               # the f32 side panel of too-wide target columns (_register_side)
        self.argmax_only = False                       # the driver's word that score tables only feed per-voxel argmaxes (_sweeps)
        self.mean_only = False                         # ... or only the argmax of their voxel mean (single_alpha)
        # what this fit ran, for the caller (NestedCVModel.last_fit; bench.py prices the roofline with it): arithmetic
        # of the sweeps, alphas scored inside the fused launch, algorithmic flops of the plain fp16x3 GEMMs.  Per
        # engine: two fits in one process do not share it.
        self.info = {"precision": None, "fused_alphas": self.A, "series_terms": 0, "plain_flops": 0.0,
                     "plain_launches": 0, "used_all": None, "fused_flops": 0.0, "fused_launches": 0, "series_flops": 0.0,
                     "series_launches": 0}
        if self.uploader is not None:
            if self._x_job is not None:
                self.uploader.wait(self._x_job)        # the design is needed now (Gram matrix)
            if not in_flight and len(jobs) == (1 if self._x_job is not None else 0):
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
            main.wait_event(ev)
            if self._assume_split:
                # the operators are prepared for the split already and a wide panel means a restart on the f32 path
                # whenever it is noticed: the host does not wait for the panel here -- it would wait for the panel's copies
                # AND for whatever shares the side stream's hardware queue (four queues for all streams of a process:
                # tools/stream_queue_probe.py), with the main stream idle until it had queued the sweeps afterwards.  The
                # flag is looked at when the range's scores are (fold_select -> _verify_target_flag).
                rg.flag_check = (ev, flag_h)
                wide = False
            else:
                ev.synchronize()
                wide = bool(int(flag_h[0]))
        else:
            side_ok = check and self._side_panel_ok(Y, rg)
            colflags = torch.empty(rg.Vp, dtype=torch.uint8, device=self.dev) if side_ok else None
            cs, flag = ops.col_scales_f16(Y, self.Ttot, rg.Vp, colflags=colflags)
            wide = False
            if check:
                self.shard.all_reduce_(flag, "max")
                wide = bool(int(flag.cpu()[0]))
            if wide and side_ok:
                # WHICH columns: a handful of them leaves the fp16 arithmetic alone (an exact-f32 side path recomputes their
                # scores, weights and test correlations, _side_sweeps_begin / _side_refit_begin); the ranks agree on that together
                cols = np.nonzero(colflags.cpu().numpy()[: rg.V])[0]
                many = ops.upload(np.asarray([int(len(cols) > self.opt.side_panel_max_cols)], dtype=np.int32), self.dev)
                self.shard.all_reduce_(many, "max")
                if not int(many.cpu()[0]):
                    self._register_side(rg, cols)
                    wide = False
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

    def _side_panel_ok(self, Y, rg):
        """The f32 side path covers the resident, un-normalised targets of a fit in the dual (n x n) form, decided once for
        all columns of the rank (begin_fit)."""
        return (Y is rg.Y and rg is self.full and not self.norm_y and not self.primal and not self.spectral
                and self.opt.side_panel_max_cols > 0)

    def _register_side(self, rg, cols):
        """The target columns ``cols`` (rank-local, sorted) whose dynamic range the fp16 split cannot carry: an f32 copy of
        them side by side (all rows, zero-padded to whole 128-column tiles), the index lists the scatters back need."""
        n = int(len(cols))
        self.info["side_panel_cols"] = n
        if n == 0:
            self.side = None
            return
        Vs = ops.pad_to(n, COL_TILE)
        idx = np.full(Vs, -1, dtype=np.int32)
        idx[:n] = cols
        d_cols = ops.upload(idx, self.dev)
        Ys = torch.empty((self.Ttot, Vs), dtype=torch.float32, device=self.dev)
        ops.gather(self.dY_full, self.dY_full.stride(0), None, self.Ttot, d_cols, Vs, Ys)
        self.side = dict(cols=np.asarray(cols, dtype=np.int64), n=n, Vp=Vs, Y=Ys, d_cols=d_cols)
        logger.info("%d target column(s) too wide for the fp16x3 sweep: recomputed on the f32 side path", n)

    def _verify_target_flag(self, rg=None):
        """The deferred look at a range's dynamic-range flag (_target_scales); ``rg`` None: every range of the fit."""
        for r in ([rg] if rg is not None else list(self._ranges.values())):
            if r.flag_check is None:
                continue
            ev, flag_h = r.flag_check
            r.flag_check = None
            ev.synchronize()
            if int(flag_h[0]):
                raise _WideTargets("target dynamic range too wide for the fp16x3 sweep")

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
                                  use_mfma=self.opt.lanczos_mfma, tol=self.opt.lanczos_tol)
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
