"""One outer fold in phases -- prepare (auxiliary stream: fp64, V-independent) -> begin (main stream: inner-CV sweeps) ->
choose / select (one host sync on the alpha histogram) -> finish (V-wide refit, test scores) -> collect -- plus the
per-fold result exchange, the mean weights and their way to the host.
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
from .common import (_ScreenMissed, SERIES_TERMS, SINGCUTOFF_REL, GROUPS_PER_LAUNCH, MAX_INNER_FOLDS, FitOptions, check_penalties, _PrimalUnsuitable, _WideTargets, _FoldResult, _aux_stream, _Range, _column_panels, _download_panels, _DeviceShapes, logger)


class FoldPhases:
    """The phase interface NestedCVModel's driver loop calls (tests/test_dist_gloo.py has an oracle-backed stand-in)."""

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
            if img.get("Ht_s") is not None:            # (the screening pass' image of the fused sweep: fewer alphas)
                hs = img["hp_s"]
                sub["imgs"][0].update(hp_s=hs, Ht_s=img["Ht_s"][s * hs * N * 2:(s + Fo) * hs * N * 2],
                                      rs_hs=img["rs_hs"][s * hs:(s + Fo) * hs])
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
                data_ready = None
                if self.primal or X is not self.dX:        # (a design of the fold's own was just made on this stream)
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
                          + [im.get(k) for im in (hat.get("imgs") or []) if im is not None
                             for k in ("Pt", "rs_p", "Ht", "rs_h", "Ht_s", "rs_hs")]):
                    if t is not None and t.is_cuda:
                        t.record_stream(main)              # allocated on aux, consumed on main
        return out

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
        hat.update(cs=cs, split=split, n_te=len(base["te"]))
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

    def fold_choose(self, st, single_alpha):
        """Alpha choice of the fold and the grouping of the voxels by it, enqueued behind the fold's sweeps; the
        histogram travels to pinned memory asynchronously, so the caller can queue the next fold's sweeps on the
        main stream BEFORE waiting for it in fold_select (the stream then never idles through the host round trip)."""
        self._enter(st)
        rs = st["hat"].get("refine_stream")
        if rs is not None and not st.get("_on_refine_stream"):
            # the step's score table is being completed on the refinement's stream (DualSweeps._after_screening): the choice
            # and the grouping follow it there; fold_finish makes the main stream wait for them
            st["_on_refine_stream"] = True
            try:
                with torch.cuda.stream(rs):
                    self.fold_choose(st, single_alpha)
                    st["choose_done"] = torch.cuda.Event()
                    st["choose_done"].record()
            finally:
                st.pop("_on_refine_stream", None)
            return st
        if getattr(self, "debug_scores", None) is not None:      # (tools/screen_probe.py: the score tables of a fit)
            self.debug_scores.append((st["fold"], st["rg"].c0, st["scores"].clone()))
        st["best"] = self.choose(st["scores"], single_alpha, check=st["hat"].get("mean_check") if single_alpha else None)
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

    def fold_choose_joint(self, sts, assign=None):
        """``single_alpha`` when a fold is worked through in several voxel ranges (host inputs arriving panel by panel):
        the ONE alpha is the argmax of the across-voxel mean of the scores (nested_cv.py:396-400), so the per-alpha sums
        of all ranges -- and of all voxel shards -- are added up on the device before any range is grouped.  Every
        range's state (``assign``: only these) gets its ``best`` vector and its grouping, as fold_choose would give it.
        Returns the (A,) device vector of the sums."""
        total = None
        for st in sts:
            self._enter(st)
            _, rowsum = ops.select_alpha(st["scores"], self.A, self.Vp, want_best=False, want_rowsum=True)
            total = rowsum if total is None else ops.accumulate_f64(rowsum, total)
        # (screening pass: the ranges' kappa sums ride along, the all-reduced result leaves for fold_select's check)
        checks = [st["hat"]["mean_check"] for st in sts if st["hat"].get("mean_check")]
        total = self._mean_sums(total, checks if len(checks) == len(sts) else [])
        for st in (sts if assign is None else assign):
            self._enter(st)
            best = torch.empty(self.Vp, dtype=torch.int32, device=self.dev)
            st["best"] = ops.fill_argmax(total, self.A, best, self.Vp)
            if not self.moments:
                st["grouping"] = self._group_async(st["best"], st["split"])
        return total

    def fold_select(self, st, single_alpha):
        """Waits for the fold's alpha histogram (fold_choose; the one host synchronisation of a fold) and puts the
        fp64 systems of the refit on the auxiliary stream -- they run beside whatever the main stream does next."""
        self._enter(st)
        self._mean_check(st)
        if self.moments:                               # nothing to factor after the choice, and no host sync
            if "best" not in st:
                self.fold_choose(st, single_alpha)
            st.update(used=[], used_all=[])
            return st
        if "grouping" not in st:
            self.fold_choose(st, single_alpha)
        self._screen_check(st, single_alpha)
        best, split = st["best"], st["split"]
        grouping = st.pop("grouping")
        perm, used, tiles, Vs, used_all = self._refit_groups(best, split, grouping)
        st["best_h"] = grouping[3].numpy() if (len(grouping) > 3 and grouping[3] is not None) else None
        self._verify_target_flag(st["rg"])             # (its event lies before the sweeps whose histogram just arrived)
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

    def _mean_check(self, st):
        """single_alpha under the screening pass: the ONE alpha is the argmax of the voxel mean of the scores
        (nested_cv.py:396-400).  The mean of V screening scores is good to ~2e-4 / sqrt(rows) x sqrt(sum kappa^2) / V when the
        voxels' errors are independent (tools/screen_probe.py: rms 1.9e-4 per voxel in those units); the lead of the best
        alpha over the runner-up must exceed  screen_tau / sqrt(rows) x max(0.4 sqrt(sum kappa^2), screen_mean_coherent x
        sum kappa)  -- ten such sigmas, or 2 % of the per-voxel bound coherent across all voxels -- else _ScreenMissed: the
        driver repeats the fit on three MFMAs.  Voxel shards: every rank decides from the same all-reduced numbers."""
        chk = st["hat"].get("mean_check")
        if not chk or "host" not in chk or chk.get("done"):
            return
        chk["ev"].synchronize()
        chk["done"] = True
        h = chk["host"].numpy()
        A = self.A
        sums, k1, k2 = h[:A], float(h[A]), float(h[A + 1])
        self.info["screened"] = self.info.get("screened", 0) + int(st["rg"].V)
        if A < 2:
            return
        order = np.argsort(-sums, kind="stable")
        lead = float(sums[order[0]] - sums[order[1]])
        thr = (self.opt.screen_tau * chk["F"] / np.sqrt(max(chk["rows"], 1))
               * max(0.4 * np.sqrt(max(k2, 0.0)), self.opt.screen_mean_coherent * k1))
        self.info["screen_mean_lead_over_threshold"] = lead / thr if thr > 0 else float("inf")
        if not np.isfinite(sums).all() or not (lead >= thr):
            raise _ScreenMissed(f"the best alpha's score sum leads by {lead:.3g}, the screening pass vouches for {thr:.3g}")

    def _screen_check(self, st, single_alpha):
        """Two-precision inner CV: how many voxels the step's screening pass left undecided (known here, at the step's one
        host synchronisation; DualSweeps._refine_undecided).  More than the refinement's panel held -- on this rank or, voxel
        shards, on any -- and the whole range is scored again with the three-MFMA products and chosen again (every rank
        alike: the choice's histogram is all-reduced); otherwise the share is remembered for the next panels' capacity."""
        chk = st["hat"].pop("screen_check", None)
        if chk is None:
            return
        ev, host, V, cap, sharded = chk
        ev.synchronize()
        found = int(host[0])                           # (voxel shards: the largest count of any rank)
        over = int(host[1]) > 0
        if self.shard.simulate:
            # one rank of a W-rank job run alone for timing (tools/scaling_model.py): its peers' operators are copies of its
            # own, its scores mean nothing -- and neither does the share of voxels they leave undecided.  The rank is timed
            # with a panel of the usual size (_refine_capacity) and no second scoring: what a real rank's step costs
            self.info["screened"] = self.info.get("screened", 0) + V
            return
        self.info["undecided"] = self.info.get("undecided", 0) + min(found, cap)
        self.info["screened"] = self.info.get("screened", 0) + V
        if not hasattr(self, "_undecided_fracs"):
            self._undecided_fracs = []
        self._undecided_fracs.append(min(found, V) / max(V, 1))
        if self._undecided_fracs[-1] > self.opt.screen_max_undecided and self.opt.screen_panel_cols == 0:
            # most voxels scored twice: no gain -- the steps queued from here on take the three-MFMA sweeps throughout (this
            # engine's own copy of the options; voxel shards: from all-reduced counts, every rank alike)
            self.opt.screen_inner = False
            self.info["screen_switched_off"] = True
        if not over:
            return
        self.info["screen_overflows"] = self.info.get("screen_overflows", 0) + 1
        args = st["hat"].get("refine_args")
        if (args is not None and not st.get("_refined_again") and found <= self.opt.screen_second_panel_max * V
                and self.opt.screen_panel_cols == 0):
            # (round 6, later) a second panel that holds them all instead of three MFMAs for the whole range: the table still
            # holds the screening scores (the first panel's columns their three-MFMA scores: they are found undecided again or
            # not, and scored to the same bits either way); the count is exact, the margin covers the columns that change sides
            logger.info("screening pass: %d of %d voxels undecided, the panel held %d: a second panel", found, V, cap)
            st["_refined_again"] = True
            self.info["undecided"] -= min(found, cap)      # (the second look at this step counts it again)
            self.info["screened"] -= V
            self._undecided_fracs.pop()
            self._enter(st)
            rs = st["hat"].get("refine_stream")
            hat_ = st["hat"]
            with torch.cuda.stream(rs if rs is not None else torch.cuda.current_stream()):
                self._refine_undecided(hat_, st["Y"], st["scores"], *args, cap=int(1.1 * found) + 512)
            st.pop("grouping", None)
            st.pop("choose_done", None)
            self.fold_choose(st, single_alpha)
            return self._screen_check(st, single_alpha)
        logger.info("screening pass: %d of %d voxels undecided, the panel held %d: the range is scored again", found, V, cap)
        hat = dict(st["hat"])
        hat["exact"] = True
        hat.pop("refine_stream", None)
        st["hat"].pop("refine_stream", None)           # (the new table is made, and chosen from, on the current stream)
        st.pop("choose_done", None)
        self._enter(st)
        st["scores"] = self._sweeps(hat, st["Y"], st["done"])
        st.pop("grouping", None)
        self.fold_choose(st, single_alpha)

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
        if st.get("choose_done") is not None:          # the choice was made on the refinement's stream (fold_choose)
            torch.cuda.current_stream().wait_event(st["choose_done"])
            for t in (best, perm):
                if isinstance(t, torch.Tensor):
                    t.record_stream(torch.cuda.current_stream())
        torch.cuda.current_stream().wait_event(st["systems_ready"])
        row0 = self.p_pad                              # first row of the test-row hat matrix inside M_alpha
        # (too-wide target columns: their refit in exact f32, on the side stream, beside the main path's below)
        side_job = (self._side_refit_begin(st, row0, n_t) if (self.side is not None and st["split"] and not self.primal) else None)
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
        if o["split"] and self.opt.refit_fused_pearson:
            r_s = self._refit_pearson(o, row0, n_t)        # the predictions stay in the contraction's registers
        else:
            pred = self._refit_product(o, row0, st["Malpha"][0].shape[0], n_t)[:n_t]
            if o.get("te_src") is not None:
                r_s = ops.pearson_cols_gather(*o["te_src"], pred, n_t, Vs)
            else:
                r_s = ops.pearson_cols(o["Ys_te"], pred, n_t, Vs)
        p_s = ops.pearson_pvalues(r_s, Vs, n_t)
        # the weight rows stay in alpha-sorted order where the contraction writes them (one matrix per fold, plus where each
        # voxel's column went); the mean over the folds is taken in one pass per voxel range once its last fold is in
        # (_combine_weights), not accumulated fold by fold
        if not self.primal and self._mo_enabled(st):
            # (round 6) no weight rows per fold: the range's mean weights come from the MEAN of the folds' operators once its
            # last fold has chosen (engine/mean_refit.py) -- one contraction of depth T instead of one of depth n_train per fold
            pend = self._publish(st, r_s, p_s, perm, Vs, best, st["info"], st["info_o"], n_t, side_job=side_job)
            self._mo_record(st, weight_scale, side_job, o=o, perm=perm, Vs=Vs)     # (may decide against the option: _mo_decide)
            self._range_finished(st)
            return pend
        ent, off = self._ws_slot(st["fold"], rg, Vs, weight_scale)
        ops.invert_perm(perm, Vs, off, ent["pos"][rg.c0:])
        # (the side columns' r / p, exact f32, go over the main path's on the communication stream, just before the fold's
        # results are exchanged: the main stream does not wait for the side refit here)
        pend = self._publish(st, r_s, p_s, perm, Vs, best, st["info"], st["info_o"], n_t, side_job=side_job)
        # the weights last: nothing the host waits for depends on them (for the last fold the host statistics then
        # run beside this part of the contraction)
        self._refit_product(o, 0, self.p_pad, self.p, out=ent["buf"][:, off:off + Vs])
        if side_job is not None:
            W_s, dst, n_s = self._side_refit_end(side_job, ent, off)
            ops.scatter_cols(W_s, self.p_pad, dst, n_s, ent["buf"][:, off:off + Vs])
        self._range_finished(st)
        return pend

    def _side_refit_begin(self, st, row0, n_t):
        """The refit of the side panel's columns of this (fold, range) step in exact f32 arithmetic (ridge_torch + the test
        predictions + Pearson r, ridge_regression.py:9-63, nested_cv.py:151-155), queued on the side stream: one column tile
        per alpha in use -- a side voxel sits in the tile of the alpha IT chose (from its corrected scores), the other
        columns of the tiles are zeros, so that nothing has to come to the host --, the grouped f32-input MFMA product with
        the same f32 operators the main path's fp16 images were split from, Pearson r / p of the test rows."""
        hit = self._side_cols_of(st["rg"])
        if hit is None:
            return None
        s0, ns, local = hit
        used, Malpha = list(st["used"]), st["Malpha"]
        G, Vg = len(used), ops.pad_to(ns, COL_TILE)
        main, ss = torch.cuda.current_stream(), self.side_stream
        d_local = ops.upload(local, self.dev)                  # (on the current stream, before the event the side stream waits for)
        d_used = ops.upload(np.asarray(used, dtype=np.int32), self.dev)
        few = ns <= ops.GEMV_MAX_COLS                          # a handful of columns: streamed products (see _side_sweeps_begin)
        if few:
            N_o = Malpha[0].shape[1]
            d_tr = ops.idx_tensor(st["tr"], N_o, self.dev)
            d_te = ops.idx_tensor(st["te"], n_t, self.dev)
        start = torch.cuda.Event()
        start.record()
        ss.wait_event(start)
        with torch.cuda.stream(ss):
            best_s = torch.full((Vg,), -2, dtype=torch.int32, device=self.dev)
            best_s[:ns] = st["best"][d_local.long()]
            if few:
                # column j of C is side column j, refitted with the operator of the alpha IT chose: one streamed product per
                # alpha in use, each taking the columns that chose it (decided on the device: nothing comes to the host)
                rows_all = Malpha[0].shape[0]
                Ysel = self.side["Y"][:, s0:]
                C = ops.zeros((rows_all, Vg), torch.float32, self.dev)
                for g, a in enumerate(used):
                    ops.gemv_cols(Malpha[g], rows_all, N_o, Ysel, d_tr, ns, C, sel=best_s, want=int(a))
                Ys_te = ops.zeros((n_t, Vg), torch.float32, self.dev)
                ops.gather(Ysel, Ysel.stride(0), d_te, n_t, None, min(Vg, Ysel.shape[1]), Ys_te)
                r = ops.pearson_cols(Ys_te, C[row0:row0 + n_t], n_t, Vg)
                pv = ops.pearson_pvalues(r, Vg, n_t)
                perm_s = torch.arange(Vg, dtype=torch.int32, device=self.dev)
                perm_s[ns:] = -1
                done = torch.cuda.Event()
                done.record()
                for t in (C, r, pv, perm_s):
                    t.record_stream(main)
                for t in (d_local, d_used, d_tr, d_te):
                    t.record_stream(ss)
                return dict(C=C, r=r, p=pv, perm=perm_s, Vss=Vg, d_local=d_local, done=done, rg=st["rg"])
            j = torch.arange(Vg, dtype=torch.int32, device=self.dev)
            perm_s = torch.where(best_s[None, :] == d_used[:, None], j[None, :], torch.full_like(j, -1)[None, :])
            perm_s = perm_s.reshape(-1).contiguous()                       # (G Vg): voxel j in its alpha's tile, else -1
            tiles = [g * (Vg // COL_TILE) for g in range(G + 1)]
            Vss = G * Vg
            Ysel = self.side["Y"][:, s0:]
            o = self._refit_operands(Ysel, st["tr"], st["te"], perm_s, tiles, Vss, Malpha, False, None)
            C = self._refit_product(o, 0, Malpha[0].shape[0], self.p + n_t)
            r = ops.pearson_cols(o["Ys_te"], C[row0:row0 + n_t], n_t, Vss)
            pv = ops.pearson_pvalues(r, Vss, n_t)
            done = torch.cuda.Event()
            done.record()
        for t in (C, r, pv, perm_s):
            t.record_stream(main)
        d_local.record_stream(ss)
        d_used.record_stream(ss)
        return dict(C=C, r=r, p=pv, perm=perm_s, Vss=Vss, d_local=d_local, done=done, rg=st["rg"])

    def _side_refit_end(self, job, ent, off):
        """Main stream, behind the main path's weight product of the step: (weights, destination columns, count) for the
        scatter of the side refit's weights over the main path's alpha-sorted weight columns."""
        torch.cuda.current_stream().wait_event(job["done"])
        rg, perm_s, Vss = job["rg"], job["perm"], job["Vss"]
        # where each side column belongs in the main path's alpha-sorted order of this step (lc_invert_perm has run)
        pos_rel = ent["pos"][rg.c0:][job["d_local"].long()] - off
        dst = torch.where(perm_s >= 0, pos_rel[perm_s.clamp(min=0).long()], torch.full_like(perm_s, -1)).to(torch.int32).contiguous()
        return job["C"][: self.p_pad], dst, Vss

    def _side_results_into(self, blk, jobs):
        """Current (communication) stream: the side refits' Pearson r / p of a fold over the main path's entries of the
        packed result block (natural voxel order: rows 0 / 1, lc_fold_pack_at)."""
        cur = torch.cuda.current_stream()
        for job in jobs:
            cur.wait_event(job["done"])
            rg, perm_s, Vss = job["rg"], job["perm"], job["Vss"]
            nat = job["d_local"] + rg.c0
            dst = torch.where(perm_s >= 0, nat[perm_s.clamp(min=0).long()], torch.full_like(perm_s, -1)).to(torch.int32).contiguous()
            ops.scatter_cols(job["r"].view(1, -1), 1, dst, Vss, blk[0:1])
            ops.scatter_cols(job["p"].view(1, -1), 1, dst, Vss, blk[1:2])
            for t in (job["r"], job["p"], perm_s, job["d_local"]):
                t.record_stream(cur)

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
        if getattr(self, "_mo", None):
            self._mean_operator_weights(st["rg"])      # (the folds left their weight rows to this point)
        elif not self.moments:                         # (the moments form accumulates voxel by voxel: lc_primal_refit)
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

    def _publish(self, st, r_s, p_s, perm, Vs, best, info, info_o, n_t, side_job=None):
        """The per-voxel results of one (fold, range) step go into the rank's packed block of the fold, natural voxel
        order (r, p, alpha index, pivot flags).  Once every range of the fold is in, the block is all-gathered over the
        voxel shards, unpacked to V_total-long vectors, and the fold's BH-FDR runs on ALL p-values -- on the
        communication stream, so that neither the collective nor the sort hold up the main stream.  Returns the pending
        host copies of the fold then, None before."""
        fold_no, rg = st["fold"], st["rg"]
        ent = self._fold_blk.get(fold_no)
        if ent is None:
            ent = self._fold_blk[fold_no] = dict(
                blk=torch.empty((4, max(self.w_max, 2)), dtype=torch.float64, device=self.dev), cols=0, keep=[], side=[])
        ops.fold_pack(r_s, p_s, perm, Vs, best, rg.V, info, info_o, ent["blk"], col0=rg.c0, clear=ent["cols"] == 0)
        ent["cols"] += rg.V
        ent["keep"] += [r_s, p_s, perm, best, info, info_o]
        if side_job is not None:
            ent["side"].append(side_job)
        if ent["cols"] < self.V_rank:
            return None
        blk = ent["blk"]
        packed = torch.cuda.Event()
        packed.record()
        self.comm.wait_event(packed)
        Vt = self.V_total
        with torch.cuda.stream(self.comm):
            if ent["side"]:
                self._side_results_into(blk, ent["side"])
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
