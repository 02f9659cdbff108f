"""The primal (p x p) form for tall designs: block products for a handful of features (lc_primal.hip), the
V-wide route with shared series terms and sums over validation blocks for hundreds to thousands of features (round 4).
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
from .common import (SERIES_TERMS, SINGCUTOFF_REL, GROUPS_PER_LAUNCH, MAX_INNER_FOLDS, FitOptions, check_penalties, _PrimalUnsuitable, _WideTargets, _FoldResult, _aux_stream, _Range, _column_panels, _download_panels, _DeviceShapes, logger)


class PrimalForm:
    """_prepare_* run on the auxiliary stream (p x p side), _sweeps_* on the main stream (V-wide side)."""

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
        for k in range(len(g)):                        # (before the Lanczos run: the host looks at this when it queues the
            self._check_feature_scales(G[F + k])       # fold's first V-wide phase, and must not wait for the run there)
        # (the systems are the leading p x p blocks of their own matrices: the streaming matvec, round 5)
        lmax = None
        if self.normalpha:
            lmax = (ops.lambda_max_dense(G, PP, PP * PP, S, PP, p, self.steps, tol=self.opt.lanczos_tol) if self.opt.lanczos_dense
                    else ops.lambda_max_strided(G, PP, PP * PP, ident, S, PP, self.steps))
        self._check_singcutoff(lmax)
        a2 = ops.penalties(None if lmax is None else lmax[:F], F, self.d_alphas, self.normalpha)
        rhs = ops.gather_rows_f64(X, va, F, M, p, PP)                        # (F, M, PP): Pstim of every inner fold
        Ac = len(cho)
        grid_id = [(j // Ac) * A + cho[j % Ac] for j in range(F * Ac)] if Ac else []

        def assemble(jobs):                                                  # job -> system fold * A + alpha of the grid
            aug = torch.empty((len(jobs), PP + M, PP), dtype=torch.float64, device=self.dev)
            sysv = ops.upload(np.asarray([grid_id[j] for j in jobs], dtype=np.int32), self.dev)
            ops.batch_assemble_sel(G, ident, None, rhs, a2, sysv, len(jobs), A, PP, M, aug, k_fold_stride=PP * PP)
            return aug

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
        # (the series operands first, the Cholesky chains after them: the series part of the sweeps -- it needs only P --
        # runs on the main stream while this stream is still in the chains, _sweeps_primal)
        tp = hp = 0
        if img is not None and use_series:
            tp = ops.pad_to(P.shape[1], 256)
            img.update(Pt=torch.empty(F * tp * PP * 2, dtype=torch.float16, device=self.dev),
                       rs_p=torch.empty(F * tp, dtype=torch.float32, device=self.dev))
            ops.split_rows_f16_groups(P.view(-1, PP), F, P.shape[1], PP, img["Pt"], img["rs_p"])
        series_ready = torch.cuda.Event()
        series_ready.record()
        if Ac:
            H, info = self._sharded_solve(F * Ac, PP, M, assemble)           # (>= F * Ac, M, PP) f32: A_alpha
        else:
            H, info = None, ops.zeros(1, torch.int32, self.dev)
        # the fp16 hi/lo images of the V-independent operands (the A sides of the V-wide contractions) are made once per fold
        # on this stream and shared by every voxel range of the fold (a host-to-host fit works through the targets panel by
        # panel)
        if img is not None:
            if Ac:
                hp = ops.pad_to(Ac * M, 256)
                img.update(Ht=torch.empty(F * hp * PP * 2, dtype=torch.float16, device=self.dev),
                           rs_h=torch.empty(F * hp, dtype=torch.float32, device=self.dev))
                ops.split_rows_f16_alphas(H.view(-1, PP), F, Ac, M, PP, img["Ht"], img["rs_h"])
            img.update(tp=tp, hp=hp)
        hat = dict(F=F, N=PP, M=M, n_v=n_v, n_i=n_i, tr=None if rows_all is None else rows_all[:F], va=va, shared=None,
                   img=img, blocks_ready=blocks_ready,
                   Hs=[(0, F, H, P)], info=info, lmax=None if lmax is None else lmax[:F], a2=a2, cho=cho, ser=ser,
                   d_ser=self.d_ser if use_series else None, moments=use_series, series_ready=series_ready, split=split,
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
        # screening arithmetic for the two sweeps of an inner fold (DualSweeps._screen_terms; the block products B = X'Y feed
        # the refit as well and stay on three MFMAs)
        panel = bool(hat.get("panel"))
        live = hat.get("live") if panel else None
        terms = self._screen_terms(hat, moments, split, PP)
        if not panel:
            self.info["screen_terms"] = min(self.info.get("screen_terms", 3), terms)
        if terms == 1 and not self.opt.screen_two_workgroups:
            terms = 101
        Vp_, V_ = self.Vp, self.V
        scores = torch.empty((A, Vp_), dtype=torch.float32, device=self.dev)
        scores_d = scores if not moments else (scores[:Ad] if cho_first else
                                               torch.empty((max(Ad, 1), Vp_), dtype=torch.float32, device=self.dev))
        part = torch.empty((max(Ad, 1) * M // LC_MB, 4, Vp_), dtype=torch.float32, device=self.dev)
        # two passes over the inner folds when the series terms are ready before the hat matrices (the p x p side forms the
        # terms first, then runs its Cholesky chains): pass 1 = validation statistics, B_f, its image, the series sweep of
        # every fold; pass 2, behind the chains = the fused sweeps, on the images pass 1 kept -- the main stream works through
        # the ~20 ms of the chains instead of waiting for them (LeBel shape: 5 x 0.98 GB of images per full-width range)
        two_pass = bool(moments and Ad and split and by_blocks and hat.get("series_ready") is not None)
        nbuf = F if two_pass else 1
        ystat = torch.empty((nbuf, 3, Vp_), dtype=torch.float32, device=self.dev)
        yblk = torch.empty((nbuf, M // LC_MB, Vp_), dtype=torch.float32, device=self.dev)
        yv = torch.empty((nbuf, M, Vp_), dtype=torch.float32, device=self.dev)
        Vt = ops.pad_to(Vp_, 256)
        # (sums of whole block products overwrite every entry; a contraction over the training rows leaves the padding
        # rows / columns to the zero fill)
        B = (torch.empty((PP, Vt), dtype=torch.float32, device=self.dev) if Xt_val is not None
             else ops.zeros((PP, Vt), torch.float32, self.dev))
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
            Bts = [torch.empty(Vt * PP * 2, dtype=torch.float16, device=self.dev) for _ in range(nbuf)]
            csBs = [None] * nbuf
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
        if two_pass:
            main.wait_event(hat["series_ready"])           # from here on: the series terms; the hat matrices in pass 2
        elif by_blocks and done is not None:
            main.wait_event(done)                          # from here on: the hat matrices / series terms of the fold

        def fused_sweep(f, b):
            Ht_f, rs_h_f = Ht, rs_inv
            if img is not None and "Ht" in img:
                Ht_f, rs_h_f = img["Ht"][f * img["hp"] * PP * 2:], img["rs_h"][f * img["hp"]:]
            else:
                ops.split_rows_f16_alphas(H[f * Ad:(f + 1) * Ad].reshape(Ad * M, PP), 1, Ad, M, PP, Ht, rs_inv)
            self.info["fused_flops"] += 2.0 * Ad * n_v[f] * self.p * V_
            self.info["fused_launches"] += 1
            ops.alpha_sweep_scores_f16x3(Ht_f, rs_h_f, Ad, M, PP, Bts[b], csBs[b][Vp_:], yv[b], Vp_, n_v[f], ystat[b], yblk[b],
                                         self.mode, part, scores_d, accumulate=f > 0, terms=terms, live=live)

        for f in range(F):
            b = f if two_pass else 0
            ops.val_stats(Y, Vp_, va[f], M, n_v[f], ystat[b], yblk[b], yv[b])
            csB = None
            if by_blocks and split:
                # B_f and its fp16 column scales in one pass (the maxima are taken while the sum is written)
                _, csB = ops.combine_colmax([Bv[q] for q in range(F) if q != f], [1.0] * (F - 1), B, Vt, want_scales_for=Vp_)
            elif by_blocks:
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
                if csB is None:
                    csB, _ = ops.col_scales_f16(B, self.p, Vp_, want_flag=False)
                Bt, csBs[b] = Bts[b], csB
                ops.split_cols_f16(B, Vp_, ident, PP, csB, Bt)
                if moments:
                    Pt_f, rs_p_f = Pt, rs_p
                    if img is not None and "Pt" in img:
                        Pt_f, rs_p_f = img["Pt"][f * img["tp"] * PP * 2:], img["rs_p"][f * img["tp"]:]
                    else:
                        ops.split_rows_f16(P[f], Tm, PP, Pt, rs_p)
                    self.info["plain_flops"] += 2.0 * SERIES_TERMS * n_v[f] * self.p * V_
                    self.info["plain_launches"] += 1
                    ops.series_sweep_scores_f16x3(Pt_f, rs_p_f, M, n_v[f], PP, Bt, self._cs_inv_padded(csB, Vt), Vt, yv[b], Vp_,
                                                  ystat[b], yblk[b], self.d_coef, hat["d_ser"], part_s, scores,
                                                  accumulate=f > 0, terms=terms, live=live)
                if Ad and not two_pass:
                    fused_sweep(f, b)
            else:
                if not by_blocks:
                    ops.gemm_grouped(Xt_f, Nmax, 0, Y, Y.stride(0), tr[f], B, Vt, PP, Vp_, Ni, [0, Vp_ // COL_TILE])
                ops.alpha_sweep_scores(H[f * A:(f + 1) * A], A, M, PP, B, Vp_, ident, yv[b], n_v[f], ystat[b], yblk[b],
                                       self.mode, part, scores, accumulate=f > 0)
        if two_pass:
            if done is not None:
                main.wait_event(done)
            for f in range(F):
                fused_sweep(f, f)
        if moments and Ad and not cho_first:
            for i, a in enumerate(cho):
                scores[a].copy_(scores_d[i])
        if terms in (1, 101):
            self._after_screening(hat, Y, scores, ystat[0], F, sum(n_v))
        if by_blocks:
            # the outer block's product = the sum of all its validation blocks': the refit's operand (_primal_refit_inputs)
            if split:
                # ... written straight into the rows it has in the refit's operand [B_o ; Y_te] (_primal_refit_inputs: the
                # refit is on the critical tail of a single-alpha fit -- no fill and no copy of 1 GB there)
                ext = torch.empty((PP + int(hat.get("n_te") or 0), Vt), dtype=torch.float32, device=self.dev)
                hat["B_all"], hat["csB_all"] = ops.combine_colmax([Bv[q] for q in range(F)], [1.0] * F, ext[:PP], Vt,
                                                                  want_scales_for=Vp_)
                hat["ext"] = ext
            else:
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
        B_all = st["hat"].get("B_all")
        ext = st["hat"].get("ext")
        if ext is not None and ext.shape[0] == PP + n_t:
            ext[PP:, self.Vp:].zero_()                                       # (B_o is in place already, _sweeps_primal)
        else:
            ext = ops.zeros((PP + n_t, Vt), torch.float32, self.dev)
            if B_all is not None:
                ext[:PP].copy_(B_all)
        if B_all is not None:
            # the inner CV of this step left  B_o = Rstim'Rresp  of the outer block (the sum of its validation blocks')
            csB = st["hat"].get("csB_all") if st["split"] else None          # (taken while B_all was written)
            if st["split"] and csB is None:
                csB = ops.col_scales_f16(ext, self.p, self.Vp, want_flag=False)[0]
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
