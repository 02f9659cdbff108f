"""The inner CV in the dual (n x n) form: hat matrices of every (inner fold, alpha) -- batched Cholesky for the small
alphas, the shared polynomial series for the large ones -- and the V-wide fused sweeps.
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


class DualSweeps:
    """ridge_corr_torch for all inner folds of an outer fold (ridge_regression.py:66-141, nested_cv.py:334-415)."""

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
                    # (round 6) the chain's products on the fp16x3 kernel: Kn's images once, Q_(j-1)'s per term
                    x3 = bool(self.opt.series_chain_f16x3 and self._split_assumed() and Mq % 256 == 0 and N % 32 == 0 and N >= 64
                              and fcl <= 64)
                    if x3:
                        Np = ops.pad_to(N, 256)                      # (every group's image is padded to whole 256-row tiles)
                        Kt = torch.empty(fcl * Np * N * 2, dtype=torch.float16, device=self.dev)
                        rs_k = torch.empty(fcl * Np, dtype=torch.float32, device=self.dev)
                        ops.split_rows_f16_groups(Kn.view(-1, N), fcl, N, N, Kt, rs_k)
                        rows_n = ops.idx_tensor(np.arange(N), N, self.dev)
                        tiles3 = [f * (Mq // 256) for f in range(fcl + 1)]
                        Qt = torch.empty(fcl * Mq * N * 2, dtype=torch.float16, device=self.dev)
                    for j in range(SERIES_TERMS):
                        if j:
                            Qn = torch.empty_like(Q)
                            if x3:
                                Q2 = Q.view(N, fcl * Mq)
                                cs_q, _ = ops.col_scales_f16(Q2, N, fcl * Mq, want_flag=False)
                                ops.split_cols_f16(Q2, fcl * Mq, rows_n, N, cs_q, Qt)
                                ops.gemm_grouped_f16x3(Kt, rs_k, N, Qt, cs_q[fcl * Mq:], Qn.view(N, fcl * Mq), fcl * Mq, fcl * Mq, N,
                                                       tiles3)
                            else:
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
                ops.split_rows_f16_alphas(H.view(-1, N), fc, Ac, M, N, img["Ht"], img["rs_h"])
                Ac_s = Ac - int(getattr(self, "scr_drop", 0))
                if 0 < Ac_s < Ac and self.opt.screen_inner:
                    # the screening pass' image: the first Ac_s alphas only (the others are screened from the series terms)
                    hp_s = ops.pad_to(Ac_s * M, 256)
                    img.update(hp_s=hp_s, Ht_s=torch.empty(fc * hp_s * N * 2, dtype=torch.float16, device=self.dev),
                               rs_hs=torch.empty(fc * hp_s, dtype=torch.float32, device=self.dev))
                    ops.split_rows_f16_alphas_sel(H.view(-1, N), fc, Ac, Ac_s, M, N, img["Ht_s"], img["rs_hs"])
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
        rg_ = self.cur                                    # (the range this call works on: the second part may run later)
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
            main.wait_event(hat["data_ready"])            # a design of the fold's own (normalize_features), made on the auxiliary stream
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
                ops.split_cols_f16(Y, Vp_, ops.idx_tensor(union, len(union), self.dev), len(union), cs, Yu,
                                   live=hat.get("live") if hat.get("panel") else None)
                Yt = [Yu] * nbuf
                views = [(len(union), g0, gl) for g0, gl in gaps]
                hat["image"] = (Yu, union)                # the refit permutes its operand out of it (_refit_operands)
            else:
                Yt = [torch.empty(Vt * N * 2, dtype=torch.float16, device=self.dev) for _ in range(nbuf)]
                views = [(0, 0, 0)] * F
        # screening arithmetic (FitOptions.screen_inner): one MFMA per product for both contractions of the inner CV
        panel = bool(hat.get("panel"))                    # this call scores the refinement's column panel (_refine_undecided)
        live = hat.get("live") if panel else None
        terms = self._screen_terms(hat, moments, split, N)
        if not panel:
            self.info["screen_terms"] = min(self.info.get("screen_terms", 3), terms)     # (1: some step of the fit screened)
        # screening pass: the trailing factorised alphas that the 4-term series serves to << the screening error are scored from
        # the series moments (FitOptions.screen_series_tol): the fused launch carries Ad_f < Ad alphas, from an image of its own
        imgs_ = hat.get("imgs") or []
        drop = int(getattr(self, "scr_drop", 0)) if (terms == 1 and moments and cho_first and self.d_ser is not None) else 0
        if drop and not (drop == Ad or (imgs_ and all(im is not None and im.get("Ht_s") is not None for im in imgs_))):
            drop = 0
        Ad_f = Ad - drop
        coef_s, dser_s = (self.d_coef_scr, self.d_ser_scr) if drop else (self.d_coef, hat["d_ser"])
        if not panel:
            self.info["fused_alphas"] = Ad_f
        if terms == 1 and not self.opt.screen_two_workgroups:
            terms = 101                                   # (lc_*_sweep_scores_f16x3_folds: the one-workgroup-per-CU kernel)
        folds = [(f0 + j, j, H, P) for f0, fc, H, P in hat["Hs"] for j in range(fc)]
        # the operators' fp16 images made with the hat matrices (_hat_matrices), per chunk: fold f0 + j is group j
        imgs = hat.get("imgs") or [None] * len(hat["Hs"])
        img_of = {f0 + j: (im, j) for (f0, fc, _, _), im in zip(hat["Hs"], imgs) if im is not None for j in range(fc)}
        fused = False
        Pt = rs_p = part_s = Tbuf = cs_inv = rowmap = slab_light = Tm = None
        # (round 5) narrow ranges -- a rank's 10 000 voxels of an 8-GPU job, the first panels of host targets -- are 1.25 rounds
        # of workgroups per inner fold and launch: ALL inner folds of the step in one launch each (lc_*_f16x3_folds: the
        # folds' stacked operator images against the one target image, every fold skipping its own block) fill the chip
        # instead of ending every fold in a partial round.  The same kernel, the same fold order of the fp32 adds: the same
        # bits.  Needs the shared target image and all folds' images in one stack; wide ranges keep the launch per fold
        # (measured slower merged at 80 000 voxels: profiles/experiments/README.md)
        one_launch = bool(moments and split and shared is not None and len(hat["Hs"]) == 1 and imgs[0] is not None
                          and imgs[0].get("Pt") is not None and (Ad == 0 or imgs[0].get("Ht") is not None)
                          and 1 < F <= 64 and self.opt.folds_in_one_launch_tiles > 0
                          # (the refinement's panel: its capacity is generous, the voxels it holds are few -- always one launch)
                          and (panel or ((max(Ad, 1) * M + 255) // 256) * (Vt // 256) < self.opt.folds_in_one_launch_tiles))

        at_once = bool(self.opt.finalize_folds_at_once and moments and split and 1 < F <= 64)

        def series_part():
            nonlocal fused, Pt, rs_p, part_s, Tbuf, cs_inv, rowmap, slab_light, Tm
            if moments:
                # ---- pass 1: validation statistics, operand split, series contraction + moment kernel
                Tm, rowmap, slab_light = self._series_layout(M)
                fused = slab_light is None                   # layout of the moments epilogue: the terms are never stored
                tp = ops.pad_to(Tm, 256)
                Pt = torch.empty(tp * N * 2, dtype=torch.float16, device=self.dev)
                rs_p = torch.empty(tp, dtype=torch.float32, device=self.dev)
                if fused:
                    part_s = torch.empty((1, M // LC_MB, 18, Vp_), dtype=torch.float32, device=self.dev)
                else:
                    Tbuf = torch.empty((Tm, Vt), dtype=torch.float32, device=self.dev)
                cs_inv = self._cs_inv_padded(cs, Vt)                               # padded to the plain GEMM's tiles
                if hat.get("series_ready") is not None:
                    main.wait_event(hat["series_ready"])
                # validation statistics of all inner folds in one launch (the blocks are independent)
                for f0 in range(0, F, 64):
                    f1 = min(F, f0 + 64)
                    ops.val_stats_folds(Y, Vp_, va[f0:f1], f1 - f0, M, n_v[f0:f1], ystat[f0:f1], yblk[f0:f1], yv[f0:f1], live=live)
                if one_launch and fused:
                    im = imgs[0]
                    part_f = torch.empty((F, M // LC_MB, 18, Vp_), dtype=torch.float32, device=self.dev)
                    self.info["plain_flops"] += sum(2.0 * SERIES_TERMS * n_v[f] * hat["n_i"][f] * V_ for f in range(F))
                    self.info["plain_launches"] += 1
                    self.info["series_flops"] = self.info.get("series_flops", 0.0) + sum(2.0 * SERIES_TERMS * n_v[f] * hat["n_i"][f] * V_ for f in range(F))
                    self.info["series_launches"] = self.info.get("series_launches", 0) + 1
                    ops.series_sweep_scores_f16x3_folds(im["Pt"], im["rs_p"], M, n_v, N, Yt[0], cs_inv, Vt, yv, Vp_, ystat, yblk,
                                                        coef_s, dser_s, part_f, scores, False, views, terms=terms,
                                                        live=live)
                    return
                part_sf = (torch.empty((F, M // LC_MB, 18, Vp_), dtype=torch.float32, device=self.dev)
                           if (fused and at_once) else None)
                for f, j, H, P in folds:
                    if shared is None:
                        ops.split_cols_f16(Y, Vp_, tr[f], N, cs, Yt[f], live=live)
                    Pt_f, rs_p_f = Pt, rs_p
                    if f in img_of:
                        im, g = img_of[f]
                        Pt_f, rs_p_f = im["Pt"][g * im["tp"] * N * 2:], im["rs_p"][g * im["tp"]:]
                    else:
                        ops.split_rows_f16(P[j], Tm, N, Pt, rs_p)
                    self.info["plain_flops"] += 2.0 * SERIES_TERMS * n_v[f] * hat["n_i"][f] * V_
                    self.info["plain_launches"] += 1
                    if fused:
                        # (the launches of the series-moments instantiation of k_sweep_f16x3: bench.py prices them with the
                        # score launches -- the same kernel, the inner CV's other contraction)
                        self.info["series_flops"] = self.info.get("series_flops", 0.0) + 2.0 * SERIES_TERMS * n_v[f] * hat["n_i"][f] * V_
                        self.info["series_launches"] = self.info.get("series_launches", 0) + 1
                        ops.series_sweep_scores_f16x3(Pt_f, rs_p_f, M, n_v[f], N, Yt[f], cs_inv, Vt, yv[f], Vp_, ystat[f], yblk[f],
                                                      coef_s, dser_s, part_s if part_sf is None else part_sf[f], scores,
                                                      accumulate=(f > 0) if part_sf is None else 2,
                                                      bview=views[f], terms=terms, live=live)
                        continue
                    ops.gemm_grouped_f16x3(Pt_f, rs_p_f, Tm, Yt[f], cs_inv, Tbuf, Vt, Vt, N, [0, Vt // 256], slab_light,
                                           bview=views[f])
                    ops.series_scores(Tbuf, Vt, SERIES_TERMS, M, n_v[f], Vp_, yv[f], ystat[f], coef_s, dser_s,
                                      scores, accumulate=f > 0, rowmap=rowmap)
                if part_sf is not None:                      # the folds' partial moments -> scores, all folds in one pass
                    ops.series_sweep_finalize_folds(part_sf, ystat, yblk, M, n_v, Vp_, coef_s, dser_s, scores, accumulate=False, live=live)

        def fused_part():
            if done is not None:
                main.wait_event(done)
            # the f32 side path of too-wide target columns runs on a stream of its own BESIDE the sweeps queued below (a
            # few dozen workgroups per launch); its scores are written over the main path's before any alpha is chosen
            side_job = self._side_sweeps_begin(hat, rg_) if (self.side is not None and split and not panel) else None
            # ---- pass 2 (the only one without the moment path): fused sweeps of the alphas that have hat matrices
            if one_launch and fused and Ad_f:
                im = imgs[0]
                part_f = torch.empty((F, Ad_f * M // LC_MB, 4, Vp_), dtype=torch.float32, device=self.dev)
                self.info["fused_flops"] += sum(2.0 * Ad_f * n_v[f] * hat["n_i"][f] * V_ for f in range(F))
                self.info["fused_launches"] += 1
                self.info["folds_per_launch"] = F
                ops.alpha_sweep_scores_f16x3_folds(im["Ht_s" if drop else "Ht"], im["rs_hs" if drop else "rs_h"], Ad_f, M, N, Yt[0],
                                                   cs[Vp_:], yv, Vp_, n_v, ystat, yblk, self.mode, part_f, scores_d, False, views,
                                                   terms=terms, live=live)
            part_ff = (torch.empty((F, Ad_f * M // LC_MB, 4, Vp_), dtype=torch.float32, device=self.dev)
                       if (at_once and fused and Ad_f and not one_launch) else None)
            for f, j, H, P in (() if (one_launch and fused) else folds):
                b = f if moments else 0
                if not moments:
                    ops.val_stats(Y, Vp_, va[f], M, n_v[f], ystat[b], yblk[b], yv[b])
                if split:
                    if not moments and shared is None:
                        ops.split_cols_f16(Y, Vp_, tr[f], N, cs, Yt[b], live=live)
                    if Ad_f:
                        self.info["fused_flops"] += 2.0 * Ad_f * n_v[f] * hat["n_i"][f] * V_
                        self.info["fused_launches"] += 1
                        Ht_f, rs_h_f = Ht, rs_inv
                        if drop:                             # (the images exist: checked where ``drop`` was decided)
                            im, g = img_of[f]
                            Ht_f, rs_h_f = im["Ht_s"][g * im["hp_s"] * N * 2:], im["rs_hs"][g * im["hp_s"]:]
                        elif moments and f in img_of and img_of[f][0]["Ht"] is not None:
                            im, g = img_of[f]
                            Ht_f, rs_h_f = im["Ht"][g * im["hp"] * N * 2:], im["rs_h"][g * im["hp"]:]
                        else:
                            ops.split_rows_f16_alphas(H[j * Ad:(j + 1) * Ad].reshape(Ad * M, N), 1, Ad, M, N, Ht, rs_inv)
                        ops.alpha_sweep_scores_f16x3(Ht_f, rs_h_f, Ad_f, M, N, Yt[b], cs[Vp_:], yv[b], Vp_, n_v[f], ystat[b],
                                                     yblk[b], self.mode, part if part_ff is None else part_ff[f], scores_d,
                                                     accumulate=(f > 0) if part_ff is None else 2, bview=views[f],
                                                     terms=terms, live=live)
                else:
                    self.info["fused_flops"] += 2.0 * A * n_v[f] * hat["n_i"][f] * V_
                    self.info["fused_launches"] += 1
                    ops.alpha_sweep_scores(H[j * A:(j + 1) * A], A, M, N, Y, Vp_, tr[f], yv[b], n_v[f], ystat[b], yblk[b],
                                           self.mode, part, scores, accumulate=f > 0)
            if part_ff is not None:                          # the folds' partial moments -> scores, all folds in one pass
                ops.alpha_sweep_finalize_folds(part_ff, ystat, yblk, Ad_f, M, n_v, Vp_, self.mode, scores_d, accumulate=False, live=live)
            if moments and Ad and not cho_first:
                for i, a in enumerate(cho):
                    scores[a].copy_(scores_d[i])
            if terms in (1, 101):
                self._after_screening(hat, Y, scores, ystat[0], F, sum(n_v))
            if side_job is not None:
                self._side_sweeps_end(side_job, scores)
            self.sweeps_done = torch.cuda.Event()
            self.sweeps_done.record()
            return scores


        if not moments:                                   # one pass only: nothing to put another step's work behind
            out = fused_part()
            return (lambda: out) if split_phase else out
        series_part()
        return fused_part if split_phase else fused_part()

    def _screen_terms(self, hat, moments, split, depth):
        """1 when this call of the sweeps runs on the screening arithmetic (one fp16 MFMA per product), else 3.  Only where
        the score table feeds an ARGMAX and nothing else, and the driver has said which: ``argmax_only`` -- every voxel's own
        (nested_cv.py:405-411): undecided voxels are scored again (_refine_undecided) --, or ``mean_only`` -- single_alpha, the
        argmax of the voxel MEAN (:396-400): checked in fold_select (_mean_check).  A caller that wants the scores themselves
        (ridge.ridge_corr = ridge_corr_torch) gets three-MFMA scores.  Needs the moments form of the series alphas, the fp16
        path, correlation scores and a contraction depth that is a multiple of 64 (two K-tiles per ring stage)."""
        panel = bool(hat.get("panel"))
        ok = (self.opt.screen_inner and (getattr(self, "argmax_only", False) or getattr(self, "mean_only", False))
              and moments and split and self.mode == LC_SCORE_CORR and depth % 64 == 0 and not panel and not hat.get("exact"))
        return 1 if ok else 3

    def _after_screening(self, hat, Y, scores, ystat0, F, n_val_rows):
        """What follows a screening pass' score table (sums over the F inner folds), before any alpha is chosen from it."""
        if getattr(self, "mean_only", False):
            # single_alpha: nothing per voxel to decide -- what the voxel MEAN's error scales with goes along with the
            # per-alpha sums (fold_choose / fold_choose_joint), the lead of the best alpha is checked in fold_select
            hat["mean_check"] = dict(ksums=ops.kappa_sums(ystat0, self.V), F=int(F), rows=int(n_val_rows))
            return
        # the voxels the screening pass leaves undecided: scored again with the three-MFMA products, their columns of the
        # table overwritten -- before the side path's columns are, and before any alpha is chosen
        if (self.opt.refine_on_side_stream and not self.shard.active and self.side is None and not hat.get("panel")
                and self.dev.type == "cuda"):
            # ... on a stream of its own: the panel is a few column tiles (64 workgroups at cfg2, ~1 ms of latency per step
            # with three quarters of the chip idle), and nothing the main stream queues next -- the next step's sweeps --
            # depends on it; the alpha choice follows on the same stream (fold_choose), the step's refit waits for both
            rs = self.refine_stream
            ev = torch.cuda.Event()
            ev.record()
            rs.wait_event(ev)
            for t in (scores, ystat0):
                t.record_stream(rs)                    # (made on the main stream, read / written over there)
            with torch.cuda.stream(rs):
                self._refine_undecided(hat, Y, scores, ystat0, F, n_val_rows)
            hat["refine_stream"] = rs
            return
        self._refine_undecided(hat, Y, scores, ystat0, F, n_val_rows)

    def _refine_capacity(self, V):
        """Columns of the refinement's panel for a voxel range of V columns: FitOptions.screen_panel_first of them (half)
        until a step of this fit has reported its share of undecided voxels, then the largest share reported so far x
        FitOptions.screen_panel_margin + 1024 columns; whole 256-column tiles."""
        if self.opt.screen_panel_cols > 0:
            return int(min(ops.pad_to(self.opt.screen_panel_cols, 256), ops.pad_to(V, 256)))
        if self.shard.simulate:                          # (timing studies: the panel a real rank's ~1 % would get, _screen_check)
            return int(min(ops.pad_to(max(V // 32, 512), 256), ops.pad_to(V, 256)))
        fracs = getattr(self, "_undecided_fracs", None)
        # (round 6, later: every pass over the panel costs what its voxels cost -- lc_gather_f32 / lc_col_scales_f16 /
        # lc_split_cols_f16 / lc_val_stats_folds / the sweeps all read the count on the device -- so the capacity is generous
        # while no step has reported: a panel that cannot hold a step's undecided voxels costs a second pass and a host round
        # trip.  Not free, though: the sweeps' launches cover the CAPACITY, and a workgroup that leaves at once still had to
        # wait for a CU with 128 KB of LDS free -- 40 000 empty columns x 5 folds cost a cfg2 fit 1.5-2.5 ms
        # (tools/panel_cap_queue_ab.sh) -- so from the first report on the capacity follows the shares seen: the largest x
        # FitOptions.screen_panel_margin + 1024 columns.  Measured on one box, margin 1.25 / 2 / capacity V/2 throughout: cfg2
        # resident 80.7 / 80.9 / 82.3 ms, cfg5's shape 162 / 163 / 168, half the voxels pure noise 134 / 143 / 136, a tenth of
        # the signal 172 / 175 / 177; a step beyond the margin costs a second panel, _screen_check)
        if fracs and self.opt.screen_panel_margin > 0:
            cap = max(int(self.opt.screen_panel_margin * max(fracs) * V) + 1024, 2048)
        else:
            cap = max(int(self.opt.screen_panel_first * V), 2048)
        return int(min(ops.pad_to(max(cap, 256), 256), ops.pad_to(V, 256)))

    def _refine_undecided(self, hat, Y, scores, ystat0, F, n_val_rows, cap=None):
        """Second half of the two-precision inner CV (FitOptions.screen_inner).  ``scores`` holds the sums
        over the F inner folds of the SCREENING scores (one fp16 MFMA per product: good to ~1e-5 of a fold-mean score).
        nested_cv.py:408-411 only takes each voxel's argmax of them, so a voxel whose two best alphas lie further apart
        than screen_tau / sqrt(validation rows scored) (scaled by rms / std of the column: an offset costs operand bits) is
        decided; the others --
        lc_undecided_cols: ~1 % at cfg2 -- are gathered into a column panel, scored again by the SAME sweeps with the
        three-MFMA products (every V-wide kernel keeps a voxel's arithmetic inside its own column: the panel's scores are,
        bit for bit, what the full-width three-MFMA sweeps give those voxels) and written over their columns of the table.
        The panel's capacity is fixed when its launches are queued; how many columns it holds only the device knows -- the
        sweeps' tiles behind the last one leave at once (live), and a count beyond the capacity is reported to fold_select,
        which scores the whole range again (hat["screen_check"])."""
        Vp_, V_, A = self.Vp, self.V, self.A
        cap = self._refine_capacity(V_) if cap is None else int(min(ops.pad_to(max(int(cap), 256), 256), ops.pad_to(V_, 256)))
        hat["refine_args"] = (ystat0, F, n_val_rows)          # (a panel that overflows is followed by one that does not: _screen_check)
        # (the table holds SUMS over the F folds: the gap of the fold means x F)
        tau_sum = self.opt.screen_tau * F / float(np.sqrt(max(int(n_val_rows), 1)))
        lst, count = ops.undecided_cols(scores, A, V_, tau_sum, ystat0, cap)
        # voxel shards: every rank must take the same decisions in fold_select -- score the range again (the choice that
        # follows all-reduces its histogram) and switch the screening off (this very all-reduce would then be missing on one
        # rank): (undecided columns found, "the panel does not hold them all") as the MAX over the ranks, in place
        if self.shard.active:
            self.shard.all_reduce_(count[1:3], "max")
        # (every pass over the panel honours count[0], the number of columns it holds: it costs what its voxels cost, not
        # what its capacity would -- which is why the capacity can be generous, _refine_capacity)
        Yp = torch.empty((self.Ttot, cap), dtype=torch.float32, device=self.dev)
        ops.gather(Y, Y.stride(0), None, self.Ttot, lst, cap, Yp, live=count[0:1])
        cs_p, _ = ops.col_scales_f16(Yp, self.Ttot, cap, want_flag=False, live=count[0:1])    # (per column, from the same values: the same scales)
        hp = dict(hat)
        hp.update(cs=cs_p, split=True, panel=True, live=count[0:1], data_ready=None, series_ready=None)
        keep = {k: self.info.get(k) for k in ("plain_flops", "plain_launches", "fused_flops", "fused_launches", "series_flops",
                                              "series_launches", "precision", "fused_alphas", "series_terms", "folds_per_launch")}
        prev = self.cur
        self.cur = _Range(0, cap, cap, Yp, None)
        try:
            sc = self._sweeps(hp, Yp, None)
        finally:
            self.cur = prev
            self.info.update(keep)                             # (the counters describe the full-width launches)
        ops.scatter_cols(sc, A, lst, cap, scores)
        host = torch.empty(2, dtype=torch.int32, pin_memory=True)
        host.copy_(count[1:3], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        hat["screen_check"] = (ev, host, V_, cap, bool(self.shard.active))
        self.info["refine_launch_cols"] = self.info.get("refine_launch_cols", 0) + cap

    def _side_cols_of(self, rg):
        """(first position in the side panel, positions' count, range-local column of each) of the side columns that lie in
        the voxel range; None when there are none."""
        cols = self.side["cols"]
        sel = np.nonzero((cols >= rg.c0) & (cols < rg.c0 + rg.V))[0]
        if len(sel) == 0:
            return None
        return int(sel[0]), int(len(sel)), (cols[sel] - rg.c0).astype(np.int32)

    def _side_sweeps_begin(self, hat, rg):
        """The inner-CV scores of the side panel's columns (target columns whose dynamic range the fp16 hi/lo split cannot
        carry, _register_side) in exact f32 arithmetic: the factorised alphas through the f32-input MFMA sweep on the same
        f32 hat matrices the fp16 images were split from, the alphas on the polynomial series through the f32 product with
        the same shared terms and the moment kernel on stored terms (lc_series_scores) -- what ridge_corr_torch computes for
        every column alike (ridge_regression.py:104-133).  Queued on the side stream behind everything the current stream
        has queued so far (the hat matrices are complete there); _side_sweeps_end writes the result over the main path's
        scores of those columns BEFORE any alpha is chosen from them.  Returns the job, or None."""
        hit = self._side_cols_of(rg)
        if hit is None:
            return None
        s0, ns, local = hit
        side = self.side
        Vs = ops.pad_to(ns, COL_TILE)
        A, N, M, tr, va, n_v = self.A, hat["N"], hat["M"], hat["tr"], hat["va"], hat["n_v"]
        moments, cho = hat["moments"], hat["cho"]
        Ad = len(cho) if moments else A
        cho_first = list(cho) == list(range(len(cho)))
        main, ss = torch.cuda.current_stream(), self.side_stream
        dst = np.full(Vs, -1, dtype=np.int32)
        dst[:ns] = local
        d_dst = ops.upload(dst, self.dev)                      # (on the current stream, before the event the side stream waits for)
        # a handful of columns (one outlier voxel among 80 000 is the usual case): their products streamed (lc_gemv_cols_f32,
        # fp64 accumulation) instead of 128-column MFMA tiles that are 127 / 128 padding and hold 15-30 CUs for ~0.3 ms per
        # launch beside the fit's own sweeps; the factorised alphas' predictions go through the moment kernel too, as
        # "terms" with unit coefficients -- the same score formula for every alpha
        few = bool(moments and ns <= ops.GEMV_MAX_COLS and Ad <= 8)
        if few and side.get("unit") is None:
            side["unit"] = (ops.upload(np.eye(max(Ad, 1), dtype=np.float64), self.dev),
                            ops.upload(np.asarray(list(cho) or [0], dtype=np.int32), self.dev))
        start = torch.cuda.Event()
        start.record()
        ss.wait_event(start)
        with torch.cuda.stream(ss):
            Ys = side["Y"][:, s0:]                             # (T, >= ns) view: the range's side columns first
            if Ys.shape[1] < Vs:                               # (the panel's last columns: a padded copy of their own)
                Yp = ops.zeros((self.Ttot, Vs), torch.float32, self.dev)
                Yp[:, : Ys.shape[1]].copy_(Ys)
                Ys = Yp
            sc = torch.empty((A, Vs), dtype=torch.float32, device=self.dev)
            sc_d = sc if not moments else (sc[:Ad] if cho_first else
                                           torch.empty((max(Ad, 1), Vs), dtype=torch.float32, device=self.dev))
            part = torch.empty((max(Ad, 1) * M // LC_MB, 4, Vs), dtype=torch.float32, device=self.dev)
            ystat = torch.empty((3, Vs), dtype=torch.float32, device=self.dev)
            yblk = torch.empty((M // LC_MB, Vs), dtype=torch.float32, device=self.dev)
            yv = torch.empty((M, Vs), dtype=torch.float32, device=self.dev)
            Tbuf = rowmap = Tm = None
            if moments:
                Tm, rowmap, _ = self._series_layout(M)
                Tbuf = torch.empty((Tm, Vs), dtype=torch.float32, device=self.dev)
            if few:
                Tbuf = ops.zeros((Tm, Vs), torch.float32, self.dev)                 # (columns >= ns stay zero)
                Tcho = ops.zeros((max(Ad, 1) * M, Vs), torch.float32, self.dev)
                unit, d_cho = side["unit"]
            for f0, fc, H, P in hat["Hs"]:
                for j in range(fc):
                    f = f0 + j
                    ops.val_stats(Ys, Vs, va[f], M, n_v[f], ystat, yblk, yv)
                    if few:
                        if Ad:
                            ops.gemv_cols(H[j * Ad:(j + 1) * Ad].reshape(Ad * M, N), Ad * M, N, Ys, tr[f], ns, Tcho)
                            ops.series_scores(Tcho, Vs, Ad, M, n_v[f], Vs, yv, ystat, unit, d_cho, sc, accumulate=f > 0)
                        ops.gemv_cols(P[j], Tm, N, Ys, tr[f], ns, Tbuf)
                        ops.series_scores(Tbuf, Vs, SERIES_TERMS, M, n_v[f], Vs, yv, ystat, self.d_coef, hat["d_ser"], sc,
                                          accumulate=f > 0, rowmap=rowmap)
                        continue
                    if Ad:
                        ops.alpha_sweep_scores(H[j * Ad:(j + 1) * Ad], Ad, M, N, Ys, Vs, tr[f], yv, n_v[f], ystat, yblk,
                                               self.mode, part, sc_d, accumulate=f > 0)
                    if moments:
                        ops.gemm_grouped(P[j], N, 0, Ys, Ys.stride(0), tr[f], Tbuf, Vs, Tm, Vs, N, [0, Vs // COL_TILE])
                        ops.series_scores(Tbuf, Vs, SERIES_TERMS, M, n_v[f], Vs, yv, ystat, self.d_coef, hat["d_ser"], sc,
                                          accumulate=f > 0, rowmap=rowmap)
            if moments and Ad and not cho_first and not few:
                for i, a in enumerate(cho):
                    sc[a].copy_(sc_d[i])
            done = torch.cuda.Event()
            done.record()
        sc.record_stream(main)
        return sc, d_dst, Vs, done

    def _side_sweeps_end(self, job, scores):
        sc, d_dst, Vs, done = job
        torch.cuda.current_stream().wait_event(done)
        ops.scatter_cols(sc, self.A, d_dst, Vs, scores)

    def _alpha_scores(self, K, Y, inner_abs):
        cs, split = self._target_scales(Y)
        hat = self._hat_matrices(K, inner_abs, moments=self._series_by_moments(split))
        hat.update(cs=cs, split=split)
        scores = self._sweeps(hat, Y)
        if hat.get("refine_stream") is not None:       # (the table is completed on the refinement's stream: the caller reads it here)
            ev = torch.cuda.Event()
            with torch.cuda.stream(hat["refine_stream"]):
                ev.record()
            torch.cuda.current_stream().wait_event(ev)
        return scores, hat["info"]
