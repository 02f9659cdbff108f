"""Alpha choice and refit (ridge_torch, ridge_regression.py:9-63): grouping of the voxels by chosen alpha, the refit
operators (explicit inverses, augmented solves, the polynomial series, the spectral route), the grouped V-wide product;
systems solved ahead of the choice (refit_ahead, fold_speculate).
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


class Refit:
    """The refit half of RidgeCVEngine."""

    # -------------------------------------------------------------- alpha selection
    def choose(self, scores, single_alpha, check=None):
        """(Vp,) int32 device vector of alpha indices: per-voxel first argmax (nested_cv.py:405-411)
        or, for ``single_alpha``, the argmax of the across-voxel mean (:396-400; the per-alpha sums
        are all-reduced over the voxel shards).  ``check``: the screening pass' note for the voxel mean
        (DualSweeps._after_screening) -- its kappa sums travel with the per-alpha sums, see _mean_sums."""
        if single_alpha:
            _, rowsum = ops.select_alpha(scores, self.A, self.Vp, want_best=False, want_rowsum=True)
            rowsum = self._mean_sums(rowsum, [check] if check else [])     # A doubles, on the device: the choice never visits the host
            best = torch.empty(self.Vp, dtype=torch.int32, device=self.dev)
            return ops.fill_argmax(rowsum, self.A, best, self.Vp)     # first maximum, like torch.argmax
        return ops.select_alpha(scores, self.A, self.Vp)[0]

    def _mean_sums(self, rowsum, checks):
        """The per-alpha score sums of all voxel shards (all-reduce, on the device).  Under the screening pass (``checks``: the
        notes of the ranges summed) the sums of kappa and kappa^2 ride along in the same all-reduce, and the result leaves for
        pinned memory: fold_select looks at the lead of the best alpha there (_mean_check).  Returns the (A,) sums."""
        A = self.A
        if not checks:
            self.shard.all_reduce_(rowsum, "sum")
            return rowsum
        tot = torch.empty(A + 2, dtype=torch.float64, device=self.dev)
        tot[:A].copy_(rowsum)
        tot[A:].copy_(checks[0]["ksums"])
        for c in checks[1:]:
            ops.accumulate_f64(c["ksums"], tot[A:])
        self.shard.all_reduce_(tot, "sum")
        host = torch.empty(A + 2, dtype=torch.float64, pin_memory=True)
        host.copy_(tot, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        for c in checks:
            c.update(host=host, ev=ev, done=False)
        return tot[:A]

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
        perm, count_h, ev = pending[:3]
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
        best_h = None
        if getattr(self, "_mo", None) is not False and self._mo_possible():
            # the alpha indices themselves travel with the histogram: the mean-operator refit groups the voxels by their alpha
            # TUPLE over the folds on the host (engine/mean_refit.py; 4 bytes per voxel)
            best_h = torch.empty(self.Vp, dtype=torch.int32, pin_memory=True)
            best_h.copy_(best[: self.Vp], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return perm, count_h, ev, best_h

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
        alphas_idx = list(alphas_idx)
        Gc, (rows, N_o) = len(alphas_idx), rhs.shape
        if self.spectral:
            d_al = ops.upload(np.asarray([self.alphas[a] for a in alphas_idx], dtype=np.float64), self.dev)
            a2_sel = ops.penalties(lmax_o, 1, d_al, self.normalpha)
            H = self._spectral_operators(K, tr_o, None, rhs.reshape(1, rows, N_o), 1, N_o, rows, a2_sel, Gc,
                                         [min(self._real_rows(tr_o), self.p)], cache_key=("refit", tr_o.data_ptr()))
            return H.view(Gc, rows, N_o), ops.zeros(max(Gc, 1), torch.int32, self.dev)
        a2_o = ops.penalties(lmax_o, 1, self.d_alphas, self.normalpha)              # (A,): grid F = 1
        inv = [a for a in alphas_idx if self._refit_by_inverse([a])]
        if inv and len(inv) < Gc:
            # a grid on both sides of refit_inverse_min_alpha: every alpha takes the route IT qualifies for, whatever else
            # is asked for in the same call -- an alpha's operator must not depend on which other alphas the voxels of a
            # range, a panel plan or a voxel shard happen to need at the same moment (round 5's shard fuzzing: a sharded
            # fit solved all factorised alphas ahead in one list, the unsharded one only those in use, and the lists fell
            # on different sides of the threshold: last-bit differences between the two)
            sol = [a for a in alphas_idx if a not in inv]
            M_i, info_i = self._refit_chol(K, tr_o, lmax_o, rhs, inv)
            M_s, info_s = self._refit_chol(K, tr_o, lmax_o, rhs, sol)
            out = torch.empty((Gc, rows, N_o), dtype=torch.float32, device=self.dev)
            one = self.shard.world == 1                # (one rank: flag k belongs to alpha k of the call, fold_speculate)
            info = torch.empty(Gc, dtype=torch.int32, device=self.dev) if one else None
            for src, flags, part in ((M_i, info_i, inv), (M_s, info_s, sol)):
                for k, a in enumerate(part):
                    out[alphas_idx.index(a)].copy_(src[k])
                    if one:
                        info[alphas_idx.index(a):alphas_idx.index(a) + 1].copy_(flags[k:k + 1])
            return out, (info if one else self._join_flags([info_i, info_s]))
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
        for alpha >= FitOptions.refit_inverse_min_alpha (0.1: measured 2.5e-5 of max|W| at alpha = 0.1, 1.7e-7 at 0.68), on the
        fp16x3 path, with normalpha (S[0] known); the solves otherwise.  Asked alpha by alpha: [a]."""
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
            if ahead == have:
                Pa = spec["P"]
            else:                                       # (device-to-device copies into one block: no framework stack kernel)
                Pa = torch.empty((len(ahead),) + tuple(spec["P"].shape[1:]), dtype=spec["P"].dtype, device=self.dev)
                for i, a in enumerate(ahead):
                    Pa[i].copy_(spec["P"][have.index(a)])
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
            rows_x = ops.idx_tensor(np.asarray(extra_rows, dtype=np.int64), n_x, self.dev) if n_x > 0 else None
            Ys_te, te_src = None, None
            if n_x > 0 and (n_x <= 640 or self.opt.refit_fused_pearson):   # Pearson r reads the test rows through (rows, perm) in place
                te_src = (Y, rows_x, perm)
            elif n_x > 0:
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

    def _refit_images(self, o, r0, r1):
        """(tiled fp16 image, row scales) of rows [r0, r1) of every alpha group's operator, groups stacked tile-aligned.
        The operators' images are the same for every voxel range of the fold whose voxels chose the same alphas: kept in
        the fold's cache."""
        Malpha, N_o, Kc = o["Malpha"], o["N_o"], o["K"]        # contraction depth Kc: the training rows (the operators'
        G, rows = len(Malpha), r1 - r0                         # padding columns beyond them are zero)
        rows_pad = ops.pad_to(rows, 256)
        key = (o.get("used"), r0, r1, Kc)
        cache = o.get("img_cache")
        if cache is not None and key in cache:
            return cache[key]
        At = torch.empty(G * rows_pad * N_o * 2, dtype=torch.float16, device=self.dev)
        rs_inv = torch.empty(G * rows_pad, dtype=torch.float32, device=self.dev)
        for g in range(G):
            ops.split_rows_f16(Malpha[g][r0:r1], rows, Kc, At[g * rows_pad * Kc * 2:], rs_inv[g * rows_pad:])
        if cache is not None and o.get("used") is not None:
            cache[key] = (At, rs_inv)
        return At, rs_inv

    def _refit_product(self, o, r0, r1, useful_rows, out=None):
        """Rows [r0, r1) of  C = [M_alpha ; H_te,alpha](group) . Ys  as an (r1 - r0, Vs) f32 matrix (fp16x3 path: the
        rows of every group split to fp16 triples, one grouped launch).  The caller takes the test predictions first
        -- what the host statistics wait for -- and the weight rows afterwards.  ``out``: a (r1 - r0, Vs) view (any row
        stride) the product is written to."""
        Malpha, Vs, N_o = o["Malpha"], o["Vs"], o["N_o"]          # one (rows, N_o) operator per alpha group
        G, rows = len(Malpha), r1 - r0
        C = out if out is not None else torch.empty((rows, Vs), dtype=torch.float32, device=self.dev)
        if o["split"]:
            At, rs_inv = self._refit_images(o, r0, r1)
            ops.gemm_grouped_f16x3(At, rs_inv, rows, o["Yt"], o["cs_s"][1], C, C.stride(0), Vs, o["K"], o["tiles"])
            self.info["plain_flops"] += 2.0 * useful_rows * o["n_o"] * self.V
            self.info["plain_launches"] += 1
        else:
            Ms = torch.empty((G, rows, N_o), dtype=torch.float32, device=self.dev)
            for g in range(G):
                Ms[g].copy_(Malpha[g][r0:r1])                       # (D2D copies: the groups' operators side by side)
            ops.gemm_grouped(Ms, N_o, Ms.stride(0), o["Ys"], Vs, None, C, C.stride(0), rows, Vs, N_o, o["tiles"])
        return C

    def _refit_pearson(self, o, r0, n_t):
        """Pearson r of the test rows (alpha-sorted voxel order) straight from the contraction: rows [r0, r0 + n_t) of the
        operators applied to the sorted targets and reduced against the test targets in the launch's epilogue
        (lc_gemm_grouped_f16x3_pearson) -- the predictions are never stored and lc_pearson_cols never reads them back
        (nested_cv.py:151-155, 251-257; fp16x3 arithmetic only)."""
        Vs = o["Vs"]
        At, rs_inv = self._refit_images(o, r0, r0 + n_t)
        r_s = torch.empty(Vs, dtype=torch.float64, device=self.dev)
        if o.get("te_src") is not None:
            y, y_rows, y_cols = o["te_src"]
        else:
            y, y_rows, y_cols = o["Ys_te"], None, None
        ops.gemm_grouped_f16x3_pearson(At, rs_inv, n_t, o["Yt"], o["cs_s"][1], Vs, o["K"], o["tiles"], y, y_rows, y_cols, r_s)
        self.info["plain_flops"] += 2.0 * n_t * o["n_o"] * self.V
        self.info["plain_launches"] += 1
        return r_s

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
        return (bool(self.cho) and len(self.cho) <= 8 and not self.primal and self.speculation_pays()
                and bool(self._ahead_alphas(self.cho)))

    def _ahead_alphas(self, alphas_idx):
        """Those of the listed alphas whose refit operators are formed BEFORE anybody has chosen them: explicit inverses
        (cheap) of alphas that are likely to be used (FitOptions.refit_ahead_min_alpha)."""
        return [a for a in alphas_idx if self._refit_by_inverse([a]) and self.alphas[a] >= self.opt.refit_ahead_min_alpha]

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
            inv = self._ahead_alphas(cho)
            if inv:
                # (the inverses of the alphas that qualify, ahead; the others when somebody has chosen them -- each alpha by
                # its own route: _refit_chol)
                cho = inv
            else:
                # no alpha qualifies for an inverse AHEAD of the choice: ahead by solves only those whose route IS the solves
                # (a grid like 0.05 / 0.12 -- nothing reaches refit_ahead_min_alpha, 0.12 is above
                # refit_inverse_min_alpha -- used to send 0.12 through the row-sliced solves here and through the explicit
                # inverse on one GPU: last-bit differences between a sharded and an unsharded fit); an inverse-route alpha
                # below the ahead threshold is built when somebody has chosen it, as on one GPU (_refit_chol)
                cho = [a for a in cho if not self._refit_by_inverse([a])]
                if not cho:
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
        if early:
            # the first fold's systems for EVERY factorised alpha, before anybody has chosen: those that come from explicit
            # inverses (N^3 flops each) and are likely to be used (refit_ahead_min_alpha); the others when somebody has
            # chosen them (fold_select)
            inv = self._ahead_alphas(todo)
            todo = inv if inv else todo
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
