"""The mean weights of a cross-validated fit from the MEAN of the folds' refit operators (round 6).

nested_cv.py:293-296 returns ``np.mean(fold_weights, axis=0)`` and nothing else of the folds' weight matrices, and a fold's
weights are linear in its training targets:  W_f[:, v] = M_f(alpha_f(v)) Y[tr_f, v]  (ridge_regression.py:46-61).  For the
voxels that chose the same alpha TUPLE (alpha_0, .., alpha_{F-1}) over the folds,

    mean_f W_f[:, v] = C Y[:, v],    C = 1/F sum_f M_f(alpha_f) scattered to the columns of its training rows   (p x T)

-- ONE contraction of depth T per tuple group instead of one of depth n_train per fold: 5 x 2400 -> 3000 rows of depth at
cfg2, where the folds' alphas of a voxel come from two neighbouring grid values (32 tuples, 40 with the rare ones).  The test
predictions of every fold (Pearson r, p) are per fold by definition and keep their own contraction (FoldPhases.fold_finish).
Tuples that a handful of voxels share (flat score curves: every fold another alpha) do not pay for an operator image of
their own: their voxels take the folds' own weight products, restricted to them -- bit for bit what a fold-by-fold refit
gives them.
"""
import numpy as np
import torch

from .. import ops
from .common import logger


class MeanOperatorRefit:
    """Mixin of RidgeCVEngine: weights of a voxel range once its last fold has chosen."""

    def _mo_possible(self):
        """Static conditions (known before any alpha is chosen): a cross-validated fit in the dual form on the fp16x3 path whose
        targets are the resident ones in every fold."""
        o = self.opt
        Kd = ops.pad_to(max(self.Ttot, 1), 32)
        return bool(o.mean_operator_refit and self.n_folds >= 2 and self.n_folds <= 16 and not self.moments and not self.primal
                    and not self.spectral and not self.norm_y and self.precision != "f32" and 64 <= Kd <= 8192
                    and self.V_rank >= o.mean_operator_min_cols)      # (this rank's voxels: a narrow voxel shard keeps the folds' own products,
                                                                      #  formed fold by fold beside the next fold's sweeps)

    def _mo_enabled(self, st):
        """Decided at the first refit of the fit (the arithmetic is known by then) and kept for all of it."""
        if getattr(self, "_mo", None) is None:
            ok = self._mo_possible() and bool(st["split"]) and st.get("best_h") is not None
            self._mo = dict(folds={}, images={}, maps={}, done=0, pending=[], decided=self.n_folds <= 2) if ok else False
            self.info["mean_operator"] = dict(on=bool(ok), ranges=0, tuples=0, tiles=0, built=0, voxels=0, other_voxels=0)
        return bool(self._mo)

    def _mo_record(self, st, weight_scale, side_job=None, o=None, perm=None, Vs=None):
        """A (fold, range) step whose weight rows are NOT formed now: the alphas its voxels chose, the fold's operators (one
        dictionary per fold, shared by its ranges: it fills as alphas are first used), the natural-order image of the range."""
        f, rg = int(st["fold"]), st["rg"]
        ent = self._mo["folds"].get(f)
        if ent is None:
            base = st.get("base", st)
            cache = base.setdefault("refit_cache", {})
            ent = self._mo["folds"][f] = dict(best=np.full(max(self.V_rank, 1), -1, dtype=np.int64), M=cache.setdefault("M", {}),
                                              imgs=cache.setdefault("imgs", {}), tr=np.asarray(st["tr"], dtype=np.int64),
                                              scale=float(weight_scale), images={}, side=[])
        if st.get("best_h") is None or not st["split"]:
            raise RuntimeError("mean-operator refit: a step without host alpha indices / on the f32 path in a fit set up for it")
        ent["best"][rg.c0:rg.c0 + rg.V] = st["best_h"][: rg.V]
        # (the alphas the step's voxels chose, from its histogram: the tuple keys' digits without a pass over the indices)
        if st.get("used") and ent.get("used") is not False:
            ent.setdefault("used", set()).update(int(a) for a in st["used"])
        else:
            ent["used"] = False
        ent["images"][rg.key] = st["hat"].get("image")
        if side_job is not None:                       # the step's exact-f32 weights of the side panel's columns (_side_refit_begin)
            ent["side"].append((side_job, rg.c0))
        if not self._mo["decided"]:
            # until the first two folds have chosen nobody knows whether the voxels' alphas repeat over the folds: the step's
            # operands are kept, so that its own weight product can still be formed (_mo_decide)
            self._mo["pending"].append(dict(fold=f, rg=rg, o=o, perm=perm, Vs=int(Vs), scale=float(weight_scale), side_job=side_job))
            self._mo_decide()

    def _mo_decide(self):
        """After the first two folds' choices: does the mean-operator refit pay on THESE data?  The (alpha_0, alpha_1) pairs of
        the rank's voxels stand in for the tuples over all folds -- a pair shared by c voxels of V is expected to split into
        tuples of ~c (c / V)^((F - 2) / 2) voxels; the voxels of pairs whose tuples would reach the size at which an operator
        pays (mean_operator_cost_ratio) must make up mean_operator_min_share of all.  Otherwise the option is dropped for the
        rest of the fit: the kept steps' own weight products are formed now, the later folds' as they always were."""
        mo, A = self._mo, self.A
        ents = [mo["folds"].get(f) for f in (0, 1)]
        if any(e is None or (e["best"][: self.V_rank] < 0).any() for e in ents):
            return                                     # (a fold still has voxel ranges to come)
        V = self.V_rank
        b0, b1 = ents[0]["best"][:V], ents[1]["best"][:V]
        c2 = np.bincount(b0 * A + b1, minlength=A * A).astype(np.float64)
        c2 = c2[c2 > 0]
        size = c2 * (c2 / V) ** ((self.n_folds - 2) / 2.0)
        n_o = float(len(ents[0]["tr"])) * self.n_folds
        Kd = ops.pad_to(self.Ttot, 32)
        build = 0.85 * (self.n_folds * ops.pad_to(len(ents[0]["tr"]), 64) + Kd)
        cnt_min = 256.0 * (Kd + build) / max(self.opt.mean_operator_cost_ratio * n_o, 1.0)
        share = float(c2[size >= cnt_min].sum() / max(V, 1))
        self.info["mean_operator"]["expected_share"] = round(share, 4)
        mo["decided"] = True
        pending, mo["pending"] = mo["pending"], []
        if share >= self.opt.mean_operator_min_share:
            return                                     # on: the kept operands are dropped
        logger.info("mean-operator refit: %.0f %% of the voxels expected in alpha tuples that pay: the folds' own weight products",
                    100.0 * share)
        self.info["mean_operator"]["on"] = False
        self._mo = False
        for s in pending:                              # what fold_finish does without the option, for the steps gone by
            ent, off = self._ws_slot(s["fold"], s["rg"], s["Vs"], s["scale"])
            ops.invert_perm(s["perm"], s["Vs"], off, ent["pos"][s["rg"].c0:])
            prev, self.cur = self.cur, s["rg"]
            try:
                self._refit_product(s["o"], 0, self.p_pad, self.p, out=ent["buf"][:, off:off + s["Vs"]])
                if s["side_job"] is not None:
                    W_s, dst, n_s = self._side_refit_end(s["side_job"], ent, off)
                    ops.scatter_cols(W_s, self.p_pad, dst, n_s, ent["buf"][:, off:off + s["Vs"]])
            finally:
                self.cur = prev

    def _mo_map(self, fold, tr_rows, Kd):
        """(Kd,) int32 device: column of fold ``fold``'s operators that belongs to target row t (-1: not a training row)."""
        m = self._mo["maps"].get(fold)
        if m is None:
            h = np.full(Kd, -1, dtype=np.int32)
            h[tr_rows] = np.arange(len(tr_rows), dtype=np.int32)
            m = self._mo["maps"][fold] = ops.upload(h, self.dev)
        return m

    def _mo_side_columns(self, rg, ents):
        """Too-wide target columns (the f32 side panel): their weights of every fold came from the side refit in exact f32
        (FoldPhases._side_refit_begin); the range's mean over them -- zero, then scale_f W_f added fold by fold, the sums a
        fold-by-fold refit forms -- goes over whatever the fp16 path left in those columns."""
        jobs = [(e, job, c0) for e in ents for job, c0 in e["side"]]
        if not jobs:
            return
        main = torch.cuda.current_stream()
        cols = np.asarray(self.side["cols"], dtype=np.int64) if self.side is not None else np.zeros(0, dtype=np.int64)
        sel = cols[(cols >= rg.c0) & (cols < rg.c0 + rg.V)] - rg.c0
        if len(sel) == 0:
            return
        ops.scatter_cols(ops.zeros((self.p, len(sel)), torch.float32, self.dev), self.p, ops.upload(sel.astype(np.int32), self.dev),
                         len(sel), rg.W)
        for e, job, c0 in jobs:
            main.wait_event(job["done"])
            perm_s, Vss = job["perm"], job["Vss"]
            nat = job["d_local"] + (c0 - rg.c0)        # columns of THIS range (a step may have worked on another range)
            dst = torch.where(perm_s >= 0, nat[perm_s.clamp(min=0).long()], torch.full_like(perm_s, -1))
            dst = torch.where((dst >= 0) & (dst < rg.V), dst, torch.full_like(dst, -1)).to(torch.int32).contiguous()
            ops.scatter_axpy(job["C"], self.p, dst, Vss, e["scale"], rg.W)
            for t in (job["C"], perm_s, job["d_local"]):
                t.record_stream(main)

    @staticmethod
    def _alpha_tuples(best, A, used=None):
        """The voxels of a range grouped by the alpha TUPLE they chose over the folds.  ``best``: per fold an integer vector of
        alpha indices (0 <= index < A), all of the same length V.  Returns (order (V,) -- the voxels tuple after tuple, tuples in
        ascending order of their mixed-radix key (digit f = rank of the alpha among those fold f uses, fold 0 least significant),
        voxels ascending inside a tuple --, counts per tuple, the tuples as alpha-index tuples), or None when the key space
        overflows 62 bits.  Host side, numpy: a radix sort of 16-bit keys for the usual few alphas per fold."""
        best = [np.asarray(b) for b in best]
        # ``used``: per fold the alphas that occur (or a superset: digits nobody has leave empty keys), sorted -- known from the
        # folds' histograms; else counted here (0.2 ms per fold at cfg2, on the fit's critical path)
        given = used is not None
        used = ([np.asarray(sorted(u), dtype=np.int64) for u in used] if given
                else [np.nonzero(np.bincount(b, minlength=A))[0] for b in best])
        radix = [max(len(u), 1) for u in used]
        if given and int(np.prod([float(r) for r in radix])) > 65535:
            return MeanOperatorRefit._alpha_tuples(best, A)       # (a superset may cost the 16-bit keys: count, then)
        stride = 1
        for r in radix:
            if stride * r > (1 << 62):
                return None
            stride *= r
        # (the host forms the tuples at the fit's tail, with nothing left to overlap them: one table look-up and one add per
        # fold, in the narrowest key type -- 0.6 ms at cfg2 where the straightforward int64 version took 1.7)
        kt = np.uint32 if stride <= 65535 else np.int64
        key, st = None, 1
        for b, u, r in zip(best, used, radix):
            # (alphas that a GIVEN list does not name: a bit above every key -- caught below, never grouped silently)
            table = np.full(A, (1 << 30) if given else 0, dtype=kt)
            table[u] = (np.arange(len(u), dtype=np.int64) * st).astype(kt)          # digit x stride, per alpha
            key = table[b] if key is None else key + table[b]
            st *= r
        if given and len(key) and int(key.max()) >= (1 << 30):
            raise RuntimeError("mean-operator refit: a voxel chose an alpha its fold's histogram does not list")
        if stride <= 65535:
            key = key.astype(np.uint16)
            order = np.argsort(key, kind="stable")           # (numpy radix-sorts 16-bit keys)
            counts = np.bincount(key, minlength=stride)
            live = np.nonzero(counts)[0]
            cnt = counts[live]
        else:
            order = np.argsort(key, kind="stable")
            live, cnt = np.unique(key, return_counts=True)
        q = np.asarray(live, dtype=np.int64)
        cols = []
        for u, r in zip(used, radix):
            cols.append(u[q % r] if len(u) else np.zeros(len(q), dtype=np.int64))
            q = q // r
        tuples = [tuple(row) for row in np.stack(cols, axis=1).tolist()] if len(live) else []
        return order, np.asarray(cnt, dtype=np.int64), tuples

    @staticmethod
    def _padded_groups(order, cnt):
        """Column list of voxels sorted into groups (``order``: the voxels group after group, ``cnt``: the groups' sizes), every
        group starting on a 256-column tile: (perm (Vs,) int32 with -1 padding, tile starts (G + 1,))."""
        cnt = np.asarray(cnt, dtype=np.int64)
        tiles_g = (cnt + 255) // 256
        start = np.concatenate([[0], np.cumsum(tiles_g)]).astype(np.int64)
        perm = np.full(int(start[-1]) * 256, -1, dtype=np.int32)
        if len(order):
            first = np.concatenate([[0], np.cumsum(cnt)])[:-1]
            # (position of voxel i of the sorted list: i + what the groups before its own were padded by)
            perm[np.arange(len(order)) + np.repeat(start[:-1] * 256 - first, cnt)] = np.asarray(order, dtype=np.int32)
        return perm, start

    def _mean_operator_weights(self, rg):
        """W of the range once its last fold has chosen: one grouped contraction of depth T over the voxels sorted by alpha
        tuple (module docstring) for the tuples that pay for their operator image, the folds' own weight products for the
        voxels of the others.  The tuples are formed on the host from the folds' alpha indices (they arrive with each fold's
        histogram, fold_choose); a fold may have been worked through in other voxel ranges than this one (upload panels at the
        start of a host-to-host fit, download panels at its end): only its alphas and its operators are needed here."""
        mo, info = self._mo, self.info["mean_operator"]
        folds = sorted(mo["folds"])
        ents = [mo["folds"][f] for f in folds]
        V, A, c0 = rg.V, self.A, rg.c0
        best = [e["best"][c0:c0 + V] for e in ents]
        if len(folds) != self.n_folds or any((b < 0).any() or (b >= A).any() for b in best):
            raise RuntimeError("mean-operator refit: a voxel range finished before every fold had chosen its alphas")
        Kd = ops.pad_to(self.Ttot, 32)
        cs, _split = self._target_scales(rg.Y, rg)
        # the natural-order image of ALL target rows of the range first: GPU work that does not depend on the tuples, queued
        # before the host forms them (0.45 ms at cfg2 beside ~1.5 ms of numpy; unused when no tuple pays: flat score curves)
        Vt = ops.pad_to(rg.Vp, 256)
        rows_all = ops.idx_tensor(np.arange(self.Ttot), Kd, self.dev)
        Yu = torch.empty(Vt * Kd * 2, dtype=torch.float16, device=self.dev)
        ops.split_cols_f16(rg.Y, rg.Vp, rows_all, Kd, cs, Yu)
        n_o = [len(e["tr"]) for e in ents]
        scales = {e["scale"] for e in ents}
        pays = None
        known = [e.get("used") for e in ents]
        grouped = (self._alpha_tuples(best, A, used=known if all(known) else None)) if len(scales) == 1 else None
        if grouped is not None:
            order, cnt, tuples = grouped
            # which tuples pay.  Costs in (column tile x depth row) units of the grouped contraction (3.7 ns at cfg2); one
            # operator image costs ~12 900 of them there (46 us: the folds' operators read, the image written), a cached one
            # a quarter (a device copy); the folds' own products cost a tuple's voxels their share of sum_f n_train depth rows
            build = 0.85 * (sum(ops.pad_to(n, 64) for n in n_o) + Kd)
            cached = np.asarray([t in mo["images"] for t in tuples], dtype=bool)
            cost_new = ((cnt + 255) // 256) * Kd + np.where(cached, 0.25, 1.0) * build
            cost_old = cnt / 256.0 * float(sum(n_o))
            pays = cost_new <= self.opt.mean_operator_cost_ratio * cost_old
            if pays.any() and not pays.all():
                # the folds' own products for the voxels of the other tuples are one small launch per fold whatever their
                # number (~20 tile-rows of latency per depth row: 170 us at cfg2): a handful of rare tuples is cheaper served
                # by operators of their own
                rest_as_folds = 20.0 * float(sum(n_o)) + float(cost_old[~pays].sum())
                if float(cost_new[~pays].sum()) <= rest_as_folds:
                    pays[:] = True
            if pays.sum() > self.opt.mean_operator_max_tuples:
                keep = np.argsort(-cnt, kind="stable")[: self.opt.mean_operator_max_tuples]
                mask = np.zeros(len(cnt), dtype=bool)
                mask[keep] = True
                pays &= mask
        parts = []
        rest = np.arange(V)
        if pays is not None and pays.any():
            sel_sorted = np.repeat(pays, cnt)          # per voxel in ``order``
            sel_groups = np.nonzero(pays)[0]
            G = len(sel_groups)
            perm_h, start = self._padded_groups(order[sel_sorted], cnt[sel_groups])
            rest = np.sort(order[~sel_sorted])
            Vs = len(perm_h)
            perm = ops.upload(perm_h, self.dev)
            # operands: the columns of the natural-order image gathered tuple by tuple
            Yt = torch.empty(Vs * Kd * 2, dtype=torch.float16, device=self.dev)
            ops.permute_cols_f16(Yu, perm, Vs, Kd, Yt)
            cs_s = torch.empty((2, Vs), dtype=torch.float32, device=self.dev)
            ops.gather(cs.reshape(2, rg.Vp), rg.Vp, None, 2, perm, Vs, cs_s)
            # ... and the groups' mean-operator images
            rows = self.p_pad
            rows_pad = ops.pad_to(rows, 256)
            At = torch.empty(G * rows_pad * Kd * 2, dtype=torch.float16, device=self.dev)
            rs_inv = torch.empty(G * rows_pad, dtype=torch.float32, device=self.dev)
            maps = [self._mo_map(f, e["tr"], Kd) for f, e in zip(folds, ents)]
            new_mats, new_slots = [], []
            for g, gq in enumerate(sel_groups):
                tup = tuples[gq]
                a_g, r_g = At[g * rows_pad * Kd * 2:(g + 1) * rows_pad * Kd * 2], rs_inv[g * rows_pad:(g + 1) * rows_pad]
                hit = mo["images"].get(tup)
                if hit is not None:
                    a_g.copy_(hit[0])
                    r_g.copy_(hit[1])
                    continue
                new_mats.append([e["M"][a][:rows] for e, a in zip(ents, tup)])
                new_slots.append(g)
                mo["images"][tup] = (a_g, r_g)
            # (all new images in one launch: one per tuple was ~45 us of host time each, 1.8 ms at the tail of a cfg2 fit)
            strides = {tuple(int(x.stride(0)) for x in ms) for ms in new_mats}
            if len(strides) <= 1:
                ops.mean_operator_images(new_mats, new_slots, maps, ents[0]["scale"], rows, Kd, At, rs_inv)
            else:                                      # (a fold's operators with different row strides: image by image)
                for ms, g in zip(new_mats, new_slots):
                    ops.mean_operator_image(ms, maps, ents[0]["scale"], rows, Kd, At[g * rows_pad * Kd * 2:(g + 1) * rows_pad * Kd * 2],
                                            rs_inv[g * rows_pad:(g + 1) * rows_pad])
            new_imgs = len(new_slots)
            C = torch.empty((rows, Vs), dtype=torch.float32, device=self.dev)
            ops.gemm_grouped_f16x3(At, rs_inv, rows, Yt, cs_s[1], C, Vs, Vs, Kd, [int(t) for t in start])
            n_comb = int(sel_sorted.sum())
            self.info["plain_flops"] += 2.0 * self.p * self.Ttot * n_comb
            self.info["plain_launches"] += 1
            pos = ops.filled((max(V, 1),), torch.int32, self.dev, 0xFF)
            ops.invert_perm(perm, Vs, 0, pos)
            parts.append((C, pos, 1.0))
            info["tuples"] += G
            info["tiles"] += int(start[-1])
            info["built"] += new_imgs
            info["voxels"] += n_comb
        del Yu
        # ---- the other voxels: the folds' own products, restricted to them (the operands and the arithmetic a fold-by-fold
        # refit gives them: alpha groups in ascending order, voxels in ascending order inside)
        if len(rest):
            for f, e, b in zip(folds, ents, best):
                a_r = b[rest]
                o_r = np.argsort(a_r.astype(np.uint16), kind="stable")
                c_r = np.bincount(a_r, minlength=A)
                used_r = [int(a) for a in np.nonzero(c_r)[0]]
                perm_r, start_r = self._padded_groups(rest[o_r], c_r[used_r])
                Vs_r = len(perm_r)
                d_perm_r = ops.upload(perm_r, self.dev)
                o = self._refit_operands(rg.Y, e["tr"], (), d_perm_r, [int(t) for t in start_r], Vs_r,
                                         [e["M"][a] for a in used_r], True, cs, image=e["images"].get(rg.key))
                o.update(used=tuple(used_r), img_cache=e["imgs"])
                buf = torch.empty((self.p_pad, Vs_r), dtype=torch.float32, device=self.dev)
                self._refit_product(o, 0, self.p_pad, self.p, out=buf)
                pos_r = ops.filled((max(V, 1),), torch.int32, self.dev, 0xFF)
                ops.invert_perm(d_perm_r, Vs_r, 0, pos_r)
                parts.append((buf, pos_r, e["scale"]))
            info["other_voxels"] += int(len(rest))
        ops.combine_folds(parts, self.p, V, rg.W)      # natural voxel order: every voxel has exactly one source
        self._mo_side_columns(rg, ents)
        info["ranges"] += 1
        mo["done"] += V
        if mo["done"] >= self.V_rank:                  # every range of the rank is final: nothing of the fit is kept
            mo["folds"].clear()
            mo["images"].clear()
            mo["maps"].clear()
            mo["done"] = 0
