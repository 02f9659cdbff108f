"""Voxel sharding across the GPUs of a node: one process per GPU, torch.distributed.

The fit is independent per voxel (column of ``targets``) given the design-matrix algebra
(``encoding/models/ridge_regression.py:104-125``, ``nested_cv.py:396-411``), so each rank fits a contiguous block
of voxel columns and the V-wide data path needs no collective.  What the ranks exchange:

  * the V-INDEPENDENT fp64 operators.  Every (fold, alpha) Cholesky system -- hat matrices of the inner folds,
    weight/prediction operators of the refit -- is needed by every rank but depends on no voxel: the systems are dealt
    out over the ranks (``RidgeCVEngine._sharded_solve``), each rank solves its share, and the f32 results are
    all-gathered (<= 74 MB per outer fold for the hat matrices at cfg2; one direct hop over xGMI).  Without this
    every rank repeats ~60 ms of fp64 work per fit and 8 GPUs cannot be more than ~2x faster than one;
  * ``single_alpha=True`` (nested_cv.py:396-400): the per-alpha score sums over voxels, all-reduced on the device
    (A doubles, once per outer fold); the histogram of the chosen alphas (A ints) the same way, so that all ranks
    solve the same refit systems;
  * one packed (4, V_local) block of per-voxel results per fold (r, p, alpha index, pivot flags;
    ``lc_fold_pack`` / ``lc_fold_unpack``), all-gathered so that the global statistics (BH-FDR ranks ALL p-values)
    see every voxel.

All of these take and return DEVICE tensors.  Backend "nccl" is RCCL over xGMI on ROCm and moves them directly;
under "gloo" (the CPU tests, and the two-ranks-on-one-GPU parity test) the same calls stage through the host.

Transport.  With the "nccl" backend the device-tensor collectives do not go through torch.distributed at all but through
the library's own thin RCCL wrappers (``lc_allgather_f32`` / ``lc_allgather_bytes`` / ``lc_allreduce``,
include/litcoder_hip.h; SURVEY 8b's export list) on communicators created from a unique id that rank 0 makes and
torch.distributed broadcasts ONCE -- PyTorch is a container and a rendezvous, nothing on the data path.  Round 5 built
this as an option; since round 6 it is the DEFAULT under "nccl" (``direct_rccl=False`` / LITCODER_AMD_RCCL_DIRECT=0 goes
back to torch.distributed's calls; under gloo -- the CPU tests, two ranks on one GPU -- torch.distributed is the only
transport).  Two knobs give a first multi-GPU run a conservative mode without a code change:

  * LITCODER_AMD_RCCL_LANES=0 -- every collective on ONE communicator (no "hat" / "refit" lanes of their own for the bulk
    all-gathers): one issue order for everything, at the price of a 20-int all-reduce queueing behind a batch gather;
  * LITCODER_AMD_RCCL_DIRECT=0 -- torch.distributed's transport (its own communicators and streams).

Ordering rule of the direct transport: a communicator executes its collectives in ISSUE ORDER; every rank issues the same
collectives in the same host order by construction (job lists, panel counts and flags are identical on all ranks), whatever
HIP stream each is ordered on.  The default communicator carries the small exchanges (flags, per-alpha sums, histograms,
the packed fold results) from the main / scales / communication streams; the lanes carry the operator gathers.  This pool
has no multi-GPU node: the one-rank RCCL test (tests/test_gpu_shards.py) is the hardware evidence there is.
``ShardContext.simulated`` runs one rank of a W-rank job alone (collectives become local copies): the results are
meaningless, the per-rank timeline is what an W-GPU run would see minus the wire time -- used by
``tools/scaling_model.py`` on the single-GPU box.
"""
from typing import Tuple

import numpy as np



def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``; the first ``n % world`` ranks get one more."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def job_share(n_jobs: int, world: int, rank: int) -> Tuple[int, range]:
    """How ``n_jobs`` independent systems are dealt out: contiguous chunks of n_per = ceil(n_jobs / world); returns
    (n_per, the jobs of ``rank``).  Slot j of the all-gathered result (world * n_per slots) is job j."""
    n_per = -(-int(n_jobs) // int(world))
    return n_per, range(min(n_jobs, rank * n_per), min(n_jobs, (rank + 1) * n_per))


_LANES = {}     # (group ranks, backend) -> {"hat": group, "refit": group}: communicators are created once per process


class ShardContext:
    """Wraps a torch.distributed process group (or nothing, for a single process).  Constructing a context whose
    collectives are active is itself COLLECTIVE over the whole default group the first time a (group, backend) pair is
    seen (torch.distributed.new_group must be entered by every process, members or not): build it on every rank, in
    the same order."""

    def __init__(self, group=None, device=None, always_collective=False, global_lists=True, direct_rccl=None):
        """``always_collective``: issue the backend's collectives even in a one-rank group (they are no-ops
        arithmetically) -- lets a single GPU exercise the RCCL calls of the sharded path.
        ``global_lists``: the per-voxel containers of the metrics dictionary (correlations, p-values, masks, alphas)
        cover ALL voxels on every rank (True, default) or only the rank's own block (False).  The scalar summaries and
        the statistics behind them (BH-FDR ranks all p-values) are global either way; the lists are V_total Python
        objects each -- ~70 ms of interpreter time per rank at 8 x 80 000 voxels, on every rank -- so a job that only
        needs each rank's own voxels (like its block of the weights) turns them off."""
        import os
        import torch.distributed as dist
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self._comms = {}                               # lane -> lc_comm_t* of the direct RCCL path (None: torch.distributed)
        self.group = group
        self.device = device
        self.rank = self._dist.get_rank(group) if self._dist else 0
        self.world = self._dist.get_world_size(group) if self._dist else 1
        # the transport per device type: "nccl" (RCCL) moves device tensors where they live, "gloo" host tensors; a
        # default group initialised without a backend reports e.g. "cpu:gloo,cuda:nccl" and moves both directly
        raw = str(self._dist.get_backend(group)).lower() if self._dist else ""
        self._cuda_direct = "nccl" in raw
        self._cpu_direct = "gloo" in raw or "mpi" in raw
        self.backend = ("nccl" if raw == "nccl" else "gloo" if raw == "gloo" else raw) if self._dist else None
        self.simulate = False
        self.bytes_received = 0                        # payload of the all-gathers so far (what the wire carries INTO this rank)
        self.global_lists = bool(global_lists)
        self.always = bool(always_collective) and self._dist is not None
        # Collectives of one communicator execute in issue order on its one internal stream.  The operator all-gathers
        # are issued far ahead (whole batches of later folds) and wait for their fp64 chains; a 20-int all-reduce that
        # fold 0 needs NOW must not sit behind them -- so the bulk traffic gets communicators ("lanes") of its own.
        self._lanes = {}
        self.use_lanes = True
        if self._dist is not None and (self.world > 1 or self.always):
            ranks = tuple(self._dist.get_process_group_ranks(group)) if group is not None else None
            key = (ranks, raw)
            # the lanes live as long as the process-group WORLD they were made in: after destroy_process_group() +
            # init_process_group() (test sessions, notebooks, repeated jobs) the cached communicators belong to a dead
            # world and a collective on them fails or hangs -- the entry remembers its world object (held
            # strongly, so its identity cannot be recycled) and entries of other worlds are dropped
            world_now = self._dist.group.WORLD
            for k in [k for k, (w, _) in _LANES.items() if w is not world_now]:
                del _LANES[k]
            self.use_lanes = os.environ.get("LITCODER_AMD_RCCL_LANES", "1") != "0"
            direct_wanted = (direct_rccl if direct_rccl is not None
                             else os.environ.get("LITCODER_AMD_RCCL_DIRECT", "1") != "0") and self._cuda_direct
            # torch.distributed lane groups only where its transport carries the bulk traffic (gloo, or nccl with the
            # direct transport switched off): the direct transport makes its own communicators (_init_direct_rccl)
            if self.use_lanes and not direct_wanted:
                if key not in _LANES:                   # the lanes span exactly the ranks of ``group``
                    be = None if raw in ("", "undefined") else raw
                    _LANES[key] = (world_now, {lane: self._dist.new_group(ranks=list(ranks) if ranks is not None else None,
                                                                           backend=be)
                                               for lane in ("hat", "refit")})  # same order on every process
                self._lanes = _LANES[key][1]
        if direct_rccl is None:
            direct_rccl = os.environ.get("LITCODER_AMD_RCCL_DIRECT", "1") != "0"
        if direct_rccl and self._dist is not None and self._cuda_direct and (self.world > 1 or self.always):
            self._init_direct_rccl()

    def close(self):
        """Destroys the direct transport's communicators (lc_comm_destroy).  The context falls back to torch.distributed's
        calls afterwards.  Collective in spirit: call it on every rank once the last fit is done (nothing
        released them before)."""
        comms, self._comms = self._comms, {}
        if comms:
            from . import _lib
            for h in {id(h): h for h in comms.values()}.values():
                try:
                    _lib.call("lc_comm_destroy", h)
                except Exception:  # noqa: BLE001 -- tearing down: the process group may be gone already
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _init_direct_rccl(self):
        """One RCCL communicator per lane through the library's own wrappers (lc_comm_create): rank 0 makes the unique ids,
        one object broadcast hands them round (collective over the group: every rank constructs its context alike)."""
        import ctypes
        import torch
        from . import _lib
        lib = _lib.load()
        n = lib.lc_comm_unique_id_bytes()
        lanes = (None, "hat", "refit") if self.use_lanes else (None,)
        ids = [None] * len(lanes)
        if self.rank == 0:
            for i in range(len(lanes)):
                buf = ctypes.create_string_buffer(n)
                _lib.call("lc_comm_unique_id", buf, n)
                ids[i] = bytes(buf.raw)
        src = self._dist.get_global_rank(self.group, 0) if self.group is not None else 0
        self._dist.broadcast_object_list(ids, src=src, group=self.group)
        dev = self.device.index if (self.device is not None and self.device.index is not None) else torch.cuda.current_device()
        for lane, raw in zip(lanes, ids):
            h = ctypes.c_void_p()
            _lib.call("lc_comm_create", ctypes.c_char_p(raw), n, self.world, self.rank, int(dev), ctypes.byref(h))
            self._comms[lane] = h

    @classmethod
    def single(cls, device=None):
        """A one-rank context that ignores any initialised process group (what a model without ``shard=`` runs in)."""
        ctx = cls.__new__(cls)
        ctx._dist, ctx.group, ctx.device, ctx.rank, ctx.world, ctx.backend = None, None, device, 0, 1, None
        ctx.simulate, ctx.always, ctx._lanes, ctx.global_lists, ctx.bytes_received = False, False, {}, True, 0
        ctx._comms = {}
        ctx.use_lanes = True
        ctx._cuda_direct, ctx._cpu_direct = False, False
        return ctx

    @classmethod
    def simulated(cls, world: int, rank: int, device=None, global_lists=True):
        """One rank of a ``world``-rank job without peers: every collective is a local copy of this rank's own
        contribution into all slots.  Timing studies only."""
        ctx = cls.single(device=device)
        ctx.rank, ctx.world, ctx.backend, ctx.simulate = int(rank), int(world), "simulated", True
        ctx.global_lists = bool(global_lists)
        return ctx

    @property
    def active(self) -> bool:
        """Collectives are issued (more than one rank, or a one-rank group with ``always_collective``)."""
        return self.world > 1 or self.always

    # ------------------------------------------------------------------ partition
    def bounds(self, n_items: int) -> Tuple[int, int]:
        return shard_bounds(n_items, self.world, self.rank)

    def all_bounds(self, n_items: int) -> np.ndarray:
        """(world + 1,) int64 offsets of every rank's block."""
        return np.asarray([shard_bounds(n_items, self.world, r)[0] for r in range(self.world)] + [int(n_items)],
                          dtype=np.int64)

    # ------------------------------------------------------------------ device-tensor collectives
    def _direct(self, t) -> bool:
        """True when the backend moves ``t`` where it lives (RCCL for device tensors, gloo for host tensors)."""
        return self._cuda_direct if t.is_cuda else self._cpu_direct

    def all_gather(self, t, lane=None):
        """(world, *t.shape): every rank's ``t`` (same shape and dtype everywhere), on ``t``'s device, ordered on the
        current stream.  One rank: a view, no copy.  ``lane``: "hat" / "refit" = the communicator of that bulk traffic
        (default: the group's own, for the small latency-critical exchanges)."""
        import torch
        if not self.active:
            return t.unsqueeze(0)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.bytes_received += (self.world - 1) * t.numel() * t.element_size()
        if self.simulate:
            out.copy_(t.unsqueeze(0).expand_as(out))
            return out
        t = t.contiguous()
        if self._comms and t.is_cuda:
            # the library's own RCCL call, ordered on the current stream (flat: rank-major blocks)
            import ctypes
            from . import _lib
            comm = self._comms.get(lane, self._comms[None])
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if t.dtype == torch.float32:
                _lib.call("lc_allgather_f32", comm, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()), t.numel(), stream)
            else:
                _lib.call("lc_allgather_bytes", comm, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                          t.numel() * t.element_size(), stream)
            return out
        group = self._lanes.get(lane, self.group)
        if self._direct(t):
            self._dist.all_gather_into_tensor(out.view(-1), t.view(-1), group=group)         # flat: rank-major blocks
            return out
        # staging: gloo given device tensors (the tests) or RCCL given host tensors
        h = t.cpu() if t.is_cuda else t.to(self.device)
        parts = torch.empty((self.world,) + tuple(h.shape), dtype=h.dtype, device=h.device)
        self._dist.all_gather_into_tensor(parts.view(-1), h.contiguous().view(-1), group=group)
        out.copy_(parts)
        return out

    def all_reduce_(self, t, op: str = "sum"):
        """In-place element-wise sum / max over ranks of a small device tensor, ordered on the current stream."""
        if not self.active or self.simulate:
            return t
        if self._comms and t.is_cuda and t.is_contiguous():
            import ctypes
            import torch
            from . import _lib
            code = {torch.float32: _lib.LC_F32, torch.float64: _lib.LC_F64, torch.int32: _lib.LC_I32}.get(t.dtype)
            if code is not None:
                _lib.call("lc_allreduce", self._comms[None], ctypes.c_void_p(t.data_ptr()), t.numel(), code, 0 if op == "sum" else 1,
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                return t
        rop = self._dist.ReduceOp.SUM if op == "sum" else self._dist.ReduceOp.MAX
        if self._direct(t):
            self._dist.all_reduce(t, op=rop, group=self.group)
            return t
        h = t.cpu() if t.is_cuda else t.to(self.device)
        self._dist.all_reduce(h, op=rop, group=self.group)
        t.copy_(h)
        return t

    # ------------------------------------------------------------------ host-array helpers (small vectors)
    def allreduce_sum(self, arr: np.ndarray) -> np.ndarray:
        """Element-wise sum over ranks of a small float64 host vector."""
        import torch
        arr = np.asarray(arr, dtype=np.float64)
        if self.world == 1 or self.simulate:
            return arr
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        if not self._cpu_direct:
            t = t.to(self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def total_of_local_blocks(self, n_local: int) -> int:
        """``local_targets``: every rank holds ITS block of the voxel columns; returns the job's total -- and checks, on every
        rank alike (from the same all-reduced vector), that the blocks are the ones ``bounds`` deals out: a rank with
        another width would cut its block into another number of panels and pair its collectives with nobody's (ADVICE r5:
        a hang instead of an error).  One rank of a simulated job stands for ``world`` equal blocks."""
        n_local = int(n_local)
        if self.world == 1:
            return n_local
        if self.simulate:
            return n_local * self.world
        mine = np.zeros(self.world)
        mine[self.rank] = n_local
        every = np.rint(self.allreduce_sum(mine)).astype(np.int64)
        total = int(every.sum())
        want = np.asarray([shard_bounds(total, self.world, r)[1] - shard_bounds(total, self.world, r)[0]
                           for r in range(self.world)], dtype=np.int64)
        if not np.array_equal(every, want):
            raise ValueError(f"local_targets: the ranks hold {every.tolist()} voxel columns; the blocks of a {total}-voxel job over "
                             f"{self.world} ranks are {want.tolist()} (ShardContext.bounds)")
        return total

    def allgather_cols(self, arr: np.ndarray, n_total: int) -> np.ndarray:
        """``arr`` is (k, n_local) for this rank's block; returns (k, n_total) on every rank (host arrays)."""
        import torch
        arr = np.ascontiguousarray(arr)
        if self.world == 1:
            return arr
        k = arr.shape[0]
        lo = self.all_bounds(n_total)
        wmax = int(np.max(np.diff(lo)))
        pad = np.zeros((k, wmax), dtype=arr.dtype)
        pad[:, : arr.shape[1]] = arr
        mine = torch.from_numpy(pad)
        if not self._cpu_direct:
            mine = mine.to(self.device)
        parts = self.all_gather(mine).cpu().numpy()
        out = np.empty((k, n_total), dtype=arr.dtype)
        for r in range(self.world):
            out[:, lo[r]:lo[r + 1]] = parts[r][:, : lo[r + 1] - lo[r]]
        return out

    def barrier(self):
        if self.world > 1 and not self.simulate:
            self._dist.barrier(group=self.group)
