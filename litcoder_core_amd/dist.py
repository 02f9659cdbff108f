"""Voxel sharding across the GPUs of a node: one process per GPU, torch.distributed.

The fit is independent per voxel (column of ``targets``) given the replicated design-matrix
algebra, so each rank fits a contiguous block of voxel columns and the data path needs no
collective.  Two small exchanges remain:

  * ``single_alpha=True`` (nested_cv.py:396-400): the per-alpha score sums over voxels are
    all-reduced (A doubles, once per outer fold);
  * the per-voxel result vectors (correlations, chosen alpha index; a few floats per voxel)
    are all-gathered once at the end so that the global statistics (FDR, medians) see every
    voxel.

Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
from typing import List, Optional, Tuple

import numpy as np


_COMM_STREAMS: dict = {}


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``; the first ``n % world`` ranks get one more."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardContext:
    """Wraps a torch.distributed process group (or nothing, for a single process)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.device = device
        self.rank = self._dist.get_rank(group) if self._dist else 0
        self.world = self._dist.get_world_size(group) if self._dist else 1
        self._comm = None

    def _comm_scope(self):
        """The exchanges carry host data (per-voxel result vectors, score sums): on a GPU they run on a stream of
        their own, so that neither the staging copies nor the collective queue behind the fit's kernels already
        enqueued on the compute streams."""
        import contextlib
        if self.device is not None and getattr(self.device, "type", None) == "cuda":
            import torch
            if self._comm is None:
                key = (self.device.type, self.device.index)
                if key not in _COMM_STREAMS:        # one per device and process: a stream's first use costs milliseconds
                    _COMM_STREAMS[key] = torch.cuda.Stream(device=self.device)
                self._comm = _COMM_STREAMS[key]
            return torch.cuda.stream(self._comm)
        return contextlib.nullcontext()

    def bounds(self, n_items: int) -> Tuple[int, int]:
        return shard_bounds(n_items, self.world, self.rank)

    def _tensor(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device) if self.device is not None else t

    def allreduce_sum(self, arr: np.ndarray) -> np.ndarray:
        """Element-wise sum over ranks of a small float64 vector."""
        if self.world == 1:
            return np.asarray(arr, dtype=np.float64)
        with self._comm_scope():
            t = self._tensor(np.asarray(arr, dtype=np.float64))
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()

    def allgather_cols(self, arr: np.ndarray, n_total: int) -> np.ndarray:
        """``arr`` is (k, n_local) for this rank's block; returns (k, n_total) on every rank."""
        arr = np.ascontiguousarray(arr)
        if self.world == 1:
            return arr
        import torch
        k = arr.shape[0]
        widths = [shard_bounds(n_total, self.world, r) for r in range(self.world)]
        wmax = max(hi - lo for lo, hi in widths)
        pad = np.zeros((k, wmax), dtype=arr.dtype)
        pad[:, : arr.shape[1]] = arr
        out = np.empty((k, n_total), dtype=arr.dtype)
        with self._comm_scope():
            mine = self._tensor(pad)
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self._dist.all_gather(parts, mine, group=self.group)
            for (lo, hi), part in zip(widths, parts):
                out[:, lo:hi] = part.cpu().numpy()[:, : hi - lo]
        return out

    def barrier(self):
        if self.world > 1:
            self._dist.barrier(group=self.group)
