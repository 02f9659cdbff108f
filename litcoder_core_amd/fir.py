"""FIR delay expander with the reference's interface, computed on the MI355X.

Mirrors ``encoding/features/FIR_expander.py:6-73`` (``@dataclass FIR`` with the static
``make_delayed`` and the ``expand / n_delays / output_dim / valid_length / summary`` helpers).
The delay stacking itself is the HIP kernel behind ``lc_fir_delay``; outputs are bit-exact
copies of the inputs.
"""
from dataclasses import dataclass
from typing import Iterable, Optional

import numpy as np
import torch

from . import ops


@dataclass
class FIR:
    delays: Optional[Iterable[int]] = None
    circpad: bool = False

    def expand(self, stim: np.ndarray) -> np.ndarray:
        if self.delays is None:
            raise ValueError("delays must be provided for instance usage of FIR")
        return FIR.make_delayed(stim, self.delays, self.circpad)

    @staticmethod
    def make_delayed(stim: np.ndarray, delays: Iterable[int], circpad: bool = False) -> np.ndarray:
        """(nt, ndim) -> (nt, ndim*len(delays)); block k is ``stim`` delayed by ``delays[k]``.
        float64 output, except that all-zero delays keep the input dtype (FIR_expander.py:41:
        ``stim.copy()`` blocks keep it and ``np.hstack`` only promotes when a float64 block exists)."""
        stim = np.asarray(stim)
        nt, ndim = stim.shape                     # same ValueError as the reference for non-2D input
        delays = [int(d) for d in delays]
        if len(delays) == 0:
            raise ValueError("need at least one array to concatenate")   # np.hstack([]) in the reference
        dev = ops.device()
        if stim.dtype == np.float32:
            src = torch.from_numpy(np.ascontiguousarray(stim)).to(dev)
        else:
            # every other real dtype is widened to float64 exactly as ``dstim[...] = stim[...]`` does
            src = torch.from_numpy(np.ascontiguousarray(stim, dtype=np.float64)).to(dev)
        out = ops.fir_delay(src, delays, circpad).cpu().numpy()
        if all(d == 0 for d in delays) and stim.dtype != np.float64:
            out = out.astype(stim.dtype)
        return out

    def n_delays(self) -> int:
        return len(self.delays) if self.delays is not None else 0

    def output_dim(self, input_dim: int) -> int:
        return input_dim * self.n_delays()

    def valid_length(self, nt: int) -> int:
        if self.delays is None:
            raise ValueError("delays must be provided")
        if self.circpad:
            return nt
        return max(0, nt - max(abs(d) for d in self.delays))

    def summary(self, input_dim: Optional[int] = None, nt: Optional[int] = None) -> str:
        msg = f"FIR(delays={list(self.delays)}, circpad={self.circpad})"
        if input_dim is not None:
            msg += f"\n- Output dim: {self.output_dim(input_dim)}"
        if nt is not None:
            msg += f"\n- Valid length: {self.valid_length(nt)}"
        return msg
