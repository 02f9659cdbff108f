"""Oracle: FIR delay stacking (test infrastructure, see oracle/__init__.py).

Follows ``encoding/features/FIR_expander.py:24-43`` (``FIR.make_delayed``) and
its duplicate ``encoding/utils.py:62-83``.
"""
import numpy as np


def make_delayed(stim, delays, circpad=False):
    """Column block k of the result holds ``stim`` shifted down by ``delays[k]``
    rows (up for negative values); vacated rows are zero, or wrap around when
    ``circpad``.  A zero delay contributes a plain copy (keeps the input dtype),
    every other block is float64 (FIR_expander.py:31-42).

    Corner the reference's slice arithmetic implies: with ``circpad`` and
    ``|d| >= nt`` both slice assignments degenerate to "copy everything", so
    the block equals ``stim`` unshifted; without ``circpad`` it is all zero.
    """
    stim = np.asarray(stim)
    nt, ndim = stim.shape
    rows = np.arange(nt)
    blocks = []
    for d in delays:
        d = int(d)
        if d == 0:
            blocks.append(stim.copy())
            continue
        blk = np.zeros((nt, ndim), dtype=np.float64)
        src = rows - d                       # out[r] = stim[r - d]
        if abs(d) >= nt:
            if circpad:
                blk[:] = stim
        elif circpad:
            blk[rows] = stim[src % nt]
        else:
            ok = (src >= 0) & (src < nt)
            blk[rows[ok]] = stim[src[ok]]
        blocks.append(blk)
    return np.hstack(blocks)
