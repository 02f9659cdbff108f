#!/usr/bin/env python3
"""Feasibility numbers of the two-precision inner CV (round 6): the score tables (sum over the inner
folds) of every outer fold from the three-MFMA products and from the one-MFMA screening pass on the same data --
how far apart they are, how many voxels' best two alphas lie closer than a gap tau, whether any voxel OUTSIDE that
set changes its alpha, and what the two passes cost.

    python tools/screen_probe.py [--voxels 80000] [--cfg cfg2|cfg4|cfg5] [--reps 3]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402
from litcoder_core_amd.engine.common import FitOptions  # noqa: E402

SHAPES = {"cfg2": dict(T=3000, F0=768, DELAYS=[1, 2, 3, 4], A=20), "cfg4": dict(T=2226, F0=768, DELAYS=[1, 2, 3, 4], A=20),
          "cfg5": dict(T=3000, F0=1280, DELAYS=[1, 2, 3, 4, 5, 6], A=32),
          "small": dict(T=900, F0=75, DELAYS=[1, 2, 3, 4], A=15), "tiny": dict(T=300, F0=40, DELAYS=[1, 2, 3, 4], A=12)}


def fit(dX, dY, p, V, alphas, screen, capture, **kw):
    m = NestedCVModel("ridge_regression", options=FitOptions(screen_inner=bool(screen), **kw))
    m.debug_scores = [] if capture else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.fit_predict_device(dX, dY, p, V, alphas=alphas, **bench.FIT_KW)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out, m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=80000)
    ap.add_argument("--cfg", default="cfg2")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    sh = SHAPES[a.cfg]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dX, dY, p = bench.synth_inputs(a.voxels, 0, dev, T=sh["T"], F0=sh["F0"], DELAYS=sh["DELAYS"])
    alphas = np.logspace(-1, 8, sh["A"])
    V = a.voxels
    n_in = bench.N_INNER
    ms3, out3, m3 = fit(dX, dY, p, V, alphas, False, True)
    ms1, out1, m1 = fit(dX, dY, p, V, alphas, True, True, screen_tau=0.0)      # tau = 0: nothing is refined
    print(f"{a.cfg}: V {V}; first fits {ms3:.1f} ms (three MFMAs per product) / {ms1:.1f} ms (screening arithmetic, no refinement); "
          f"screen_terms {m1.last_fit.get('screen_terms')}")
    taus = [2e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3]
    for (f3, c3, s3), (f1, c1, s1) in zip(m3.debug_scores, m1.debug_scores):
        assert (f3, c3) == (f1, c1)
        s3 = s3[:, :V].double() / n_in
        s1 = s1[:, :V].double() / n_in
        d = (s1 - s3).abs()
        top3 = torch.topk(s3, 2, dim=0)
        top1 = torch.topk(s1, 2, dim=0)
        gap1 = top1.values[0] - top1.values[1]
        b3, b1 = s3.argmax(0), s1.argmax(0)
        flips = b3 != b1
        rows = n_in * ((sh["T"] - sh["T"] // bench.N_OUTER) // n_in)
        worst = gap1[flips].max().item() if flips.any() else 0.0
        print(f"         in units of 1/sqrt(validation rows = {rows}): rms error {d.pow(2).mean().sqrt().item() * rows ** 0.5:.2e}, "
              f"max error {d.max().item() * rows ** 0.5:.2e}, largest screening gap of a voxel whose argmax is wrong {worst * rows ** 0.5:.2e}")
        # pairs of alphas on the polynomial series share the screening pass' moments: the error of a DIFFERENCE of their scores
        # should shrink like 1 / alpha_min^2 -- per adjacent pair: max |d(light diff) - d(exact diff)| x alpha_min^2
        ser = [i for i, al in enumerate(alphas) if al >= 7.9]
        pair = []
        for i, j in zip(ser[:-1], ser[1:]):
            e = ((s1[i] - s1[j]) - (s3[i] - s3[j])).abs().max().item()
            pair.append(e * float(alphas[i]) ** 2 * rows ** 0.5)
        hat = [i for i, al in enumerate(alphas) if al < 7.9]
        pair_h = [((s1[i] - s1[j]) - (s3[i] - s3[j])).abs().max().item() * rows ** 0.5 for i, j in zip(hat[:-1] + hat[-1:], hat[1:] + ser[:1])]
        print(f"         error of score DIFFERENCES, max over voxels, units of 1/sqrt(rows): adjacent series pairs x alpha_min^2: "
              + " ".join(f"{x:.1e}" for x in pair) + ";  pairs with a factorised alpha (no factor): " + " ".join(f"{x:.1e}" for x in pair_h))
        line = (f"fold {f3}: |d score| (fold mean) max {d.max().item():.2e}  99.9 % {torch.quantile(d.flatten()[::7].float(), 0.999).item():.2e}"
                f"  rms {d.pow(2).mean().sqrt().item():.2e};  argmax differs for {int(flips.sum())} voxels"
                f" (largest screening gap among them {gap1[flips].max().item() if flips.any() else 0.0:.2e})")
        print(line)
        print("         tau: " + "  ".join(f"{t:.0e}: {int((gap1 < t).sum())} undecided ({100.0 * float((gap1 < t).float().mean()):.2f} %), "
                                           f"{int((flips & (gap1 >= t)).sum())} wrong outside" for t in taus))
    al3, al1 = np.stack(m3.last_fold_alphas), np.stack(m1.last_fold_alphas)
    print(f"alpha agreement over all folds (screening alone, NO refinement): {float((al3 == al1).mean()):.6f}")
    # the two-precision inner CV as the product runs it: screening + refinement of the undecided voxels
    for tau in (5e-3, 1.5e-2):
        ms2, out2, m2 = fit(dX, dY, p, V, alphas, True, True, screen_tau=tau)
        al2 = np.stack(m2.last_fold_alphas)
        nd = sum(int((a[2][:, :V] != b[2][:, :V]).any(0).sum()) for a, b in zip(m3.debug_scores, m2.debug_scores))
        dW = float((out2[1][:, :V] - out3[1][:, :V]).abs().max()) if torch.is_tensor(out2[1]) else float(np.abs(out2[1] - out3[1]).max())
        dc = float(np.abs(np.asarray(out2[0]["correlations"]) - np.asarray(out3[0]["correlations"])).max())
        print(f"screen_tau {tau:g} (gap {tau / np.sqrt(n_in * (sh['T'] - sh['T'] // bench.N_OUTER) // n_in):.1e}): alpha agreement with the three-MFMA fit {float((al3 == al2).mean()):.6f} ({int((al3 != al2).sum())} of {al3.size} differ); "
              f"max |dW| {dW:.3e}, max |d corr| {dc:.3e}; undecided {m2.last_fit.get('undecided')} of {m2.last_fit.get('screened')} "
              f"(overflows {m2.last_fit.get('screen_overflows', 0)}, panel columns queued {m2.last_fit.get('refine_launch_cols')}); "
              f"columns of the five score tables equal, bit for bit, to the three-MFMA tables': {5 * V - nd}")
    # timings: interleaved
    for screen in (False, True, "one workgroup per CU", False, True, "one workgroup per CU"):
        ops.timing_enable(True)
        ts = []
        for _ in range(a.reps):
            ms, _, m = fit(dX, dY, p, V, alphas, screen, False, **({"screen_two_workgroups": False} if isinstance(screen, str) else {}))
            ts.append(ms)
        tm = ops.timing_read()
        ops.timing_enable(False)
        keys = sorted(tm)
        print(f"screen={screen}: fits {', '.join(f'{t:.1f}' for t in ts)} ms; per fit: " +
              "; ".join(f"{k} {tm[k][0] / a.reps:.2f} ms / {tm[k][1] // a.reps}" for k in keys))


if __name__ == "__main__":
    main()
