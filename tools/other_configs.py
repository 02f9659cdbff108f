#!/usr/bin/env python3
"""One fit at the shapes of BASELINE.json configs 4 and 5 on a single GPU (synthetic data, device-resident inputs):
cfg1 = a word-rate feature (p = 4) over T 9000, V 80 000 (p << n: the dual form still applies);
cfg4 = T 2226, p 3072, V 200 000 (the whole volume on one GPU instead of 8 shards), 20 alphas;
cfg5 = T 3000, p 1280 x 6 delays = 7680, V 80 000, 32 alphas, two feature bands with different penalty scales.
Prints time, voxels/s and the median score; checks that every result is finite; and (round 5) the same fit HOST TO HOST --
float64 numpy arrays in pageable host memory in, metrics + float32 host weights out, the metric's own definition
-- with the config's link floor: max(float32 bytes up / H2D rate, weight bytes down / D2H rate), the rates measured here
with 1 GiB page-locked copies.   python tools/other_configs.py [cfg1 cfg4 cfg5 search]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import NestedCVModel, ops  # noqa: E402

dev = ops.device(0)
KW = dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, single_alpha=False, normalpha=True, use_corr=True)


def synth(T, F0, delays, V, band_scale=None, seed=0):
    rng = np.random.default_rng(seed)
    Xd = ops.fir_delay(torch.from_numpy(rng.standard_normal((T, F0))).to(dev), delays, False)
    p = Xd.shape[1]
    dX = torch.zeros((T, ops.pad_to(p, 32)), dtype=torch.float32, device=dev)
    dX[:, :p] = Xd.to(torch.float32)
    if band_scale is not None:                     # banded ridge with fixed scales == ridge on the rescaled design
        dX[:, :p] /= torch.as_tensor(band_scale, dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    dY = torch.zeros((T, ops.pad_to(V, 128)), dtype=torch.float32, device=dev)
    W = 0.02 * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
    dY[:, :V] = dX[:, :p] @ W + torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    return dX, dY, p


def link_rates():
    n = 1 << 30
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    rates = []
    for src, dst in ((h, d), (d, h)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        rates.append(3 * n / (time.perf_counter() - t))
    return rates


RATES = link_rates()
print(f"page-locked link rates on this box: H2D {RATES[0] / 1e9:.1f} GB/s, D2H {RATES[1] / 1e9:.1f} GB/s", flush=True)
only = sys.argv[1:]
for name, args, alphas in (
        ("cfg1-like: wordrate, T 9000, p 4 (1 x 4 delays), V 80000, 20 alphas", dict(T=9000, F0=1, delays=[1, 2, 3, 4], V=80000),
         np.logspace(-1, 8, 20)),
        ("cfg4-like: T 2226, p 3072, V 200000, 20 alphas", dict(T=2226, F0=768, delays=[1, 2, 3, 4], V=200000),
         np.logspace(-1, 8, 20)),
        ("cfg5-like: T 3000, p 7680 (1280 x 6), V 80000, 32 alphas, 2 bands",
         dict(T=3000, F0=1280, delays=[1, 2, 3, 4, 5, 6], V=80000,
              band_scale=np.r_[np.full(3840, 1.0), np.full(3840, 2.0)]), np.logspace(-1, 8, 32))):
    if only and not any(o in name for o in only):
        continue
    V = args["V"]
    dX, dY, p = synth(**args)
    model = NestedCVModel("ridge_regression")
    fit = lambda: model.fit_predict_device(dX, dY, p, V, alphas=alphas, **KW)
    fit(); fit(); torch.cuda.synchronize()
    t = time.perf_counter(); m, W, a = fit(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    name += f" [{model.last_form} form]"
    ok = bool(np.isfinite(np.asarray(m["correlations"])).all() and torch.isfinite(W).all() and np.isfinite(a).all())
    print(f"{name}: resident {1e3 * dt:.0f} ms = {V / dt:.0f} voxels/s, median score {m['median_score']:.4f}, all finite: {ok}; "
          f"mean-operator refit {model.last_fit.get('mean_operator')}; screening undecided {model.last_fit.get('undecided')} of "
          f"{model.last_fit.get('screened')}",
          flush=True)
    # ---- host to host: the reference's call
    r_res = np.asarray(m["correlations"])
    del W
    T = dX.shape[0]
    Xh = dX[:, :p].cpu().numpy().astype(np.float64)
    Yh = np.empty((T, V), dtype=np.float64)
    for c in range(0, V, 16384):
        Yh[:, c:c + 16384] = dY[:, c:min(V, c + 16384)].cpu().numpy()
    del dX, dY
    torch.cuda.empty_cache()
    hfit = lambda: model.fit_predict(Xh, Yh, alphas=alphas, **KW)
    hfit()
    ts = []
    for _ in range(3):
        out = None
        torch.cuda.synchronize(); t = time.perf_counter(); out = hfit(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt_h = float(np.median(ts))
    up, down = (T * V + T * p) * 4, p * V * 4
    floor = max(up / RATES[0], down / RATES[1])
    same = bool(np.array_equal(np.asarray(out[0]["correlations"]), r_res))
    print(f"    host to host: {1e3 * dt_h:.0f} ms (median of 3: {', '.join('%.0f' % (1e3 * x) for x in ts)}) = {V / dt_h:.0f} voxels/s; "
          f"{up / 1e9:.2f} GB up as float32, {down / 1e9:.2f} GB of weights down: link floor {1e3 * floor:.0f} ms -> "
          f"{dt_h / floor:.2f} x the floor; panels {model.last_fit.get('panels')}; per-voxel correlations equal to the resident "
          f"fit's: {same}", flush=True)
    del Xh, Yh, out
    torch.cuda.empty_cache()

if not only or any("search" in o for o in only):
    # cfg5's shape with a SEARCH over band scales (three candidates): every candidate is a pass of the inner CV
    import time as _t
    from litcoder_core_amd import BandedNestedCVModel
    rng = np.random.default_rng(3)
    T, F0, V = 3000, 1280, 80000
    Xh = ops.fir_delay(torch.from_numpy(rng.standard_normal((T, F0))).to(dev), [1, 2, 3, 4, 5, 6], False).cpu().numpy()
    g = torch.Generator(device=dev); g.manual_seed(7)
    Wd = 0.02 * torch.randn((Xh.shape[1], V), generator=g, device=dev, dtype=torch.float32)
    Wd[3840:, : V // 2] = 0
    Yh = (torch.from_numpy(Xh).to(dev, torch.float32) @ Wd + torch.randn((T, V), generator=g, device=dev)).cpu().numpy()
    del Wd
    bm = BandedNestedCVModel("ridge_regression")
    args = (Xh, Yh, [(0, 3840), (3840, 7680)], [[1.0, 1.0], [1.0, 4.0], [4.0, 1.0]])
    skw = dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, alphas=np.logspace(-1, 8, 32))
    bm.fit_predict_search(*args, **skw)
    torch.cuda.synchronize(); t = _t.perf_counter()
    m, W, a = bm.fit_predict_search(*args, **skw)
    torch.cuda.synchronize(); dt = _t.perf_counter() - t
    share = [float((bm.last_fold_candidates == c).mean()) for c in range(3)]
    print(f"cfg5-like with a search over 3 band-scale candidates (host arrays in, host weights out): {1e3 * dt:.0f} ms = "
          f"{V / dt:.0f} voxels/s, median score {m['median_score']:.4f}, candidate shares {share}", flush=True)
