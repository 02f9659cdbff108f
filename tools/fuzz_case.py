#!/usr/bin/env python3
"""One case of tools/fuzz_vs_oracle.py again, with a float64 ground truth beside both fits: who is off, and by how much.
    python tools/fuzz_case.py seed case [large | tall] [FitOptions field=value ...]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import litcoder_core_amd as lc  # noqa: E402
import oracle.folds as ofolds  # noqa: E402
import oracle.nested_cv as onc  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
large = len(sys.argv) > 3 and sys.argv[3] == "large"
tall = len(sys.argv) > 3 and sys.argv[3] == "tall"
rng = np.random.default_rng(seed)
for case in range(want + 1):
    T = int(rng.integers(90, 420))
    p = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 40, 70, 130, 300]))
    V = int(rng.choice([1, 3, 17, 64, 100, 129, 257, 300]))
    if large:
        T = int(rng.integers(500, 1400))
        p = int(rng.choice([40, 300, 517, 768, 1000, 1536]))
        V = int(rng.choice([300, 1025, 2000, 3333, 5000]))
    if tall:
        T = int(rng.integers(1800, 3400))
        p = int(rng.choice([256, 300, 320, 384, 517, 640]))
        V = int(rng.choice([300, 1025, 2000, 3000]))
    fold = str(rng.choice(["kfold", "chunked", "kfold_trimmed", "chunked_trimmed", "timeseries", "group"]))
    use_corr = bool(rng.random() < 0.8)
    kw = dict(folding_type=fold, n_outer_folds=int(rng.integers(2, 4)), n_inner_folds=int(rng.integers(2, 4)),
              alphas=np.logspace(rng.uniform(-2, 0), rng.uniform(1, 5), int(rng.integers(1, 9))),
              normalpha=bool(rng.random() < 0.7), use_corr=use_corr, single_alpha=bool(rng.random() < 0.25),
              normalize_features=bool(rng.random() < 0.2), normalize_targets=bool(rng.random() < 0.2))
    if "chunked" in fold:
        kw["chunk_length"] = int(rng.integers(5, 30))
    tt = int(rng.integers(30, 90)) if rng.random() < 0.3 else 0
    if fold == "group":
        kw["groups"] = rng.integers(0, 8, size=T - tt)
    signal = 1.0 if not use_corr else float(rng.choice([0.3, 1.0]))
    X = rng.standard_normal((T, p)) * rng.uniform(0.5, 2.0, p)
    Y = X @ (rng.standard_normal((p, V)) * (signal / np.sqrt(p))) + rng.standard_normal((T, V)) + rng.uniform(-3, 3)
    precision = str(rng.choice(["auto", "auto", "f32"]))
if "spikes" in sys.argv[3:]:
    rs = np.random.default_rng(1000003 * seed + want)
    for c in rs.choice(V, size=min(V, int(rs.integers(1, 4))), replace=False):
        Y[int(rs.integers(0, T - tt)), int(c)] = float(rs.choice([-1.0, 1.0]) * 10.0 ** rs.uniform(4, 6))
print(f"T{T} p{p} V{V} {fold} tt{tt} {precision}", {k: v for k, v in kw.items() if k != "groups"})
args = (X[:T - tt], Y[:T - tt])
extra = dict(X_test=X[T - tt:], y_test=Y[T - tt:]) if tt else {}
kw_run = {k: v for k, v in kw.items() if not (tt and k == "n_outer_folds")}
random.seed(want); np.random.seed(want)
detail = {}
m_o, W_o, a_o = onc.fit_predict(*args, detail=detail, **extra, **kw_run)
random.seed(want); np.random.seed(want)
if os.environ.get("FUZZ_TRACE_LMAX"):       # print every lambda_max the fit computes (S0 = its square root)
    from litcoder_core_amd import ops as _ops
    _real = _ops.lambda_max_masked

    def _traced(*a, **k):
        out = _real(*a, **k)
        print("  lambda_max_masked tol", k.get("tol"), "-> S0", np.sqrt(out.cpu().numpy()))
        return out
    _ops.lambda_max_masked = _traced
opt_kw = {}
for item in sys.argv[3:]:
    if "=" in item:
        k_, v_ = item.split("=")
        opt_kw[k_] = float(v_) if "." in v_ or "e" in v_ else int(v_)
from litcoder_core_amd.engine.common import FitOptions  # noqa: E402
model = lc.NestedCVModel("r", precision=precision, options=FitOptions(**opt_kw) if opt_kw else None)
if opt_kw:
    print("options", opt_kw)
m, W, a = model.fit_predict(*args, **extra, **kw_run)
if tt:                                       # train/test: one "fold" = all training rows, the oracle's alphas
    detail = dict(outer=[(np.arange(T - tt), None)], fold_alphas=[np.asarray(a_o, dtype=np.float64)])
X, Y = args
print("form", model.last_form, model.last_fit.get("precision"))
# float64 ground truth: the reference's arithmetic (fp32 inputs, train-statistics normaliser, S[0] of the normalised
# training design) carried out in float64, at the ORACLE's alphas
X32, Y32 = X.astype(np.float32).astype(np.float64), Y.astype(np.float32).astype(np.float64)
Wt = np.zeros((p, V))
for f, (tr, te) in enumerate(detail["outer"]):
    tr = np.asarray(tr)
    Xtr, Ytr = X32[tr], Y32[tr]
    if kw["normalize_features"]:
        Xtr = (Xtr - Xtr.mean(0)) / (Xtr.std(0, ddof=1) + 1e-8)
    if kw["normalize_targets"]:
        Ytr = (Ytr - Ytr.mean(0)) / (Ytr.std(0, ddof=1) + 1e-8)
    U, S, Vh = np.linalg.svd(Xtr, full_matrices=False)
    al = np.asarray(detail["fold_alphas"][f], dtype=np.float64)
    na = al * S[0] if kw["normalpha"] else al
    UR = U.T @ Ytr
    Wf = np.empty((p, V))
    for v in range(V):
        Wf[:, v] = Vh.T @ ((S / (S ** 2 + na[v] ** 2)) * UR[:, v])
    Wt += Wf / len(detail["outer"])
    print(f"fold {f}: n_train {len(tr)}  S0 {S[0]:.4g}  Smin {S[-1]:.4g}  alphas {np.unique(al)}")
scale = np.abs(Wt).max()
print(f"max|W_true| {scale:.4g}")
print(f"ours   - truth: max abs {np.abs(W - Wt).max():.3g}   oracle - truth: max abs {np.abs(W_o - Wt).max():.3g}   "
      f"ours - oracle: {np.abs(W - W_o).max():.3g}")
same = np.isclose(np.asarray(a), np.asarray(a_o), rtol=1e-6)
tol = 2e-4 * np.abs(W_o) + 3e-6 * max(1.0, float(np.abs(W_o).max()))
for name, Wx in (("ours", W), ("oracle", W_o)):
    bad = np.abs(Wx - Wt) > tol
    print(f"{name}: {int(bad[:, same].sum())} elements beyond the test's tolerance of the float64 truth (voxels with the oracle's alphas)")
print(f"alphas equal for {same.mean():.3f} of the voxels")
if same.any():
    eo, er = np.abs(W - Wt)[:, same], np.abs(W_o - Wt)[:, same]
    print(f"voxels with the oracle's alphas: ours - truth max {eo.max():.3g} (rms {np.sqrt((eo ** 2).mean()):.3g}), "
          f"oracle - truth max {er.max():.3g} (rms {np.sqrt((er ** 2).mean()):.3g}); max|W_true| {scale:.3g}")
co, cr = np.asarray(m["correlations"], dtype=np.float64), np.asarray(m_o["correlations"], dtype=np.float64)
print(f"correlations: ours - oracle max {np.nanmax(np.abs(co - cr)):.3g} over all voxels, "
      f"{np.nanmax(np.abs(co - cr)[same]) if same.any() else float('nan'):.3g} over those with the oracle's alphas")
