#!/usr/bin/env python3
"""Stage-by-stage numerical diagnosis of the HIP path against numpy / the oracle on a GPU box.
Prints one line per stage (max abs error) and never stops at the first failure.
    python tools/gpu_diag.py [--big]
"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from litcoder_core_amd import ops  # noqa: E402
from litcoder_core_amd._lib import LC_MB, LC_NB  # noqa: E402
import oracle  # noqa: E402

dev = ops.device()
rng = np.random.default_rng(0)
results = []


def stage(name):
    def deco(fn):
        t = time.time()
        try:
            msg = fn()
            torch.cuda.synchronize()
            results.append((name, "ok", msg, time.time() - t))
        except Exception as e:  # noqa: BLE001
            results.append((name, "FAIL", f"{type(e).__name__}: {e}", time.time() - t))
            traceback.print_exc()
        print(f"[{results[-1][1]:4s}] {name}: {results[-1][2]}  ({results[-1][3]:.2f}s)", flush=True)
        return fn
    return deco


def dv(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


@stage("fir")
def _():
    s = rng.standard_normal((333, 77))
    out = ops.fir_delay(dv(s), [1, 2, 3, 4, -2, 0, 400], False).cpu().numpy()
    ref = oracle.fir.make_delayed(s, [1, 2, 3, 4, -2, 0, 400], False)
    out2 = ops.fir_delay(dv(s.astype(np.float32)), [3, -1], True).cpu().numpy()
    ref2 = oracle.fir.make_delayed(s.astype(np.float32), [3, -1], True)
    return f"exact={np.array_equal(out, ref)} exact_circ_f32={np.array_equal(out2, ref2)}"


@stage("lanczos")
def _():
    n_old, D, n_new = 2500, 768, 350
    ot = np.sort(rng.uniform(0, 700, n_old))
    nt = 1.0 + 2.0 * np.arange(n_new)
    d = rng.standard_normal((n_old, D))
    cutoff = 1 / np.mean(np.diff(nt))
    out = ops.lanczos_interp(dv(d), dv(ot), dv(nt), cutoff, 3, False).cpu().numpy()
    ref = oracle.lanczos.lanczos_interp(d, ot, nt, 3, 1.0)
    out2 = ops.lanczos_interp(dv(d[:, :50].copy()), dv(ot), dv(nt), cutoff, 3, True).cpu().numpy()
    ref2 = oracle.lanczos.lanczos_interp(d[:, :50], ot, nt, 3, 1.0, True)
    return f"maxerr={np.abs(out - ref).max():.3e} rect={np.abs(out2 - ref2).max():.3e}"


@stage("cast+gather+colstats+pearson")
def _():
    T, V = 157, 301
    y = rng.standard_normal((T, V)) * 3 + 5
    ld = ops.pad_to(V, 128)
    dy = ops.upload_f32(y, ld, dev)
    e1 = np.abs(dy[:, :V].cpu().numpy() - y.astype(np.float32)).max()
    padz = float(dy[:, V:].abs().max())
    rows = np.array([5, 7, 100, 3, 9, 11, 150])
    m, s = ops.col_mean_std(dy, ops.idx_tensor(rows, len(rows), dev), len(rows), V)
    y32 = y.astype(np.float32).astype(np.float64)
    e2 = np.abs(m.cpu().numpy() - y32[rows].mean(0)).max()
    e3 = np.abs(s.cpu().numpy() - y32[rows].std(0, ddof=1)).max()
    b = y + rng.standard_normal((T, V))
    db = ops.upload_f32(b, ld, dev)
    r = ops.pearson_cols(dy, db, T, V).cpu().numpy()
    rr = np.array([np.corrcoef(y32[:, i], b.astype(np.float32).astype(np.float64)[:, i])[0, 1] for i in range(V)])
    return f"cast={e1:.1e} pad={padz} mean={e2:.1e} std={e3:.1e} pearson={np.abs(r - rr).max():.1e}"


@stage("gram")
def _():
    T, p = 331, 203
    x = rng.standard_normal((T, p)).astype(np.float32)
    dx = ops.upload_f32(x, ops.pad_to(p, 32), dev)
    k = ops.gram(dx, T, p).cpu().numpy()
    ref = x.astype(np.float64) @ x.astype(np.float64).T
    return f"relerr={np.abs(k - ref).max() / np.abs(ref).max():.2e} sym={np.abs(k - k.T).max():.1e}"


def make_k(T, p, ar=0.0):
    x = rng.standard_normal((T, p))
    if ar:
        for t in range(1, T):
            x[t] = ar * x[t - 1] + np.sqrt(1 - ar * ar) * x[t]
    x = x.astype(np.float32)
    return x, x.astype(np.float64) @ x.astype(np.float64).T


@stage("lambda_max")
def _():
    msgs = []
    for (T, p, ar, steps) in [(300, 500, 0.0, 64), (300, 500, 0.0, 192), (300, 40, 0.7, 192), (100, 300, 0.9, 192),
                              (1920, 3072, 0.0, 192)]:
        x, k = make_k(T, p, ar)
        dk = dv(k)
        rows1 = np.arange(0, T, 2)
        rows2 = np.arange(T // 3, T)
        N = ops.pad_to(max(len(rows1), len(rows2)), LC_NB)
        idx = torch.stack([ops.idx_tensor(rows1, N, dev), ops.idx_tensor(rows2, N, dev)])
        lm = ops.lambda_max(dk, idx, 2, N, steps).cpu().numpy()
        ref = [np.linalg.eigvalsh(k[np.ix_(r, r)])[-1] for r in (rows1, rows2)]
        msgs.append(f"T{T}p{p}s{steps}:{max(abs(lm[i] - ref[i]) / ref[i] for i in range(2)):.1e}")
    return " ".join(msgs)


@stage("chol_solve")
def _():
    T, p = 400, 600
    x, k = make_k(T, p, 0.5)
    dk = dv(k)
    tr_list = [np.r_[0:100, 180:400], np.r_[0:250, 330:400]]
    va_list = [np.r_[100:180], np.r_[250:330]]
    F, A = 2, 3
    N = ops.pad_to(max(map(len, tr_list)), LC_NB)
    M = ops.pad_to(max(map(len, va_list)), LC_MB)
    tr = torch.stack([ops.idx_tensor(t, N, dev) for t in tr_list])
    va = torch.stack([ops.idx_tensor(v, M, dev) for v in va_list])
    a2 = np.array([[0.5, 30.0, 4000.0], [2.0, 100.0, 1e5]])
    aug = torch.empty((F * A, N + M, N), dtype=torch.float64, device=dev)
    H = torch.empty((F * A, M, N), dtype=torch.float32, device=dev)
    ops.batch_assemble(dk, tr, va, None, dv(a2.reshape(-1)), F, A, N, M, aug)
    info = ops.batch_chol_solve(aug, F * A, N, M, H).cpu().numpy()
    Hh = H.cpu().numpy()
    err = 0.0
    for f in range(F):
        for a in range(A):
            t, v = tr_list[f], va_list[f]
            ref = k[np.ix_(v, t)] @ np.linalg.inv(k[np.ix_(t, t)] + a2[f, a] * np.eye(len(t)))
            got = Hh[f * A + a][: len(v), : len(t)]
            err = max(err, np.abs(got - ref).max() / np.abs(ref).max())
            padmax = max(np.abs(Hh[f * A + a][len(v):]).max(initial=0), np.abs(Hh[f * A + a][:, len(t):]).max(initial=0))
    return f"relerr={err:.2e} info={info.tolist()} pad={padmax}"


@stage("alpha_sweep (fused GEMM) vs oracle")
def _():
    import oracle.ridge as oridge
    msgs = []
    for (T, p, V, use_corr) in [(300, 400, 200, True), (300, 400, 200, False), (700, 64, 1000, True)]:
        X = rng.standard_normal((T, p))
        Wt = rng.standard_normal((p, V)) / np.sqrt(p)
        Y = X @ Wt + rng.standard_normal((T, V))
        Y[:, 3] = 2.0
        Y[:, 4] += 100.0
        tr_rows = np.r_[0:T // 2, T // 2 + T // 5:T]
        va_rows = np.r_[T // 2:T // 2 + T // 5]
        alphas = np.logspace(-1, 4, 6)
        Xt, Yt = torch.tensor(X, dtype=torch.float32), torch.tensor(Y, dtype=torch.float32)
        ref = oridge.alpha_sweep_scores(Xt[tr_rows], Xt[va_rows], Yt[tr_rows], Yt[va_rows], alphas, 1e-10, use_corr,
                                        True).numpy()
        from litcoder_core_amd.nested_cv import RidgeCVEngine
        eng = RidgeCVEngine(X, Y, alphas, True, use_corr, False, False)
        scores, info = eng._alpha_scores(eng.K, eng.dY, [(tr_rows, va_rows)])
        got = scores[:, :V].cpu().numpy()
        d = np.abs(got - ref)
        d[np.abs(ref) > 1e30] = 0
        msgs.append(f"T{T}p{p}V{V}corr{int(use_corr)}:max={d.max():.2e}@{np.unravel_index(d.argmax(), d.shape)} "
                    f"argmax_agree={np.mean(got.argmax(0) == ref.argmax(0)):.3f}")
    return " | ".join(msgs)


@stage("alpha_sweep f16x3 vs oracle")
def _():
    import oracle.ridge as oridge
    from litcoder_core_amd.nested_cv import RidgeCVEngine
    msgs = []
    for (T, p, V, use_corr) in [(300, 400, 200, True), (300, 400, 200, False), (700, 64, 1000, True), (900, 1100, 700, True)]:
        X = rng.standard_normal((T, p))
        Y = X @ (rng.standard_normal((p, V)) / np.sqrt(p)) + rng.standard_normal((T, V))
        Y[:, 3] = 2.0
        Y[:, 4] += 100.0
        Y[:, 5] *= 1e-4
        Y[:, 6] *= 3e4
        tr_rows = np.r_[0:T // 2, T // 2 + T // 5:T]
        va_rows = np.r_[T // 2:T // 2 + T // 5]
        alphas = np.logspace(-1, 4, 6)
        Xt, Yt = torch.tensor(X, dtype=torch.float32), torch.tensor(Y, dtype=torch.float32)
        ref = oridge.alpha_sweep_scores(Xt[tr_rows], Xt[va_rows], Yt[tr_rows], Yt[va_rows], alphas, 1e-10, use_corr,
                                        True).numpy()
        eng = RidgeCVEngine(X, Y, alphas, True, use_corr, False, False, precision="f16x3")
        scores, info = eng._alpha_scores(eng.K, eng.dY, [(tr_rows, va_rows)])
        got = scores[:, :V].cpu().numpy()
        eng32 = RidgeCVEngine(X, Y, alphas, True, use_corr, False, False)
        got32 = eng32._alpha_scores(eng32.K, eng32.dY, [(tr_rows, va_rows)])[0][:, :V].cpu().numpy()
        d = np.abs(got - ref)
        d[np.abs(ref) > 1e30] = 0
        d32 = np.abs(got - got32)
        d32[np.abs(got32) > 1e30] = 0
        msgs.append(f"T{T}p{p}V{V}corr{int(use_corr)}:vs_oracle={d.max():.2e} vs_f32path={d32.max():.2e} "
                    f"argmax_agree={np.mean(got.argmax(0) == ref.argmax(0)):.3f}")
    return " | ".join(msgs)


@stage("full fit f16x3 vs oracle (kfold)")
def _():
    from litcoder_core_amd import NestedCVModel
    import oracle.nested_cv as onc
    T, p, V = 240, 320, 64
    X = rng.standard_normal((T, p))
    Y = X @ (rng.standard_normal((p, V)) * 0.05) + rng.standard_normal((T, V))
    alphas = np.logspace(-1, 4, 6)
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=alphas)
    m0, W0, a0 = onc.fit_predict(X, Y, **kw)
    m1, W1, a1 = NestedCVModel("ridge_regression", precision="f16x3").fit_predict(X, Y, **kw)
    return (f"corr={np.abs(np.array(m0['correlations']) - np.array(m1['correlations'])).max():.2e} "
            f"W={np.abs(W0 - W1).max():.2e} alphas_equal={np.mean(a0 == a1):.3f}")


@stage("full fit vs oracle (kfold)")
def _():
    import random
    from litcoder_core_amd import NestedCVModel
    import oracle.nested_cv as onc
    T, p, V = 240, 320, 64
    X = rng.standard_normal((T, p))
    Y = X @ (rng.standard_normal((p, V)) * 0.05) + rng.standard_normal((T, V))
    alphas = np.logspace(-1, 4, 6)
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=alphas)
    m0, W0, a0 = onc.fit_predict(X, Y, **kw)
    m1, W1, a1 = NestedCVModel("ridge_regression").fit_predict(X, Y, **kw)
    return (f"corr={np.abs(np.array(m0['correlations']) - np.array(m1['correlations'])).max():.2e} "
            f"W={np.abs(W0 - W1).max():.2e} alphas_equal={np.mean(a0 == a1):.3f} "
            f"p={np.abs(np.array(m0['p_values']) - np.array(m1['p_values'])).max():.2e} "
            f"median {m0['median_score']:.5f} vs {m1['median_score']:.5f}")


if "--big" in sys.argv:
    @stage("cfg2-sized timing (V=8192)")
    def _():
        from litcoder_core_amd import NestedCVModel
        T, F0, V = 3000, 768, 8192
        X0 = rng.standard_normal((T, F0))
        X = oracle.fir.make_delayed(X0, [1, 2, 3, 4])
        Y = X @ (0.02 * rng.standard_normal((X.shape[1], V))) + rng.standard_normal((T, V))
        t = time.time()
        m, W, a = NestedCVModel("r").fit_predict(X, Y, folding_type="kfold", alphas=np.logspace(-1, 8, 20))
        torch.cuda.synchronize()
        return f"time={time.time() - t:.2f}s median={m['median_score']:.4f}"

print("\nSUMMARY")
for r in results:
    print(f"  {r[1]:4s} {r[0]}: {r[2]}")
