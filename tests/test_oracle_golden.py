"""The oracle (oracle/) against the golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import fir, folds, harness, lanczos, nested_cv, ridge, stats


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_fir_bit_exact(golden_dir):
    g = load(golden_dir, "fir.npz")
    for tag in "abcdefg":
        out = fir.make_delayed(g[f"{tag}_stim"], g[f"{tag}_delays"].tolist(), bool(g[f"{tag}_circpad"]))
        want = g[f"{tag}_out"]
        assert out.dtype == want.dtype and out.shape == want.shape, tag
        assert np.array_equal(out, want), tag


def test_lanczos_kat_and_interp(golden_dir):
    g = load(golden_dir, "downsample.npz")
    assert np.array_equal(lanczos.lanczos_kernel(0.5, g["kat_t"], 3), g["kat_val"])
    d, ot, nt = g["data"], g["oldtime"], g["newtime"]
    np.testing.assert_allclose(lanczos.lanczos_interp(d, ot, nt, 3, 1.0), g["lanczos_w3"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(lanczos.lanczos_interp(d, ot, nt, 2, 0.5), g["lanczos_w2_c05"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(lanczos.lanczos_interp(d, ot, nt, 3, 1.0, True), g["lanczos_w3_rect"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(lanczos.lanczos_interp(g["data_f32"], ot, nt, 3, 1.0), g["lanczos_f32"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(lanczos.sinc_interp(d, ot, nt, 1.0, 3), g["sinc_w3"], rtol=0, atol=1e-13)


def test_simple_downsamplers(golden_dir):
    g = load(golden_dir, "downsample.npz")
    d, ot, nt = g["data"], g["oldtime"], g["newtime"]
    np.testing.assert_allclose(lanczos.rect(d, ot, nt), g["rect"], atol=1e-15)
    for m in ("average", "sum", "last"):
        np.testing.assert_allclose(lanczos.by_label(d, g["labels"], m), g[m], atol=1e-15)
        np.testing.assert_allclose(lanczos.by_chunks(d, g["bounds"], m), g["legacy_" + m], atol=1e-15)


def test_folds_identical(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "folds.json")))
    for c in cases:
        if c["seed"] is not None:
            random.seed(c["seed"])
            np.random.seed(c["seed"])
        groups = np.array(c["groups"]) if "groups" in c else None
        sp = folds.create_folds(c["n"], c["fold_type"], c["n_folds"], c["chunk_length"], c["trim_size"], groups)
        got = [[list(map(int, a)), list(map(int, b))] for a, b in sp]
        assert got == c["splits"], c["fold_type"]


def test_fold_known_answers():
    assert [(len(a), len(b)) for a, b in folds.create_folds(3000, "kfold", 5)] == [(2400, 600)] * 5
    assert [(len(a), len(b)) for a, b in folds.create_folds(3000, "kfold_trimmed", 5)] == [(2400, 590)] * 5
    random.seed(0)
    assert [(len(a), len(b)) for a, b in folds.create_folds(3000, "chunked_trimmed", 5, 20)] == [(2400, 300)] * 5
    assert [len(a) for a, _ in folds.create_folds(3000, "timeseries", 5)] == [500, 1000, 1500, 2000, 2500]
    assert max(max(b) for _, b in folds.create_folds(3005, "chunked_contiguous", 5, 20)) == 2999
    with pytest.raises(ValueError):          # groups landing in trim_size (nested_cv.py:130-132 positional bug)
        folds.create_folds(300, "kfold_trimmed", 3, 20, np.arange(300))
    with pytest.raises(ValueError):
        folds.create_folds(10, "nope", 2)


def test_ridge_solvers(golden_dir):
    g = load(golden_dir, "ridge.npz")
    alphas = g["alphas"]
    for tag in ("wide", "tall"):
        X = torch.tensor(g[f"{tag}_X"], dtype=torch.float32)
        Y = torch.tensor(g[f"{tag}_Y"], dtype=torch.float32)
        tr, va = g[f"{tag}_tr"], g[f"{tag}_va"]
        for uc in (1, 0):
            for na in (1, 0):
                got = ridge.alpha_sweep_scores(X[tr], X[va], Y[tr], Y[va], alphas, 1e-10, bool(uc), bool(na)).numpy()
                np.testing.assert_allclose(got, g[f"{tag}_scores_corr{uc}_norm{na}"], rtol=0, atol=1e-6)
        val = torch.tensor(g[f"{tag}_valphas"])
        for na in (1, 0):
            got = ridge.ridge_weights(X[tr], Y[tr], val, 1e-10, bool(na)).numpy()
            np.testing.assert_allclose(got, g[f"{tag}_W_norm{na}"], rtol=0, atol=1e-6)
        got = ridge.ridge_weights(X[tr], Y[tr], 2.5, 1e-10, True).numpy()
        np.testing.assert_allclose(got, g[f"{tag}_W_scalar"], rtol=0, atol=1e-6)


def test_singcutoff_on_rank_deficient_design(golden_dir):
    """The oracle truncates like the reference (ridge_utils.py:44-63): scores / weights / a full fit of the reference
    on a rank-25 design with p = 40 for singcutoff 1e-30, 1e-10 (nothing dropped) and 1e-6 (7 of the 15 noise-level
    singular values dropped)."""
    g = load(golden_dir, "singcutoff.npz")
    X, Y = torch.tensor(g["X"], dtype=torch.float32), torch.tensor(g["Y"], dtype=torch.float32)
    tr, va, alphas = g["tr"], g["va"], g["alphas"]
    assert [int(g[f"kept_{i}"]) for i in range(3)] == [40, 40, 33]
    for i, sc in enumerate(g["cutoffs"]):
        assert ridge.thin_svd(X[tr], float(sc))[1].numel() == int(g[f"kept_{i}"])
        for na in (1, 0):
            got = ridge.alpha_sweep_scores(X[tr], X[va], Y[tr], Y[va], alphas, float(sc), True, bool(na)).numpy()
            np.testing.assert_allclose(got, g[f"scores_{i}_norm{na}"], rtol=0, atol=1e-6)
            got = ridge.ridge_weights(X[tr], Y[tr], 3.0, float(sc), bool(na)).numpy()
            np.testing.assert_allclose(got, g[f"W_{i}_norm{na}"], rtol=0, atol=1e-6)
        m, W, a = nested_cv.fit_predict(g["X"], g["Y"], alphas=alphas, folding_type="kfold", n_outer_folds=3,
                                        n_inner_folds=3, singcutoff=float(sc))
        np.testing.assert_allclose(a, g[f"fit_{i}_alphas"], rtol=1e-6)
        np.testing.assert_allclose(W, g[f"fit_{i}_W"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(np.asarray(m["correlations"]), g[f"fit_{i}_correlations"], rtol=0, atol=2e-6)


def test_alpha_zero_and_biting_singcutoff(golden_dir):
    """alpha = 0 (D = 1 / S on the kept directions, ridge_regression.py:56,117) and cutoffs that really drop singular
    values (ridge_utils.py:44-63), as the reference computes them: rank-deficient design (18 of 25 real directions kept
    at the larger cutoff), wide full-rank design (alpha = 0 interpolates), tall design (alpha = 0 = least squares)."""
    g = load(golden_dir, "spectral.npz")
    X, Y = torch.tensor(g["rd_X"], dtype=torch.float32), torch.tensor(g["rd_Y"], dtype=torch.float32)
    tr, va, alphas = g["rd_tr"], g["rd_va"], g["rd_alphas"]
    assert alphas[0] == 0.0 and [int(g["rd_kept_0"]), int(g["rd_kept_1"])] == [25, 18]
    for i, sc in enumerate(g["rd_cutoffs"]):
        assert ridge.thin_svd(X[tr], float(sc))[1].numel() == int(g[f"rd_kept_{i}"])
        for na in (1, 0):
            got = ridge.alpha_sweep_scores(X[tr], X[va], Y[tr], Y[va], alphas, float(sc), True, bool(na)).numpy()
            np.testing.assert_allclose(got, g[f"rd_scores_{i}_norm{na}"], rtol=0, atol=2e-6)
            for a_ in (0, 3):
                got = ridge.ridge_weights(X[tr], Y[tr], float(a_), float(sc), bool(na)).numpy()
                np.testing.assert_allclose(got, g[f"rd_W_{i}_norm{na}_a{a_}"], rtol=0, atol=2e-6)
        for single in (0, 1):
            m, W, a = nested_cv.fit_predict(g["rd_X"], g["rd_Y"], alphas=alphas, folding_type="kfold", n_outer_folds=3,
                                            n_inner_folds=3, singcutoff=float(sc), single_alpha=bool(single))
            tag = f"rd_fit_{i}_s{single}"
            np.testing.assert_allclose(a, g[tag + "_alphas"], rtol=1e-6)
            np.testing.assert_allclose(W, g[tag + "_W"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(np.asarray(m["correlations"]), g[tag + "_correlations"], rtol=0, atol=2e-6)
    for tag in ("wide", "tall"):
        Xc, Yc, al = g[f"{tag}_X"], g[f"{tag}_Y"], g[f"{tag}_alphas"]
        Xt, Yt = torch.tensor(Xc, dtype=torch.float32), torch.tensor(Yc, dtype=torch.float32)
        tr, va = g[f"{tag}_tr"], g[f"{tag}_va"]
        for na in (1, 0):
            got = ridge.alpha_sweep_scores(Xt[tr], Xt[va], Yt[tr], Yt[va], al, 1e-10, True, bool(na)).numpy()
            np.testing.assert_allclose(got, g[f"{tag}_scores_norm{na}"], rtol=0, atol=2e-6)
            got = ridge.alpha_sweep_scores(Xt[tr], Xt[va], Yt[tr], Yt[va], al, 1e-10, False, bool(na)).numpy()
            np.testing.assert_allclose(got, g[f"{tag}_scores_r2_norm{na}"], rtol=0, atol=2e-6)
        got = ridge.ridge_weights(Xt[tr], Yt[tr], 0.0, 1e-10, True).numpy()
        np.testing.assert_allclose(got, g[f"{tag}_W_a0"], rtol=0, atol=2e-6)
        m, W, a = nested_cv.fit_predict(Xc, Yc, alphas=al, folding_type="kfold", n_outer_folds=3, n_inner_folds=3,
                                        singcutoff=1e-10)
        np.testing.assert_allclose(a, g[f"{tag}_fit_alphas"], rtol=1e-6)
        np.testing.assert_allclose(W, g[f"{tag}_fit_W"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(np.asarray(m["correlations"]), g[f"{tag}_fit_correlations"], rtol=0, atol=2e-6)
        m, W, a = nested_cv.fit_predict(Xc[:160], Yc[:160], X_test=Xc[160:], y_test=Yc[160:], alphas=al,
                                        folding_type="kfold", n_inner_folds=3, singcutoff=1e-10)
        np.testing.assert_allclose(a, g[f"{tag}_tt_alphas"], rtol=1e-6)
        np.testing.assert_allclose(W, g[f"{tag}_tt_W"], rtol=0, atol=5e-6)


def _check_fit(g, spec, name):
    s = spec[name]
    X, Y = g[f"X_{s['data']}"], g[f"Y_{s['data']}"]
    random.seed(s["random_seed"])
    np.random.seed(s["random_seed"])
    kw = dict(s["kwargs"], alphas=g["alphas"])
    if s["train_test"]:
        m, W, a = nested_cv.fit_predict(X[:180], Y[:180], X_test=X[180:], y_test=Y[180:], **kw)
    else:
        m, W, a = nested_cv.fit_predict(X, Y, **kw)
    pre = name + "__"
    assert str(W.dtype) == s["types"]["W"] and str(a.dtype) == s["types"]["alphas"]
    assert type(m["correlations"][0]).__name__ == s["types"]["corr_elem"]
    np.testing.assert_allclose(a, g[pre + "alphas"], rtol=1e-6)
    np.testing.assert_allclose(W, g[pre + "W"], rtol=0, atol=2e-6)
    keys = sorted(k[len(pre) + 2:] for k in g.files if k.startswith(pre + "m_"))
    assert keys == sorted(m.keys()), name
    for k in keys:
        want = g[pre + "m_" + k]
        got = np.asarray(m[k])
        if want.dtype == bool or want.dtype.kind in "iu":
            assert np.array_equal(got, want), (name, k)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6, err_msg=f"{name}:{k}")


def test_full_fits(golden_dir):
    g = load(golden_dir, "fits.npz")
    spec = json.load(open(os.path.join(golden_dir, "fits.json")))
    for name in spec:
        _check_fit(g, spec, name)


def test_harness(golden_dir):
    g = load(golden_dir, "harness.npz")
    assert np.array_equal(harness.zs(g["zs_in"]), g["zs_out"])
    stories = ["s0", "s1", "s2", "s3"]
    feats = harness.delay_all({s: g[f"feat_{s}"] for s in stories}, [1, 2, 3, 4])
    brain = {s: g[f"brain_{s}"] for s in stories}
    trimming = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0,
                "train_targets_end": None, "test_features_start": 50, "test_features_end": -5,
                "test_targets_start": 40, "test_targets_end": None}
    d = harness.train_test_matrices(feats, brain, trimming)
    for k in ("Rstim", "Rresp", "Pstim", "Presp"):
        assert np.array_equal(d[k], g[k]), k
    d = harness.concatenated_matrices(feats, brain, stories,
                                      {"features_start": 10, "features_end": -5, "targets_start": 3, "targets_end": -12})
    assert np.array_equal(d["X"], g["cat_X"]) and np.array_equal(d["Y"], g["cat_Y"])


def test_bh_fdr_known_answers():
    # hand-checkable: n=5, alpha=0.05 -> thresholds .01 .02 .03 .04 .05
    p = np.array([0.04, 0.001, 0.03, 0.5, 0.011])
    rej, adj = stats.bh_fdr(p, 0.05)
    # sorted: .001 .011 .03 .04 .5 ; p*n/i: .005 .0275 .05 .05 .5 ; largest i with p<=i/n*a: i=4 (.04<=.04)
    assert rej.tolist() == [True, True, True, False, True]
    np.testing.assert_allclose(adj, [0.05, 0.005, 0.05, 0.5, 0.0275], atol=1e-15)
    rej, adj = stats.bh_fdr(np.array([0.9, 0.8, 1.0]), 0.05)
    assert not rej.any() and np.allclose(adj, [1.0, 1.0, 1.0])
    rej, adj = stats.bh_fdr(np.array([0.2, 0.2, 0.2, 0.2]), 0.25)   # ties
    assert rej.all() and np.allclose(adj, 0.2)


def test_fisher_known_answers():
    from scipy.stats import chi2
    got = stats.fisher_combine([[0.01, 1.0], [0.2, 1.0], [0.5, 1.0]])
    assert got[1] == 1.0
    assert abs(got[0] - chi2.sf(-2 * np.log(0.01 * 0.2 * 0.5), 6)) < 1e-15


def test_oracle_at_a_config_shape_against_the_reference(golden_dir):
    """The oracle at a BASELINE config's own shape (cfg4: T = 2226, p = 3072, 20 alphas, 5 x 5 K-folds, 64 of the
    fixture's 256 voxels) against what the REFERENCE returned there (tests/golden/configs.npz, made by
    make_golden_configs.py): per-fold alphas identical, per-fold test correlations and score tables to fp32 rounding.
    (~25 s: thirty SVDs of 1780 x 3072; the other configs' fixtures are held to by the HIP path in the GPU tests.)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _config_problems as cp
    import _fixtures as fx
    import oracle.nested_cv as onc
    g, spec = fx.load(golden_dir)
    X, Y, kw = cp.matrix_problem("cfg4")
    fx.check_inputs(g, "cfg4__checks", X, Y)
    nv = 64
    detail = {}
    m, W, a = onc.fit_predict(X, Y[:, :nv], detail=detail, **kw)
    assert np.array_equal(np.asarray(detail["fold_alphas"]), g["cfg4__fold_alphas"][:, :nv])
    np.testing.assert_allclose(detail["fold_mean_scores"], g["cfg4__fold_tables"][:, :, :nv], rtol=0, atol=2e-6)
    np.testing.assert_allclose(detail["fold_scores"], g["cfg4__fold_r"][:, :nv], rtol=0, atol=2e-6)
    np.testing.assert_allclose(W[:, :32], g["cfg4__W"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(np.asarray(m["correlations"]), g["cfg4__correlations"][:nv], rtol=0, atol=1e-6)


def test_bh_fdr_against_scipys_implementation():
    """statsmodels' ``fdrcorrection`` (nested_cv.py:158,263,282) is absent here, so the Benjamini-Hochberg step is restated
    (parity unpinned at that boundary) -- but scipy >= 1.11 ships an independent implementation of the same
    published procedure, ``scipy.stats.false_discovery_control(ps, method="bh")``: the oracle's adjusted p-values must be
    its, and the rejection mask ``p_adj <= alpha`` (what ``fdrcorrection(method="indep")`` returns).  Ties, ones, a single
    value, 5 000 values."""
    scipy_stats = pytest.importorskip("scipy.stats")
    if not hasattr(scipy_stats, "false_discovery_control"):
        pytest.skip("scipy < 1.11")
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 100, 5000):
        p = rng.uniform(0, 1, n) ** 3
        p[rng.integers(0, n, max(1, n // 10))] = 1.0
        if n > 3:
            p[1] = p[2]
        for alpha in (0.05, 0.2):
            rej, padj = stats.bh_fdr(p, alpha)
            ref = scipy_stats.false_discovery_control(p, method="bh")
            np.testing.assert_allclose(padj, ref, rtol=1e-13, atol=0)
            assert np.array_equal(rej, ref <= alpha)
