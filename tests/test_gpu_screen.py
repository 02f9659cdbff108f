"""The two-precision inner CV (round 6; FitOptions.screen_inner): a screening pass with ONE fp16 MFMA per
product decides the alpha of every voxel whose best two alphas are clearly apart, the others are scored again with the
three-MFMA products.  nested_cv.py:408-411 only takes each voxel's argmax of the fold-mean scores, so the fit must come out
as the three-MFMA fit does -- alphas identical, hence weights and test scores bit for bit -- and both must match the oracle.
Every test here needs a real MI355X:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lc():
    import litcoder_core_amd as pkg
    from litcoder_core_amd import ops
    ops.device()                     # raises (no CPU fallback) when there is no gfx950
    return pkg


def _problem(rng, T, p, V, noise_cols=0.3, signal=0.4):
    X = rng.standard_normal((T, p))
    W = rng.standard_normal((p, V)) * (signal / np.sqrt(p))
    W[:, rng.uniform(size=V) < noise_cols] = 0.0                 # pure-noise voxels: flat score curves, near-ties
    return X, X @ W + rng.standard_normal((T, V))


def test_undecided_columns_kernel_against_numpy(lc):
    """lc_undecided_cols: the list (ascending, -1 behind), the counts and the overflow flag against numpy, incl. the offset
    scaling (kappa = rms / std of the validation rows), all-zero columns (decided), non-finite sums (undecided), a capacity
    smaller than the number found, and widths that are not multiples of the block."""
    from litcoder_core_amd import ops
    dev = ops.device()
    rng = np.random.default_rng(11)
    for (A, V, ld, cap, tau) in ((20, 5000, 5120, 1024, 0.02), (7, 257, 384, 256, 0.05), (3, 999, 1024, 256, 0.3), (1, 300, 384, 256, 0.1)):
        sc = rng.uniform(-1, 1, (A, ld)).astype(np.float32)
        sc[:, 5] = 0.0                                           # constant column: decided
        sc[0, 6] = np.inf                                        # non-finite: undecided
        if A > 1:
            sc[:, 7] = 0.25                                      # exact tie of non-zero sums: undecided
            sc[:, 8] = -3.0
            sc[1, 8] = 2.0                                       # clear winner: decided
        mean = rng.uniform(-3, 3, ld).astype(np.float32)
        std = rng.uniform(0.5, 2, ld).astype(np.float32)
        std[9] = 0.0
        ystat = np.stack([mean, std, std * std]).astype(np.float32)
        for use_stat in (False, True):
            lst, cnt = ops.undecided_cols(torch.from_numpy(sc).to(dev), A, V, tau, torch.from_numpy(ystat).to(dev) if use_stat else None, cap)
            lst, cnt = lst.cpu().numpy(), cnt.cpu().numpy()
            srt = np.sort(sc[:, :V].astype(np.float32), axis=0)
            kappa = np.ones(V, dtype=np.float32)
            if use_stat:
                with np.errstate(divide="ignore", invalid="ignore"):
                    k = np.sqrt(mean[:V] * mean[:V] + ystat[2, :V]) / std[:V]
                kappa = np.where((std[:V] > 0) & (k >= 1) & np.isfinite(k), k, 1.0).astype(np.float32)
            with np.errstate(invalid="ignore"):
                gap = (srt[-1] - srt[-2]) if A > 1 else np.full(V, np.inf, dtype=np.float32)
                und = ~(gap >= np.float32(tau) * kappa) if A > 1 else np.zeros(V, dtype=bool)
            und |= ~np.isfinite(sc[:, :V]).all(0)
            und &= ~(sc[:, :V] == 0).all(0)
            want = np.nonzero(und)[0]
            assert cnt[1] == len(want) and cnt[0] == min(len(want), cap) and cnt[2] == int(len(want) > cap)
            np.testing.assert_array_equal(lst[:cnt[0]], want[:cap])
            assert (lst[cnt[0]:] == -1).all()
            assert not und[5] and und[6]
            if A > 1:
                assert und[7] and not und[8]


def test_screening_scores_are_close_and_the_panel_is_bitwise(lc):
    """The screening pass' score table against the three-MFMA table (same operators, same images): close (the error of
    11-bit operands averaged over the rows), never exact garbage; and the refinement's panel -- the same sweeps on a
    gathered copy of some columns -- reproduces the three-MFMA scores of those columns bit for bit, wherever they sit in the
    panel (every V-wide kernel keeps a voxel's arithmetic inside its own column)."""
    from litcoder_core_amd.nested_cv import RidgeCVEngine
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(5)
    T, p, V = 640, 200, 1500
    X, Y = _problem(rng, T, p, V)
    alphas = np.logspace(-1, 5, 13)
    tr_o = np.r_[0:128, 256:640]
    inner = [(np.delete(tr_o, np.s_[k * 128:(k + 1) * 128]), tr_o[k * 128:(k + 1) * 128]) for k in range(4)]
    e3 = RidgeCVEngine(X, Y, alphas, True, True, False, False, precision="f16x3")
    s3, _ = e3._alpha_scores(e3.K, e3.dY, inner)
    e1 = RidgeCVEngine(X, Y, alphas, True, True, False, False, precision="f16x3", options=FitOptions(screen_inner=True, screen_tau=0.0))
    e1.argmax_only = True                                        # (what NestedCVModel's driver tells its engine)
    s1, _ = e1._alpha_scores(e1.K, e1.dY, inner)
    assert e1.info["screen_terms"] == 1 and e3.info["screen_terms"] == 3
    # the two forms of the screening sweeps -- 4-wave workgroups on 256 x 128 tiles, two per CU (k_sweep_hi2, the default) and
    # the HI2 mode of the 8-wave kernel -- issue the same products in the same order: the same table bit for bit
    e1b = RidgeCVEngine(X, Y, alphas, True, True, False, False, precision="f16x3",
                        options=FitOptions(screen_inner=True, screen_tau=0.0, screen_two_workgroups=False))
    e1b.argmax_only = True
    s1b, _ = e1b._alpha_scores(e1b.K, e1b.dY, inner)
    assert torch.equal(s1, s1b)
    d = (s1[:, :V] - s3[:, :V]).abs().cpu().numpy() / len(inner)
    assert 1e-8 < d.max() < 3e-4 and np.sqrt((d ** 2).mean()) < 3e-5, (d.max(), np.sqrt((d ** 2).mean()))
    # tau = +inf-like: every voxel undecided -> with a panel as wide as the range the table IS the three-MFMA table
    eall = RidgeCVEngine(X, Y, alphas, True, True, False, False, precision="f16x3",
                         options=FitOptions(screen_inner=True, screen_tau=1e3, screen_panel_cols=2048))
    eall.argmax_only = True
    sall, _ = eall._alpha_scores(eall.K, eall.dY, inner)
    assert torch.equal(sall[:, :V], s3[:, :V])
    # a realistic tau: refined columns carry the three-MFMA bits, the others the screening pass' bits
    emix = RidgeCVEngine(X, Y, alphas, True, True, False, False, precision="f16x3",
                         options=FitOptions(screen_inner=True, screen_tau=2e-3 * np.sqrt(512.0), screen_panel_cols=1024))
    emix.argmax_only = True
    smix, _ = emix._alpha_scores(emix.K, emix.dY, inner)
    eq3 = (smix[:, :V] == s3[:, :V]).all(0).cpu().numpy()
    eq1 = (smix[:, :V] == s1[:, :V]).all(0).cpu().numpy()
    assert (eq3 | eq1).all() and 10 < eq3.sum() < V and eq1.sum() > 10
    top = torch.topk(s1[:, :V], 2, dim=0).values
    und = ((top[0] - top[1]) < 2e-3 * len(inner)).cpu().numpy()
    assert eq3[und].all(), "every undecided voxel must have been scored again"


@pytest.mark.parametrize("case", ["cv", "cv_offset", "traintest", "small_panel"])
def test_screened_fit_equals_the_three_mfma_fit(lc, case):
    """Full fits: FitOptions.screen_inner on / off -> the same alphas in every fold, hence the same weights, correlations and
    p-values bit for bit.  cv_offset: targets riding on large per-voxel offsets (the rms / std scaling of the gap sends them
    to the refinement); small_panel: a forced 256-column panel that cannot hold the undecided voxels -- the overflow path
    (the range is scored again) must give the same fit."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng({"cv": 1, "cv_offset": 2, "traintest": 3, "small_panel": 4}[case])
    T, p, V = 900, 300, 3000
    X, Y = _problem(rng, T, p, V, noise_cols=0.05)
    if case == "cv_offset":
        Y = Y + rng.uniform(-300, 300, V)[None, :] * (rng.uniform(size=V) < 0.5)
    kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=4, alphas=np.logspace(-1, 6, 15))
    opts = dict(screen_inner=True)
    if case == "small_panel":
        opts.update(screen_tau=0.1, screen_panel_cols=256)
    extra = {}
    if case == "traintest":
        Xt, Yt = X[700:], Y[700:]
        X, Y = X[:700], Y[:700]
        extra = dict(X_test=Xt, y_test=Yt)
        kw.pop("n_outer_folds")
    m3 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(screen_inner=False))
    out3 = m3.fit_predict(X, Y, **extra, **kw)
    m1 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(**opts))
    out1 = m1.fit_predict(X, Y, **extra, **kw)
    assert m1.last_fit["screen_terms"] == 1 and m3.last_fit["screen_terms"] == 3
    # (how many voxels the screening leaves undecided depends on how flat the data's score curves are -- 1 % at cfg2, 20-40 %
    # on small weak-signal problems like this one: tools/screen_probe.py -- the first steps' panels may overflow, later ones
    # are sized from the shares seen; the FIT must not depend on any of that)
    assert 0 < m1.last_fit["undecided"] < m1.last_fit["screened"], m1.last_fit
    if case == "small_panel":
        assert m1.last_fit.get("screen_overflows", 0) >= 1
    for a3, a1 in zip(m3.last_fold_alphas, m1.last_fold_alphas):
        np.testing.assert_array_equal(a1, a3)
    np.testing.assert_array_equal(out1[2], out3[2])
    np.testing.assert_array_equal(out1[1], out3[1])
    for key in ("correlations", "p_values", "corrected_p_values", "significant_mask"):
        if key in out3[0]:
            np.testing.assert_array_equal(np.asarray(out1[0][key]), np.asarray(out3[0][key]), err_msg=key)
    assert out1[0]["median_score"] == out3[0]["median_score"]


def test_screening_switches_itself_off_on_flat_score_curves(lc):
    """Pure-noise voxels have flat score curves: on the plateau of the large alphas the scores of neighbouring alphas agree
    to fp32 rounding, the argmax there is rounding noise in ANY arithmetic (the reference's included), and the screening
    pass cannot decide it -- such a voxel is scored again.  When most voxels are like that the second pass costs more than
    the screening saves: the engine notices (the share of undecided voxels comes back with every step's histogram) and
    scores the remaining folds on three MFMAs throughout.  The fit is the three-MFMA fit either way."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(9)
    T, p, V = 900, 300, 3000
    X, Y = _problem(rng, T, p, V, noise_cols=0.9)
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=np.logspace(-1, 8, 15))
    m3 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(screen_inner=False))
    out3 = m3.fit_predict(X, Y, **kw)
    m1 = NestedCVModel("ridge_regression", precision="f16x3")
    out1 = m1.fit_predict(X, Y, **kw)
    assert m1.last_fit.get("screen_switched_off"), m1.last_fit
    assert m1.last_fit["screened"] < 4 * V                          # not every fold went through the screening pass
    for a3, a1 in zip(m3.last_fold_alphas, m1.last_fold_alphas):
        np.testing.assert_array_equal(a1, a3)
    np.testing.assert_array_equal(out1[1], out3[1])
    np.testing.assert_array_equal(np.asarray(out1[0]["correlations"]), np.asarray(out3[0]["correlations"]))


def test_screened_fit_against_the_oracle(lc):
    """... and against the CPU oracle (the reference's algorithm), with the rule of tests/_oracle_check.py: every alpha
    that differs from the oracle's must be a near-tie of the ORACLE's own fold-mean score table."""
    import oracle.nested_cv as onc
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    from _oracle_check import assert_matches_oracle
    rng = np.random.default_rng(21)
    T, p, V = 600, 192, 700
    X, Y = _problem(rng, T, p, V, noise_cols=0.4)
    kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 6, 12))
    detail = {}
    orc = onc.fit_predict(X, Y, detail=detail, **kw)
    model = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(screen_inner=True))
    ours = model.fit_predict(X, Y, **kw)
    assert model.last_fit["screen_terms"] == 1 and model.last_fit["screened"] >= V, model.last_fit
    assert_matches_oracle(lc, model, ours, orc, detail, X, Y, kw, "screened fit vs oracle")


@pytest.mark.parametrize("case", ["dual_single", "dual_single_missed", "primal_pervoxel", "primal_single", "primal_single_traintest"])
def test_screening_single_alpha_and_primal_form(lc, case):
    """single_alpha: the ONE alpha is the argmax of the voxel MEAN of the scores (nested_cv.py:396-400) -- screened sweeps, no
    per-voxel refinement, the winner's lead checked against what the screening pass vouches for (_mean_check); a lead it cannot
    vouch for (forced here with a huge screen_tau) repeats the fit on three MFMAs.  Primal (p x p) form: the two sweeps of an
    inner fold on the screening arithmetic, the block products X'Y (which the refit shares) on three MFMAs; per-voxel alpha
    with the undecided voxels' panel through the primal sweeps.  Every fit equals the unscreened one bit for bit."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng({"dual_single": 31, "dual_single_missed": 32, "primal_pervoxel": 33, "primal_single": 34,
                                 "primal_single_traintest": 35}[case])
    primal = case.startswith("primal")
    T, p, V = (1500, 250, 2000) if primal else (900, 300, 2500)
    X, Y = _problem(rng, T, p, V, noise_cols=0.1, signal=0.5)
    single = "single" in case
    kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 6, 15), single_alpha=single)
    extra = {}
    if case.endswith("traintest"):
        extra = dict(X_test=X[1200:], y_test=Y[1200:])
        X, Y = X[:1200], Y[:1200]
        kw.pop("n_outer_folds")
    opts = dict(screen_inner=True)
    if case == "dual_single_missed":
        opts["screen_tau"] = 1e6
    m3 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(screen_inner=False))
    out3 = m3.fit_predict(X, Y, **extra, **kw)
    m1 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(**opts))
    out1 = m1.fit_predict(X, Y, **extra, **kw)
    assert m1.last_form == m3.last_form == ("primal" if primal else "dual")
    if case == "dual_single_missed":
        assert m1.last_fit.get("screen_mean_repeated") and m1.last_fit["screen_terms"] == 3, m1.last_fit
    else:
        assert m1.last_fit["screen_terms"] == 1 and m3.last_fit["screen_terms"] == 3, (m1.last_fit, m3.last_fit)
        assert not m1.last_fit.get("screen_mean_repeated")
        if single:
            assert m1.last_fit["screen_mean_lead_over_threshold"] >= 1.0
    for a3, a1 in zip(m3.last_fold_alphas, m1.last_fold_alphas):
        np.testing.assert_array_equal(a1, a3)
    np.testing.assert_array_equal(out1[2], out3[2])
    np.testing.assert_array_equal(out1[1], out3[1])
    np.testing.assert_array_equal(np.asarray(out1[0]["correlations"]), np.asarray(out3[0]["correlations"]))
    np.testing.assert_array_equal(np.asarray(out1[0]["p_values"]), np.asarray(out3[0]["p_values"]))


@pytest.mark.parametrize("case", ["single_missed_in_panels", "pervoxel_overflow_in_panels"])
def test_screening_fallbacks_with_host_panels(lc, case):
    """The two fall-backs of the screening pass while the targets are still arriving from the host in voxel panels (explicit
    256-column panels: nine of them): single_alpha with a lead the screening cannot vouch for (forced: huge screen_tau) -- the
    early panels' weights may already be on their way back when the fit is given up and repeated on three MFMAs --, and
    per-voxel alpha with a refinement panel that cannot hold a step's undecided voxels (forced: 256 columns, generous gap) --
    the step is scored again.  Either way the fit equals the unscreened one bit for bit."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(77 if case.startswith("single") else 78)
    T, p, V = 700, 300, 2300
    X, Y = _problem(rng, T, p, V, noise_cols=0.1, signal=0.5)
    single = case.startswith("single")
    kw = dict(folding_type="kfold", n_inner_folds=3, alphas=np.logspace(-1, 6, 15), single_alpha=single)
    extra = dict(X_test=X[560:], y_test=Y[560:])
    X, Y = X[:560], Y[:560]
    opts = dict(screen_inner=True, screen_tau=1e6) if single else dict(screen_inner=True, screen_tau=0.2, screen_panel_cols=256)
    m3 = NestedCVModel("ridge_regression", precision="f16x3", panel_cols=256, options=FitOptions(screen_inner=False))
    out3 = m3.fit_predict(X, Y, **extra, **kw)
    m1 = NestedCVModel("ridge_regression", precision="f16x3", panel_cols=256, options=FitOptions(**opts))
    out1 = m1.fit_predict(X, Y, **extra, **kw)
    assert len(m3.last_fit["panels"]) >= 3, m3.last_fit["panels"]            # (the plan both models start with)
    if single:
        # (last_fit describes the REPEATED fit: resident by then, one range)
        assert m1.last_fit.get("screen_mean_repeated"), m1.last_fit
    else:
        assert len(m1.last_fit["panels"]) >= 3, m1.last_fit["panels"]
        assert m1.last_fit.get("screen_overflows", 0) >= 1, m1.last_fit
    np.testing.assert_array_equal(out1[2], out3[2])
    np.testing.assert_array_equal(out1[1], out3[1])
    np.testing.assert_array_equal(np.asarray(out1[0]["correlations"]), np.asarray(out3[0]["correlations"]))
