"""The mean-operator refit (round 6, engine/mean_refit.py; FitOptions.mean_operator_refit): the mean weights of a
cross-validated fit (nested_cv.py:293-296) from the MEAN of the folds' refit operators -- one contraction of depth T per group
of voxels with the same alpha in every fold -- instead of the folds' own weight products.  Same alphas, same test scores
(bit for bit: they come from the folds' test-row contractions either way), weights equal to fp32 rounding, everything within
the oracle's tolerances; the operator image kernel against numpy; the fallback when every voxel has an alpha tuple of its own.
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
    W[:, rng.uniform(size=V) < noise_cols] = 0.0
    return X, X @ W + rng.standard_normal((T, V))


def _image_to_matrix(img, rs_inv, rows, K):
    """Undo the tiled fp16 hi/lo layout of lc_split_rows_f16 (tile, K-tile, plane, k-group, row, 8 halves) on the host."""
    rows_pad = -(-rows // 256) * 256
    a = img.reshape(rows_pad // 256, K // 16, 2, 2, 256, 8).astype(np.float64)      # [tile][kt][plane][kg][row][8]
    full = a[:, :, 0] + a[:, :, 1]                                                   # hi + lo: [tile][kt][kg][row][8]
    m = full.transpose(0, 3, 1, 2, 4).reshape(rows_pad, K)                           # [tile, row][kt, kg, 8]
    return (m * rs_inv[:, None].astype(np.float64))[:rows]


def test_mean_operator_image_against_numpy(lc):
    """lc_mean_operator_image_f16: scale * sum_f scatter_f(M_f) as fp16 hi + lo with per-row power-of-two scales -- against the
    float32 sum formed in numpy (22-bit split: 2^-21 of the row maximum), incl. unaligned runs, rows that no fold trains on,
    operators of different widths and a row count that is not a multiple of 256."""
    from litcoder_core_amd import ops
    dev = ops.device()
    rng = np.random.default_rng(5)
    for (rows, T, nf) in ((300, 1000, 5), (96, 130, 3), (513, 640, 2)):
        K = -(-T // 32) * 32
        folds = np.array_split(rng.permutation(T) if nf == 3 else np.arange(T), nf)
        mats, maps, ref = [], [], np.zeros((rows, K), dtype=np.float32)
        scale = np.float32(1.0 / nf)
        for f in range(nf):
            tr = np.sort(np.concatenate([folds[g] for g in range(nf) if g != f]))
            if f == 1:
                tr = tr[tr != tr[len(tr) // 2]]                       # a hole: rows around it are unaligned runs
            N_o = -(-len(tr) // 64) * 64
            M = (rng.standard_normal((rows + 32, N_o)) * (10.0 ** rng.uniform(-3, 3, (rows + 32, 1)))).astype(np.float32)
            M[:, len(tr):] = 0.0
            mp = np.full(K, -1, dtype=np.int32)
            mp[tr] = np.arange(len(tr), dtype=np.int32)
            ref[:, tr] = ref[:, tr] + M[:rows, : len(tr)]             # float32 adds, folds in order
            mats.append(torch.from_numpy(M).to(dev))
            maps.append(ops.upload(mp, dev))
        ref = ref * scale
        rows_pad = -(-rows // 256) * 256
        img = torch.empty(rows_pad * K * 2, dtype=torch.float16, device=dev)
        rs = torch.empty(rows_pad, dtype=torch.float32, device=dev)
        ops.mean_operator_image([m[:rows] for m in mats], maps, float(scale), rows, K, img, rs)
        got = _image_to_matrix(img.cpu().numpy(), rs.cpu().numpy(), rows, K)
        rowmax = np.abs(ref).max(axis=1, keepdims=True)
        err = np.abs(got - ref.astype(np.float64)) / np.maximum(rowmax, 1e-30)
        assert err.max() <= 2.0 ** -20, (rows, T, nf, err.max())
        # the same image as lc_split_rows_f16 makes of the matrix itself (bit for bit: the same scale, the same split)
        img2 = torch.empty_like(img)
        rs2 = torch.empty_like(rs)
        ops.split_rows_f16(torch.from_numpy(ref).to(dev), rows, K, img2, rs2)
        assert torch.equal(rs[:rows], rs2[:rows])
        assert torch.equal(img, img2)
        # several images in one launch (lc_mean_operator_images_f16): bit for bit the single launches', at the slots asked for
        alt = [[m[:rows] for m in mats], [(m * 0.5)[:rows] for m in mats], [torch.flip(m, dims=(0,))[:rows].contiguous() for m in mats]]
        singles = []
        for ms in alt:
            i1, r1 = torch.empty_like(img), torch.empty_like(rs)
            ops.mean_operator_image(ms, maps, float(scale), rows, K, i1, r1)
            singles.append((i1, r1))
        slots = [2, 0, 3]
        imgs = torch.zeros(4 * rows_pad * K * 2, dtype=torch.float16, device=dev)
        rss = torch.zeros(4 * rows_pad, dtype=torch.float32, device=dev)
        ops.mean_operator_images(alt, slots, maps, float(scale), rows, K, imgs, rss)
        for (i1, r1), sl in zip(singles, slots):
            assert torch.equal(imgs[sl * rows_pad * K * 2:(sl + 1) * rows_pad * K * 2], i1)
            assert torch.equal(rss[sl * rows_pad:sl * rows_pad + rows], r1[:rows])
        assert not imgs[rows_pad * K * 2:2 * rows_pad * K * 2].any()          # (slot 1: nobody's)


@pytest.mark.parametrize("case", ["kfold", "chunked_norm_x", "single_alpha", "single_alpha_panels", "panels", "own_choice"])
def test_mean_operator_fit_equals_fold_by_fold_fit(lc, case):
    """The same fit with and without the mean-operator refit: alphas, correlations, p-values identical (the folds' test-row
    contractions are untouched), weights equal to fp32 rounding of the two summation orders."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng({"kfold": 1, "chunked_norm_x": 2, "single_alpha": 3, "panels": 4, "own_choice": 5, "single_alpha_panels": 6}[case])
    T, p, V = 600, 200, 1900
    if case == "own_choice":                                   # (the cost rule decides which tuples get an operator: wide enough
        T, p, V = 640, 330, 24000                               #  for some to pay, noise voxels for the others)
    X, Y = _problem(rng, T, p, V, **(dict(noise_cols=0.2, signal=2.5) if case == "own_choice" else {}))
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=np.logspace(-1, 5, 13))
    mkw = {}
    if case == "chunked_norm_x":
        kw.update(folding_type="chunked", chunk_length=25, normalize_features=True, n_outer_folds=5)
    if case in ("single_alpha", "single_alpha_panels"):
        kw.update(single_alpha=True)
    if case in ("panels", "single_alpha_panels"):              # (single alpha + host panels: the joint choice over the ranges,
        mkw.update(panel_cols=512)                               #  early panels refitted with the alpha they guess)
    import random
    random.seed(7)                                             # (chunked folds shuffle with Python's global generator)
    m0 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(mean_operator_refit=False), **mkw)
    out0 = m0.fit_predict(X, Y, **kw)
    random.seed(7)
    opt1 = (FitOptions(mean_operator_min_cols=0, mean_operator_min_share=0.0) if case == "own_choice"
            else FitOptions(mean_operator_min_cols=0, mean_operator_cost_ratio=1e9))
    m1 = NestedCVModel("ridge_regression", precision="f16x3", options=opt1, **mkw)
    out1 = m1.fit_predict(X, Y, **kw)
    mo = m1.last_fit.get("mean_operator")
    assert mo and mo["on"] and mo["ranges"] >= 1, mo
    if not case.startswith("single_alpha"):
        assert mo["voxels"] > 0 and mo["other_voxels"] > 0, mo    # both routes ran (30 % noise voxels: tuples of their own)
    assert not (m0.last_fit.get("mean_operator") or {}).get("on")
    if case in ("panels", "single_alpha_panels"):
        assert len(m1.last_fit["panels"]) >= 3 and mo["ranges"] >= 2, (m1.last_fit["panels"], mo)
    np.testing.assert_array_equal(out1[2], out0[2])
    for k in ("correlations", "p_values"):
        np.testing.assert_array_equal(np.asarray(out1[0][k]), np.asarray(out0[0][k]))
    W0, W1 = np.asarray(out0[1], dtype=np.float64), np.asarray(out1[1], dtype=np.float64)
    assert np.abs(W1 - W0).max() <= 2e-6 * np.abs(W0).max(), np.abs(W1 - W0).max() / np.abs(W0).max()


def test_mean_operator_fit_matches_oracle(lc):
    """... and against the CPU oracle (float64 truth of the reference's algorithm), like every other fit of the suite."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    from oracle import nested_cv as onc
    from _oracle_check import assert_matches_oracle
    rng = np.random.default_rng(21)
    T, p, V = 500, 160, 700
    X, Y = _problem(rng, T, p, V, noise_cols=0.2, signal=0.6)
    kw = dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=4, alphas=np.logspace(-1, 4, 11))
    m = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(mean_operator_min_cols=0, mean_operator_cost_ratio=1e9))
    ours = m.fit_predict(X, Y, **kw)
    assert m.last_fit["mean_operator"]["ranges"] >= 1, m.last_fit["mean_operator"]
    detail = {}
    orc = onc.fit_predict(X.astype(np.float32), Y.astype(np.float32), detail=detail, **kw)
    assert_matches_oracle(lc, m, ours, orc, detail, X, Y, kw, "mean_operator_refit")


def test_mean_operator_layouts_agree_bit_for_bit(lc):
    """Voxel panels and voxel ranges do not change a voxel's weights: its mean operator depends on ITS alpha tuple alone, and
    every V-wide kernel keeps a voxel's arithmetic inside its own column."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(31)
    T, p, V = 560, 180, 2300
    X, Y = _problem(rng, T, p, V, noise_cols=0.1, signal=0.5)
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=np.logspace(-1, 5, 9))
    outs = []
    for pc in (0, 256, 768):
        m = NestedCVModel("ridge_regression", precision="f16x3", panel_cols=pc,
                          options=FitOptions(mean_operator_min_cols=0, mean_operator_cost_ratio=1e9, mean_operator_max_tuples=10 ** 6))
        outs.append(m.fit_predict(X, Y, **kw))
        assert m.last_fit["mean_operator"]["other_voxels"] == 0, m.last_fit["mean_operator"]
    for o in outs[1:]:
        np.testing.assert_array_equal(o[2], outs[0][2])
        np.testing.assert_array_equal(np.asarray(o[1]), np.asarray(outs[0][1]))
        np.testing.assert_array_equal(np.asarray(o[0]["correlations"]), np.asarray(outs[0][0]["correlations"]))


def test_mean_operator_leaves_scattered_tuples_to_the_folds(lc):
    """Pure-noise targets: the folds' alphas of a voxel are unrelated, nearly every voxel has an alpha tuple of its own and no
    tuple pays for an operator image -- every voxel takes the folds' own weight products (formed when the range's last fold
    has chosen), bit for bit the fit without the option."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(41)
    T, p, V = 480, 150, 1500
    X = rng.standard_normal((T, p))
    Y = rng.standard_normal((T, V))
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=np.logspace(-1, 6, 15))
    m0 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(mean_operator_refit=False))
    out0 = m0.fit_predict(X, Y, **kw)
    m1 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(mean_operator_min_cols=0, mean_operator_min_share=0.0))
    out1 = m1.fit_predict(X, Y, **kw)
    mo = m1.last_fit["mean_operator"]
    assert mo["on"] and mo["voxels"] == 0 and mo["other_voxels"] == V, mo
    np.testing.assert_array_equal(out1[2], out0[2])
    np.testing.assert_array_equal(np.asarray(out1[1]), np.asarray(out0[1]))
    np.testing.assert_array_equal(np.asarray(out1[0]["correlations"]), np.asarray(out0[0]["correlations"]))


def test_mean_operator_with_a_wide_target_column(lc):
    """A target column dominated by an outlier (the f32 side panel, round 5) in a fit on the mean-operator refit: the column's
    weights are the folds' exact-f32 weights averaged fold by fold -- bit for bit the fit without the option --, the other
    voxels' weights equal to fp32 rounding; alphas and test scores identical."""
    from litcoder_core_amd import NestedCVModel, ops
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(51)
    T, p, V = 600, 200, 1900
    X, Y = _problem(rng, T, p, V, noise_cols=0.1, signal=0.8)
    Y[7, 1234] = 1e6
    Y[300, 17] = -3e5
    kw = dict(folding_type="kfold", n_outer_folds=4, n_inner_folds=3, alphas=np.logspace(-1, 5, 13))
    dev = ops.device()
    dX = ops.upload_f32(X, ops.pad_to(p, 32), dev)
    dY = ops.upload_f32(Y, ops.pad_to(V, 128), dev)
    outs = []
    for opt in (FitOptions(mean_operator_refit=False), FitOptions(mean_operator_min_cols=0, mean_operator_cost_ratio=1e9)):
        m = NestedCVModel("ridge_regression", options=opt)
        outs.append(m.fit_predict_device(dX, dY, p, V, weights_on_host=True, **kw))
        assert m.last_fit.get("side_panel_cols") == 2 and m.last_fit["precision"] == "f16x3", m.last_fit
    assert outs and m.last_fit["mean_operator"]["on"] and m.last_fit["mean_operator"]["voxels"] > 0, m.last_fit["mean_operator"]
    (m0, W0, a0), (m1, W1, a1) = outs
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_array_equal(np.asarray(m1["correlations"]), np.asarray(m0["correlations"]))
    W0, W1 = np.asarray(W0), np.asarray(W1)
    np.testing.assert_array_equal(W1[:, [17, 1234]], W0[:, [17, 1234]])
    assert np.abs(W1.astype(np.float64) - W0).max() <= 2e-6 * np.abs(W0).max()


@pytest.mark.parametrize("layout", ["resident", "host_panels"])
def test_mean_operator_is_dropped_when_the_first_two_folds_say_so(lc, layout):
    """Weak-signal targets: after the first two folds' choices the expected share of voxels in alpha tuples that pay is far below
    mean_operator_min_share -- the option is dropped, the steps gone by form their own weight products at once, the later folds
    theirs as they always did: bit for bit the fit without the option (host panels: the first fold's steps were voxel ranges)."""
    from litcoder_core_amd import NestedCVModel
    from litcoder_core_amd.engine.common import FitOptions
    rng = np.random.default_rng(61)
    T, p, V = 520, 170, 2100
    X, Y = _problem(rng, T, p, V, noise_cols=0.5, signal=0.25)
    kw = dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=3, alphas=np.logspace(-1, 6, 15))
    mkw = dict(panel_cols=512) if layout == "host_panels" else {}
    m0 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(mean_operator_refit=False), **mkw)
    out0 = m0.fit_predict(X, Y, **kw)
    m1 = NestedCVModel("ridge_regression", precision="f16x3", options=FitOptions(mean_operator_min_cols=0), **mkw)
    out1 = m1.fit_predict(X, Y, **kw)
    mo = m1.last_fit["mean_operator"]
    assert not mo["on"] and mo["expected_share"] < FitOptions().mean_operator_min_share and mo["ranges"] == 0, mo
    np.testing.assert_array_equal(out1[2], out0[2])
    np.testing.assert_array_equal(np.asarray(out1[1]), np.asarray(out0[1]))
    np.testing.assert_array_equal(np.asarray(out1[0]["correlations"]), np.asarray(out0[0]["correlations"]))
