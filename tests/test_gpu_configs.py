"""The configurations BASELINE.json names, at their own shapes (``-m gpu``).

cfg2 (the bench workload) at full size through size-independent properties; cfg3 (LeBel UTS03-like: Lanczos
downsampling of word-level GPT-2 features -> 4 FIR delays -> per-story z-scoring -> train/test fit, p = 3072), cfg4
(Narratives-like: T = 2226, V = 200 000 voxels) and cfg5 (Whisper-like: 1280-d x 6 delays = 7680 features, 32 alphas,
two feature bands) at their own shapes against the REFERENCE ITSELF on 256 voxels each: tests/golden/configs.npz holds
what /root/reference returned in the build container (tests/golden/make_golden_configs.py; inputs are rebuilt here from
seeds, tests/_config_problems.py), so no CPU SVD runs on the GPU box (round 3 re-ran the oracle here: 660 of the suite's
700 s) -- with every alpha that differs from the reference's proven to be a near-tie of the reference's own score table
(tests/_oracle_check.py).
"""
import random

import numpy as np
import pytest
import torch

from _oracle_check import assert_matches_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lc():
    import litcoder_core_amd as pkg
    from litcoder_core_amd import ops
    ops.device()
    return pkg


def _device_problem(lc, T, F0, delays, V, seed, band_scale=None, noise=1.0, wscale=0.02):
    """bench.py's generator on the device: X0 ~ N(0,1) -> FIR delays (HIP) -> X; Y = X W + noise, fp32, padded."""
    from litcoder_core_amd import ops
    dev = ops.device(0)
    rng = np.random.default_rng(seed)
    Xd = ops.fir_delay(torch.from_numpy(rng.standard_normal((T, F0))).to(dev), delays, False)
    p = Xd.shape[1]
    dX = torch.zeros((T, ops.pad_to(p, 32)), dtype=torch.float32, device=dev)
    dX[:, :p] = Xd.to(torch.float32)
    if band_scale is not None:
        dX[:, :p] /= torch.as_tensor(band_scale, dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    dY = torch.zeros((T, ops.pad_to(V, 128)), dtype=torch.float32, device=dev)
    W = wscale * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
    dY[:, :V] = dX[:, :p] @ W + noise * torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    return dX, dY, p


def test_cfg2_full_size_properties(lc):
    """The bench workload itself (T 3000, p 3072, V 80 000, 20 alphas, 5 x 5 K-folds): every result finite, planted
    known answers in place (a constant voxel -> r 0 / p 1 / alphas[0] in every fold; a noiseless voxel on top), the
    metrics consistent with each other, and a 10 000-voxel block fitted alone equal to its slice of the big fit bit for
    bit (what makes an 8-GPU shard of this job return the 1-GPU numbers)."""
    V = 80000
    alphas = np.logspace(-1, 8, 20)
    kw = dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, alphas=alphas)
    dX, dY, p = _device_problem(lc, 3000, 768, [1, 2, 3, 4], V, seed=0)
    dY[:, 123] = 0.75                                                     # constant voxels (one inside the block
    dY[:, 30007] = -2.0                                                   # fitted alone below: the reference's
    # np.mean(fold_scores) is float64 when some r was NaN and float32 otherwise, nested_cv.py:276 -- keep both alike)
    g = torch.Generator(device=dY.device)
    g.manual_seed(5)
    dY[:, 456] = dX[:, :p] @ (0.05 * torch.randn(p, generator=g, device=dY.device))   # noiseless voxel
    model = lc.NestedCVModel("r")
    m, W, a = model.fit_predict_device(dX, dY, p, V, weights_on_host=True, **kw)
    r = np.asarray(m["correlations"])
    assert r.shape == (V,) and np.isfinite(r).all() and np.isfinite(W).all() and np.isfinite(a).all()
    assert np.isfinite(np.asarray(m["p_values"])).all() and np.isfinite(np.asarray(m["corrected_p_values"])).all()
    assert r[123] == 0.0 and m["p_values"][123] == 1.0 and a[123] == np.float32(alphas[0]) and not m["significant_mask"][123]
    # noiseless, but p = 3072 > n = 2400 and the smallest penalty is 0.1 S[0]: far above every noisy voxel, not 1
    assert r[456] > 0.7 and r[456] >= np.quantile(r, 0.999) and m["significant_mask"][456]
    assert 0.3 < m["median_score"] < 0.55 and abs(m["median_score"] - float(np.median(r))) < 1e-6    # ~0.43
    assert m["n_significant"] == int(np.sum(m["significant_mask"])) and m["n_significant"] > 0.9 * V
    assert set(np.unique(np.concatenate(model.last_fold_alphas))) <= set(alphas.astype(np.float32))
    lo, hi = 30000, 40000
    blk = torch.zeros((3000, 10112), dtype=torch.float32, device=dY.device)
    blk[:, : hi - lo] = dY[:, lo:hi]
    m_b, W_b, a_b = lc.NestedCVModel("r").fit_predict_device(dX, blk, p, hi - lo, weights_on_host=True, **kw)
    assert np.array_equal(np.asarray(m_b["correlations"]), r[lo:hi]) and np.array_equal(a_b, a[lo:hi])
    # (round 6) the big fit forms its mean weights from the MEAN of the folds' operators (engine/mean_refit.py: one product
    # of depth T per alpha tuple), the block -- too narrow for a tuple to pay for its operator -- from the folds' own
    # products: the same numbers to fp32 rounding of the two summation orders; bit for bit when both take the folds' products
    assert model.last_fit["mean_operator"]["voxels"] > 0.9 * V, model.last_fit["mean_operator"]
    assert np.abs(W_b - W[:, lo:hi]).max() <= 2e-6 * np.abs(W).max()
    from litcoder_core_amd.engine.common import FitOptions
    off = FitOptions(mean_operator_refit=False)
    _, W0, a0 = lc.NestedCVModel("r", options=off).fit_predict_device(dX, dY, p, V, weights_on_host=True, **kw)
    assert np.array_equal(a0, a) and np.abs(W0 - W).max() <= 2e-6 * np.abs(W).max()
    assert np.array_equal(W_b, W0[:, lo:hi])


def test_cfg3_synthetic_lebel_like_story_pipeline_train_test(lc):
    """LeBel-UTS03-like end to end: per story, word-level 768-d features at irregular word times -> Lanczos resampling
    to the TR grid (lc_lanczos_interp) -> 4 FIR delays -> trim + per-story z-scoring -> stories[:-1] train / last story
    tests -> nested-CV fit in train/test mode with single_alpha (example.py:104-117) and CHUNKED folds, V = 8192
    (StoryPipeline.fit on downsampled features).  Against the oracle's own pipeline (oracle.lanczos / fir / harness /
    nested_cv) on the first 48 voxels; the full LeBel shape against the reference is the next test."""
    import oracle.harness as oh
    import oracle.lanczos as olz
    import oracle.nested_cv as onc
    rng = np.random.default_rng(17)
    V, D = 8192, 192                                # (p = 768: the oracle's six SVDs stay at a second each)
    stories = {"s%d" % i: n for i, n in enumerate((330, 290, 360, 310, 345, 300, 325, 240))}      # TRs per story
    feats_ds, feats_ds_o, brain = {}, {}, {}
    Wtrue = rng.standard_normal((D * 4, V)) * 0.02
    for name, n_tr in stories.items():
        n_words = int(7.2 * n_tr)
        wt = np.sort(rng.uniform(0, 2.0 * n_tr, n_words))
        tr_times = 1.0 + 2.0 * np.arange(n_tr + 15)                                            # features 15 TRs longer
        emb = rng.standard_normal((n_words, D))
        emb[1:] = 0.6 * emb[:-1] + 0.8 * emb[1:]                                                # smooth like LM states
        ds = lc.Downsampler().downsample(emb, wt, tr_times, method="lanczos", window=3, cutoff_mult=1.0)
        ds_o = olz.lanczos_interp(emb, wt, tr_times, window=3, cutoff_mult=1.0)
        np.testing.assert_allclose(ds, ds_o, rtol=0, atol=1e-10)
        feats_ds[name], feats_ds_o[name] = ds, ds_o
    trimming = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0, "train_targets_end": None,
                "test_features_start": 10, "test_features_end": -5, "test_targets_start": 0, "test_targets_end": None}
    delayed_o = oh.delay_all(feats_ds_o, [1, 2, 3, 4])
    for name, n_tr in stories.items():
        brain[name] = delayed_o[name][10:-5] @ Wtrue + rng.standard_normal((n_tr, V))
    kw = dict(folding_type="chunked", n_inner_folds=5, chunk_length=20, alphas=np.logspace(1, 4, 10), single_alpha=True)
    import random
    random.seed(3)
    model = lc.NestedCVModel("r")
    ours = lc.StoryPipeline([1, 2, 3, 4], trimming, model=model).fit(feats_ds, brain, **kw)
    mats = oh.train_test_matrices(delayed_o, {k: v[:, :48] for k, v in brain.items()}, trimming)
    assert mats["Rstim"].shape == (sum(stories.values()) - 240, 4 * D) and mats["Pstim"].shape == (240, 4 * D)
    # single_alpha couples the voxels (argmax of the across-voxel mean): the oracle on 48 voxels would choose from
    # another mean, so it is GIVEN the alpha the full fit chose and must reproduce weights / correlations at it ...
    chosen = float(ours[2][0])
    assert np.all(ours[2] == ours[2][0]) and ours[2].dtype == np.float64
    random.seed(3)
    oracle = onc.fit_predict(mats["Rstim"], mats["Rresp"], X_test=mats["Pstim"], y_test=mats["Presp"],
                             **dict(kw, alphas=[chosen]))
    np.testing.assert_allclose(np.asarray(ours[0]["correlations"])[:48], np.asarray(oracle[0]["correlations"]), atol=1e-4)
    np.testing.assert_allclose(ours[1][:, :48], oracle[1], rtol=1e-3, atol=2e-5)
    assert ours[0]["median_score"] > 0.15
    # ... and per-voxel alpha on the same data is compared in full, ties proven
    kw2 = dict(kw, single_alpha=False)
    random.seed(3)
    ours2 = lc.StoryPipeline([1, 2, 3, 4], trimming, model=model).fit(feats_ds, brain, **kw2)
    detail = {}
    random.seed(3)
    oracle2 = onc.fit_predict(mats["Rstim"], mats["Rresp"], X_test=mats["Pstim"], y_test=mats["Presp"], detail=detail, **kw2)
    assert_matches_oracle(lc, model, ours2, oracle2, detail, mats["Rstim"], np.hstack([mats["Rresp"]]), kw2, "cfg3",
                          corr_atol=1e-4, w_rtol=1e-3, w_atol=1e-4, X_test=mats["Pstim"], y_test=mats["Presp"],
                          cols=np.arange(48))


def test_cfg3_lebel_shape_story_pipeline_against_reference_fixture(lc, golden_dir):
    """BASELINE cfg3 at its own shape -- 26 training stories + 1 test story of 260-440 TRs (T = 9222 / 251), word-level
    768-d float32 features -> Lanczos -> 4 FIR delays (p = 3072) -> trim + per-story zs -> train/test fit with the
    kwargs of example.py:104-117 (K-folds, default 10-alpha grid) -- through StoryPipeline.fit_words: the stories' Lanczos
    resampling and their design matrix in one launch each, the brain data z-scored in the fit's native upload threads and
    landing story by story, panel by panel, while the sweeps run.  Against what the REFERENCE's own Downsampler / FIR /
    AbstractTrainer._create_train_test_split / NestedCVModel returned for the first 256 voxels (configs.npz):
    * the design matrix against samples of the reference's Rstim / Pstim (float32 of its float64 values: the Lanczos
      kernel's sums differ from np.dot's in the last bits of the float64, nothing more);
    * single_alpha on the 256 voxels alone: the reference's alpha, r to 1e-4, weights to 1e-3;
    * per-voxel alphas with the 256 voxels in front of a volume of 80 000, host to host in voxel panels: alpha flips
      proven near-ties of the reference's score table, r / weights as above;
    * single_alpha on the full volume: finite, one alpha, and equal BIT FOR BIT to NestedCVModel.fit_predict on the
      matrices the two-step route builds (harness.structure_train_test: the pipeline changes when bytes move, not what
      is computed)."""
    import _config_problems as cp
    import _fixtures as fx
    from litcoder_core_amd import harness, ops
    g, spec = fx.load(golden_dir)
    pr = cp.story_problem()
    names = list(pr["words"])
    fx.check_inputs(g, "cfg3__checks", *[pr["brain"][s] for s in names[:3]], pr["words"][names[0]])
    kw = dict(pr["kw"], normalpha=True, use_corr=True)
    nv = cp.N_FIX
    # ---- 256 voxels, single alpha: the reference's own problem
    model = lc.NestedCVModel("r")
    pipe = lc.StoryPipeline(pr["delays"], pr["trimming"], model=model)
    m, W, a = pipe.fit_words(pr["words"], pr["wtimes"], pr["trtimes"], pr["brain"], window=3, cutoff_mult=1.0,
                             single_alpha=True, **kw)
    dX, T, Tt, p = pipe.last_design
    shapes = g["cfg3__shapes"]
    assert (T, p) == tuple(shapes[0]) and (Tt, p) == tuple(shapes[2]) and W.shape == (p, nv)
    Xh = dX[:, :p].cpu().numpy()
    np.testing.assert_allclose(Xh[:T][::41, ::53], g["cfg3__Rstim_sample"].astype(np.float32), rtol=0, atol=2e-6)
    np.testing.assert_allclose(Xh[T:][::11, ::53], g["cfg3__Pstim_sample"].astype(np.float32), rtol=0, atol=2e-6)
    (m_o, W_o, a_o), detail = fx.reference_fit(g, "cfg3s")
    assert a.dtype == a_o.dtype and np.array_equal(a, a_o), "single alpha differs from the reference's"
    np.testing.assert_allclose(np.asarray(m["correlations"]), m_o["correlations"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(W[:, :W_o.shape[1]], W_o, rtol=1e-3, atol=1e-4 * float(np.abs(W_o).max()))
    assert abs(m["median_score"] - spec["cfg3s"]["median_score"]) < 1e-4
    # ---- the full volume: 80 000 voxels of brain data per story (float64 host arrays), the fixture's 256 in front
    V = 80000
    dev = ops.device()
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    Wsig = 0.004 * torch.randn((p, V), generator=gen, device=dev)
    brain, row = {}, 0
    for i, s in enumerate(names):
        n_tr = pr["brain"][s].shape[0]
        y = torch.randn((n_tr, V), generator=gen, device=dev, dtype=torch.float32)
        if i < len(names) - 1:                          # brain row t of a training story pairs with design row row + t
            y += dX[row:row + n_tr, :p] @ Wsig
            row += n_tr
        else:                                           # test story: targets [40:] pair with the Tt test rows
            y[40:] += dX[T:T + Tt, :p] @ Wsig
        b = (y * 3.0 + 100.0).cpu().numpy().astype(np.float64)          # un-normalised, like BOLD data
        b[:, :nv] = pr["brain"][s]
        brain[s] = b
    del Wsig, y
    model_v = lc.NestedCVModel("r")
    pipe_v = lc.StoryPipeline(pr["delays"], pr["trimming"], model=model_v)
    ours_v = pipe_v.fit_words(pr["words"], pr["wtimes"], pr["trtimes"], brain, single_alpha=False, **kw)
    assert len(model_v.last_fit["panels"]) > 1, "the full volume must arrive in voxel panels"
    # the matrices of the reference's two-step route (per-story Lanczos, FIR, numpy zs + vstack: the oracle's harness)
    import oracle.fir as ofir
    import oracle.harness as oh
    from _oracle_check import assert_matches_oracle
    delayed = {s: ofir.make_delayed(lc.Downsampler().downsample(pr["words"][s], pr["wtimes"][s], pr["trtimes"][s],
                                                                method="lanczos", window=3, cutoff_mult=1.0), pr["delays"])
               for s in names}
    mt = oh.train_test_matrices(delayed, brain, pr["trimming"])
    oracle_v, detail_v = fx.reference_fit(g, "cfg3v")
    kw_v = dict(kw, single_alpha=False)
    flips = assert_matches_oracle(lc, model_v, ours_v, oracle_v, detail_v, mt["Rstim"], mt["Rresp"][:, :nv], kw_v,
                                  "cfg3 per-voxel vs reference", corr_atol=1e-4, w_rtol=1e-3, w_atol=1e-4, min_same=0.95,
                                  X_test=mt["Pstim"], y_test=mt["Presp"][:, :nv], cols=np.arange(nv),
                                  w_cols=spec["cfg3v"]["w_cols"])
    r_v = np.asarray(ours_v[0]["correlations"])
    assert r_v.shape == (V,) and np.isfinite(r_v).all() and np.isfinite(ours_v[1]).all(), f"{flips} flips"
    # ---- single alpha on the full volume == the two-step route, bit for bit
    ours_s = pipe_v.fit_words(pr["words"], pr["wtimes"], pr["trtimes"], brain, single_alpha=True, **kw)
    assert len(model_v.last_fit["panels"]) > 1 and np.all(ours_s[2] == ours_s[2][0])
    Xs = pipe_v.last_design[0][:, :p].cpu().numpy()
    assert np.array_equal(Xs[:T], mt["Rstim"].astype(np.float32)) and np.array_equal(Xs[T:], mt["Pstim"].astype(np.float32)), \
        "batched Lanczos + fused design kernel differ from the per-story route"
    two = lc.NestedCVModel("r").fit_predict(mt["Rstim"], mt["Rresp"], X_test=mt["Pstim"], y_test=mt["Presp"],
                                            single_alpha=True, **kw)
    assert np.array_equal(ours_s[2], two[2]) and np.array_equal(ours_s[1], two[1])
    assert np.array_equal(np.asarray(ours_s[0]["correlations"]), np.asarray(two[0]["correlations"]))
    assert ours_s[0]["n_significant"] == two[0]["n_significant"]


def test_cfg1_full_size_properties(lc):
    """BASELINE cfg1 at its full shape -- train_simple.py's word-rate model: one feature x 4 FIR delays (p = 4), T = 9000
    training + 600 test TRs, V = 80 000 voxels, train/test mode, 10 alphas: the block-product form (csrc/lc_primal.hip).
    Size-independent properties: everything finite, constant voxel -> (r 0, p 1, alpha[0]), a planted noiseless voxel
    recovered with its weights, a 10 000-voxel block fitted alone equals its slice bit for bit, host-to-host call in
    panels equals the resident fit bit for bit; the first 256 voxels against the oracle (ties proven)."""
    import oracle.nested_cv as onc
    V, T, Tt = 80000, 9000, 600
    dX, dY, p = _device_problem(lc, T + Tt, 1, [1, 2, 3, 4], V, seed=9, wscale=0.15)
    assert p == 4
    dY[:, 77] = 0.5
    wtrue = torch.tensor([0.3, -0.2, 0.1, 0.05], device=dY.device)
    dY[:, 4242] = dX[:, :p] @ wtrue
    alphas = np.logspace(-1, 8, 10)
    kw = dict(folding_type="chunked_contiguous", n_inner_folds=5, chunk_length=20, alphas=alphas)
    model = lc.NestedCVModel("r")
    m, W, a = model.fit_predict_device(dX, dY, p, V, n_test_rows=Tt, weights_on_host=True, **kw)
    assert model.last_form == "primal"
    r = np.asarray(m["correlations"])
    assert r.shape == (V,) and np.isfinite(r).all() and np.isfinite(W).all() and W.shape == (4, V)
    assert r[77] == 0.0 and m["p_values"][77] == 1.0 and a[77] == np.float32(alphas[0])
    assert r[4242] > 0.999 and np.allclose(W[:, 4242], wtrue.cpu().numpy(), atol=6e-3)
    assert abs(m["median_score"] - float(np.median(r))) < 1e-6 and m["median_score"] > 0.05
    lo, hi = 30000, 40000
    blk = torch.zeros((T + Tt, 10112), dtype=torch.float32, device=dY.device)
    blk[:, : hi - lo] = dY[:, lo:hi]
    m_b, W_b, a_b = lc.NestedCVModel("r").fit_predict_device(dX, blk, p, hi - lo, n_test_rows=Tt, weights_on_host=True, **kw)
    assert np.array_equal(np.asarray(m_b["correlations"]), r[lo:hi]) and np.array_equal(a_b, a[lo:hi])
    assert np.array_equal(W_b, W[:, lo:hi])
    # the reference's own call: float64 host arrays in, host weights out (moved in voxel panels)
    X = dX[:, :p].cpu().numpy().astype(np.float64)
    Y = dY[:, :V].cpu().numpy().astype(np.float64)
    m_h, W_h, a_h = lc.NestedCVModel("r").fit_predict(X[:T], Y[:T], X_test=X[T:], y_test=Y[T:], **kw)
    assert np.array_equal(np.asarray(m_h["correlations"]), r) and np.array_equal(a_h, a) and np.array_equal(W_h, W)
    nv = 256
    detail = {}
    oracle = onc.fit_predict(X[:T], Y[:T, :nv], X_test=X[T:], y_test=Y[T:, :nv], detail=detail, **kw)
    assert_matches_oracle(lc, model, (m, W[:, :nv], a), oracle, detail, X[:T], Y[:T], kw, "cfg1", corr_atol=1e-4,
                          w_rtol=1e-3, w_atol=1e-4, X_test=X[T:], y_test=Y[T:], cols=np.arange(nv), min_same=0.95)


def _fixture_volume(lc, name, V, seed):
    """The config's fixture problem (tests/_config_problems.py) resident on the device, its 256 voxels in front of a
    volume of V: (X host f64, Y host f64 (T, 256), kwargs, dX, dY, p)."""
    import _config_problems as cp
    from litcoder_core_amd import ops
    dev = ops.device(0)
    X, Y, kw = cp.matrix_problem(name)
    p = X.shape[1]
    dX = ops.upload_f32(X, ops.pad_to(p, 32), dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    dY = torch.zeros((len(X), ops.pad_to(V, 128)), dtype=torch.float32, device=dev)
    dY[:, :V] = dX[:, :p] @ (0.02 * torch.randn((p, V), generator=gen, device=dev)) + torch.randn((len(X), V), generator=gen,
                                                                                                   device=dev)
    dY[:, :cp.N_FIX] = torch.from_numpy(Y.astype(np.float32)).to(dev)
    return X, Y, kw, dX, dY, p


def test_cfg4_narratives_shape_full_volume(lc, golden_dir):
    """Narratives-like: T = 2226, p = 3072, V = 200 000 voxels on one GPU (the 8-GPU job's whole volume): finite
    everywhere; a 25 000-voxel shard fitted alone equals its slice bit for bit, and so does a column-permuted copy
    (which other voxels share a launch never matters); the first 256 voxels against what the REFERENCE returned for them
    (configs.npz, 5 x 5 K-folds as the config implies; flips proven near-ties of the reference's score table)."""
    import _config_problems as cp
    import _fixtures as fx
    g, spec = fx.load(golden_dir)
    V = 200000
    X, Y, kw, dX, dY, p = _fixture_volume(lc, "cfg4", V, seed=4)
    T = len(X)
    fx.check_inputs(g, "cfg4__checks", X, Y)
    # (the fixture holds a constant voxel, whose NaN r makes the reference's np.mean(fold_scores) a float64 mean for
    # EVERY voxel, nested_cv.py:276 -- the block fitted alone below gets one too, so that both fits average alike)
    dY[:, 130007] = -2.0
    model = lc.NestedCVModel("r")
    m, W, a = model.fit_predict_device(dX, dY, p, V, weights_on_host=False, **kw)
    r = np.asarray(m["correlations"])
    assert r.shape == (V,) and np.isfinite(r).all() and bool(torch.isfinite(W).all()) and np.isfinite(a).all()
    from litcoder_core_amd.dist import shard_bounds
    lo, hi = shard_bounds(V, 8, 5)
    blk = torch.zeros((T, 25088), dtype=torch.float32, device=dY.device)
    blk[:, : hi - lo] = dY[:, lo:hi]
    m_b, W_b, a_b = lc.NestedCVModel("r").fit_predict_device(dX, blk, p, hi - lo, weights_on_host=False, **kw)
    assert np.array_equal(np.asarray(m_b["correlations"]), r[lo:hi]) and np.array_equal(a_b, a[lo:hi])
    # (round 6: the mean weights of the wide fit come from the MEAN of the folds' operators per alpha tuple, those of the
    # narrow block -- where fewer tuples pay for an operator image -- partly from the folds' own products: equal to fp32
    # rounding of the two summation orders; bit for bit with the option off, test_cfg2_full_size_properties)
    assert float((W_b - W[:, lo:hi]).abs().max()) <= 2e-6 * float(W.abs().max())
    perm = torch.randperm(hi - lo, device=dY.device, generator=torch.Generator(device=dY.device).manual_seed(1))
    blk[:, : hi - lo] = dY[:, lo:hi][:, perm]
    m_p, W_p, a_p = lc.NestedCVModel("r").fit_predict_device(dX, blk, p, hi - lo, weights_on_host=False, **kw)
    ph = perm.cpu().numpy()
    assert np.array_equal(np.asarray(m_p["correlations"]), r[lo:hi][ph]) and np.array_equal(a_p, a[lo:hi][ph])
    assert torch.equal(W_p, W_b[:, perm])                                    # (the same block, its voxels shuffled: bit for bit)
    nv = cp.N_FIX
    oracle, detail = fx.reference_fit(g, "cfg4", n_rows=T)
    flips = assert_matches_oracle(lc, model, (m, W[:, :nv].cpu().numpy(), a), oracle, detail, X, Y, kw, "cfg4 vs reference",
                                  corr_atol=1e-4, w_rtol=1e-3, w_atol=1e-4, cols=np.arange(nv), min_same=0.95,
                                  w_cols=spec["cfg4"]["w_cols"])
    assert abs(np.median(r[:nv]) - spec["cfg4"]["median_score"]) < 1e-3, f"{flips} flipped (fold, voxel) pairs"
    # ---- the metric's own call at this config: float64 numpy features / targets in pageable
    # host memory in (3.56 GB of targets, cast to float32 in the staging threads, crossing the link in voxel panels while the
    # first fold runs), metrics + float32 HOST weights out (2.46 GB, leaving panel by panel during the last folds) -- equal to
    # the resident fit bit for bit
    W_res = W.cpu().numpy()
    del W, W_b, W_p, blk
    Yh = np.empty((T, V), dtype=np.float64)
    for c in range(0, V, 16384):
        Yh[:, c:c + 16384] = dY[:, c:min(V, c + 16384)].cpu().numpy()
    del dY
    torch.cuda.empty_cache()
    model_h = lc.NestedCVModel("r")
    m_h, W_h, a_h = model_h.fit_predict(X, Yh, **kw)
    assert len(model_h.last_fit["panels"]) > 1, "200 000 voxels must cross the link in panels"
    assert W_h.dtype == np.float32 and W_h.shape == (p, V)
    assert np.array_equal(np.asarray(m_h["correlations"]), r) and np.array_equal(a_h, a)
    assert np.array_equal(np.asarray(m_h["p_values"]), np.asarray(m["p_values"]))
    assert np.array_equal(np.asarray(m_h["significant_mask"]), np.asarray(m["significant_mask"]))
    # (round 6: which alpha tuples get a mean operator is decided per voxel range from the range's own counts; a tuple near the
    # break-even point may go one way in a download panel and the other way at full width -- the same weights to fp32 rounding)
    assert np.abs(W_h - W_res).max() <= 2e-6 * np.abs(W_res).max(), "host-to-host weights differ from the resident fit's"


def test_cfg5_whisper_shape_banded(lc, golden_dir):
    """Whisper-like: 1280-d speech features x 6 FIR delays = 7680 columns (p > n), 32 alphas logspace(-1, 8), two feature
    bands with penalty scales (1, 2) (BandedNestedCVModel: ridge on the rescaled design), T = 3000, at the
    config's full width V = 80 000 and HOST TO HOST (round 5; 2 048 voxels before): float64 numpy arrays in -- 1.92 GB of
    targets in voxel panels -- metrics + 2.46 GB of float32 host weights out.  The first 256 voxels against what the
    REFERENCE returned on the rescaled design (configs.npz; 12-13 distinct alphas chosen per fold), 5 x 5 K-folds; the rest
    of the volume finite and, for a 4 096-voxel block fitted alone, equal to its slice bit for bit."""
    import _config_problems as cp
    import _fixtures as fx
    from litcoder_core_amd import ops
    g, spec = fx.load(golden_dir)
    Xs, Y, kw = cp.matrix_problem("cfg5")                      # the fixture's design is the RESCALED one, X / gamma
    fx.check_inputs(g, "cfg5__checks", Xs, Y)
    T, p, nv, V = len(Xs), Xs.shape[1], cp.N_FIX, 80000
    gamma = np.r_[np.full(p // 2, 1.0), np.full(p - p // 2, 2.0)]
    X = Xs * gamma                                             # exact (powers of two): the banded model divides it back
    dev = ops.device(0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(23)
    dXs = torch.from_numpy(Xs.astype(np.float32)).to(dev)
    Yw = np.empty((T, V), dtype=np.float64)
    Yw[:, :nv] = Y
    for c in range(nv, V, 16384):                              # (data synthesis on the device, column blocks)
        w = min(16384, V - c)
        blk = dXs @ (0.015 * torch.randn((p, w), generator=gen, device=dev)) + torch.randn((T, w), generator=gen, device=dev)
        Yw[:, c:c + w] = blk.cpu().numpy()
    del dXs, blk
    torch.cuda.empty_cache()
    lo, hi = 40960, 45056
    # (the fixture holds a constant voxel, whose NaN r makes the reference's np.mean(fold_scores) a float64 mean for EVERY
    # voxel, nested_cv.py:276 -- the block fitted alone below gets one too, so that both fits average alike)
    Yw[:, lo + 7] = -2.0
    model = lc.BandedNestedCVModel("r")
    m, W, a = model.fit_predict(X, Yw, bands=[(0, p // 2), (p // 2, p)], band_scales=[1.0, 2.0], **kw)
    assert W.shape == (p, V) and W.dtype == np.float32 and np.isfinite(W).all() and np.isfinite(np.asarray(m["correlations"])).all()
    assert len(model.last_fit["panels"]) > 1, "80 000 voxels must cross the link in panels"
    m_b, W_b, a_b = lc.BandedNestedCVModel("r").fit_predict(X, Yw[:, lo:hi], bands=[(0, p // 2), (p // 2, p)],
                                                            band_scales=[1.0, 2.0], **kw)
    assert np.array_equal(np.asarray(m_b["correlations"]), np.asarray(m["correlations"])[lo:hi]) and np.array_equal(a_b, a[lo:hi])
    assert np.abs(W_b - W[:, lo:hi]).max() <= 2e-6 * np.abs(W).max()         # (round 6: see test_cfg4_narratives_shape_full_volume)
    oracle, detail = fx.reference_fit(g, "cfg5", n_rows=T)
    # weights come back on the ORIGINAL feature scale: w_b = w'_b / gamma_b
    assert_matches_oracle(lc, model, (m, W * gamma[:, None].astype(np.float32), a), oracle, detail, Xs, Yw, kw,
                          "cfg5 vs reference", corr_atol=1e-4, w_rtol=1e-3, w_atol=1e-4, cols=np.arange(nv), min_same=0.95,
                          w_cols=spec["cfg5"]["w_cols"])
    assert abs(np.median(np.asarray(m["correlations"])[:nv]) - spec["cfg5"]["median_score"]) < 1e-3


def test_primal_form_for_tall_designs(lc):
    """cfg1-like designs (train_simple.py:21-31: a word-rate feature x 4 delays, p = 4; example.py:104-117 with
    single_alpha) and other tall ones, T >= 2000: ``form="auto"`` takes the primal (p x p) route, which must give the
    oracle's answer (ties proven) and the dual route's -- full CV and train/test, per-voxel and single alpha, both
    arithmetic paths.  Up to 16 features with correlation scores the primal route is the moments form (block products
    X'y, csrc/lc_primal.hip): p = 4, 8 (trimmed folds: inner training sets that are NOT the outer block minus the
    validation rows), 12; the same small designs also go through the V-wide primal contraction (moments form switched
    off), as p = 64 and R^2 scoring always do.  A design whose feature scales are too far apart for the fp16 split
    falls back to the dual form by itself where fp16 operands would be used."""
    import oracle.nested_cv as onc
    import litcoder_core_amd.nested_cv as ncv
    rng = np.random.default_rng(41)
    for p0, delays, T, V, kw in (
            (1, [1, 2, 3, 4], 2400, 1500, dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 4, 8))),
            (3, [1, 2, 3, 4], 2000, 700, dict(folding_type="chunked", n_outer_folds=3, n_inner_folds=2, chunk_length=25,
                                              alphas=np.logspace(-1, 3, 6), single_alpha=True)),
            (2, [1, 2, 3, 4], 2100, 450, dict(folding_type="kfold_trimmed", n_outer_folds=2, n_inner_folds=3,
                                              alphas=np.logspace(0, 5, 7), normalpha=False)),
            (16, [1, 2, 3, 4], 2100, 600, dict(folding_type="kfold_trimmed", n_outer_folds=2, n_inner_folds=3,
                                               alphas=np.logspace(0, 5, 7), normalpha=False)),
            (64, [0], 2050, 333, dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=np.logspace(-1, 2, 5),
                                      use_corr=False)),
    ):
        from oracle.fir import make_delayed
        X = make_delayed(rng.standard_normal((T, p0)), delays)
        p = X.shape[1]
        Y = X @ (rng.standard_normal((p, V)) * (0.5 / np.sqrt(p))) + rng.standard_normal((T, V))
        Y[:, 2] = -1.0
        r2 = not kw.get("use_corr", True)
        tol = dict(corr_atol=1e-3 if r2 else 3e-5, gap_tol=2e-3 if r2 else 2e-6)
        import random
        for tt in (False, True):
            kw_run = {k: v for k, v in kw.items() if not (tt and k == "n_outer_folds")}
            args = (X[:-300], Y[:-300]) if tt else (X, Y)
            extra = dict(X_test=X[-300:], y_test=Y[-300:]) if tt else {}
            detail = {}
            random.seed(11)
            oracle = onc.fit_predict(*args, detail=detail, **extra, **kw_run)
            moments = p <= ncv.FitOptions().primal_moments_max_p and not r2
            for precision, gemm in (("auto", False), ("f32", False)) + ((("auto", True),) if moments else ()):
                tag = f"p={p} tt={tt} {precision} gemm={gemm}"
                model = lc.NestedCVModel("r", precision=precision,
                                         options=ncv.FitOptions(primal_moments_max_p=0) if gemm else None)
                random.seed(11)
                ours = model.fit_predict(*args, **extra, **kw_run)
                assert model.last_form == "primal", tag
                assert (model.last_fit["precision"] == "f64 block products") == (moments and not gemm), tag
                assert_matches_oracle(lc, model, ours, oracle, detail, args[0], args[1], kw_run, tag, min_same=0.97, **tol,
                                      **({k: extra[k] for k in extra} if tt else {}))
            dual = lc.NestedCVModel("r", form="dual")
            random.seed(11)
            m_d, W_d, a_d = dual.fit_predict(*args, **extra, **kw_run)
            assert dual.last_form == "dual"
            same = np.isclose(a_d, ours[2], rtol=1e-6)
            assert same.mean() >= 0.97
            np.testing.assert_allclose(np.asarray(m_d["correlations"])[same], np.asarray(ours[0]["correlations"])[same],
                                       atol=1e-3 if r2 else 3e-5)
            np.testing.assert_allclose(W_d[:, same], ours[1][:, same], rtol=2e-4, atol=1e-5)
    # feature scales 1 : 2^-12 : the fp16 split cannot carry both through a contraction over features -> dual by itself
    X = rng.standard_normal((2000, 6))
    X[:, 3] *= 2.0 ** -12
    Y = X @ rng.standard_normal((6, 50)) + rng.standard_normal((2000, 50))
    kw = dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=[0.1, 1.0])
    gemm_only = ncv.FitOptions(primal_moments_max_p=0)                     # the V-wide primal contraction (fp16 operands)
    model = lc.NestedCVModel("r", options=gemm_only)
    ours = model.fit_predict(X, Y, **kw)
    assert model.last_form == "dual"
    with pytest.raises(ValueError, match="form='primal' is not usable"):
        lc.NestedCVModel("r", form="primal", options=gemm_only).fit_predict(X, Y, **kw)
    model32 = lc.NestedCVModel("r", precision="f32", options=gemm_only)    # exact-fp32 arithmetic: primal is fine
    ours32 = model32.fit_predict(X, Y, **kw)
    assert model32.last_form == "primal"
    np.testing.assert_allclose(np.asarray(ours32[0]["correlations"]), np.asarray(ours[0]["correlations"]), atol=3e-5)
    model64 = lc.NestedCVModel("r")                                        # fp64 block products: any feature scales
    ours64 = model64.fit_predict(X, Y, **kw)
    assert model64.last_form == "primal" and model64.last_fit["precision"] == "f64 block products"
    np.testing.assert_allclose(np.asarray(ours64[0]["correlations"]), np.asarray(ours[0]["correlations"]), atol=3e-5)
    np.testing.assert_allclose(ours64[1], ours[1], rtol=2e-4, atol=1e-5)


def test_primal_form_with_shared_series_terms_and_block_sums(lc):
    """Round 4: tall designs of HUNDREDS of features (the LeBel-style train/test shape is 9000 rows x 3072 features: the
    p x p side is 13x less fp64 work than the n x n one and the sweeps contract over p instead of n).  From 256 features
    on the primal form (a) scores the alphas on the polynomial series from the moments of shared terms
    P'_j = Pstim G^j / lambda^(j+1), and (b) takes the Gram matrix and the block product Rstim'Rresp of every inner
    training set as the sum over the OTHER folds' validation blocks when the folds partition the training block.
    Against the oracle (ties proven) and against the dual form: K-folds and chunked folds (block sums), trimmed K-folds
    (no partition: the route that contracts over the training rows, still with the series), R^2 scores and exact-fp32
    arithmetic (no moments: every alpha a p x p factorisation), train/test + single_alpha, p not a multiple of 128, and a
    host-to-host fit in voxel panels equal to the resident one bit for bit."""
    import random
    import oracle.nested_cv as onc
    from litcoder_core_amd import ops
    rng = np.random.default_rng(43)
    cases = (
        # (the primal form needs 2 p <= the smallest inner training set)
        (256, 1800, 900, dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=3, alphas=np.logspace(-1, 5, 9))),
        (300, 2600, 700, dict(folding_type="chunked", n_outer_folds=2, n_inner_folds=2, chunk_length=25,
                              alphas=np.logspace(-1, 4, 7), single_alpha=True)),
        (256, 1900, 500, dict(folding_type="kfold_trimmed", n_outer_folds=2, n_inner_folds=3, alphas=np.logspace(0, 5, 7))),
        (384, 3300, 400, dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=np.logspace(-1, 3, 5),
                              use_corr=False)),
    )
    for p, T, V, kw in cases:
        X = rng.standard_normal((T, p))
        X[1:] = 0.5 * X[:-1] + 0.87 * X[1:]                        # correlated rows, like delayed LM features
        Y = X @ (rng.standard_normal((p, V)) * (0.4 / np.sqrt(p)) * np.exp(rng.uniform(np.log(0.05), np.log(2.0), V))) \
            + rng.standard_normal((T, V))
        Y[:, 2] = -1.0
        r2 = not kw.get("use_corr", True)
        tol = dict(corr_atol=1e-3 if r2 else 3e-5, gap_tol=2e-3 if r2 else 2e-6)
        for tt in (False, True):
            kw_run = {k: v for k, v in kw.items() if not (tt and k == "n_outer_folds")}
            args = (X[:-200], Y[:-200]) if tt else (X, Y)
            extra = dict(X_test=X[-200:], y_test=Y[-200:]) if tt else {}
            detail = {}
            random.seed(11)
            oracle = onc.fit_predict(*args, detail=detail, **extra, **kw_run)
            for precision in ("auto", "f32"):
                tag = f"p={p} {kw['folding_type']} tt={tt} {precision}"
                model = lc.NestedCVModel("r", precision=precision)
                random.seed(11)
                ours = model.fit_predict(*args, **extra, **kw_run)
                assert model.last_form == "primal", tag
                want_terms = 4 if (precision == "auto" and not r2 and max(kw["alphas"]) >= 8.0) else 0
                assert model.last_fit["series_terms"] == want_terms, (tag, model.last_fit)
                assert_matches_oracle(lc, model, ours, oracle, detail, args[0], args[1], kw_run, tag, min_same=0.95, **tol,
                                      **({k: extra[k] for k in extra} if tt else {}))
                if precision == "auto":
                    first = ours
            dual = lc.NestedCVModel("r", form="dual")
            random.seed(11)
            m_d, W_d, a_d = dual.fit_predict(*args, **extra, **kw_run)
            assert dual.last_form == "dual"
            same = np.isclose(a_d, first[2], rtol=1e-6)
            assert same.mean() >= 0.95
            np.testing.assert_allclose(np.asarray(m_d["correlations"])[same], np.asarray(first[0]["correlations"])[same],
                                       atol=1e-3 if r2 else 3e-5)
            np.testing.assert_allclose(W_d[:, same], first[1][:, same], rtol=2e-4, atol=1e-5)
    # host-to-host in voxel panels == resident, bit for bit (the block sums are per range)
    p, T, V = 256, 2300, 3000
    X = rng.standard_normal((T, p))
    Y = X @ (rng.standard_normal((p, V)) * 0.03) + rng.standard_normal((T, V))
    kw = dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=np.logspace(-1, 4, 6))
    dev = ops.device()
    dX, dY = ops.upload_f32(X, ops.pad_to(p, 32), dev), ops.upload_f32(Y, ops.pad_to(V, 128), dev)
    res = lc.NestedCVModel("r").fit_predict_device(dX, dY, p, V, weights_on_host=True, **kw)
    for single in (False, True):
        kws = dict(kw, single_alpha=single)
        res = lc.NestedCVModel("r").fit_predict_device(dX, dY, p, V, weights_on_host=True, **kws)
        host = lc.NestedCVModel("r", panel_cols=1024)
        out = host.fit_predict(X, Y, **kws)
        assert host.last_form == "primal" and len(host.last_fit["panels"]) > 1
        assert np.array_equal(out[1], res[1]) and np.array_equal(out[2], res[2])
        assert np.array_equal(np.asarray(out[0]["correlations"]), np.asarray(res[0]["correlations"]))


def test_block_product_form_on_odd_folds_against_oracle(lc):
    """The moments form of the primal route (p <= 16, correlation scores: csrc/lc_primal.hip) where its bookkeeping
    differs from the plain K-fold case: time-series and trimmed folds (inner training sets that are NOT the outer
    block minus the validation rows -> row sets of their own), group folds (ragged), train-statistics normalisation
    of features and targets (per-fold data), a single alpha for all voxels, train/test mode, one voxel, raw alphas --
    against the oracle with the near-tie proof."""
    import oracle.nested_cv as onc
    from _oracle_check import assert_matches_oracle
    rng = np.random.default_rng(2024)
    cases = [
        dict(T=310, p=3, V=37, kw=dict(folding_type="timeseries", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 3, 6))),
        dict(T=402, p=6, V=131, kw=dict(folding_type="chunked_trimmed", n_outer_folds=3, n_inner_folds=2, chunk_length=12,
                                        alphas=np.logspace(0, 4, 5), single_alpha=True)),
        dict(T=277, p=5, V=64, kw=dict(folding_type="group", n_outer_folds=3, n_inner_folds=2, alphas=[0.3, 3.0, 30.0],
                                       groups=True)),
        dict(T=350, p=9, V=90, kw=dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 2, 4),
                                       normalize_features=True, normalize_targets=True)),
        dict(T=300, p=2, V=1, kw=dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=np.logspace(-2, 4, 7),
                                      normalpha=False)),
        dict(T=420, p=16, V=203, tt=120, kw=dict(folding_type="chunked", n_inner_folds=3, chunk_length=20,
                                                 alphas=np.logspace(-1, 3, 5))),
    ]
    for i, c in enumerate(cases):
        T, p, V = c["T"], c["p"], c["V"]
        X = rng.standard_normal((T, p)) * rng.uniform(0.2, 3.0, p) + rng.uniform(-1, 1, p)
        Y = X @ (rng.standard_normal((p, V)) * (0.4 / np.sqrt(p))) + rng.standard_normal((T, V)) + 5.0
        kw = dict(c["kw"])
        if kw.pop("groups", False):
            kw["groups"] = rng.integers(0, 11, size=T - c.get("tt", 0))
        tt = c.get("tt", 0)
        args = (X[:T - tt], Y[:T - tt])
        extra = dict(X_test=X[T - tt:], y_test=Y[T - tt:]) if tt else {}
        random.seed(50 + i); np.random.seed(50 + i)
        detail = {}
        oracle = onc.fit_predict(*args, detail=detail, **extra, **kw)
        random.seed(50 + i); np.random.seed(50 + i)
        model = lc.NestedCVModel("r")
        ours = model.fit_predict(*args, **extra, **kw)
        tag = f"moments case {i}"
        assert model.last_form == "primal" and model.last_fit["precision"] == "f64 block products", tag
        assert_matches_oracle(lc, model, ours, oracle, detail, args[0], args[1], kw, tag, corr_atol=3e-5, gap_tol=2e-6,
                              min_same=0.97, **extra)


def test_block_product_form_degenerate_voxels(lc):
    """NaN, inf and constant target voxels through the moments form: alpha, correlation 0 / p 1, NaN weight columns exactly
    where the oracle has them; the other voxels unaffected."""
    import oracle.nested_cv as onc
    rng = np.random.default_rng(3)
    T, p, V = 320, 4, 12
    X = rng.standard_normal((T, p))
    Y = X @ rng.standard_normal((p, V)) * 0.4 + rng.standard_normal((T, V))
    Y[5, 3] = np.nan
    Y[7, 6] = np.inf
    Y[:, 9] = 0.0
    kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=2, alphas=[0.1, 1.0, 10.0])
    random.seed(1)
    mo, Wo, ao = onc.fit_predict(X, Y, **kw)
    random.seed(1)
    model = lc.NestedCVModel("r")
    m, W, a = model.fit_predict(X, Y, **kw)
    assert model.last_fit["precision"] == "f64 block products"
    assert np.array_equal(a, ao)
    c, co = np.asarray(m["correlations"]), np.asarray(mo["correlations"])
    assert (c[[3, 6, 9]] == 0).all() and (np.asarray(m["p_values"])[[3, 6, 9]] == 1).all()
    assert np.array_equal(np.isnan(W).any(0), np.isnan(Wo).any(0)) and np.isnan(W).any(0).nonzero()[0].tolist() == [3, 6]
    ok = ~np.isnan(W).any(0)
    np.testing.assert_allclose(c[ok], co[ok], atol=3e-6)
    np.testing.assert_allclose(W[:, ok], Wo[:, ok], rtol=1e-5, atol=1e-6)


def test_inner_fold_without_validation_rows_is_skipped(lc):
    """The reference scores every alpha NaN -> 0 on an empty validation block (ridge_regression.py:124-133), which adds
    nothing to the sum the alpha is chosen from: the engine drops such a fold -- identical scores; when NO inner fold has
    validation rows the reference takes alphas[0] for every voxel, and so does the engine."""
    from litcoder_core_amd.nested_cv import RidgeCVEngine
    rng = np.random.default_rng(9)
    T, p, V = 260, 30, 70
    X = rng.standard_normal((T, p))
    Y = X @ (rng.standard_normal((p, V)) * 0.1) + rng.standard_normal((T, V))
    tr, te = np.r_[0:200], np.r_[200:260]
    good = [(np.r_[0:100], np.r_[100:200]), (np.r_[100:200], np.r_[0:100])]
    empty = (np.r_[0:200], np.r_[0:0])
    alphas = np.logspace(-1, 3, 5)
    scores = []
    for inner in (good, good + [empty], [empty] + good):
        eng = RidgeCVEngine(X, Y, alphas, True, True, False, False)
        eng.begin_fit(1)
        st = eng.fold_begin(tr, te, inner)
        scores.append(st["scores"][:, :V].cpu().numpy())
    assert np.array_equal(scores[0], scores[1]) and np.array_equal(scores[0], scores[2])
    # no validation rows in ANY inner fold: every alpha scores 0, the first-maximum argmax takes alphas[0] for every
    # voxel (ridge_regression.py:124-133, nested_cv.py:405-411) -- a whole fit equals the oracle's, which follows the
    # reference there
    eng = RidgeCVEngine(X, Y, alphas, True, True, False, False)
    eng.begin_fit(1)
    st = eng.fold_begin(tr, te, [empty])
    assert not st["scores"][:, :V].any()
    import oracle.nested_cv as onc
    import litcoder_core_amd.folding as folding
    real = folding.create_folds

    def all_empty(n, kind, k, *a, **kw_):                      # every inner split: all rows train, none validate
        out = real(n, kind, k, *a, **kw_)
        return [(np.arange(n), np.arange(0)) for _ in out] if n == 200 else out

    import oracle.folds as ofolds
    real_o = ofolds.create_folds
    kw = dict(folding_type="kfold", n_inner_folds=3, alphas=alphas)
    try:
        folding.create_folds = all_empty
        ofolds.create_folds = all_empty
        import litcoder_core_amd.nested_cv as ncv
        keep = ncv.create_folds
        ncv.create_folds = all_empty
        m, W, a = lc.NestedCVModel("r").fit_predict(X[:200], Y[:200], X_test=X[200:], y_test=Y[200:], **kw)
        m_o, W_o, a_o = onc.fit_predict(X[:200], Y[:200], X_test=X[200:], y_test=Y[200:], **kw)
    finally:
        folding.create_folds = real
        ofolds.create_folds = real_o
        ncv.create_folds = keep
    assert np.all(a == np.float32(alphas[0])) and np.array_equal(a, a_o)
    np.testing.assert_allclose(np.asarray(m["correlations"]), np.asarray(m_o["correlations"]), atol=3e-5)
    np.testing.assert_allclose(W, W_o, rtol=2e-4, atol=3e-6)


def test_sweeps_are_bit_reproducible_beside_a_coresident_workgroup(lc):
    """The fp16x3 sweep kernel fills its LDS ring by LDS-DMA behind counted waits and one barrier per K-tile.  A small
    LDS-using kernel of ANOTHER stream fits on the same CU beside its workgroup (31 KB of LDS and a few VGPRs are left);
    rounds 1-2 leaned on a DMA landing later than a queued ds_read returns, which such a neighbour broke in ~1 sweep of
    5: one wave multiplied 32 rows of one K-tile with the next ring turn's bytes, a handful of voxels of the first outer
    fold chose another alpha (found by the full-size bit-identity test below turning flaky; fixed by waiting for the
    fragment reads before the barrier).  Fold 0's sweeps at cfg2 size, repeated with the auxiliary stream doing what it
    does beside them in a real fit (fold 0's own Cholesky chain, then the series chain of folds 1-4): every repetition
    equals the undisturbed one bit for bit."""
    from litcoder_core_amd import nested_cv as ncv, ops
    from litcoder_core_amd.folding import create_folds
    V, T = 80000, 3000
    dX, dY, p = _device_problem(lc, T, 768, [1, 2, 3, 4], V, seed=0)
    alphas = np.logspace(-1, 8, 20)
    eng = ncv.RidgeCVEngine(ncv._DeviceShapes(dX, p), ncv._DeviceShapes(dY, V), alphas, True, True, False, False)
    eng.begin_fit(5)
    outer = [(tr, te, create_folds(len(tr), "kfold", 5)) for tr, te in create_folds(T, "kfold", 5)]
    lmax_pre = eng.precompute_lmax(outer)
    base = eng.prepare_folds(outer[:1], lmax_pre[:1])[0]
    torch.cuda.synchronize()

    def sweep(done=None):
        hat = dict(base["hat"])
        cs, split = eng._target_scales(eng.dY_full, eng.full)
        hat.update(cs=cs, split=split)
        return eng._sweeps(hat, eng.dY_full, done)

    ref = sweep()
    torch.cuda.synchronize()
    ref = ref.clone()
    N, M, B = 1920, 480, 20
    aug0 = torch.randn((B, N + M, N), dtype=torch.float64, device=dX.device) * 0.01
    aug0[:, :N] += torch.eye(N, dtype=torch.float64, device=dX.device) * 50.0
    for it in range(10):
        gate = torch.cuda.Event()
        with torch.cuda.stream(eng.aux):
            aug = aug0.clone()
            H = torch.empty((B, M, N), dtype=torch.float32, device=dX.device)
            ops.batch_chol_solve(aug, B, N, M, H)
            done = torch.cuda.Event()
            done.record()
        eng.prepare_folds(outer[1:], lmax_pre[1:], chol_after=gate)
        s = sweep(done)
        gate.record()
        torch.cuda.synchronize()
        assert torch.equal(s[:, :V], ref[:, :V]), f"repetition {it}: {int((s[:, :V] != ref[:, :V]).sum())} scores differ"


@pytest.mark.gpu
def test_banded_search_picks_the_planted_band_scales(lc):
    """Search over band scales at a medium shape (T 900, two bands of 128 features, 4096 voxels, three candidates):
    voxels driven by ONE band choose the candidate that penalises the other band, and their out-of-sample r is not
    worse than under the uniform scaling; every result is finite; the fold-mean weights of a voxel sit in its band."""
    rng = np.random.default_rng(77)
    T, p, V = 900, 256, 4096
    X = rng.standard_normal((T, p))
    Wt = np.zeros((p, V))
    Wt[:128, : V // 2] = rng.standard_normal((128, V // 2)) * 0.12          # first half of the voxels: band 0 only
    Wt[128:, V // 2:] = rng.standard_normal((128, V // 2)) * 0.12           # second half: band 1 only
    Y = X @ Wt + rng.standard_normal((T, V))
    bands = [(0, 128), (128, 256)]
    cands = [[1.0, 1.0], [1.0, 8.0], [8.0, 1.0]]                            # uniform / shrink band 1 / shrink band 0
    kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 3, 9))
    model = lc.BandedNestedCVModel("ridge_regression")
    m, W, a = model.fit_predict_search(X, Y, bands, cands, **kw)
    r = np.asarray(m["correlations"], dtype=np.float64)
    assert np.isfinite(r).all() and np.isfinite(W).all() and np.isfinite(a).all()
    c = model.last_fold_candidates                                          # (3 folds, V)
    first, second = c[:, : V // 2], c[:, V // 2:]
    assert (first == 1).mean() > 0.8 and (second == 2).mean() > 0.8, ((first == 1).mean(), (second == 2).mean())
    m0, W0, a0 = lc.NestedCVModel("ridge_regression", form="dual").fit_predict(X, Y, **kw)
    r0 = np.asarray(m0["correlations"], dtype=np.float64)
    assert np.median(r) > np.median(r0) + 0.005, (np.median(r), np.median(r0))
    own = np.r_[np.abs(W[:128, : V // 2]).mean(), np.abs(W[128:, V // 2:]).mean()]
    other = np.r_[np.abs(W[128:, : V // 2]).mean(), np.abs(W[:128, V // 2:]).mean()]
    assert np.all(own > 5 * other), (own, other)
