"""Host-side logic of litcoder_core_amd that needs no GPU: fold generation, the statistics tail,
the Downsampler front-end's host methods, the C-ABI library's exported symbols, sharding helpers."""
import ctypes
import json
import os
import random
import re

import numpy as np
import pytest

import litcoder_core_amd as lc
from litcoder_core_amd import _lib, folding, stats
from oracle import folds as ofolds
from oracle import stats as ostats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports every function include/litcoder_hip.h declares;
    the ctypes table names exactly the same set."""
    hdr = open(os.path.join(ROOT, "include", "litcoder_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(lc_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().lc_version() >= 100


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.LitcoderHipError, match="no CPU fallback"):
        lc.FIR.make_delayed(np.zeros((4, 2)), [1])
    with pytest.raises(_lib.LitcoderHipError):
        lc.NestedCVModel("r").fit_predict(np.zeros((40, 3)), np.zeros((40, 2)), folding_type="kfold")


def test_penalty_grid_validation_and_route():
    """check_penalties validates the grid before the device is touched and says which route the fit takes: Cholesky
    (every shipped caller's values) or the spectral one -- alpha = 0 in the grid, or a singcutoff that is not negligible
    against the smallest penalty, where the reference's truncated SVD has to be reproduced as such
    (ridge_utils.py:44-63, ridge_regression.py:56,117)."""
    from litcoder_core_amd.nested_cv import check_penalties
    assert not check_penalties(np.logspace(-1, 8, 20), 1e-10, True, 5)   # example.py / train_simple.py / unified.py
    assert not check_penalties([0.1, 1.0], 1e-30, False)                 # ridge_corr_torch's own default
    assert not check_penalties([1.0], 1e-3, False)                       # (1e-3 / 1)^2 = 1e-6: invisible in fp32
    assert check_penalties([0.0, 1.0], 1e-10, True) and check_penalties([1.0, 10.0], 1e-2, False)
    assert check_penalties([0.1], 1e-6, True)                            # normalpha: S[0] unknown yet -> the safe route
    model = lc.NestedCVModel("r")
    X, Y = np.zeros((40, 3)), np.zeros((40, 2))
    for bad in ([np.nan], [np.inf]):
        with pytest.raises(ValueError, match="alphas must be finite"):
            model.fit_predict(X, Y, alphas=bad, folding_type="kfold")
    # the reference squares alpha (ridge_regression.py:56,117) and takes any number of them (:46-50,115): no refusal
    assert not check_penalties([-1.0, 2.0], 1e-10, True) and not check_penalties(np.logspace(-1, 8, 200), 1e-10, True)
    assert check_penalties([-1.0, 0.0], 1e-10, True)                     # ... and 0 still means the spectral route
    with pytest.raises(ValueError, match="singcutoff must be a finite number"):
        model.fit_predict(X, Y, alphas=[1.0], singcutoff=-1.0, folding_type="kfold")
    with pytest.raises(ValueError, match="n_inner_folds must be >= 1"):
        model.fit_predict(X, Y, alphas=[1.0], n_inner_folds=0, folding_type="kfold")
    with pytest.raises(ValueError):
        check_penalties([], 1e-10, True)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "litcoder_core_amd")
    seen = 0
    for base, _, files in os.walk(pkg):                 # the package and its engine/ sub-package
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(base, fn)).read()
                seen += 1
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
    assert seen >= 20


def test_folds_match_reference_golden(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "folds.json"))):
        if c["seed"] is not None:
            random.seed(c["seed"])
            np.random.seed(c["seed"])
        groups = np.array(c["groups"]) if "groups" in c else None
        sp = folding.create_folds(c["n"], c["fold_type"], c["n_folds"], c["chunk_length"], c["trim_size"], groups)
        assert [[list(map(int, a)), list(map(int, b))] for a, b in sp] == c["splits"], c["fold_type"]


def test_folds_match_oracle_on_more_shapes():
    rng = np.random.default_rng(3)
    for n, k in [(10, 2), (11, 3), (97, 5), (100, 7), (2400, 5)]:
        for ft in ("kfold", "kfold_trimmed", "timeseries", "chunked_contiguous"):
            a = folding.create_folds(n, ft, k, 7)
            b = ofolds.create_folds(n, ft, k, 7)
            assert all(list(x[0]) == list(y[0]) and list(x[1]) == list(y[1]) for x, y in zip(a, b)), (n, k, ft)
        g = rng.integers(0, k + 3, size=n)
        a, b = folding.create_folds(n, "group", k, groups=g), ofolds.create_folds(n, "group", k, groups=g)
        assert all(list(x[0]) == list(y[0]) and list(x[1]) == list(y[1]) for x, y in zip(a, b))
        for ft in ("chunked", "chunked_trimmed"):
            random.seed(n); np.random.seed(n)
            a = folding.create_folds(n, ft, k, 7)
            random.seed(n); np.random.seed(n)
            b = ofolds.create_folds(n, ft, k, 7)
            assert all(list(x[0]) == list(y[0]) and list(x[1]) == list(y[1]) for x, y in zip(a, b)), (n, k, ft)
    with pytest.raises(ValueError, match="Unknown folding type"):
        folding.create_folds(10, "nope", 2)
    with pytest.raises(ValueError, match="Groups must be provided"):
        folding.create_folds(10, "group", 2)
    with pytest.raises(ValueError):        # the reference's groups-into-trim_size positional slip
        folding.create_folds(300, "kfold_trimmed", 3, 20, np.arange(300))


def test_pearson_pvalues_match_scipy():
    from scipy.stats import pearsonr
    rng = np.random.default_rng(0)
    for n in (3, 10, 600):
        x = rng.standard_normal((n, 40)).astype(np.float32)
        y = (0.3 * x + rng.standard_normal((n, 40))).astype(np.float32)
        rp = [pearsonr(x[:, i], y[:, i]) for i in range(40)]
        r = np.array([a for a, _ in rp])                      # float32, like the reference's fold scores
        p = np.array([float(b) for _, b in rp])
        assert r.dtype == np.float32
        np.testing.assert_allclose(stats.pearson_pvalues(r, n), p, rtol=1e-9, atol=1e-300)
        r64 = np.array([float(pearsonr(x[:, i].astype(np.float64), y[:, i].astype(np.float64))[0]) for i in range(40)])
        p64 = np.array([float(pearsonr(x[:, i].astype(np.float64), y[:, i].astype(np.float64))[1]) for i in range(40)])
        np.testing.assert_allclose(stats.pearson_pvalues(r64, n), p64, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(stats.pearson_pvalues(np.array([np.nan, 1.0, -1.0, 0.0]), 50), [1.0, 0.0, 0.0, 1.0], atol=1e-15)
    assert stats.pearson_pvalues(np.array([0.5]), 2).tolist() == [1.0]
    # numpy-only continued fraction agrees with scipy's betainc
    from scipy.special import betainc
    x = np.linspace(0, 1, 101)
    for ab in (0.5, 4.0, 299.0):
        np.testing.assert_allclose(stats._betainc_cf(ab, ab, x), betainc(ab, ab, x), rtol=1e-10, atol=1e-300)


def test_fisher_and_fdr_match_oracle():
    rng = np.random.default_rng(1)
    P = rng.uniform(0, 1, size=(5, 300)) ** 3
    P[:, 7] = 1.0
    P[2, 9] = 0.0
    np.testing.assert_allclose(stats.fisher_combine(P), ostats.fisher_combine([list(r) for r in P]), rtol=1e-12, atol=0)
    for alpha in (0.05, 0.2):
        r1, a1 = stats.fdrcorrection(P[0], alpha)
        r2, a2 = ostats.bh_fdr(P[0], alpha)
        assert np.array_equal(r1, r2) and np.array_equal(a1, a2)
    rej, adj = stats.fdrcorrection(np.array([0.04, 0.001, 0.03, 0.5, 0.011]), 0.05)     # hand-checked KAT
    assert rej.tolist() == [True, True, True, False, True]
    np.testing.assert_allclose(adj, [0.05, 0.005, 0.05, 0.5, 0.0275], atol=1e-15)


def test_metrics_dicts_have_reference_keys(golden_dir):
    g = np.load(os.path.join(golden_dir, "fits.npz"))
    r = np.array([0.5, 0.1, -0.2, 0.9], dtype=np.float32)
    p = stats.pearson_pvalues(r.astype(np.float64), 60)
    sig, adj = stats.fdrcorrection(p, 0.05)
    m = stats.train_test_metrics(list(r), list(p), adj, sig, np.array([1, 2, 3, 4], dtype=np.float32), sig.sum())
    want = sorted(k.split("__m_")[1] for k in g.files if k.startswith("tt_kfold_p__m_"))
    assert sorted(m) == want
    m = stats.full_cv_metrics(r.astype(np.float64), p, adj, sig, sig, np.ones(4), sig.sum(), sig.sum())
    want = sorted(k.split("__m_")[1] for k in g.files if k.startswith("cv_kfold_p__m_"))
    assert sorted(m) == want
    assert isinstance(m["correlations"], list) and isinstance(m["median_score"], float)


def test_downsampler_front_end_validation(golden_dir):
    """Method registry and kwarg validation (downsampling.py:330-393) need no GPU: errors come first."""
    g = np.load(os.path.join(golden_dir, "downsample.npz"))
    ds = lc.Downsampler()
    d, ot, nt = g["data"], g["oldtime"], g["newtime"]
    assert set(ds.available_methods) == set(lc.Downsampler.METHOD_PARAMS)
    assert ds.get_method_params("lanczos") == {"required": ["window", "cutoff_mult"], "optional": ["rectify"]}
    with pytest.raises(ValueError, match="Required parameter 'split_indices' missing for method 'average'"):
        ds.downsample(d, ot, nt, method="average")
    with pytest.raises(ValueError, match="Required parameter 'cutoff_mult' missing for method 'sinc'"):
        ds.downsample(d, ot, nt, method="sinc", window=3)
    with pytest.raises(ValueError, match="Unsupported downsampling method"):
        ds.get_method_params("zzz")
    with pytest.raises(ValueError, match="Unsupported downsampling method: nope"):
        ds.downsample(d, ot, nt, method="nope")


def test_fir_helpers_without_gpu():
    f = lc.FIR(delays=[1, 2, 3, 4])
    assert f.n_delays() == 4 and f.output_dim(10) == 40 and f.valid_length(100) == 96
    assert lc.FIR(delays=[-3, 2], circpad=True).valid_length(50) == 50
    assert "FIR(delays=[1, 2, 3, 4], circpad=False)" in f.summary(10, 100)
    with pytest.raises(ValueError, match="delays must be provided"):
        lc.FIR().expand(np.zeros((3, 3)))


def test_shard_bounds_partition():
    for n, w in [(80000, 8), (10, 3), (7, 8), (200000, 8)]:
        b = [lc.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


# ------------------------------------------------------------------ result files
def test_model_saver_matches_reference_format(golden_dir, tmp_path):
    import pickle
    from litcoder_core_amd import ModelSaver
    g = json.load(open(os.path.join(golden_dir, "saver.json")))
    hp = g["hyperparams"]
    assert ModelSaver.run_hash(hp) == g["dir_suffix"]
    metrics = {"median_score": 0.25, "correlations": [np.float32(0.5), 0.0, np.float32(-0.125)],
               "best_alphas": [1.0, 1.0, 10.0], "significant_mask": [True, False, False]}
    W = np.arange(12, dtype=np.float32).reshape(4, 3)
    sv = ModelSaver(base_dir=str(tmp_path / "results"))
    run = sv.save_encoding_model(W, np.array([1.0, 1.0, 10.0]), hp, metrics, save_weights=True)
    parts = run.name.split("_")
    assert parts[0] == "run" and len(parts[1]) == 8 and len(parts[2]) == 6 and parts[3] == g["dir_suffix"]
    assert sorted(p.name for p in run.iterdir()) == g["files"]
    assert (run / "hyperparams.json").read_text() == g["hyperparams_json_text"]
    assert sorted(pickle.load(open(run / "metrics.pkl", "rb")).keys()) == g["metrics_keys"]
    assert list(np.load(run / "weights.npy").shape) == g["weights_shape"]
    W2, a2, hp2, m2 = sv.load_encoding_model(run)
    assert np.array_equal(W2, W) and np.array_equal(a2, [1.0, 1.0, 10.0]) and hp2 == hp and m2["median_score"] == 0.25
    run_nw = sv.save_encoding_model(W, None, dict(hp, layer_idx=3), metrics)        # default: no weights file
    assert sorted(p.name for p in run_nw.iterdir()) == ["hyperparams.json", "metrics.pkl"]
    runs = sv.list_runs()
    assert len(runs) == 2 and {r["hyperparams"]["layer_idx"] for r in runs} == {3, 9}


def test_band_column_scales_validation():
    from litcoder_core_amd.banded import band_column_scales
    np.testing.assert_array_equal(band_column_scales(5, [(0, 2), (2, 5)], [1.0, 4.0]), [1, 1, 4, 4, 4])
    np.testing.assert_array_equal(band_column_scales(4, np.array([1, 0, 1, 0]), [2.0, 3.0]), [3, 2, 3, 2])
    for bad in (([(0, 2), (1, 5)], [1.0, 2.0]), ([(0, 2)], [1.0]), ([(0, 2), (2, 5)], [1.0, -1.0]),
                ([(0, 2), (2, 5)], [1.0])):
        with pytest.raises(ValueError):
            band_column_scales(5, *bad)


# ------------------------------------------------------------------ polynomial form of the ridge inverse
def test_minimax_inverse_polynomial():
    from litcoder_core_amd import series
    xs = np.linspace(0.0, 1.0, 4001)
    for alpha in (2.64, 7.85, 23.4, 616.0, 1e8):
        for terms in (3, 4, 5, 6):
            c = series.minimax_inverse_coefficients(alpha, terms)
            res = np.max(np.abs(1.0 - (xs + alpha ** 2) * np.polynomial.polynomial.polyval(xs, c)))
            bound = series.residual_bound(alpha, terms)
            assert res <= 1.02 * bound + 5e-16, (alpha, terms, res, bound)
            taylor = np.array([(-1) ** j * alpha ** (-2.0 * (j + 1)) for j in range(terms)])
            res_t = np.max(np.abs(1.0 - (xs + alpha ** 2) * np.polynomial.polynomial.polyval(xs, taylor)))
            assert res <= res_t + 5e-16                      # never worse than the truncated Neumann series
            np.testing.assert_allclose(c[0], taylor[0], rtol=1e-3)
    # the engine's membership rule at cfg2: 4 terms serve every alpha >= 7.85 of logspace(-1, 8, 20)
    al = np.logspace(-1, 8, 20)
    assert [i for i, a in enumerate(al) if series.residual_bound(a, 4) <= 2e-9] == list(range(4, 20))


def test_oracle_banded_search_reduces_to_the_plain_fit():
    """oracle/banded.py (the self-defined search over band scales) with ONE candidate of unit scales is the reference
    algorithm's full nested CV; with one non-trivial candidate it is the fit on the rescaled design with the weights
    mapped back to the original features."""
    import oracle.banded as oband
    import oracle.nested_cv as onc
    rng = np.random.default_rng(5)
    X = rng.standard_normal((90, 12))
    Y = X @ (rng.standard_normal((12, 9)) * 0.3) + rng.standard_normal((90, 9))
    kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 2, 4))
    bands = [(0, 5), (5, 12)]
    m0, W0, a0 = onc.fit_predict(X, Y, **kw)
    m1, W1, a1 = oband.fit_predict_search(X, Y, bands, [[1.0, 1.0]], **kw)
    np.testing.assert_allclose(W1, W0, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_allclose(m1["correlations"], m0["correlations"], rtol=0, atol=1e-7)
    g = np.r_[np.full(5, 0.5), np.full(7, 3.0)]
    m2, W2, a2 = onc.fit_predict(X / g, Y, **kw)
    m3, W3, a3 = oband.fit_predict_search(X, Y, bands, [[0.5, 3.0]], **kw)
    np.testing.assert_allclose(W3, W2 / g[:, None], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(a3, a2)
    det = {}
    oband.fit_predict_search(X, Y, bands, [[1.0, 1.0], [0.5, 3.0]], detail=det, **kw)
    assert det["fold_candidates"].shape == (3, 9) and det["fold_tables"].shape == (3, 8, 9)


def test_host_rows_accepts_any_view():
    """torch.tensor(x, dtype=float32) (nested_cv.py:99-100) takes any array view; the native uploader walks rows of a
    positive whole-element stride, so HostRows copies what is not that: row-reversed, broadcast,
    column-strided and Fortran-ordered views -- and keeps plain row slices as views."""
    from litcoder_core_amd.ops import HostRows
    Y = np.arange(20.0).reshape(5, 4)
    views = [Y[::-1], np.broadcast_to(Y[0], (5, 4)), Y[:, ::-1], np.asfortranarray(Y), Y[:, ::2], Y[::2], Y[1:4],
             Y.astype(np.float32)[::-1], np.arange(20).reshape(5, 4)]
    for v in views:
        (r0, b), = HostRows([v]).blocks
        assert r0 == 0 and np.array_equal(b, v) and b.dtype in (np.float32, np.float64)
        item = b.dtype.itemsize
        assert b.strides[1] == item and (b.shape[0] < 2 or (b.strides[0] >= b.shape[1] * item and b.strides[0] % item == 0))
    assert np.shares_memory(HostRows([Y[::2]]).blocks[0][1], Y) and np.shares_memory(HostRows([Y[1:4]]).blocks[0][1], Y)
    h = HostRows([Y[:3], Y[::-1]])
    assert h.shape == (8, 4) and [r for r, _ in h.blocks] == [0, 3]
    with pytest.raises(RuntimeError, match="shape mismatch"):
        HostRows([Y, Y[:, :2]])


def test_host_zscore_of_a_story_is_numpys_bit_for_bit():
    """What a z-scored upload job (LC_UPLOAD_ZSCORE, csrc/lc_upload.hip) stages for one story: utils.zs
    (encoding/utils.py:23-29: population std, zero-std columns only de-meaned) in the data's own precision followed by the
    float32 cast of nested_cv.py:99-100.  numpy reduces axis 0 of a C-ordered matrix row by row, one running sum per
    column; the staging threads add in that order without fused multiply-adds -- so the bytes that cross PCIe are the
    reference's.  Host code: runs without a GPU (lc_host_zscore_story)."""
    from litcoder_core_amd import ops
    import oracle.harness as oh
    rng = np.random.default_rng(0)
    for dt in (np.float64, np.float32):
        for shape in ((337, 1000), (5, 3), (291, 3001), (2, 17), (1, 9)):
            v = (rng.standard_normal(shape) * rng.uniform(0.1, 30, shape[1]) + rng.uniform(-100, 100, shape[1])).astype(dt)
            v[:, 1] = 0.5                                            # zero std: de-meaned, not divided
            if shape[0] > 1 and shape[1] > 5:
                v[1, 4] = np.nan
            with np.errstate(all="ignore"):
                want = oh.zs(v.copy()).astype(np.float32)
            got = ops.host_zscore_story(v)
            assert got.dtype == np.float32 and np.array_equal(got, want, equal_nan=True), (dt, shape)
            assert not got[:, 1].any()
        big = rng.standard_normal((400, 5000)).astype(dt)
        view = big[10:-5, 100:4000]                                  # a trimmed story inside a wider matrix
        assert np.array_equal(ops.host_zscore_story(view), oh.zs(view.copy()).astype(np.float32))


def test_host_side_of_the_library_under_sanitizers(tmp_path):
    """csrc/lc_upload.hip (staging threads, slot ring, coordinator) and csrc/lc_core.hip (host casts, 2-D copies, event timers)
    built by g++ with -fsanitize=address,undefined and with -fsanitize=thread against the HIP stand-in of
    tools/sanitize/hip_stub/ and run through the upload job matrix (plain / z-scored / lead jobs / device staging / abandoned
    uploads / injected copy failures, tools/sanitize/host_upload_test.cpp): clean runs, every destination byte equal to a scalar
    restatement (sanitizers run on the CPU build only)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh"), str(tmp_path), "1"], capture_output=True,
                       text=True, timeout=600)
    logs = "".join(open(os.path.join(tmp_path, f)).read()[-3000:] for f in sorted(os.listdir(tmp_path)) if f.endswith(".log"))
    assert r.returncode == 0, r.stdout + r.stderr + logs
    assert "asan_ubsan clean" in r.stdout and "tsan clean" in r.stdout


def test_mean_operator_tuple_grouping_and_column_layout():
    """engine/mean_refit.py's host side (round 6): voxels grouped by the alpha tuple they chose over the folds -- against a
    dictionary-based grouping --, tuples in ascending mixed-radix order, voxels ascending inside a tuple, every voxel exactly
    once; the padded column layout (every group on a 256-column tile boundary, -1 padding); the 16-bit and the wide key path."""
    from litcoder_core_amd.engine.mean_refit import MeanOperatorRefit as MO
    rng = np.random.default_rng(3)
    for (F, A, V, spread) in ((5, 20, 5000, 2), (4, 13, 777, 5), (3, 64, 300, 64), (7, 40, 2000, 9), (2, 3, 1, 1)):
        pool = rng.choice(A, size=min(spread, A), replace=False)
        best = [pool[rng.integers(0, len(pool), V)].astype(np.int64) for _ in range(F)]
        order, cnt, tuples = MO._alpha_tuples(best, A)
        want = {}
        for v in range(V):
            want.setdefault(tuple(int(b[v]) for b in best), []).append(v)
        assert sorted(order.tolist()) == list(range(V)) and int(cnt.sum()) == V and len(tuples) == len(want) == len(cnt)
        pos = 0
        for t, c in zip(tuples, cnt):
            assert order[pos:pos + c].tolist() == want[t], (F, A, V, t)         # ascending voxel order inside the tuple
            pos += int(c)
        used = [np.unique(b) for b in best]
        keys = [sum(int(np.searchsorted(u, a)) * int(np.prod([len(x) for x in used[:f]], dtype=np.int64)) for f, (u, a) in
                    enumerate(zip(used, t))) for t in tuples]
        assert keys == sorted(keys)                                              # ascending mixed-radix keys
        perm, start = MO._padded_groups(order, cnt)
        assert len(perm) == int(start[-1]) * 256 and start[0] == 0 and len(start) == len(cnt) + 1
        for g, c in enumerate(cnt):
            blk = perm[int(start[g]) * 256:int(start[g + 1]) * 256]
            assert blk[:c].tolist() == want[tuples[g]] and (blk[c:] == -1).all()
        # the folds' alphas handed in (from their histograms): exactly those that occur, or a superset -- the same grouping;
        # a list that misses an alpha somebody chose is an error, never a silent regrouping
        for extra in (0, 1):
            known = [set(u.tolist()) | ({int(rng.integers(0, A))} if extra else set()) for u in used]
            o2, c2, t2 = MO._alpha_tuples(best, A, used=known)
            assert np.array_equal(o2, order) and np.array_equal(c2, cnt) and t2 == tuples
        if V > 1 and len(used[0]) > 1 and np.prod([len(u) for u in used], dtype=np.float64) <= 65535:
            short = [set(u.tolist()) for u in used]
            short[0].discard(int(best[0][0]))
            with pytest.raises(RuntimeError):
                MO._alpha_tuples(best, A, used=short)
    # more key space than 16 bits: the sort-based path, same contract
    best = [rng.integers(0, 30, 400).astype(np.int64) for _ in range(5)]
    order, cnt, tuples = MO._alpha_tuples(best, 30)
    assert sorted(order.tolist()) == list(range(400)) and all(c >= 1 for c in cnt)
    pos = 0
    for t, c in zip(tuples, cnt):
        assert all(tuple(int(b[v]) for b in best) == t for v in order[pos:pos + c])
        pos += int(c)
