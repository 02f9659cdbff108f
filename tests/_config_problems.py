"""The inputs of the config-shape fixtures (tests/golden/configs.npz), rebuilt from seeds with numpy alone.

tests/golden/make_golden_configs.py runs the REFERENCE on these inputs in the build container and stores only seeds +
outputs; the ``-m gpu`` config tests rebuild the same inputs on the GPU box and compare the HIP path with the stored
reference outputs -- no CPU SVD on the GPU box.  ``N_FIX`` fixture voxels per config; a test embeds them as the first
columns of a volume of the config's full width (voxels are independent under per-voxel alphas).

``checks(problem)`` = a few float64 sums of the inputs, stored with the fixture: the rebuilt inputs must agree to 1e-9
relative (BLAS builds may differ in the last bits of X @ W, nothing more).
"""
import numpy as np

import oracle.fir as ofir
import oracle.harness as oharness
import oracle.lanczos as olanczos

N_FIX = 256
TRIM = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0, "train_targets_end": None,
        "test_features_start": 50, "test_features_end": -5, "test_targets_start": 40, "test_targets_end": None}

CONFIGS = {
    # BASELINE.json configs[1]: synthetic T=3000, F=768 x 4 delays, 20 alphas, 5 x 5 K-folds
    "cfg2": dict(T=3000, F0=768, delays=[1, 2, 3, 4], wscale=0.02, seed=1002,
                 kw=dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, alphas=np.logspace(-1, 8, 20))),
    # configs[3]: Narratives-like T=2226
    "cfg4": dict(T=2226, F0=768, delays=[1, 2, 3, 4], wscale=0.02, seed=1004,
                 kw=dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, alphas=np.logspace(-1, 8, 20))),
    # configs[4]: Whisper-like 1280 x 6 delays, 32 alphas, two bands with penalty scales (1, 2) = ridge on X / gamma
    "cfg5": dict(T=3000, F0=1280, delays=[1, 2, 3, 4, 5, 6], wscale=0.015, seed=1005, band_scales=(1.0, 2.0),
                 kw=dict(folding_type="kfold", n_outer_folds=5, n_inner_folds=5, alphas=np.logspace(-1, 8, 32))),
}
# configs[2]: LeBel-like story pipeline, train/test, example.py:104-117 with its argparse defaults (K-folds, default grid)
CFG3 = dict(seed=1003, n_train=26, D=768, delays=[1, 2, 3, 4], wscale=0.012,
            kw=dict(folding_type="kfold", n_inner_folds=5, chunk_length=20, alphas=np.logspace(-1, 8, 10)))


def matrix_problem(name, n_vox=N_FIX):
    """(X (T, p) float64, Y (T, n_vox) float64, kwargs) of cfg2 / cfg4 / cfg5 (the latter on the rescaled design)."""
    c = CONFIGS[name]
    rng = np.random.default_rng(c["seed"])
    X = ofir.make_delayed(rng.standard_normal((c["T"], c["F0"])), c["delays"])
    p = X.shape[1]
    if "band_scales" in c:
        half = p // 2
        X = X / np.r_[np.full(half, c["band_scales"][0]), np.full(p - half, c["band_scales"][1])]
    # signal strengths spread over two decades, so that the voxels' alpha choices spread over the grid; a constant and
    # a pure-noise voxel on top (the reference's (r 0, p 1, alphas[0]) case and a tie-prone one)
    W = c["wscale"] * rng.standard_normal((p, n_vox)) * np.exp(rng.uniform(np.log(0.03), np.log(3.0), n_vox))
    Y = X @ W + rng.standard_normal((c["T"], n_vox))
    if n_vox > 6:
        Y[:, 5] = 1.25
        Y[:, 6] = rng.standard_normal(c["T"])
    return X, Y, dict(c["kw"])


def story_lengths(seed, n_train):
    rng = np.random.default_rng(seed)
    return [int(n) for n in rng.integers(260, 440, n_train)] + [291]


def story_problem(n_vox=N_FIX, cfg=CFG3, n_train=None):
    """cfg3: per story word-level float32 features at irregular word times, TR times (15 more feature TRs than brain TRs:
    LeBel trimming [10:-5]), and float64 brain data of ``n_vox`` voxels = z-scored delayed features (the oracle's
    Lanczos + FIR + zs) @ W + noise.  Returns dict(words, wtimes, trtimes, brain, kw, trimming, delays)."""
    n_train = cfg["n_train"] if n_train is None else n_train
    rng = np.random.default_rng(cfg["seed"])
    lengths = story_lengths(cfg["seed"] + 1, n_train)
    D = cfg["D"]
    W = cfg["wscale"] * rng.standard_normal((D * len(cfg["delays"]), n_vox)) * np.exp(
        rng.uniform(np.log(0.03), np.log(3.0), n_vox))
    words, wtimes, trtimes, brain = {}, {}, {}, {}
    for i, n_tr in enumerate(lengths):
        name = "story%02d" % i
        n_words = int(7.2 * n_tr)
        wt = np.sort(rng.uniform(0, 2.0 * (n_tr + 15), n_words))
        emb = rng.standard_normal((n_words, D)).astype(np.float32)
        emb[1:] = 0.6 * emb[:-1] + 0.8 * emb[1:]                         # smooth like LM states
        tr_t = 1.0 + 2.0 * np.arange(n_tr + 15)
        words[name], wtimes[name], trtimes[name] = emb, wt, tr_t
        ds = olanczos.lanczos_interp(emb, wt, tr_t, window=3, cutoff_mult=1.0)
        xs = oharness.zs(ofir.make_delayed(ds, cfg["delays"])[10:-5])
        brain[name] = xs @ W + rng.standard_normal((n_tr, n_vox))
    return dict(words=words, wtimes=wtimes, trtimes=trtimes, brain=brain, kw=dict(cfg["kw"]), trimming=dict(TRIM),
                delays=list(cfg["delays"]))


def checks(*arrays):
    """Float64 fingerprints of input arrays: (sum, sum of squares, a strided sample's sum)."""
    out = []
    for a in arrays:
        a = np.asarray(a, dtype=np.float64)
        out += [float(a.sum()), float((a * a).sum()), float(a.reshape(-1)[::97].sum())]
    return np.asarray(out)
