#!/usr/bin/env python3
"""Random story-structured problems through harness.StoryPipeline.fit_words (all stories' Lanczos resampling and their
design matrix in one launch each, brain data z-scored in the upload threads, voxel panels, the panelled single-alpha path)
against the TWO-STEP route on the same inputs: per story Downsampler.downsample(method="lanczos") -> FIR.make_delayed ->
trim -> numpy zs -> vstack -> nan_to_num (the oracle's harness: trainer.py:203-262 restated), then NestedCVModel.fit_predict
on those matrices.  The design must be equal BIT FOR BIT, and so must alphas, weights and correlations: the pipeline changes
when bytes move, not what is computed.  A bug hunt, not a test.     python tools/fuzz_pipeline.py [n_cases [seed]]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import litcoder_core_amd as lc  # noqa: E402
import oracle.fir as ofir  # noqa: E402
import oracle.harness as oh  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
fails = 0
forms = {}
for case in range(n_cases):
    n_st = int(rng.integers(3, 10))
    D = int(rng.choice([8, 32, 64, 100]))
    delays = sorted(int(d) for d in rng.choice(np.arange(1, 7), size=int(rng.integers(1, 5)), replace=False))
    if rng.random() < 0.2:
        delays = delays[::-1]
    V = int(rng.choice([100, 300, 700, 1300]))
    fs, fe = int(rng.integers(0, 13)), int(rng.integers(0, 7))
    ts = int(rng.choice([0, 0, 3]))
    trim = {"train_features_start": fs, "train_features_end": -fe if fe else None, "train_targets_start": ts,
            "train_targets_end": None, "test_features_start": fs, "test_features_end": -fe if fe else None,
            "test_targets_start": ts, "test_targets_end": None}
    wdt = np.float32 if rng.random() < 0.7 else np.float64
    p = D * len(delays)
    Wt = rng.standard_normal((p, V)) * 0.05 * rng.uniform(0.0, 2.0, V)
    words, wtimes, trtimes, brain = {}, {}, {}, {}
    for i in range(n_st):
        nm = f"s{i}"
        n_tr = int(rng.integers(40, 200))
        n_all = n_tr + fs + fe
        nw = int(rng.uniform(3.0, 8.0) * n_all)
        t = np.sort(rng.uniform(0, 2.0 * n_all, nw))
        if rng.random() < 0.1:
            t[[3, 7]] = t[[7, 3]]                          # unsorted sample times: the full-scan path of the kernel
        wtimes[nm] = t
        emb = rng.standard_normal((nw, D)).astype(wdt)
        emb[1:] = 0.6 * emb[:-1] + 0.8 * emb[1:]
        words[nm] = emb
        trtimes[nm] = 1.0 + 2.0 * np.arange(n_all)
        brain[nm] = 3.0 * rng.standard_normal((n_tr + ts, V)) + 100.0
    if rng.random() < 0.3:
        brain[f"s{int(rng.integers(0, n_st))}"][:, int(rng.integers(0, V))] = 7.0     # constant within one story
    kw = dict(folding_type="kfold", n_inner_folds=int(rng.choice([2, 3, 5])),
              alphas=np.logspace(-1, rng.uniform(2, 8), int(rng.integers(3, 11))), single_alpha=bool(rng.random() < 0.6),
              use_corr=bool(rng.random() < 0.85), normalpha=bool(rng.random() < 0.8))
    precision = str(rng.choice(["auto", "auto", "f32"]))
    panel_cols = int(rng.choice([0, 256, 256]))
    tag = (f"case {case}: {n_st} stories D{D} delays{delays} V{V} trim({fs},{-fe if fe else None};{ts}) {np.dtype(wdt).name} "
           f"{precision} panels{panel_cols} " + " ".join(f"{k}={v}" for k, v in kw.items() if k not in ("alphas", "folding_type"))
           + f" A={len(kw['alphas'])}")
    try:
        names = list(words)
        # the planted signal lives in the design of the two-step route: build that first
        delayed = {s: ofir.make_delayed(lc.Downsampler().downsample(words[s], wtimes[s], trtimes[s], method="lanczos", window=3,
                                                                    cutoff_mult=1.0), delays) for s in names}
        mt0 = oh.train_test_matrices(delayed, brain, trim)
        sig = np.concatenate([mt0["Rstim"], mt0["Pstim"]]) @ Wt
        row = 0
        for s in names:
            n = brain[s].shape[0] - ts
            brain[s][ts:] += 3.0 * sig[row:row + n]
            row += n
        mt = oh.train_test_matrices(delayed, brain, trim)
        model = lc.NestedCVModel("r", precision=precision, panel_cols=panel_cols)
        pipe = lc.StoryPipeline(delays, trim, model=model)
        ours = pipe.fit_words(words, wtimes, trtimes, brain, window=3, cutoff_mult=1.0, **kw)
        dX, T, Tt, p_ = pipe.last_design
        Xs = dX[:, :p_].cpu().numpy()
        assert (T, Tt, p_) == (mt["Rstim"].shape[0], mt["Pstim"].shape[0], p), "shapes of the design"
        assert np.array_equal(Xs[:T], mt["Rstim"].astype(np.float32)) and np.array_equal(Xs[T:], mt["Pstim"].astype(np.float32)), \
            "the design differs from the per-story route"
        two_model = lc.NestedCVModel("r", precision=precision)
        two = two_model.fit_predict(mt["Rstim"], mt["Rresp"], X_test=mt["Pstim"], y_test=mt["Presp"], **kw)
        assert np.array_equal(ours[2], two[2]), f"alphas differ at {int((np.asarray(ours[2]) != np.asarray(two[2])).sum())} voxels"
        assert np.array_equal(ours[1], two[1], equal_nan=True), "weights differ"
        for k, v in two[0].items():
            g = ours[0][k]
            same = (np.array_equal(np.asarray(g), np.asarray(v), equal_nan=True) if isinstance(v, list) else (g == v or (g != g and v != v)))
            assert same, f"metrics[{k}] differs"
        key = f"{model.last_form} {model.last_fit.get('precision')} panels {len(model.last_fit.get('panels') or [])}" + (
            f" guess {model.last_fit.get('single_alpha_guess')}" if kw["single_alpha"] else "")
        forms[key] = forms.get(key, 0) + 1
        print("ok  ", tag, "->", key, flush=True)
    except Exception as e:                                   # noqa: BLE001
        fails += 1
        print("FAIL", tag, "\n     ", type(e).__name__, str(e)[:400], flush=True)
        if not isinstance(e, AssertionError):
            traceback.print_exc()
print(f"{n_cases - fails} of {n_cases} story pipelines equal the two-step route bit for bit; {forms}")
sys.exit(1 if fails else 0)
