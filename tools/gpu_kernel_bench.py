#!/usr/bin/env python3
"""Micro-benchmarks of individual kernels at the BASELINE cfg2 shapes (one inner fold).
    python tools/gpu_kernel_bench.py [sweep] [lanczos] [chol]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from litcoder_core_amd import ops  # noqa: E402
from litcoder_core_amd._lib import LC_MB, LC_NB, LC_SCORE_CORR  # noqa: E402

dev = ops.device()
what = set(sys.argv[1:]) or {"sweep", "sweep16", "lanczos", "chol", "hbm"}


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if "sweep" in what:
    A, n_v, n_i, V, T = 20, 480, 1920, 80000, 3000
    M, N = ops.pad_to(n_v, LC_MB), ops.pad_to(n_i, LC_NB)
    g = torch.Generator(device=dev); g.manual_seed(0)
    H = torch.randn((A * M, N), generator=g, device=dev, dtype=torch.float32) * 0.02
    Y = torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    tr = ops.idx_tensor(np.r_[0:1920], N, dev)
    va = ops.idx_tensor(np.r_[1920:2400], M, dev)
    ystat = torch.empty((3, V), dtype=torch.float32, device=dev)
    yblk = torch.empty((M // LC_MB, V), dtype=torch.float32, device=dev)
    part = torch.empty((A * M // LC_MB, 4, V), dtype=torch.float32, device=dev)
    scores = torch.empty((A, V), dtype=torch.float32, device=dev)
    yv = torch.empty((M, V), dtype=torch.float32, device=dev)
    ops.val_stats(Y, V, va, M, n_v, ystat, yblk, yv)
    fn = lambda: ops.alpha_sweep_scores(H, A, M, N, Y, V, tr, yv, n_v, ystat, yblk, LC_SCORE_CORR, part, scores, False)
    ms = timeit(fn)
    fl = 2.0 * A * n_v * n_i * V
    print(f"alpha_sweep (gemm+finalize): {ms:.2f} ms  -> {fl / ms / 1e9:.1f} TFLOP/s algorithmic")
    # correctness spot check against torch fp64 on a few columns
    cols = [0, 17, 40000, 79999]
    pred = (H.double() @ Y[:1920][:, cols].double()).reshape(A, M, len(cols))[:, :n_v]
    yref = Y[1920:2400][:, cols].double()
    zy = (yref - yref.mean(0)) / (yref.std(0) + 1e-8)
    zp = (pred - pred.mean(1, keepdim=True)) / (pred.std(1, keepdim=True) + 1e-8)
    ref = (zy.unsqueeze(0) * zp).mean(1)
    print("   max |score - fp64 ref| on sample columns:", float((scores[:, cols].double() - ref).abs().max()))

if "sweep16" in what:
    A, n_v, n_i, V, T = 20, 480, 1920, 80000, 3000
    M, N = ops.pad_to(n_v, LC_MB), ops.pad_to(n_i, LC_NB)
    g = torch.Generator(device=dev); g.manual_seed(0)
    H = torch.randn((A * M, N), generator=g, device=dev, dtype=torch.float32) * 0.02
    Y = torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    tr = ops.idx_tensor(np.r_[0:1920], N, dev)
    va = ops.idx_tensor(np.r_[1920:2400], M, dev)
    ystat = torch.empty((3, V), dtype=torch.float32, device=dev)
    yblk = torch.empty((M // LC_MB, V), dtype=torch.float32, device=dev)
    part = torch.empty((A * M // LC_MB, 4, V), dtype=torch.float32, device=dev)
    scores = torch.empty((A, V), dtype=torch.float32, device=dev)
    yv = torch.empty((M, V), dtype=torch.float32, device=dev)
    ops.val_stats(Y, V, va, M, n_v, ystat, yblk, yv)
    rows_pad = ops.pad_to(A * M, 256)
    Ht = torch.empty(rows_pad * N * 2, dtype=torch.float16, device=dev)
    rs_inv = torch.empty(rows_pad, dtype=torch.float32, device=dev)
    Yt = torch.empty(ops.pad_to(V, 256) * N * 2, dtype=torch.float16, device=dev)
    cs, _flag = ops.col_scales_f16(Y, T, V)
    prep = lambda: (ops.split_rows_f16_alphas(H, 1, A, M, N, Ht, rs_inv), ops.split_cols_f16(Y, V, tr, N, cs, Yt))
    print(f"f16 operand split (H rows + Y cols): {timeit(prep):.2f} ms")
    fn = lambda: ops.alpha_sweep_scores_f16x3(Ht, rs_inv, A, M, N, Yt, cs[V:], yv, V, n_v, ystat, yblk, LC_SCORE_CORR,
                                              part, scores, False)
    ms = timeit(fn)
    fl = 2.0 * A * n_v * n_i * V
    print(f"alpha_sweep f16x3 (gemm+finalize): {ms:.2f} ms  -> {fl / ms / 1e9:.1f} TFLOP/s algorithmic, "
          f"{3 * fl / ms / 1e9:.0f} TFLOP/s of fp16 MFMA")
    cols = [0, 17, 40000, 79999]
    pred = (H.double() @ Y[:1920][:, cols].double()).reshape(A, M, len(cols))[:, :n_v]
    yref = Y[1920:2400][:, cols].double()
    zy = (yref - yref.mean(0)) / (yref.std(0) + 1e-8)
    zp = (pred - pred.mean(1, keepdim=True)) / (pred.std(1, keepdim=True) + 1e-8)
    ref = (zy.unsqueeze(0) * zp).mean(1)
    print("   max |score - fp64 ref| on sample columns:", float((scores[:, cols].double() - ref).abs().max()))
    if "stamps" in what:
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tools", "debug_kernels"))
        import build as debug_build                      # tools/bin/liblitcoder_debug.so: the diagnostics left the product ABI
        dbg = debug_build.load()
        st = torch.zeros(32, dtype=torch.int64, device=dev)
        p_ = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = dbg.lc_debug_sweep16_stamps(p_(Ht), p_(rs_inv), A, M, N, p_(Yt), p_(cs[V:]), p_(yv), ctypes.c_int64(V), n_v,
                                         p_(ystat), p_(part), p_(st), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        torch.cuda.synchronize()
        h = st.cpu().numpy().reshape(2, 16)
        for g_ in range(2):
            nw = max(int(h[g_, 5]), 1) / 120              # waves x tiles that contributed (KT = 120 each)
            ghz = h[g_, 0] / max(h[g_, 14], 1) * 0.1
            print(f"   waves {4 * g_}-{4 * g_ + 3}: per tile and wave: prologue {h[g_, 6] / nw:.0f}, main loop {h[g_, 0] / nw:.0f} "
                  f"({h[g_, 0] / nw / 120:.0f} per K-tile; in-kernel clock {ghz:.2f} GHz), epilogue {h[g_, 7] / nw:.0f} cycles "
                  f"(step 0 {h[g_, 9] / nw:.0f}, again {h[g_, 13] / nw:.0f}, steps 1-6 {h[g_, 10] / nw:.0f}, "
                  f"step 7 {h[g_, 11] / nw:.0f}, drain {h[g_, 12] / nw:.0f})")
        # the same stamps for the screening pass' form (HI2: one MFMA per product, 60 ring steps of two K-tiles; 4 alphas as in a fit)
        A4 = 4
        st = torch.zeros(32, dtype=torch.int64, device=dev)
        rc = dbg.lc_debug_sweep16_stamps_hi2(p_(Ht), p_(rs_inv), A4, M, N, p_(Yt), p_(cs[V:]), p_(yv), ctypes.c_int64(V), n_v,
                                             p_(ystat), p_(part), p_(st), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        torch.cuda.synchronize()
        h = st.cpu().numpy().reshape(2, 16)
        for g_ in range(2):
            nw = max(int(h[g_, 5]), 1) / 60
            ghz = h[g_, 0] / max(h[g_, 14], 1) * 0.1
            print(f"   HI2 (screening) waves {4 * g_}-{4 * g_ + 3}: per tile and wave: prologue {h[g_, 6] / nw:.0f}, main loop {h[g_, 0] / nw:.0f} "
                  f"({h[g_, 0] / nw / 60:.0f} per ring step of two K-tiles, MFMA floor 1024; in-kernel clock {ghz:.2f} GHz), epilogue {h[g_, 7] / nw:.0f} cycles "
                  f"(step 0 {h[g_, 9] / nw:.0f}, again {h[g_, 13] / nw:.0f}, steps 1-6 {h[g_, 10] / nw:.0f}, "
                  f"step 7 {h[g_, 11] / nw:.0f}, drain {h[g_, 12] / nw:.0f})")

if "plain16" in what:
    # the plain (store) launches of k_sweep_f16x3: series terms (2400 x 1920) and refit (3680 x 2400), V = 80000
    g = torch.Generator(device=dev); g.manual_seed(1)
    V = 80000
    Vt = ops.pad_to(V, 256)
    for rows, K, label in [(2048, 1920, "series terms 4 x 480 in 128-row slabs"), (3680, 2400, "refit p_pad + test rows")]:
        Amat = torch.randn((rows, K), generator=g, device=dev, dtype=torch.float32) * 0.02
        Y = torch.randn((K, V), generator=g, device=dev, dtype=torch.float32)
        rows_pad = ops.pad_to(rows, 256)
        At = torch.empty(rows_pad * K * 2, dtype=torch.float16, device=dev)
        rs = torch.empty(rows_pad, dtype=torch.float32, device=dev)
        Yt = torch.empty(Vt * K * 2, dtype=torch.float16, device=dev)
        cs, _ = ops.col_scales_f16(Y, K, V)
        cs_inv = torch.ones(Vt, dtype=torch.float32, device=dev); cs_inv[:V] = cs[V:]
        ops.split_rows_f16(Amat, rows, K, At, rs)
        ops.split_cols_f16(Y, V, torch.arange(K, dtype=torch.int32, device=dev), K, cs, Yt)
        C = torch.empty((rows, Vt), dtype=torch.float32, device=dev)
        fn = lambda: ops.gemm_grouped_f16x3(At, rs, rows, Yt, cs_inv, C, Vt, Vt, K, [0, Vt // 256])
        ms = timeit(fn)
        if rows == 2048:      # the series layout: 8 tiles, each a heavy (terms 0, 1) and a light (terms 2, 3) slab
            cls = torch.tensor([0, 1] * 8, dtype=torch.uint8, device=dev)
            fl_ = lambda: ops.gemm_grouped_f16x3(At, rs, rows_pad, Yt, cs_inv, C2, Vt, Vt, K, [0, Vt // 256], cls)
            C2 = torch.empty((rows_pad, Vt), dtype=torch.float32, device=dev)
            ms_l = timeit(fl_)
            ref_l = Amat[128:132].double() @ Y[:, :512].double()
            err_l = float((C2[128:132, :512].double() - ref_l).abs().max() / ref_l.abs().max())
            print(f"   with light slabs (8 of 16): {ms_l:.2f} ms; rel err of a light row {err_l:.1e}")
        fl = 2.0 * rows * K * V
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tools", "debug_kernels"))
        import build as debug_build
        dbg = debug_build.load()
        Cw = torch.zeros((rows, Vt), dtype=torch.float32, device=dev)
        i64 = ctypes.c_int64
        fw = lambda: dbg.lc_debug_gemm_f16x3_wide(ops._p(At), ops._p(rs), i64(rows), ops._p(Yt), ops._p(cs_inv), ops._p(Cw),
                                                  i64(Vt), i64(Vt), i64(K), ops._s())
        ms_w = timeit(fw)
        dw = float((Cw[:, :V] - C[:, :V]).abs().max() / C[:, :V].abs().max())
        print(f"   the same on v_mfma_f32_16x16x32_f16 (experiment kernel): {ms_w:.2f} ms; max difference to the 32x32x16 "
              f"result {dw:.1e} of the largest entry")
        ref = Amat[:4].double() @ Y[:, :512].double()
        err = float((C[:4, :512].double() - ref).abs().max() / ref.abs().max())
        print(f"plain f16x3 GEMM {label} ({rows} x {K} x {V}): {ms:.2f} ms -> {fl / ms / 1e9:.1f} TFLOP/s algorithmic, "
              f"{3 * fl / ms / 1e9:.0f} TF of fp16 MFMA; rel err {err:.1e}")

if "preproc" in what:
    # the preprocessing launches BASELINE configs[2] actually issues (harness.StoryPipeline.fit_words): ALL 27 stories'
    # Lanczos resampling in one launch (lc_lanczos_interp_stories) and their design matrix in one launch
    # (lc_story_design_f32), against the HBM roof on their algorithmic bytes (n_old D in_bytes + n_new D 8 for
    # the resampler; features read once + the float32 design written for the design kernel); and one story of the speech
    # shape (7000 x 1280 -> 350) through the same resampler
    import bench
    rng = np.random.default_rng(0)
    words, wtimes, trtimes, _brain = bench.synth_stories(256, dev)
    names = list(words)
    dW = torch.cat([torch.from_numpy(words[s_]).to(dev) for s_ in names])
    olds, news = [wtimes[s_] for s_ in names], [trtimes[s_] for s_ in names]
    n_old, n_new, D = dW.shape[0], sum(len(t) for t in news), dW.shape[1]

    def report_p(name, nbytes, ms):
        print(f"{name}: {ms * 1e3:.1f} us, {nbytes / 1e6:.1f} MB algorithmic -> {nbytes / ms / 1e6:.0f} GB/s "
              f"({nbytes / ms / 1e6 / 8000:.2f} of 8 TB/s)")
    # (the launch alone: tables and times are uploaded once here, as a caller that resamples repeatedly would)
    ms = timeit(lambda: ops.lanczos_interp_stories(dW, olds, news, 3, 1.0, False), reps=10)
    print(f"lanczos_interp_stories incl. its host-side table uploads: {ms * 1e3:.0f} us")
    import ctypes
    from litcoder_core_amd import _lib
    old_off = np.concatenate([[0], np.cumsum([len(t) for t in olds])])
    new_off = np.concatenate([[0], np.cumsum([len(t) for t in news])])
    cutoff = np.asarray([1.0 / np.mean(np.diff(t)) for t in news])
    table, _stride = ops.story_table([(np.int64, old_off[:-1]), (np.int64, [len(t) for t in olds]), (np.int64, new_off[:-1]),
                                      (np.float64, cutoff), (np.int32, np.ones(len(olds), dtype=np.int32))], dev)
    row_story = ops.upload(np.repeat(np.arange(len(news), dtype=np.int32), [len(t) for t in news]), dev)
    d_old, d_new = ops.upload(np.concatenate(olds), dev), ops.upload(np.concatenate(news), dev)
    feat = torch.empty((n_new, D), dtype=torch.float64, device=dev)
    launch = lambda: _lib.call("lc_lanczos_interp_stories", ops._p(dW), 0, D, dW.stride(0), ops._p(d_old), ops._p(d_new), n_new,
                               ops._p(row_story), ops._p(table), len(news), 3.0, 0, ops._p(feat), D, ops._s())
    ms = timeit(launch, reps=20)
    report_p(f"k_lanczos_rows, {len(news)} stories: {n_old} words x {D} f32 -> {n_new} TRs f64 (one launch)",
             n_old * D * 4 + n_new * D * 8, ms)
    pipe_ = __import__("litcoder_core_amd").StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM)
    n_in = [len(t) for t in news]
    ms = timeit(lambda: pipe_.design(feat, new_off, n_in, names), reps=10)
    dX_, T_, Tt_, p_, _ = pipe_.design(feat, new_off, n_in, names)
    report_p(f"k_story_design, {len(news)} stories: FIR x 4 + trim + zs + nan_to_num + f32 cast -> ({T_ + Tt_}, {p_}) "
             "(incl. the zero fill of the design and the table upload)", n_new * D * 8 + (T_ + Tt_) * p_ * 4, ms)
    # one story at the speech shape
    n_o, Ds = 7000, 1280
    ot_ = np.sort(rng.uniform(0, 700, n_o))
    nt_ = 1.0 + 2.0 * np.arange(350)
    ds_ = torch.randn((n_o, Ds), device=dev, dtype=torch.float64)
    ms = timeit(lambda: ops.lanczos_interp_stories(ds_, [ot_], [nt_], 3, 1.0, False), reps=20)
    report_p(f"lanczos, one story {n_o}x{Ds}->350 f64 (incl. table uploads)", n_o * Ds * 8 + 350 * Ds * 8, ms)

if "series" in what:
    # lc_series_scores at the cfg2 shape: 4 terms x 480 rows x 80000 voxels, 16 alphas
    g = torch.Generator(device=dev); g.manual_seed(2)
    V, M, n_v, TERMS = 80000, 480, 480, 4
    Vt = ops.pad_to(V, 256)
    rows = 2560
    Tb = torch.randn((rows, Vt), generator=g, device=dev, dtype=torch.float32)
    Y = torch.randn((3000, V), generator=g, device=dev, dtype=torch.float32)
    va = ops.idx_tensor(np.r_[1920:2400], M, dev)
    ystat = torch.empty((3, V), dtype=torch.float32, device=dev)
    yblk = torch.empty((M // LC_MB, V), dtype=torch.float32, device=dev)
    yv = torch.empty((M, V), dtype=torch.float32, device=dev)
    ops.val_stats(Y, V, va, M, n_v, ystat, yblk, yv)
    from litcoder_core_amd import series as lc_series
    coef = torch.tensor(np.stack([lc_series.minimax_inverse_coefficients(a, TERMS) for a in np.logspace(-1, 8, 20)[4:]]),
                        dtype=torch.float64, device=dev)
    aidx = torch.arange(4, 20, dtype=torch.int32, device=dev)
    scores = torch.zeros((20, V), dtype=torch.float32, device=dev)
    rowmap_h = np.concatenate([np.arange(480) + 512 * j for j in range(TERMS)]).astype(np.int32)
    rowmap = torch.from_numpy(rowmap_h).to(dev)
    for rm, label in ((None, "contiguous rows"), (rowmap, "row map")):
        ms = timeit(lambda: ops.series_scores(Tb, Vt, TERMS, M, n_v, V, yv, ystat, coef, aidx, scores, False, rm))
        print(f"series_scores {TERMS} x 480 x 80000, 16 alphas, {label}: {ms * 1e3:.0f} us "
              f"({TERMS * n_v * V * 4 / ms / 1e6:.0f} GB/s of term reads)")
    ms = timeit(lambda: ops.val_stats(Y, V, va, M, n_v, ystat, yblk, yv))
    print(f"val_stats 480 x 80000: {ms * 1e3:.0f} us")

if "lanczos" in what:
    rng = np.random.default_rng(0)
    for (T, p, ar) in [(2400, 3072, 0.0), (2400, 3072, 0.8), (2400, 768, 0.5)]:
        x = rng.standard_normal((T, p))
        if ar:
            for t in range(1, T):
                x[t] = ar * x[t - 1] + np.sqrt(1 - ar * ar) * x[t]
        x = x.astype(np.float32)
        dx = ops.upload_f32(x, ops.pad_to(p, 32), dev)
        K = ops.gram(dx, T, p)
        rows = np.arange(0, 1920)
        N = ops.pad_to(len(rows), LC_NB)
        idx = ops.idx_tensor(rows, N, dev).reshape(1, N)
        ref = float(torch.linalg.eigvalsh(K[:1920, :1920].cpu())[-1])
        errs = []
        for steps in (16, 24, 32, 48, 64, 96, 128, 192):
            lm = float(ops.lambda_max(K, idx, 1, N, steps).cpu()[0])
            errs.append(f"{steps}:{abs(lm - ref) / ref:.1e}")
        print(f"lambda_max T{T} p{p} ar{ar}: " + " ".join(errs))
    idx5 = torch.stack([ops.idx_tensor(np.r_[0:480 * f, 480 * (f + 1):2400], 1920, dev) for f in range(5)])
    for steps in (64, 192):
        ms = timeit(lambda: ops.lambda_max(K, idx5, 5, 1920, steps), reps=3, warm=1)
        print(f"lambda_max F=5 N=1920 steps={steps}: {ms:.2f} ms")
    # all 30 systems of a 5x5 nested CV over T=3000 in one masked run
    T3 = 3000
    x3 = rng.standard_normal((T3, 3072)).astype(np.float32)
    K3 = ops.gram(ops.upload_f32(x3, 3072, dev), T3, 3072)
    sets = []
    for o in range(5):
        tr_o = np.r_[0:o * 600, (o + 1) * 600:T3]
        sets += [np.delete(tr_o, np.s_[i * 480:(i + 1) * 480]) for i in range(5)] + [tr_o]
    bits = np.zeros(T3, dtype=np.uint32)
    for f_, rows_ in enumerate(sets):
        bits[rows_] |= np.uint32(1 << f_)
    member = torch.from_numpy(bits.view(np.int32)).to(dev)
    ms = timeit(lambda: ops.lambda_max_masked(K3, T3, member, len(sets), 64), reps=3, warm=1)
    got = ops.lambda_max_masked(K3, T3, member, len(sets), 64).cpu().numpy()
    Kh = K3.cpu().numpy()
    ref = np.array([np.linalg.eigvalsh(Kh[np.ix_(r_, r_)])[-1] for r_ in sets[:6]])
    print(f"lambda_max_masked 30 systems T=3000 steps=64 (fp64 MFMA matvec): {ms:.2f} ms; max rel err vs eigvalsh "
          f"(first 6): {np.max(np.abs(got[:6] / ref - 1)):.2e}")
    ms0 = timeit(lambda: ops.lambda_max_masked(K3, T3, member, len(sets), 64, use_mfma=False), reps=3, warm=1)
    got0 = ops.lambda_max_masked(K3, T3, member, len(sets), 64, use_mfma=False).cpu().numpy()
    print(f"   vector-ALU matvec (round 1): {ms0:.2f} ms; max rel difference of the 30 values: "
          f"{np.max(np.abs(got / got0 - 1)):.2e}")

if "chol" in what:
    rng = np.random.default_rng(1)
    T, p = 3000, 3072
    x = rng.standard_normal((T, p)).astype(np.float32)
    dx = ops.upload_f32(x, p, dev)
    K = ops.gram(dx, T, p)
    F, A, N, M = 5, 20, 1920, 480
    tr = torch.stack([ops.idx_tensor(np.r_[0:480 * f, 480 * (f + 1):2400], N, dev) for f in range(F)])
    va = torch.stack([ops.idx_tensor(np.r_[480 * f:480 * (f + 1)], M, dev) for f in range(F)])
    lm = ops.lambda_max(K, tr, F, N, 64)
    a2 = ops.penalties(lm, F, torch.tensor(np.logspace(-1, 8, A), device=dev), True)
    aug = torch.empty((F * A, N + M, N), dtype=torch.float64, device=dev)
    H = torch.empty((F * A, M, N), dtype=torch.float32, device=dev)
    def run():
        ops.batch_assemble(K, tr, va, None, a2, F, A, N, M, aug)
        ops.batch_chol_solve(aug, F * A, N, M, H)
    ms = timeit(run, reps=2, warm=1)
    fl = F * A * (N ** 3 / 3 + 2.0 * N * N * M)
    print(f"assemble + batch_chol_solve B={F * A} N={N} M={M}: {ms:.1f} ms -> {fl / ms / 1e9:.1f} TFLOP/s fp64")
    # the shapes of the fit: inner batch (5 folds x 4 Cholesky alphas) and the refit systems (4 alphas, 3680 rows)
    for (B_, N_, M_, label) in ((20, 1920, 480, "inner folds"), (4, 2432, 3680, "refit")):
        aug2 = torch.randn((B_, N_ + M_, N_), dtype=torch.float64, device=dev)
        aug2[:, :N_] = torch.eye(N_, dtype=torch.float64, device=dev) * (4.0 * N_) + aug2[:, :N_] * 0.0 + 1.0
        H2 = torch.empty((B_, M_, N_), dtype=torch.float32, device=dev)
        base = aug2.clone()
        copt = [None]
        def run2():
            aug2.copy_(base)
            ops.batch_chol_solve(aug2, B_, N_, M_, H2, options=copt[0])
        fl2 = B_ * (N_ ** 3 / 3 + 2.0 * N_ * N_ * M_)
        for valu, ob in ((2, 256), (2, 512), (1, 256), (1, 512), (0, 256)):
            copt[0] = ops.chol_options(outer_block=ob, big_kernel=valu)
            ops.timing_enable(True); ops.timing_read()
            ms2 = timeit(run2, reps=3, warm=1)
            kt = ops.timing_read(); ops.timing_enable(False)
            chol_ms = kt.get("batch_chol_solve", (0, 1))
            print(f"batch_chol_solve {label} B={B_} N={N_} M={M_} outer block {ob} deep updates on {('MFMA 16x16x4', 'VALU', 'MFMA 4x4x4')[valu]}: {chol_ms[0] / chol_ms[1]:.2f} ms -> "
                  f"{fl2 / (chol_ms[0] / chol_ms[1]) / 1e9:.1f} TFLOP/s fp64")
    a64 = torch.randn((4096, 4096), dtype=torch.float64, device=dev)
    b64 = torch.randn((4096, 4096), dtype=torch.float64, device=dev)
    ms = timeit(lambda: torch.matmul(a64, b64), reps=5, warm=2)
    print(f"for reference, torch.matmul fp64 4096^3 (vendor BLAS): {ms:.2f} ms -> {2 * 4096 ** 3 / ms / 1e9:.1f} TFLOP/s")

if "hbm" in what:
    # HBM-bound kernels: algorithmic bytes / time against the 8 TB/s datasheet rate (6.3 TB/s achievable copy)
    rng = np.random.default_rng(3)
    g = torch.Generator(device=dev); g.manual_seed(1)
    def report(name, nbytes, ms):
        print(f"{name}: {ms * 1e3:.1f} us, {nbytes / 1e6:.1f} MB algorithmic -> {nbytes / ms / 1e6:.0f} GB/s "
              f"({nbytes / ms / 1e6 / 8000:.2f} of 8 TB/s)")
    # FIR: one LeBel story (350 TRs) and the concatenated design (3000 rows), 768 features x 4 delays, f64
    for nt in (350, 3000):
        x = torch.randn((nt, 768), generator=g, device=dev, dtype=torch.float64)
        ms = timeit(lambda: ops.fir_delay(x, [1, 2, 3, 4], False), reps=20)
        report(f"fir_delay {nt}x768x4 f64", nt * 768 * (8 + 4 * 8), ms)
    # Lanczos: 2500 words x 768 -> 350 TRs (LeBel story), speech 7000 x 1280 -> 350
    for (n_old, D) in ((2500, 768), (7000, 1280)):
        ot = torch.from_numpy(np.sort(rng.uniform(0, 700, n_old))).to(dev)
        nt_ = torch.from_numpy(1.0 + 2.0 * np.arange(350)).to(dev)
        d = torch.randn((n_old, D), generator=g, device=dev, dtype=torch.float64)
        ms = timeit(lambda: ops.lanczos_interp(d, ot, nt_, 0.5, 3, False), reps=20)
        report(f"lanczos {n_old}x{D}->350 f64", n_old * D * 8 + 350 * D * 8, ms)
    # Pearson r + p over 80000 voxels, 600 test rows
    a = torch.randn((600, 80000), generator=g, device=dev, dtype=torch.float32)
    b = a * 0.3 + torch.randn((600, 80000), generator=g, device=dev, dtype=torch.float32)
    ms = timeit(lambda: ops.pearson_cols(a, b, 600, 80000), reps=10)
    report("pearson_cols 600x80000 f32", 2 * 600 * 80000 * 4, ms)
    r = ops.pearson_cols(a, b, 600, 80000)
    ms = timeit(lambda: ops.pearson_pvalues(r, 80000, 600), reps=10)
    print(f"pearson_pvalues 80000 voxels: {ms * 1e3:.1f} us")
    # validation-target statistics and the alpha-sorted gather
    Y = torch.randn((3000, 80000), generator=g, device=dev, dtype=torch.float32)
    va = ops.idx_tensor(np.r_[1920:2400], 480, dev)
    ystat = torch.empty((3, 80000), dtype=torch.float32, device=dev)
    yblk = torch.empty((15, 80000), dtype=torch.float32, device=dev)
    yv_ = torch.empty((480, 80000), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.val_stats(Y, 80000, va, 480, 480, ystat, yblk, yv_), reps=10)
    report("val_stats 480x80000 f32", 480 * 80000 * 4, ms)
    perm = torch.randperm(80000, generator=g, device=dev).to(torch.int32)
    rows = ops.idx_tensor(np.r_[0:3000], 3000, dev)
    out = torch.empty((3000, 80000), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.gather(Y, 80000, rows, 3000, perm, 80000, out), reps=5)
    report("gather (alpha-sorted copy) 3000x80000 f32", 2 * 3000 * 80000 * 4, ms)
    # ---- round 4: the operand passes of the LeBel-style fit (cfg3: 1 844 validation rows, p = 3072), one voxel panel wide
    del out, Y
    Vp_ = 30720
    Yb = torch.randn((9513, Vp_), generator=g, device=dev, dtype=torch.float32)
    vab = ops.idx_tensor(np.r_[1844:3688], 1856, dev)
    ystat = torch.empty((3, Vp_), dtype=torch.float32, device=dev)
    yblk = torch.empty((58, Vp_), dtype=torch.float32, device=dev)
    yv_ = torch.empty((1856, Vp_), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.val_stats(Yb, Vp_, vab, 1856, 1844, ystat, yblk, yv_), reps=10)
    report(f"val_stats 1844x{Vp_} f32 (chunked: two reads, one write of the rows)", 3 * 1844 * Vp_ * 4, ms)
    ms = timeit(lambda: ops.col_scales_f16(Yb, 9513, Vp_, want_flag=True), reps=10)
    report(f"col_scales + flag 9513x{Vp_} f32 (two passes)", 2 * 9513 * Vp_ * 4, ms)
    Bs = [torch.randn((3072, Vp_), generator=g, device=dev, dtype=torch.float32) for _ in range(4)]
    Bo = torch.empty((3072, Vp_), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.combine_colmax(Bs, [1.0] * 4, Bo, Vp_, want_scales_for=Vp_), reps=10)
    report(f"combine_colmax 4 terms 3072x{Vp_} f32 (sum + column maxima + scales)", 5 * 3072 * Vp_ * 4, ms)
    ms = timeit(lambda: (ops.combine_many(Bs, [1.0] * 4, Bo), ops.col_scales_f16(Bo, 3072, Vp_, want_flag=False)), reps=10)
    report("   the two passes it replaces (combine_terms, col_scales)", 6 * 3072 * Vp_ * 4, ms)
    csb, _ = ops.col_scales_f16(Bo, 3072, Vp_, want_flag=False)
    Bt_ = torch.empty(Vp_ * 3072 * 2, dtype=torch.float16, device=dev)
    ident_ = ops.idx_tensor(np.arange(3072), 3072, dev)
    ms = timeit(lambda: ops.split_cols_f16(Bo, Vp_, ident_, 3072, csb, Bt_), reps=10)
    report(f"split_cols_f16 3072x{Vp_} (f32 in, fp16 hi + lo image out)", 2 * 3072 * Vp_ * 4, ms)
    # the test rows of a cfg2 refit: product + Pearson r, stored route against the reduced one
    Kr, n_t, Vr = 2400, 600, 80128
    Ar = (torch.randn((n_t, Kr), generator=g, device=dev, dtype=torch.float32) / 49.0)
    Yr = torch.randn((Kr, Vr), generator=g, device=dev, dtype=torch.float32)
    Yte = torch.randn((n_t, Vr), generator=g, device=dev, dtype=torch.float32)
    Atr = torch.empty(ops.pad_to(n_t, 256) * Kr * 2, dtype=torch.float16, device=dev)
    rsr = torch.empty(ops.pad_to(n_t, 256), dtype=torch.float32, device=dev)
    ops.split_rows_f16(Ar, n_t, Kr, Atr, rsr)
    csr, _ = ops.col_scales_f16(Yr, Kr, Vr, want_flag=False)
    Ytr_ = torch.empty(Vr * Kr * 2, dtype=torch.float16, device=dev)
    ops.split_cols_f16(Yr, Vr, ops.idx_tensor(np.arange(Kr), Kr, dev), Kr, csr, Ytr_)
    csi = csr[Vr:].contiguous()
    predr = torch.empty((n_t, Vr), dtype=torch.float32, device=dev)
    tiles_r = [0, Vr // 256]
    ms_a = timeit(lambda: (ops.gemm_grouped_f16x3(Atr, rsr, n_t, Ytr_, csi, predr, Vr, Vr, Kr, tiles_r),
                           ops.pearson_cols(Yte, predr, n_t, Vr)), reps=10)
    r_out = torch.empty(Vr, dtype=torch.float64, device=dev)
    ms_b = timeit(lambda: ops.gemm_grouped_f16x3_pearson(Atr, rsr, n_t, Ytr_, csi, Vr, Kr, tiles_r, Yte, None, None, r_out), reps=10)
    print(f"test rows of a refit ({n_t} x {Kr} x {Vr}): product stored + pearson_cols {ms_a * 1e3:.0f} us; reduced in the epilogue "
          f"{ms_b * 1e3:.0f} us ({2.0 * n_t * Kr * Vr / ms_b / 1e9:.0f} TF algorithmic); max |r difference| "
          f"{float((r_out - ops.pearson_cols(Yte, predr, n_t, Vr)).abs().max()):.1e}")
