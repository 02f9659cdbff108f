#!/usr/bin/env python3
"""Headline benchmark: voxels/sec of one full nested-CV ridge fit (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): synthetic T=3000, F=768 x 4 FIR delays
(p=3072), V=80000 voxels PER GPU, 20 alphas logspace(-1, 8), 5 outer x 5 inner contiguous K-folds,
per-voxel alpha, normalpha, correlation scoring.  One "step" = one complete call of the reference's
entry point, SURVEY.md 8d's metric as written: NestedCVModel.fit_predict(features, targets) with float64
numpy arrays in pageable host memory in -> metrics dict + float32 host weights + alphas out (Gram, 25
inner alpha sweeps, 5 refits, test scoring, statistics; H2D of the inputs and D2H of the weights
INSIDE the timed region, every step fenced by a device synchronisation, so `ms_per_step` is one call's
latency).  `value` = voxels of all ranks x steps / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--voxels V] [--scaling weak|strong] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 and no launcher around it (WORLD_SIZE unset) the process starts its N ranks itself (launch_ranks: N child
processes, one GPU each, 127.0.0.1 rendezvous), relays rank 0's line and exits with the worst rank's return code.

Scaling: "weak" (default) = every rank fits `--voxels` voxels; "strong" = `--voxels` voxels IN TOTAL,
split over the ranks (north_star's 80 000-voxel job on 1/2/4/8 GPUs).  At N > 1 the line carries both:
the headline `value` in the chosen mode and the other mode under `other_scaling`.

Extra legs in the same JSON line (N = 1): `resident_path` = the same fit with the fp32 inputs already in
HBM and the weights left there (what `value` was in rounds 1-2), with `roofline_full_width_launches` = the dominant
kernel's roofline over that leg's 25 full-width launches per step; `f32_path` = the resident fit with the
exact-fp32 MFMA sweep (precision="f32"), with a roofline of its own, and `parity_vs_f32_path` = how far
the headline's f16x3 results are from it over ALL voxels of the bench data.

`roofline.traffic` (N = 1): measured for THIS run -- before the process touches the GPU, two child passes
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over one host-to-host fit (`measured_traffic`; about a
minute; `--no-traffic` skips them, and so does running under a profiler): bytes per launch averaged over the same
mixed-width launches `avg_launch_ms` averages over; the full-width launches' figure sits in
`resident_path.roofline_full_width_launches.traffic` next to THEIR average duration, with the committed profile's
figure beside it (`traffic_in_committed_profile`, also the fallback there).

`cfg3_pipeline` (N = 1): BASELINE configs[2] -- the config north_star's target sentence names -- end to end and host to
host through harness.StoryPipeline.fit_words (word features -> Lanczos -> FIR -> per-story zs -> train/test fit,
single_alpha), with its link floor measured on this box and a `cpu_baseline` of its own (`--no-cfg3` skips it).

Prints ONE JSON line on rank 0.  The `roofline` object is for the dominant kernel, the fp16 MFMA contraction
k_sweep_f16x3, in the two launches an inner fold of the inner CV makes of it -- the fused score launch (the hat matrices
of the alphas that go through the batched Cholesky, their predictions reduced to scores in the epilogue) and the
series-moments launch (the four shared series terms of the large alphas, reduced to moments) -- since round 6 in the
screening arithmetic (ONE MFMA per product, hi planes): the flops those launches contract / their HIP-event time over
the timed steps, against the 2.5 PFLOP/s dense fp16 MFMA peak; `roofline.score_launches` alone is what rounds 1-5 reported.
The same kernel's other launches (test-row products, the mean-operator weight product, the undecided voxels' panels) are
summarised beside it.  `cpu_baseline` times the CPU oracle (the reference algorithm restated, SVD route) on a bounded
sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, F0, DELAYS, A, N_OUTER, N_INNER = 3000, 768, [1, 2, 3, 4], 20, 5, 5
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0         # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
FIT_KW = dict(folding_type="kfold", n_outer_folds=N_OUTER, n_inner_folds=N_INNER, chunk_length=20,
              single_alpha=False, normalpha=True, use_corr=True, normalize_features=False,
              normalize_targets=False)


def synth_inputs(V, rank, dev, T=T, F0=F0, DELAYS=DELAYS):
    """SURVEY.md 8d generator: X0 ~ N(0,1) (T, 768) -> FIR delays (HIP kernel) -> X (T, 3072);
    Y = X (0.02 N(0,1)) + N(0,1), made on the device in fp32 (data synthesis only).  (T / F0 / DELAYS: the other
    configs' shapes, tools/scaling_model.py.)"""
    from litcoder_core_amd import ops
    rng = np.random.default_rng(0)
    X0 = rng.standard_normal((T, F0))
    Xd = ops.fir_delay(torch.from_numpy(X0).to(dev), DELAYS, False)          # (T, 3072) f64 on device
    p = Xd.shape[1]
    dX = torch.zeros((T, ops.pad_to(p, 32)), dtype=torch.float32, device=dev)
    dX[:, :p] = Xd.to(torch.float32)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    Vp = ops.pad_to(V, 128)
    dY = torch.zeros((T, Vp), dtype=torch.float32, device=dev)
    W = 0.02 * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
    dY[:, :V] = dX[:, :p] @ W + torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    del W
    return dX, dY, p


_LIVE_CHILDREN = []


def _kill_children(*_):
    """The rocprofv3 child passes run in process groups of their own: when this process ends early (harness timeout,
    SIGTERM) they must not stay behind on the GPU and perturb the next timed run."""
    import signal
    for proc in list(_LIVE_CHILDREN):
        if proc.poll() is None:
            try:
                os.killpg(proc.pid, signal.SIGKILL)      # the group this process started, nothing else
            except OSError:
                pass


def _die_with_parent():
    """preexec of a child pass: its own session (so that the whole group can be killed) and SIGKILL when the parent dies
    without running its handlers (PR_SET_PDEATHSIG = 1)."""
    import ctypes
    import signal
    os.setsid()
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)
    except OSError:
        pass


def measured_traffic(voxels, per_pass_timeout=240):
    """L2-side traffic per fused launch of k_sweep_f16x3, measured for this run: two child processes --
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes, counters only, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes) over ONE HOST-TO-HOST fit of the bench workload
    (tools/host_fit_loop.py: the same 60 launches of 12 288-80 000 columns a timed step issues) -- started BEFORE this
    process touches the GPU.  Corrections of the guide: the counters are in KiB; FETCH_SIZE reports half of a
    16 B/lane streaming read on gfx950 -> doubled.
    Returns ({"all": bytes per launch averaged over ALL fused launches of the step -- the population `avg_launch_ms` of
    the headline roofline averages over --, "full_width": the same over the full-width launches only, "launches": ...}
    or None, source text)."""
    import atexit
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "not measured: this process itself runs under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "not measured: rocprofv3 not found"
    child = os.path.join(ROOT, "tools", "host_fit_loop.py")
    work = tempfile.mkdtemp(prefix="lc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    atexit.register(_kill_children)
    old_term = signal.signal(signal.SIGTERM, lambda *a: (_kill_children(), sys.exit(143)))
    means = {}
    clock = {}
    try:
        # (GRBM_GUI_ACTIVE and the SQ counter ride along in the FETCH_SIZE pass: GRBM / SQ slots are independent of the
        # TCC's, MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters only, no other trace domain)
        for counter, extra in (("FETCH_SIZE", ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"]), ("WRITE_SIZE", [])):
            out = os.path.join(work, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, *extra, "--output-format", "csv", "-d", out, "--",
                   sys.executable, child, "1", str(voxels)]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    preexec_fn=_die_with_parent)
            _LIVE_CHILDREN.append(proc)
            try:
                rc = proc.wait(timeout=per_pass_timeout)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)          # the process group this call started, nothing else
                proc.wait()
                return None, f"not measured: the {counter} pass exceeded {per_pass_timeout}s"
            finally:
                if proc.poll() is not None:
                    _LIVE_CHILDREN.remove(proc)
            if rc != 0:
                return None, f"not measured: the {counter} pass exited with code {rc}"
            rows = []
            per_dispatch = {}
            paths = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            # the fused score launches: with the screening pass on (round 6) those are the HI2 instantiation's (the few
            # three-MFMA launches of the undecided voxels' panel are another kernel: not mixed into the averages)
            # (round 6) ... and the series-moments launches of the same kernel, the inner CV's other contraction: the roofline's
            # population is both kinds (sweep_roofline); the most frequent instantiation of each kind
            names = {"score": {}, "series": {}}
            for path in paths:
                with open(path) as f:
                    for row in csv.DictReader(f):
                        nm = row.get("Kernel_Name", "")
                        kind = ("score" if "k_sweep_f16x3<true" in nm else
                                "series" if "k_sweep_f16x3<false, false, true, true" in nm else None)
                        if kind:
                            names[kind][nm] = names[kind].get(nm, 0) + 1
            wanted = {max(d_, key=d_.get): kind for kind, d_ in names.items() if d_}
            for path in paths:
                with open(path) as f:
                    for row in csv.DictReader(f):
                        kind = wanted.get(row.get("Kernel_Name", ""))
                        if kind is None:
                            continue
                        if row.get("Counter_Name") == counter:
                            rows.append((int(row["Grid_Size"]), float(row["Counter_Value"]), kind))
                        elif row.get("Counter_Name") in extra:
                            d = per_dispatch.setdefault(row["Dispatch_Id"], {"grid": int(row["Grid_Size"]), "kind": kind})
                            d[row["Counter_Name"]] = float(row["Counter_Value"])
                            d["ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            if extra:
                # the clock the chip holds under the dominant kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs
                # (MI355X_MICROARCH.md "DVFS give-back"); the matrix pipe's busy share of those cycles over 1024 SIMDs
                full_grid = {k: max((d["grid"] for d in per_dispatch.values() if d["kind"] == k), default=0) for k in names}
                ds = [d for d in per_dispatch.values() if d.get("ns", 0) > 0 and "GRBM_GUI_ACTIVE" in d]
                for tag, sel in (("all", ds), ("full_width", [d for d in ds if d["grid"] == full_grid[d["kind"]]]),
                                 ("score_launches", [d for d in ds if d["kind"] == "score"]),
                                 ("series_launches", [d for d in ds if d["kind"] == "series"])):
                    if sel:
                        cyc = sum(d["GRBM_GUI_ACTIVE"] for d in sel) / 8.0
                        ns = sum(d["ns"] for d in sel)
                        busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in sel)
                        clock[tag] = {"ghz": cyc / ns, "launches": len(sel), "avg_launch_ms_in_pmc_pass": 1e-6 * ns / len(sel),
                                      "mfma_pipe_busy": busy / (1024.0 * cyc) if cyc > 0 and busy > 0 else None}
            if not rows:
                return None, f"not measured: no k_sweep_f16x3 launch in the {counter} pass"
            full = {k: max((g for g, _, kk in rows if kk == k), default=0) for k in names}
            wide = [v for g, v, k in rows if g == full[k]]
            means[counter] = (sum(v for _, v, _ in rows) / len(rows), len(rows), sum(wide) / len(wide), len(wide))
    finally:
        signal.signal(signal.SIGTERM, old_term)
        shutil.rmtree(work, ignore_errors=True)
    fa, na, ff, nf = means["FETCH_SIZE"]
    wa, nwa, wf, nwf = means["WRITE_SIZE"]
    res = {"all": fa * 1024 * 2 + wa * 1024, "full_width": ff * 1024 * 2 + wf * 1024, "launches": na, "launches_full_width": nf,
           "clock": clock or None}
    return res, (f"measured in this run, before the timed fits: child passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` "
                 f"and `--pmc WRITE_SIZE` over one host-to-host fit (tools/host_fit_loop.py), mean over its {na} score + "
                 f"series-moments launches of all widths (the population avg_launch_ms averages over): FETCH_SIZE {fa:.0f} KiB x2 (gfx950 "
                 f"wide-read correction) + WRITE_SIZE {wa:.0f} KiB; the {nf} full-width launches alone: {ff:.0f} KiB x2 + "
                 f"{wf:.0f} KiB (resident_path.roofline_full_width_launches.traffic); counts Infinity-Cache hits (traffic "
                 f"leaving L2, an upper bound on HBM bytes)")


def cpu_baseline(dX, dY, p, V_full, alphas, v_sample=2000):
    """The oracle (reference algorithm, fp32 torch-CPU SVD route + scipy pearsonr loop) on ONE of
    the five outer folds and `v_sample` voxels; SVD time is V-independent, the rest is
    proportional to V, so  T_cpu(V) = 5 * (t_svd + t_rest * V / v_sample)."""
    import oracle.folds as ofolds
    import oracle.nested_cv as onc
    import oracle.ridge as oridge
    import oracle.stats as ostats
    X = dX[:, :p].cpu()
    Y = dY[:, :v_sample].cpu()
    svd_time = [0.0]
    raw_svd = oridge.thin_svd

    def timed_svd(M, cutoff):
        t = time.perf_counter()
        out = raw_svd(M, cutoff)
        svd_time[0] += time.perf_counter() - t
        return out

    oridge.thin_svd = timed_svd
    try:
        t0 = time.perf_counter()
        tr, te = ofolds.create_folds(T, "kfold", N_OUTER)[0]
        inner = ofolds.create_folds(len(tr), "kfold", N_INNER)
        chosen, _ = onc.select_alphas(X[tr], Y[tr], inner, alphas, False, True, True, 1e-10)
        W = oridge.ridge_weights(X[tr], Y[tr], chosen, normalpha=True, singcutoff=1e-10)
        ostats.pearson_per_voxel(Y[te].numpy(), (X[te] @ W).numpy())
        total = time.perf_counter() - t0
    finally:
        oridge.thin_svd = raw_svd
    t_svd, t_rest = svd_time[0], total - svd_time[0]
    t_full = N_OUTER * (t_svd + t_rest * V_full / v_sample)
    return {
        "value": V_full / t_full, "unit": "voxels/sec", "cores": int(torch.get_num_threads()), "kind": "port",
        "sample": (f"oracle (reference algorithm restated: fp32 torch-CPU SVD route, scipy pearsonr loop) on 1 of "
                   f"{N_OUTER} outer folds x {v_sample} of {V_full} voxels, full T={T} p={p} A={len(alphas)}: "
                   f"{total:.1f}s measured ({t_svd:.1f}s in 6 V-independent SVDs, {t_rest:.1f}s proportional to V); "
                   f"extrapolated to the whole job as {N_OUTER}*(t_svd + t_rest*V/{v_sample}) = {t_full:.0f}s"),
    }


# ------------------------------------------------------------------ cfg3: the LeBel-style story pipeline
CFG3_TRIM = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0, "train_targets_end": None,
             "test_features_start": 50, "test_features_end": -5, "test_targets_start": 40, "test_targets_end": None}
CFG3_KW = dict(folding_type="kfold", n_inner_folds=5, chunk_length=20, single_alpha=True, normalpha=True, use_corr=True)


def synth_stories(V, dev, seed=0, n_train=26, D=768, rank=0):
    """BASELINE configs[2] as synthetic data: 26 training stories + 1 test story of 260-440 TRs (T ~ 9000), per story
    word-level 768-d float32 features (AR(1)-smoothed, ~3.6 words/s) at irregular word times, TR times (15 more feature
    TRs than brain TRs: LeBel trimming [10:-5]) and float64 brain data of V voxels in pageable host memory =
    (z-scored delayed features) W + noise, un-normalised like BOLD data (3 y + 100).  Made on the device (data synthesis
    only; the Lanczos / FIR kernels used here are the product's own).  ``rank``: the stories (lengths, word times, word
    features) are the same on every rank of a sharded run, the V voxels (true weights, noise) are the rank's own."""
    from litcoder_core_amd import Downsampler, ops
    rng = np.random.default_rng(seed)
    n_trs = [int(n) for n in rng.integers(260, 440, n_train)] + [291]
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1 + 7919 * int(rank))
    Wtrue = 0.004 * torch.randn((4 * D, V), generator=g, device=dev, dtype=torch.float32)
    words, wtimes, trtimes, brain = {}, {}, {}, {}
    for i, n_tr in enumerate(n_trs):
        name = "story%02d" % i
        n_words = int(7.2 * n_tr)
        wt = np.sort(rng.uniform(0, 2.0 * (n_tr + 15), n_words))
        emb = rng.standard_normal((n_words, D)).astype(np.float32)
        emb[1:] = 0.6 * emb[:-1] + 0.8 * emb[1:]
        tr_t = 1.0 + 2.0 * np.arange(n_tr + 15)
        words[name], wtimes[name], trtimes[name] = emb, wt, tr_t
        ds = Downsampler().downsample(emb, wt, tr_t, method="lanczos", window=3, cutoff_mult=1.0)
        Xd = ops.fir_delay(torch.from_numpy(ds).to(dev), [1, 2, 3, 4], False)[10:-5]
        Xd = ((Xd - Xd.mean(0)) / Xd.std(0, unbiased=False)).to(torch.float32)
        y = Xd @ Wtrue + torch.randn((n_tr, V), generator=g, device=dev, dtype=torch.float32)
        brain[name] = (3.0 * y + 100.0).cpu().numpy().astype(np.float64)
    return words, wtimes, trtimes, brain


def cfg3_cpu_baseline(words, wtimes, trtimes, brain, v_sample=1500):
    """The oracle's own pipeline (oracle.lanczos / fir / harness / nested_cv: the reference's algorithm restated) on
    `v_sample` voxels: the V-independent part (Lanczos, FIR, feature zs, the six SVDs) is timed apart from the part
    proportional to V (brain zs, projections, sweeps, refit, pearsonr loop)."""
    import oracle.fir as ofir
    import oracle.harness as oh
    import oracle.lanczos as olz
    import oracle.nested_cv as onc
    import oracle.ridge as oridge
    names = list(words)
    V_full = brain[names[0]].shape[1]
    t0 = time.perf_counter()
    feats = {s: ofir.make_delayed(olz.lanczos_interp(words[s], wtimes[s], trtimes[s], window=3, cutoff_mult=1.0), [1, 2, 3, 4])
             for s in names}
    t_pre = time.perf_counter() - t0
    small = {s: brain[s][:, :v_sample] for s in names}
    t0 = time.perf_counter()
    mats = oh.train_test_matrices(feats, small, CFG3_TRIM)
    t_struct = time.perf_counter() - t0
    # the six SVDs (five inner training sets + the outer one: ~11 s each here) are V-independent and differ in shape only
    # by a row: each distinct shape is TIMED once and that time counted for every SVD of the shape -- the decomposition
    # handed back for a repeated shape is the cached one, which makes this sample's numbers meaningless and leaves its
    # timing (the only thing read) what it would be
    svd_time, svd_run, cache = [0.0], [0.0], {}
    raw_svd = oridge.thin_svd

    def timed_svd(M, cutoff):
        key = tuple(M.shape)
        if key not in cache:
            t = time.perf_counter()
            out = raw_svd(M, cutoff)
            cache[key] = (out, time.perf_counter() - t)
            svd_run[0] += cache[key][1]
        svd_time[0] += cache[key][1]
        return cache[key][0]

    oridge.thin_svd = timed_svd
    try:
        t0 = time.perf_counter()
        onc.fit_predict(mats["Rstim"], mats["Rresp"], X_test=mats["Pstim"], y_test=mats["Presp"], **CFG3_KW)
        t_fit = time.perf_counter() - t0 - svd_run[0] + svd_time[0]      # as if every SVD had been computed
    finally:
        oridge.thin_svd = raw_svd
    n_shapes = len(cache)
    t_fixed = t_pre + svd_time[0]
    t_prop = t_struct + t_fit - svd_time[0]
    t_full = t_fixed + t_prop * V_full / v_sample
    return {"value": V_full / t_full, "unit": "voxels/sec", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": (f"oracle pipeline (Lanczos + FIR + per-story zs + train/test nested-CV fit, reference algorithm restated) "
                       f"on {v_sample} of {V_full} voxels: {t_pre + t_struct + t_fit:.1f}s ({t_fixed:.1f}s V-independent: preprocessing "
                       f"{t_pre:.1f}s + six SVDs {svd_time[0]:.1f}s -- {n_shapes} distinct shapes timed once each, "
                       f"{svd_run[0]:.1f}s run; {t_prop:.1f}s proportional to V); extrapolated "
                       f"as t_fixed + t_prop*V/{v_sample} = {t_full:.0f}s")}


def cfg3_leg(V_total, dev, steps=5, warmup=2, cpu=True, shard=None, world=1, rank=0):
    """BASELINE configs[2] end to end, host to host: per-story word features + float64 brain data in pageable host memory ->
    Lanczos -> 4 FIR delays -> trim + per-story zs -> train/test nested-CV fit (example.py:104-117: K-folds, default
    10-alpha grid, single_alpha) -> metrics + float32 host weights, through harness.StoryPipeline.fit_words.
    ``world`` > 1: the V_total voxels split over the ranks (every rank holds its own block of the brain data,
    ``local_targets``; the per-alpha sums all-reduced, nested_cv.py:396-400), barrier + MAX over ranks like the headline."""
    from litcoder_core_amd import NestedCVModel, StoryPipeline, ops
    from litcoder_core_amd.dist import shard_bounds
    lo, hi = shard_bounds(V_total, world, rank)
    V = hi - lo
    words, wtimes, trtimes, brain = synth_stories(V, dev, rank=rank)
    names = list(words)
    model = NestedCVModel("ridge_regression", shard=shard, local_targets=world > 1)
    pipe = StoryPipeline([1, 2, 3, 4], CFG3_TRIM, model=model)

    def step():
        out = pipe.fit_words(words, wtimes, trtimes, brain, **CFG3_KW)
        torch.cuda.synchronize()
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out = None
        ts = time.perf_counter()
        out = step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    # per-kernel ms from ONE more fit with the library's event timers on (two event records per launch: they stay out of
    # the timed fits above)
    ops.timing_enable(True)
    ops.timing_read()
    step()
    kern = ops.timing_read()
    ops.timing_enable(False)
    dX, T, Tt, p = pipe.last_design
    # the link: float32 of every trimmed brain row up, word features up, float32 weights down -- against the page-locked
    # rate of THIS box, measured here with a 1 GiB copy each way
    n = 1 << 30
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    rates = []
    for src, dst in ((h, d), (d, h)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        rates.append(3 * n / (time.perf_counter() - t))
    del h, d
    up = (T + Tt) * V * 4 + sum(w.nbytes for w in words.values())
    down = p * V * 4
    floor_ms = 1e3 * max(up / rates[0], down / rates[1])
    ms = 1e3 * elapsed / steps
    fit = dict(model.last_fit)
    leg = {
        "value": V_total * steps / elapsed, "unit": "voxels/sec", "ms_per_step": ms, "steps": steps, "warmup": warmup,
        "ms_per_step_distribution": step_stats(step_ms, V_total),
        "n_gpus": world, "scaling": "strong" if world > 1 else None, "voxels_total": V_total, "voxels_rank0": V,
        "workload": (f"cfg3 synthetic LeBel-like: {len(names) - 1} training stories + 1 test story (T={T} / {Tt} TRs after "
                     f"trimming), word-level 768-d float32 features -> Lanczos(window 3) -> 4 FIR delays (p={p}) -> per-story zs "
                     f"-> train/test fit, V={V_total}" + (f" in total over {world} GPUs" if world > 1 else "")
                     + ", 10 alphas logspace(-1,8), 5 inner K-folds, single_alpha, normalpha, corr (example.py:104-117)"),
        "inputs": "per-story float32 word features + float64 brain data in pageable host memory -> metrics dict + float32 "
                  "host weights (harness.StoryPipeline.fit_words); every step fenced",
        "form": model.last_form, "arithmetic": fit.get("precision"), "chosen_alpha": float(out[2][0]),
        "median_score": out[0]["median_score"], "panels": [list(c) for c in (fit.get("panels") or [])],
        "link": {"host_brain_bytes_float64": int(sum(b.nbytes for b in brain.values())), "h2d_bytes_float32": int(up),
                 "d2h_bytes": int(down), "measured_h2d_GBps": rates[0] / 1e9, "measured_d2h_GBps": rates[1] / 1e9,
                 "link_floor_ms": floor_ms, "ms_per_step_over_link_floor": ms / floor_ms,
                 "note": "link_floor = max(up bytes / H2D rate, down bytes / D2H rate) of RANK 0's voxel block: what the "
                         "transfers alone take on this box; the fit's V-wide MFMA work at this shape is several times that "
                         "(DESIGN.md 5b)"},
        "sweep_flops_per_step": {"fused": fit.get("fused_flops"), "plain": fit.get("plain_flops")},
        "kernel_ms_per_step": {k: round(v[0], 3) for k, v in sorted(kern.items())},
    }
    if cpu:
        leg["cpu_baseline"] = cfg3_cpu_baseline(words, wtimes, trtimes, brain)
    return leg


def timed_fits(model, dX, dY, p, V, V_total, alphas, steps, warmup, world, dev, collect_kernels=False, host=None):
    """`warmup` untimed + `steps` timed fits, barrier + synchronize on both sides, MAX over ranks.  ``host`` = (X, Y)
    float64 numpy arrays: the host-to-host call (the headline), every step followed by a device synchronisation;
    None: device-resident inputs, weights left resident.  Returns (seconds, last (metrics, W, alphas), kernel timing
    dict or None, {plain, fused} flops counted during the timed steps)."""
    from litcoder_core_amd import ops

    def step():
        if host is not None:
            out = model.fit_predict(host[0], host[1], alphas=alphas, **FIT_KW)
            torch.cuda.synchronize()
            return out
        return model.fit_predict_device(dX, dY, p, V, n_voxels_total=None if world == 1 else V_total,
                                        alphas=alphas, **FIT_KW)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = step()
    if collect_kernels:
        # "sweep": the dominant kernel's launches only (the roofline's avg_launch_ms, measured inside the timed region) --
        # every timed launch is bracketed by two event records that keep it from overlapping its neighbours, and timing all
        # ~900 launches of a fit cost the headline 4 % (145 ms where an untimed loop of the same fits took 139.5)
        ops.timing_enable(True, only=["alpha_sweep_gemm", "series_sweep_gemm"] if collect_kernels == "sweep" else None)
        ops.timing_read()
    flops = {"plain": 0.0, "fused": 0.0, "fused_launches": 0, "series": 0.0, "series_launches": 0, "step_ms": []}
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = None                                           # the previous step's weights are released first
        ts = time.perf_counter()
        out = step()
        flops["step_ms"].append(1e3 * (time.perf_counter() - ts))     # this rank's own (host) view of each step
        flops["plain"] += model.last_fit.get("plain_flops", 0.0)
        flops["fused"] += model.last_fit.get("fused_flops", 0.0)
        flops["fused_launches"] += model.last_fit.get("fused_launches", 0)
        flops["series"] += model.last_fit.get("series_flops", 0.0)
        flops["series_launches"] += model.last_fit.get("series_launches", 0)
    fence()
    elapsed = time.perf_counter() - t0
    kern = None
    if collect_kernels:
        kern = ops.timing_read()
        ops.timing_enable(False)
    flops["rank_seconds"] = [elapsed]
    if world > 1:
        cdev = dev if torch.distributed.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)                # every rank's own clock around the same K steps
        flops["rank_seconds"] = [float(x.item()) for x in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out, kern, flops


def host_arrays(dX, dY, p, V):
    """The bench inputs as the reference's caller holds them: float64 numpy arrays in pageable host memory."""
    X = dX[:, :p].cpu().numpy().astype(np.float64)
    Y = np.empty((dY.shape[0], V), dtype=np.float64)
    step = 8192
    for c in range(0, V, step):                              # column blocks: no second 2 GB temporary
        Y[:, c:c + step] = dY[:, c:min(V, c + step)].cpu().numpy()
    return X, Y


def parity_of(res16, res32, dev):
    """How far the headline's results (f16x3 arithmetic) are from the same fit on exact-fp32 MFMA arithmetic, over all
    voxels: per-voxel correlations, chosen alphas (mean over folds), weights."""
    (m16, W16, a16), (m32, W32, a32) = res16, res32
    c16, c32 = np.asarray(m16["correlations"], dtype=np.float64), np.asarray(m32["correlations"], dtype=np.float64)
    dc = np.abs(c16 - c32)
    w16 = torch.as_tensor(W16).to(dev) if not isinstance(W16, torch.Tensor) else W16
    w32 = torch.as_tensor(W32).to(dev) if not isinstance(W32, torch.Tensor) else W32
    same = np.asarray(a16) == np.asarray(a32)
    cols = torch.as_tensor(np.nonzero(same)[0], device=dev)
    dW = float((w16[:, cols] - w32[:, cols]).abs().max() / w32.abs().max()) if len(cols) else None
    return {"max_abs_dcorr": float(dc.max()), "median_abs_dcorr": float(np.median(dc)),
            "max_abs_dcorr_where_alphas_agree": float(dc[same].max()) if same.any() else None,
            "alpha_agreement": float(np.mean(same)), "max_rel_dW_where_alphas_agree": dW,
            "median_score": [m16["median_score"], m32["median_score"]], "voxels": int(len(c16)),
            "note": "alpha_agreement = voxels whose MEAN alpha over the 5 outer folds is identical; a voxel whose two best "
                    "alphas score within fp32 rounding may flip in either arithmetic (tests/_oracle_check.py proves such "
                    "ties against the oracle)"}



def preproc_leg(dev, cpu=True, reps=20):
    """The two preprocessing steps that feed the fit, through the PUBLIC API, host arrays in -> host arrays out, with the
    kernels' own HIP-event time (the library's timers, on the launch stream) priced against the HBM roof on the ALGORITHMIC
    bytes of SURVEY 8d, and the oracle's CPU time for the same call on this box:
      * FIR.make_delayed, 3000 x 768 float64, delays 1..4  (FIR_expander.py:24-43);  bytes = nt ndim (8 + nd 8)
      * Downsampler.downsample(method="lanczos"), one story: 2500 words x 768 float64 -> 350 TRs  (interpdata.py:87-126);
        bytes = n_old D 8 + n_new D 8
      * the same resampling for the 27 stories of cfg3 in ONE launch (what harness.StoryPipeline issues)."""
    from litcoder_core_amd import FIR, Downsampler, ops
    import oracle.fir as ofir
    import oracle.lanczos as olz
    rng = np.random.default_rng(4)
    PEAK = 8000.0                                              # GB/s, MI355X_MICROARCH.md "HBM3E ~8 TB/s"
    out = {"hbm_peak_gbs": PEAK}

    def run(tag, fn, slot, nbytes, cpu_fn, what):
        fn()                                                   # warm-up (module load, page-locking)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        host_ms = 1e3 * (time.perf_counter() - t0) / 3
        ops.timing_enable(True, only=[slot])
        ops.timing_read()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms, n = ops.timing_read().get(slot, (0.0, 0))
        ops.timing_enable(False)
        per = ms / max(n, 1) * max(n // reps, 1)               # kernel time of ONE call (its launches together)
        ent = {"what": what, "api_host_to_host_ms": host_ms, "kernel_ms": per, "kernel_launches_per_call": n // reps,
               "algorithmic_bytes": nbytes, "achieved_gbs": (nbytes / (per * 1e-3) / 1e9) if per > 0 else None}
        ent["frac_of_hbm_peak"] = ent["achieved_gbs"] / PEAK if ent["achieved_gbs"] else None
        if cpu and cpu_fn is not None:
            t0 = time.perf_counter()
            cpu_fn()
            ent["cpu_oracle_ms"] = 1e3 * (time.perf_counter() - t0)
            ent["cpu_cores"] = int(torch.get_num_threads())
        out[tag] = ent

    X0 = rng.standard_normal((T, F0))
    run("fir_make_delayed", lambda: FIR.make_delayed(X0, DELAYS), "fir_delay", T * F0 * (8 + len(DELAYS) * 8),
        lambda: ofir.make_delayed(X0, DELAYS), f"FIR.make_delayed({T}x{F0} float64, delays {DELAYS}) -> ({T}, {F0 * len(DELAYS)}) float64")
    n_old, n_new = 2500, 350
    wt = np.sort(rng.uniform(0, 700, n_old))
    tr = 1.0 + 2.0 * np.arange(n_new)
    emb = rng.standard_normal((n_old, F0))
    ds = Downsampler()
    run("lanczos_one_story", lambda: ds.downsample(emb, wt, tr, method="lanczos", window=3, cutoff_mult=1.0), "lanczos_interp",
        (n_old + n_new) * F0 * 8, lambda: olz.lanczos_interp(emb, wt, tr, 3, 1.0),
        f"Downsampler.downsample(lanczos, window 3): {n_old} words x {F0} float64 -> {n_new} TRs")
    # cfg3's 27 stories in one launch: resident float32 word features (as harness.StoryPipeline holds them)
    words, wtimes, trtimes, _ = synth_stories(256, dev)
    names = list(words)
    cat = torch.from_numpy(np.concatenate([words[s] for s in names])).to(dev)
    nb = sum(words[s].shape[0] * words[s].shape[1] * 4 + len(trtimes[s]) * words[s].shape[1] * 8 for s in names)
    run("lanczos_27_stories_one_launch",
        lambda: ops.lanczos_interp_stories(cat, [wtimes[s] for s in names], [trtimes[s] for s in names], 3, 1.0, False),
        "lanczos_interp", nb, None,
        f"lc_lanczos_interp_stories: {cat.shape[0]} words x {cat.shape[1]} float32 (resident) -> {sum(len(trtimes[s]) for s in names)} TRs float64, "
        "27 stories, one launch (device in -> device out: no host leg)")
    out["lanczos_27_stories_one_launch"]["api_host_to_host_ms"] = None
    return out


def step_stats(step_ms, voxels):
    """min / median / max of the timed steps as rank 0's host saw them (every host-to-host step is fenced, so a step's time is
    one call's latency), and the rate at the median step: a slow outlier step moves `value` (total / total), not this."""
    if not step_ms:
        return None
    a = np.sort(np.asarray(step_ms, dtype=np.float64))
    med = float(np.median(a))
    return {"min": float(a[0]), "median": med, "max": float(a[-1]), "steps": int(len(a)),
            "value_at_median_step": voxels / (1e-3 * med), "unit": "ms; voxels/sec"}


def sweep_roofline(sweep, kern, flops, steps, split):
    """roofline object of the fused alpha-sweep kernel from the library's HIP events around its launches (on the launch
    stream) and the algorithmic flops the engine counted for them."""
    ms_f, launches_f = kern.get("alpha_sweep_gemm", (0.0, 0))
    ms_s, launches_s = kern.get("series_sweep_gemm", (0.0, 0))
    screened = bool(split and sweep.get("screen_terms", 3) == 1)
    mfma_per_product = (1 if screened else 3) if split else 1
    peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    score_tflops = flops["fused"] / (ms_f * 1e-3) / 1e12 if launches_f and ms_f > 0 else None
    series_tflops = flops.get("series", 0.0) / (ms_s * 1e-3) / 1e12 if launches_s and ms_s > 0 and flops.get("series") else None
    # (round 6) the dominant kernel is k_sweep_f16x3 in BOTH its inner-CV instantiations: the fused score launches (the
    # factorised alphas' hat matrices of an inner fold) and the series-moments launches (the four shared series terms of the
    # same fold, the larger share of a fit since the fused launch carries three alphas).  Priced together on the flops each
    # actually contracts; `score_launches` keeps the quantity rounds 1-5 reported alone
    both = series_tflops is not None and split
    ms, launches = (ms_f + ms_s, launches_f + launches_s) if both else (ms_f, launches_f)
    total = flops["fused"] + (flops.get("series", 0.0) if both else 0.0)
    alg_tflops = total / (ms * 1e-3) / 1e12 if launches and ms > 0 else None
    issued = alg_tflops * mfma_per_product if alg_tflops else None
    per_kind = {"score_launches": {"what": "fused alpha sweep: 2 x A_fused x n_val x n_train x V flop per inner fold",
                                   "achieved": score_tflops, "frac": (score_tflops / peak) if score_tflops else None,
                                   "avg_launch_ms": ms_f / max(launches_f, 1), "launches_per_step": launches_f / max(steps, 1),
                                   "flops_per_step": flops["fused"] / max(steps, 1)},
                "series_launches": ({"what": "series-moments sweep: 2 x 4 terms x n_val x n_train x V flop per inner fold",
                                     "achieved": series_tflops, "frac": series_tflops / peak,
                                     "avg_launch_ms": ms_s / max(launches_s, 1), "launches_per_step": launches_s / max(steps, 1),
                                     "flops_per_step": flops["series"] / max(steps, 1)} if both else None)}
    return {"bound": "mfma",
            "kernel": (("k_sweep_f16x3<.., HI2>: the two contractions of the inner CV's SCREENING pass (fused score launches: the "
                        "factorised alphas' hat matrices; series-moments launches: the shared series terms), 1 fp16 MFMA per "
                        "product; the undecided voxels' three-MFMA launches are inside the same timers, their flops are not counted"
                        if both else
                        "k_sweep_f16x3<score, HI2> (fused alpha sweep of the inner CV's SCREENING pass: 1 fp16 MFMA per product; "
                        "the undecided voxels' three-MFMA launches are inside the same timer, their flops are not counted)")
                       if screened else ("k_sweep_f16x3 (fused score + series-moments sweeps, 3 fp16 MFMAs per product)" if both
                                         else "k_sweep_f16x3 (fused alpha sweep, 3 fp16 MFMAs per product)")) if split
                      else "k_gemm_f32<score> (fused alpha sweep, f32-input MFMA)",
            **per_kind,
            "screening": ({"undecided_voxel_folds": sweep.get("undecided"), "screened_voxel_folds": sweep.get("screened"),
                           "undecided_frac": (sweep.get("undecided", 0) / max(sweep.get("screened", 0), 1)),
                           "overflows": sweep.get("screen_overflows", 0)} if screened else None),
            "achieved": alg_tflops, "peak": peak, "unit": "TFLOP/s", "frac": (alg_tflops / peak) if alg_tflops else None,
            "mfma_per_product": mfma_per_product, "mfma_issue_tflops": issued,
            "mfma_issue_frac": (issued / peak) if issued else None,
            "frac_vs_f32_mfma_peak": (alg_tflops / PEAK_F32_MFMA_TFLOPS) if alg_tflops else None,
            "flops_per_step": total / max(steps, 1), "launches_per_step": launches / max(steps, 1),
            "avg_launch_ms": ms / max(launches, 1), "launches": launches,
            "fused_alphas_per_launch": sweep.get("fused_alphas"), "inner_folds_per_launch": sweep.get("folds_per_launch", 1)}


def launch_ranks(n, argv):
    """`python3 bench.py --gpus N` by itself (no launcher around it): this process -- which never touches the GPU -- starts N
    fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, one GPU each),
    relays rank 0's JSON line and exits with the worst rank's return code.  No exec of anything: the ranks are children."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:                               # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True,
                                      start_new_session=True))

    def stop(*_):
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGKILL)          # exactly the groups started here
                except OSError:
                    pass
    signal.signal(signal.SIGTERM, lambda *a: (stop(), sys.exit(143)))
    try:
        out, _ = procs[0].communicate()
        rcs = [procs[0].returncode]
        for q in procs[1:]:
            try:
                rcs.append(q.wait(timeout=120))               # (a rank that outlives rank 0 by two minutes hangs)
            except subprocess.TimeoutExpired:
                rcs.append(124)
    finally:
        stop()
    sys.stdout.write(out or "")
    sys.stdout.flush()
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        print(f"bench.py: rank return codes {rcs}", file=sys.stderr)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--voxels", type=int, default=80000,
                    help="voxels per GPU (weak scaling) / voxels in total (strong scaling)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: 'strong' (default) = --voxels IN TOTAL split over the ranks -- the job BASELINE.json's configs "
                         "name (80 000 voxels); 'weak' = --voxels per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip host_path / f32_path / other_scaling")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the cfg3_pipeline leg (N = 1 only)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "f16x3"],
                    help="arithmetic of the alpha sweep (auto = f16x3 unless the targets' dynamic range forbids it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the bare `python3 bench.py --gpus N` form: this process becomes the launcher of its N ranks (it has not touched,
        # and never touches, the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.scaling is None:
        # BASELINE.json's configs fix the job (80 000 voxels in total): at N > 1 the headline is that job split over the
        # ranks; the weak-scaled job (80 000 voxels PER rank, a volume no config names) goes to `other_scaling`
        args.scaling = "weak" if world == 1 else "strong" 
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")
    # LITCODER_BENCH_ONE_GPU=1 (smoke test of the N > 1 code path on a single-GPU box): every rank on device 0, gloo
    # instead of RCCL (which refuses two ranks on one device); the numbers of such a run mean nothing
    one_gpu = os.environ.get("LITCODER_BENCH_ONE_GPU", "0") == "1"
    live_traffic = None
    if world == 1 and not args.no_traffic and not args.no_extra_legs and args.precision != "f32":
        live_traffic = measured_traffic(args.voxels)         # child processes; this one has not touched the GPU yet
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))

    from litcoder_core_amd import NestedCVModel, ShardContext, ops
    from litcoder_core_amd.dist import shard_bounds
    dev = ops.device(local)
    alphas = np.logspace(-1, 8, A)
    # every rank keeps the per-voxel lists of its OWN voxels (like its block of the weights); the statistics behind them
    # are global.  With global lists every rank would spend ~70 ms of interpreter time on 640 000-entry Python lists.
    shard = ShardContext(device=dev, global_lists=False) if world > 1 else None
    model = NestedCVModel("ridge_regression", shard=shard, precision=args.precision, local_targets=world > 1)

    def inputs(mode):
        """(dX, dY, p, V_local, V_total) of this rank: weak = `--voxels` each; strong = its block of `--voxels`."""
        if mode == "weak" or world == 1:
            dX, dY, p = synth_inputs(args.voxels, rank, dev)
            return dX, dY, p, args.voxels, args.voxels * world
        lo, hi = shard_bounds(args.voxels, world, rank)
        dX, dY, p = synth_inputs(hi - lo, rank, dev)
        return dX, dY, p, hi - lo, args.voxels

    dX, dY, p, V, V_total = inputs(args.scaling)
    host = host_arrays(dX, dY, p, V)                         # what the reference's caller holds: float64, pageable
    elapsed, res, kern, flops = timed_fits(model, dX, dY, p, V, V_total, alphas, args.steps, args.warmup, world, dev,
                                           collect_kernels="sweep", host=host)
    # the per-class breakdown (kernel_ms_per_step, the plain launches' rate) from ONE more fit with every timer on
    _, _, kern_all, flops_all = timed_fits(model, dX, dY, p, V, V_total, alphas, 1, 0, world, dev, collect_kernels=True,
                                           host=host)
    metrics = res[0]
    sweep = dict(model.last_fit)
    other = None
    if world > 1 and not args.no_extra_legs:
        mode2 = "strong" if args.scaling == "weak" else "weak"
        del dX, dY, host
        torch.cuda.empty_cache()
        dX, dY, p, V2, V2_total = inputs(mode2)
        host = host_arrays(dX, dY, p, V2)
        e2, r2, _, _ = timed_fits(model, dX, dY, p, V2, V2_total, alphas, args.steps, args.warmup, world, dev, host=host)
        other = {"scaling": mode2, "value": V2_total * args.steps / e2, "unit": "voxels/sec",
                 "ms_per_step": 1e3 * e2 / args.steps, "voxels_total": V2_total, "median_score": r2[0]["median_score"]}
    cfg3_sharded = None
    if world > 1 and not args.no_extra_legs and not args.no_cfg3:
        # BASELINE configs[2] -- the config north_star's >= 6x-at-8-GPUs sentence is on -- strong-scaled: the story pipeline on
        # every rank's block of the 80 000 voxels (every rank calls: the leg is collective)
        try:
            del dX, dY, host
        except NameError:
            pass
        torch.cuda.empty_cache()
        cfg3_sharded = cfg3_leg(args.voxels, dev, cpu=False, shard=shard, world=world, rank=rank)

    if rank == 0:
        split = sweep["precision"] == "f16x3"
        roof = sweep_roofline(sweep, kern, flops, args.steps, split)
        plain_ms, plain_n = kern_all.get("grouped_gemm", (0.0, 0))
        plain = None
        if split and plain_n:
            plain = {"launches": plain_n, "ms_per_step": plain_ms,
                     "algorithmic_tflops": flops_all["plain"] / (plain_ms * 1e-3) / 1e12,
                     "mfma_tflops": 3 * flops_all["plain"] / (plain_ms * 1e-3) / 1e12,
                     "what": "the other launches of the same kernel per step: the folds' test-row products, the mean-operator "
                             "weight product (round 6) and the undecided voxels' panels; includes the f32 MFMA launches of "
                             "that timer slot (the series-moments launches have a slot of their own since round 6)"}
        traffic = traffic_src = committed = None
        tpath = os.path.join(ROOT, "profiles", "alpha_sweep_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                committed = tj.get(sweep["precision"], tj).get("hbm_bytes_per_launch")
            except Exception:
                committed = None
        traffic_full = None
        if live_traffic is not None and live_traffic[0] is not None and split:
            traffic, traffic_full, traffic_src = live_traffic[0]["all"], live_traffic[0]["full_width"], live_traffic[1]
        elif committed is not None:
            # the committed figure is per FULL-WIDTH launch: it belongs with the resident leg's full-width roofline, whose
            # avg_launch_ms it can be divided by; the headline's mixed-width launches get no figure then
            traffic_full = committed
            traffic_src = ("not measured in this run ("
                           + (live_traffic[1] if live_traffic is not None else "child passes skipped")
                           + "); resident_path.roofline_full_width_launches.traffic carries profiles/alpha_sweep_traffic.json")
        roof.update({
            "note": ("achieved/frac = ALGORITHMIC flops of the caller's contractions (score launches: 2 x A_fused x n_val x "
                     "n_train x V per inner fold; series-moments launches: 2 x 4 terms x n_val x n_train x V; summed over the "
                     "launches of the timed steps) / the launches' HIP-event time -- `score_launches` alone is the quantity of "
                     "rounds 1-5; the kernel "
                     "issues mfma_per_product fp16 MFMAs per product (mfma_issue_tflops / mfma_issue_frac).  peak = dense "
                     "fp16 MFMA datasheet figure at 2.4 GHz; under this kernel the chip holds 1.4-1.8 GHz (in-kernel "
                     "s_memtime/s_memrealtime, profiles/).  The first fold's launches are panel-wide (the targets are still "
                     "arriving), the others full width.  Inside the timed region ONLY this kernel's launches are "
                     "bracketed by HIP events (avg_launch_ms); kernel_ms_per_step comes from one more fit with every "
                     "class timed, outside the timed region") if split else None,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_population": "bytes per launch averaged over the same launches avg_launch_ms averages over",
            "plain_launches_same_kernel": plain})
        clk = (live_traffic[0] or {}).get("clock") if live_traffic is not None else None
        if clk and clk.get("all") and roof.get("frac"):
            # what the box's clock explains of a run-to-run difference: the same kernel on a chip that holds a lower clock
            # under it reads a lower `frac`; against the peak AT THE CLOCK HELD the figure should not move
            ghz = clk["all"]["ghz"]
            roof.update({
                "clock_ghz": ghz, "mfma_pipe_busy": clk["all"]["mfma_pipe_busy"],
                "roofline_at_clock": roof["frac"] * 2.4 / ghz,
                "mfma_issue_frac_at_clock": (roof["mfma_issue_frac"] * 2.4 / ghz) if roof.get("mfma_issue_frac") else None,
                "clock_full_width_launches": clk.get("full_width"),
                "clock_source": "GRBM_GUI_ACTIVE / 8 / dispatch duration of this kernel's launches in the FETCH_SIZE child pass "
                                "of this run (counters serialise the dispatches: the kernel runs alone there; a profiled pass "
                                "holds 2-3 % less clock than an un-profiled one); roofline_at_clock = frac x 2.4 GHz / clock_ghz "
                                "= fraction of the fp16 MFMA peak at the clock the chip holds under this kernel; "
                                "mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x those cycles)"})
        renamed = {"batch_chol_solve": "batch_chol_solve_stream_ms_incl_waits_for_cus"}
        out = {
            "metric": "voxels/sec full nested-CV ridge fit (LeBel UTS03, GPT-2 768x4 delays, ~80k voxels)",
            "value": V_total * args.steps / elapsed, "unit": "voxels/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "ms_per_step_distribution": step_stats(flops["step_ms"], V_total),
            "ms_per_step_by_rank": [1e3 * t / args.steps for t in flops["rank_seconds"]],
            "rccl_ranks": (torch.distributed.get_world_size() if world > 1 else 1),
            "collective_backend": (torch.distributed.get_backend() if world > 1 else None),
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("f16x3+f32acc (fp16 hi+lo operands, fp32 accumulate; Gram/Cholesky in f64)"
                      + ("; inner-CV scores screened on fp16 hi operands, the undecided voxels re-scored on hi+lo: the fit "
                         "equals the all-hi+lo fit (unscreened_path.same_alphas_as_headline_fit)"
                         if sweep.get("screen_terms", 3) == 1 else "")) if split
                     else "f32 (Gram/Cholesky in f64)", "data": "synthetic",
            "config": {"workload": f"cfg2 synthetic T={T} F={F0}x{len(DELAYS)} delays (p={p}) "
                                   + (f"V={args.voxels}/GPU" if args.scaling == "weak"
                                      else f"V={args.voxels} in total" + (f" split over {world} GPUs (BASELINE configs[1] strong-scaled)"
                                                                          if world > 1 else ""))
                                   + f" A={A} alphas {N_OUTER}x{N_INNER} kfold, per-voxel alpha, normalpha, corr",
                       "voxels_total": V_total, "voxels_rank0": V,
                       "inputs": "float64 numpy features/targets in pageable host memory -> metrics dict + float32 host "
                                 "weights + alphas (NestedCVModel.fit_predict, SURVEY 8d); H2D (float32 after a host-side "
                                 "cast in the staging threads) and D2H inside the timed region, every step fenced",
                       "panels": [list(c) for c in (sweep.get("panels") or [])],
                       "parallelism": f"voxel-shard x{world}", "median_score": metrics["median_score"]},
            "roofline": roof,
            "kernel_ms_per_step": {renamed.get(k, k): round(v[0], 3) for k, v in sorted(kern_all.items())},
        }
        if other is not None:
            out["other_scaling"] = other
        if cfg3_sharded is not None:
            out["cfg3_pipeline"] = cfg3_sharded
        if world == 1 and not args.no_extra_legs:
            del host
            e_res, r_res, k_res, f_res = timed_fits(model, dX, dY, p, V, V_total, alphas, 3, 1, 1, dev, collect_kernels="sweep")
            roof_res = sweep_roofline(dict(model.last_fit), k_res, f_res, 3, split)
            out["resident_path"] = {"value": V * 3 / e_res, "unit": "voxels/sec", "ms_per_step": 1e3 * e_res / 3, "steps": 3,
                                    "ms_per_step_distribution": step_stats(f_res["step_ms"], V),
                                    "what": "fp32 inputs resident in HBM, weights left resident (the headline of rounds 1-2)",
                                    "median_score": r_res[0]["median_score"],
                                    "roofline_full_width_launches": dict(
                                        {k: roof_res.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac",
                                                                      "mfma_issue_frac", "avg_launch_ms", "launches_per_step")},
                                        traffic=traffic_full, traffic_in_committed_profile=committed,
                                        traffic_note="bytes leaving L2 per full-width fused launch (comparable with this "
                                                     "object's avg_launch_ms)"),
                                    "note": "the same dominant kernel in launches of all 80 000 voxels (the headline's steps cut "
                                            "the first fold and the last two into voxel panels: launches of 12 288-80 000 columns)"}
            if split and sweep.get("screen_terms", 3) == 1:
                # the same resident fit with every inner-CV product on three fp16 MFMAs (FitOptions.screen_inner off: rounds
                # 2-5's arithmetic throughout) -- what the screening pass buys, in the same run on the same box
                from litcoder_core_amd.engine.common import FitOptions
                m3 = NestedCVModel("ridge_regression", precision=args.precision, options=FitOptions(screen_inner=False))
                e3, r3, k3, f3 = timed_fits(m3, dX, dY, p, V, V_total, alphas, 3, 1, 1, dev, collect_kernels="sweep")
                out["unscreened_path"] = {
                    "value": V * 3 / e3, "unit": "voxels/sec", "ms_per_step": 1e3 * e3 / 3, "steps": 3,
                    "what": "resident fit, inner CV on three fp16 MFMAs per product for every voxel (no screening pass)",
                    "roofline": {k: v for k, v in sweep_roofline(dict(m3.last_fit), k3, f3, 3, True).items()
                                 if k in ("kernel", "achieved", "peak", "unit", "frac", "mfma_per_product", "mfma_issue_frac",
                                          "avg_launch_ms", "launches_per_step")},
                    "same_alphas_as_headline_fit": bool(np.array_equal(np.asarray(r3[2].cpu() if torch.is_tensor(r3[2]) else r3[2]),
                                                                       np.asarray(r_res[2].cpu() if torch.is_tensor(r_res[2]) else r_res[2]))),
                    "max_abs_weight_difference_to_resident_fit": float((r3[1] - r_res[1]).abs().max())
                    if torch.is_tensor(r3[1]) else float(np.abs(np.asarray(r3[1]) - np.asarray(r_res[1])).max())}
            if args.precision != "f32":
                m32 = NestedCVModel("ridge_regression", precision="f32")
                e32, r32, k32, f32 = timed_fits(m32, dX, dY, p, V, V_total, alphas, 2, 1, 1, dev, collect_kernels="sweep")
                out["f32_path"] = {"value": V * 2 / e32, "unit": "voxels/sec", "ms_per_step": 1e3 * e32 / 2, "steps": 2,
                                   "dtype": "f32 (f32-input MFMA sweep and refit; Gram/Cholesky in f64)",
                                   "median_score": r32[0]["median_score"],
                                   "roofline": sweep_roofline(dict(m32.last_fit), k32, f32, 2, False)}
                out["parity_vs_f32_path"] = parity_of(res, r32, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dX, dY, p, V, alphas)
        if world == 1 and not args.no_extra_legs:
            out["preproc"] = preproc_leg(dev, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_extra_legs and not args.no_cfg3:
            del dX, dY
            torch.cuda.empty_cache()
            out["cfg3_pipeline"] = cfg3_leg(args.voxels, dev, cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
