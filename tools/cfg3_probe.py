#!/usr/bin/env python3
"""cfg3 (LeBel UTS03-like) at full size on one GPU: where the time goes.   python tools/cfg3_probe.py [V] [what ...]

26 training stories + 1 test story of 240-440 TRs (T_train ~ 9000), word-level 768-d float32 features at irregular word
times -> Lanczos -> 4 FIR delays -> trim + per-story zs -> train/test fit, single_alpha, 10 alphas, 5 K-folds
(example.py:104-117 with its argparse defaults).  Times each piece of the path, the host-to-host fit_predict on the
structured matrices, the resident fit, and the page-locked H2D rate of the box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import Downsampler, NestedCVModel, StoryPipeline, ops  # noqa: E402
from litcoder_core_amd import harness  # noqa: E402

TRIM = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0, "train_targets_end": None,
        "test_features_start": 50, "test_features_end": -5, "test_targets_start": 40, "test_targets_end": None}
KW = dict(folding_type="kfold", n_inner_folds=5, chunk_length=20, single_alpha=True, normalpha=True, use_corr=True)


def make_stories(V, seed=0, n_train=26, D=768, dev=None):
    """Per story: word features (n_words, D) f32, word times, TR times (n_tr + 15 feature TRs), brain (n_tr, V) f64 host."""
    rng = np.random.default_rng(seed)
    n_trs = list(rng.integers(260, 440, n_train)) + [291]
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
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


def timed(fn, n=3, warm=1):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        out = None
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return min(ts), float(np.median(ts)), out


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    only = sys.argv[2:]
    dev = ops.device(0)
    t0 = time.perf_counter()
    words, wtimes, trtimes, brain = make_stories(V, dev=dev)
    names = list(words)
    T_train = sum(brain[s].shape[0] for s in names[:-1])
    print(f"stories: {len(names)} ({T_train} training TRs, {brain[names[-1]].shape[0]} test TRs), V = {V}; "
          f"brain float64 on the host: {sum(b.nbytes for b in brain.values()) / 1e9:.2f} GB; made in {time.perf_counter() - t0:.1f} s",
          flush=True)

    if os.environ.get("LITCODER_PROBE_MIGRATE"):
        # the scheduler's worst case, forced: the caller's thread moves to the OTHER socket after it filled its arrays
        def cpus(node):
            out = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                lo, _, hi = part.partition("-")
                out |= set(range(int(lo), int(hi or lo) + 1))
            return out
        import ctypes
        here = 0 if ctypes.CDLL(None).sched_getcpu() in cpus(0) else 1
        try:
            os.sched_setaffinity(0, cpus(1 - here) & os.sched_getaffinity(0))
            print(f"caller moved from node {here} to node {1 - here}", flush=True)
        except (OSError, ValueError) as e:
            print(f"could not move the caller: {e}", flush=True)
        only = [k for k in only if k != "migrate"]

    def want(k):
        return not only or k in only

    if want("h2d"):
        n = 1 << 30
        h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        d = torch.empty(n, dtype=torch.uint8, device=dev)
        for _ in range(2):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"page-locked H2D: {n / dt / 1e9:.1f} GB/s; D2H: ", end="")
        t = time.perf_counter()
        for _ in range(5):
            h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        print(f"{n / ((time.perf_counter() - t) / 5) / 1e9:.1f} GB/s", flush=True)
        del h, d

    ds = {}
    if want("pre") or want("pipe") or want("fit"):
        def lanczos_all():
            for s in names:
                ds[s] = Downsampler().downsample(words[s], wtimes[s], trtimes[s], method="lanczos", window=3, cutoff_mult=1.0)
            return ds
        best, med, _ = timed(lanczos_all)
        print(f"Downsampler.downsample x {len(names)} stories (host in, host out): {1e3 * med:.1f} ms", flush=True)
        best, med, delayed = timed(lambda: harness.apply_fir_delays(ds, [1, 2, 3, 4]))
        print(f"apply_fir_delays (host in, device out): {1e3 * med:.1f} ms", flush=True)
        best, med, d = timed(lambda: harness.structure_train_test_device(delayed, brain, TRIM), n=2)
        print(f"structure_train_test_device (float64 uploads + zs): {1e3 * med:.1f} ms", flush=True)
        del d, delayed

    model = NestedCVModel("ridge_regression")
    if want("pipe"):
        pipe = StoryPipeline([1, 2, 3, 4], TRIM, model=model)
        ops.timing_enable(True)
        ops.timing_read()
        best, med, out = timed(lambda: pipe.fit_words(words, wtimes, trtimes, brain, **KW), n=3)
        kern = ops.timing_read()
        ops.timing_enable(False)
        print(f"StoryPipeline.fit_words (word features + brain in, metrics + host weights out): {1e3 * med:.1f} ms "
              f"(best {1e3 * best:.1f}) = {V / med:.0f} voxels/s; form {model.last_form}, alpha {out[2][0]:.4g}, "
              f"median r {out[0]['median_score']:.4f}, panels {model.last_fit.get('panels')}", flush=True)
        for k, (ms, n) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
            print(f"    {k:40s} {ms / 4:9.2f} ms/fit  {n // 4:5d} launches/fit")
        best, med, out2 = timed(lambda: pipe.fit(ds, brain, **KW), n=2)
        print(f"StoryPipeline.fit (downsampled host features + brain in): {1e3 * med:.1f} ms (best {1e3 * best:.1f})", flush=True)
        del out, out2

    if want("fit"):
        import oracle.harness as oh
        t = time.perf_counter()
        delayed_h = {s: harness.apply_fir_delays({s: ds[s]}, [1, 2, 3, 4])[s].cpu().numpy() for s in names}
        mats = oh.train_test_matrices(delayed_h, brain, TRIM)
        print(f"host structuring by the oracle (numpy zs + vstack): {time.perf_counter() - t:.1f} s; Rstim {mats['Rstim'].shape}, "
              f"Rresp {mats['Rresp'].shape}, Pstim {mats['Pstim'].shape}", flush=True)
        ops.timing_enable(True)
        ops.timing_read()
        best, med, out = timed(lambda: model.fit_predict(mats["Rstim"], mats["Rresp"], X_test=mats["Pstim"],
                                                          y_test=mats["Presp"], **KW), n=2)
        kern = ops.timing_read()
        ops.timing_enable(False)
        print(f"fit_predict host to host (train/test, single alpha): {1e3 * med:.1f} ms (best {1e3 * best:.1f}) = {V / med:.0f} "
              f"voxels/s; form {model.last_form}, {model.last_fit.get('precision')}, alpha {out[2][0]:.4g}, "
              f"median r {out[0]['median_score']:.4f}", flush=True)
        for k, (ms, n) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
            print(f"    {k:40s} {ms / 3:9.2f} ms/fit  {n // 3:5d} launches/fit")
        # resident
        T, Tt = mats["Rstim"].shape[0], mats["Pstim"].shape[0]
        p = mats["Rstim"].shape[1]
        dX = ops.upload_f32(np.vstack([mats["Rstim"], mats["Pstim"]]), ops.pad_to(p, 32), dev)
        dY = ops.upload_f32(ops.HostRows([mats["Rresp"], mats["Presp"]]), ops.pad_to(V, 128), dev)
        best, med, out = timed(lambda: model.fit_predict_device(dX, dY, p, V, n_test_rows=Tt, **KW), n=2)
        print(f"fit_predict_device (resident): {1e3 * med:.1f} ms (best {1e3 * best:.1f}) = {V / med:.0f} voxels/s", flush=True)


if __name__ == "__main__":
    main()
