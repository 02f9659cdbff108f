#!/usr/bin/env python3
"""Random inputs through the preprocessing front-ends -- FIR.make_delayed and every Downsampler method that has a kernel
(lanczos, sinc, rect, average / sum / last and their legacy_* chunk versions) -- against the CPU oracle's restatement of
the reference (oracle/fir.py, oracle/lanczos.py): FIR bit for bit, the resamplers to 1e-12 of the largest output (fp64;
device sin and summation order differ from libm / BLAS in the last ulps), the reducers to 1e-13 -- sums and means of float32
data to 4e-6 of the largest output: numpy sums a float32 array in float32, the kernels in float64 (tests/test_gpu_parity.py
allows 1e-6 there).  Shapes down to one row
and one column, float32 / float64 data, negative / repeated / over-long delays, circpad, unsorted sample times, windows
and cut-offs at random.  A bug hunt, not a test.      python tools/fuzz_preproc.py [n_cases [seed]]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import litcoder_core_amd as lc  # noqa: E402
import oracle.fir as ofir  # noqa: E402
import oracle.lanczos as olz  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
fails = 0
done = {}
ds = lc.Downsampler()


def close(got, want, atol, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} vs {want.dtype}"
    scale = max(1.0, float(np.nanmax(np.abs(want))) if want.size and np.isfinite(want).any() else 1.0)
    bad = ~((np.abs(got - want) <= atol * scale) | ((got != got) & (want != want)) | (got == want))
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} entries differ, max |d| {np.nanmax(np.abs(got - want)[bad]):.3g}"


for case in range(n_cases):
    kind = str(rng.choice(["fir", "fir", "lanczos", "lanczos", "sinc", "rect", "label", "chunks"]))
    dt = np.float32 if rng.random() < 0.4 else np.float64
    tag = f"case {case}: {kind} {np.dtype(dt).name}"
    try:
        if kind == "fir":
            nt, nd = int(rng.choice([1, 2, 5, 37, 350, 1200])), int(rng.choice([1, 3, 64, 300, 768]))
            x = rng.standard_normal((nt, nd)).astype(dt)
            delays = [int(d) for d in rng.integers(-6, 12, size=int(rng.integers(1, 9)))]
            if rng.random() < 0.2:
                delays.append(nt + int(rng.integers(0, 3)))              # as long as the stimulus, or longer
            circ = bool(rng.random() < 0.3)
            tag += f" {nt}x{nd} delays{delays} circpad{int(circ)}"
            got = lc.FIR.make_delayed(x, delays, circpad=circ)
            want = ofir.make_delayed(x, delays, circpad=circ)
            assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want, equal_nan=True), "FIR differs"
        else:
            n_old, D = int(rng.choice([1, 2, 9, 120, 900, 2500])), int(rng.choice([1, 3, 16, 100, 768]))
            n_new = int(rng.choice([1, 2, 7, 60, 350]))
            span = float(rng.uniform(5, 700))
            ot = np.sort(rng.uniform(0, span, n_old))
            if rng.random() < 0.2 and n_old > 3:
                ot = ot[rng.permutation(n_old)]                             # unsorted sample times (legal for the reference)
            nt_ = np.sort(rng.uniform(0, span, n_new)) if rng.random() < 0.3 else rng.uniform(0, 2) + (span / max(n_new, 1)) * np.arange(n_new)
            d = (rng.standard_normal((n_old, D)) * 3 + 1).astype(dt)
            tag += f" {n_old}x{D} -> {n_new}"
            with np.errstate(all="ignore"):
                if kind == "lanczos":
                    w, c, r = int(rng.integers(1, 5)), float(rng.choice([0.5, 1.0, 1.7])), bool(rng.random() < 0.3)
                    tag += f" window {w} cutoff_mult {c} rectify {int(r)}"
                    want = olz.lanczos_interp(d, ot, nt_, window=w, cutoff_mult=c, rectify=r)
                    close(ds.downsample(d, ot, nt_, method="lanczos", window=w, cutoff_mult=c, rectify=r), want, 1e-12, kind)
                elif kind == "sinc":
                    w, c = int(rng.integers(1, 4)), float(rng.choice([0.7, 1.0]))
                    causal, renorm = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
                    tag += f" window {w} cutoff_mult {c} causal {int(causal)} renorm {int(renorm)}"
                    want = olz.sinc_interp(d, ot, nt_, c, w, causal, renorm)
                    close(ds.downsample(d, ot, nt_, method="sinc", window=w, cutoff_mult=c, causal=causal, renorm=renorm), want,
                          1e-12, kind)
                elif kind == "rect":
                    close(ds.downsample(d, ot, nt_), olz.rect(d, ot, nt_), 1e-13 if dt == np.float64 else 4e-6, kind)
                elif kind == "label":
                    labels = rng.integers(0, n_new, size=n_old)
                    for how in ("average", "sum", "last"):
                        close(ds.downsample(d, ot, nt_, method=how, split_indices=list(labels)), olz.by_label(d, labels, how),
                              4e-6 if (dt == np.float32 and how != "last") else 1e-13, how)
                else:
                    bounds = np.sort(rng.choice(np.arange(1, max(n_old, 2)), size=min(n_new, max(n_old - 1, 1)), replace=False))
                    for how in ("average", "sum", "last"):
                        close(ds.downsample(d, ot, nt_, method="legacy_" + how, split_indices=bounds), olz.by_chunks(d, bounds, how),
                              4e-6 if (dt == np.float32 and how != "last") else 1e-13, "legacy_" + how)
        done[kind] = done.get(kind, 0) + 1
    except Exception as e:                                   # noqa: BLE001
        fails += 1
        print("FAIL", tag, "\n     ", type(e).__name__, str(e)[:300], flush=True)
        if not isinstance(e, AssertionError):
            traceback.print_exc()
print(f"{n_cases - fails} of {n_cases} preprocessing cases agree with the oracle; {done}")
sys.exit(1 if fails else 0)
