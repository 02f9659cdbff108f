#!/usr/bin/env python3
"""Rate of the native uploader alone (no fit): float64 host stories -> float32 device matrix, plain cast vs z-scored in the
staging threads.  python tools/upload_probe.py [V]      (threads / sub-tile through LITCODER_AMD_UPLOAD_THREADS /
LITCODER_AMD_ZS_TILE, one process per setting: the sub-tile width is read once)
    python tools/upload_probe.py V busy    the z-scored upload once more beside ~120 ms of MFMA work on another stream (fp16x3
                                           sweeps of the fit's own kernel): what the fit's V-wide phases do to the copies"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
dev = ops.device(0)
rng = np.random.default_rng(0)
lens = [int(n) for n in rng.integers(260, 440, 26)] + [291]
g = torch.Generator(device=dev)
g.manual_seed(1)
stories = [(3.0 * torch.randn((n, V), generator=g, device=dev) + 100.0).cpu().numpy().astype(np.float64) for n in lens]
T = sum(lens)
dY = torch.empty((T, ops.pad_to(V, 128)), dtype=torch.float32, device=dev)
panels = [(0, 12288), (12288, 36864), (36864, 67584), (67584, V)] if V == 80000 else [(0, V)]
nbytes = sum(s.nbytes for s in stories)
print(f"{len(stories)} stories, {T} rows x {V} voxels: {nbytes / 1e9:.2f} GB float64 on the host; cpu_count {os.cpu_count()}, "
      f"threads {os.environ.get('LITCODER_AMD_UPLOAD_THREADS', 'default')}, zs tile {os.environ.get('LITCODER_AMD_ZS_TILE', 'default')}")
for zs in (False, True):
    host = ops.HostRows(stories, zscore=zs)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        up = ops.PanelUploader([(host, dY, a, b) for a, b in panels], dev)
        marks = []
        for j in range(len(panels)):
            up.wait(j)
            torch.cuda.current_stream().synchronize()
            marks.append(1e3 * (time.perf_counter() - t0))
        up.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"  zscore={zs}: {1e3 * dt:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s of float64 read, {nbytes / 2 / dt / 1e9:.1f} GB/s on the "
              f"link; panels resident after {[round(m, 1) for m in marks]} ms", flush=True)

if "busy" in sys.argv[2:]:
    # the same upload while the chip is busy: plain fp16x3 contractions (the fit's dominant kernel) back to back on the
    # current stream, the upload on its own stream beside them
    Vb, M, K = 32768, 3072, 1856
    a = torch.randn((M, K), device=dev)
    b = torch.randn((K, Vb), device=dev)
    At = torch.empty(ops.pad_to(M, 256) * K * 2, dtype=torch.float16, device=dev)
    rs = torch.empty(ops.pad_to(M, 256), dtype=torch.float32, device=dev)
    ops.split_rows_f16(a, M, K, At, rs)
    cs, _ = ops.col_scales_f16(b, K, Vb, want_flag=False)
    Bt = torch.empty(Vb * K * 2, dtype=torch.float16, device=dev)
    ops.split_cols_f16(b, Vb, ops.idx_tensor(np.arange(K), K, dev), K, cs, Bt)
    cs_inv = cs[Vb:].contiguous()
    C = torch.empty((M, Vb), dtype=torch.float32, device=dev)

    hbm = "hbm" in sys.argv[2:]          # ... or HBM-bound passes instead (sums of 0.4 GB matrices: the fit's B_f)
    terms = [torch.randn((M, Vb), device=dev) for _ in range(4)] if hbm else None

    def work(n):
        for _ in range(n):
            if hbm:
                ops.combine_colmax(terms, [1.0] * 4, C, Vb)
            else:
                ops.gemm_grouped_f16x3(At, rs, M, Bt, cs_inv, C, Vb, Vb, K, [0, Vb // 256])

    work(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); work(20); torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 20
    print(f"busy work: {1e3 * per:.2f} ms per launch alone (" + (f"{5 * M * Vb * 4 / per / 1e12:.2f} TB/s)" if hbm else
                                                                f"{2.0 * M * K * Vb / per / 1e12:.0f} TF)"))
    host = ops.HostRows(stories, zscore=True)
    for beside in (False, True, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        up = ops.PanelUploader([(host, dY, a_, b_) for a_, b_ in panels], dev)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        if beside:
            e0.record(); work(int(0.12 / per)); e1.record()
        up.join()
        side = torch.cuda.Stream()
        up.wait(len(panels) - 1, side)
        side.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        extra = f"; the MFMA work took {e0.elapsed_time(e1):.1f} ms ({int(0.12 / per)} launches: {1e3 * per * int(0.12 / per):.1f} alone)" if beside else ""
        print(f"  z-scored upload {'beside MFMA work' if beside else 'alone'}: all panels resident after {1e3 * dt:.1f} ms{extra}", flush=True)
