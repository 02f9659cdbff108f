#!/usr/bin/env python3
"""Rate of the native uploader alone (no fit): float64 host stories -> float32 device matrix, plain cast vs z-scored in the
staging threads.  python tools/upload_probe.py [V]      (threads / sub-tile through LITCODER_AMD_UPLOAD_THREADS /
LITCODER_AMD_ZS_TILE, one process per setting: the sub-tile width is read once)"""
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
