#!/usr/bin/env python3
"""n cfg3 story-pipeline fits in a row (bench.synth_stories -> StoryPipeline.fit_words): the workload under a profiler.
    python tools/cfg3_fit_loop.py [n] [V]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from litcoder_core_amd import NestedCVModel, StoryPipeline, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
V = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
dev = ops.device(0)
words, wtimes, trtimes, brain = bench.synth_stories(V, dev)
model = NestedCVModel("ridge_regression")
pipe = StoryPipeline([1, 2, 3, 4], bench.CFG3_TRIM, model=model)
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.fit_words(words, wtimes, trtimes, brain, **bench.CFG3_KW)
    torch.cuda.synchronize()
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.1f} ms ({model.last_form}, alpha {out[2][0]:g})", flush=True)
    out = None
