#!/usr/bin/env python3
"""Which of the engine's streams end up serialised behind one another by the runtime (more streams than hardware queues)?
For every pair (busy stream X, probe stream Y): ~20 ms of kernels on X, then one tiny kernel on Y and a host wait for it --
a wait of the order of X's work means Y's launches queue behind X's.     python tools/stream_queue_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402
from litcoder_core_amd.engine.common import _aux_stream  # noqa: E402

dev = ops.device(0)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
streams = {"main (default)": torch.cuda.current_stream(), "upload": ops.upload_stream(dev)}
for i, name in enumerate(("aux", "aux2", "comm", "aux3", "dl", "scales")):
    streams[name] = _aux_stream(dev, i)
a = torch.randn((4096, 4096), device=dev)
tiny = torch.zeros(64, device=dev)
for s in streams.values():                       # every stream has run something (queues exist)
    with torch.cuda.stream(s):
        tiny.add_(1.0)
torch.cuda.synchronize()


def busy(s, n=40):
    with torch.cuda.stream(s):
        for _ in range(n):
            torch.mm(a, a)


busy(streams["aux"]); torch.cuda.synchronize()
t = time.perf_counter(); busy(streams["aux"]); torch.cuda.synchronize()
print(f"busy work alone: {1e3 * (time.perf_counter() - t):.1f} ms")
names = list(streams)
print("rows: busy stream; columns: probe stream; entry: ms until the probe's tiny kernel had run")
print(" " * 16 + "".join(f"{n[:8]:>9s}" for n in names))
for x in names:
    row = []
    for y in names:
        if x == y:
            row.append("     -   ")
            continue
        torch.cuda.synchronize()
        busy(streams[x])
        t = time.perf_counter()
        with torch.cuda.stream(streams[y]):
            tiny.add_(1.0)
            e = torch.cuda.Event(); e.record()
        e.synchronize()
        row.append(f"{1e3 * (time.perf_counter() - t):9.1f}")
        torch.cuda.synchronize()
    print(f"{x:16s}" + "".join(row))
