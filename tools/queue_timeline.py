#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 --kernel-trace CSV: the LAST fit of the trace (from the last k_gram / first kernel after the
longest idle gap on), for the queue that runs the MFMA sweeps: every kernel with start, duration and the idle time
before it, consecutive launches of one kernel merged; then the kernel-time totals of the other queues in that window.
    python tools/queue_timeline.py <kernel_trace.csv> [min_us_to_list] [all]      (all: list the other queues the same way)"""
import collections
import csv
import re
import sys

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0


def short(n):
    m = re.search(r"(k_[a-zA-Z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else re.sub(r"\(.*", "", n)[-40:]


rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
rows.sort()
# the last fit: from the last k_gram launch (one per fit) on
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_widen") or r[2].startswith("k_gram")]
i0 = starts[-1] if starts else 0
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200_000:      # include what was queued right before it (uploads, casts)
    i0 -= 1
fit = rows[i0:]
t0 = fit[0][0]
sweep_q = collections.Counter(r[3] for r in fit if r[2].startswith("k_sweep")).most_common(1)[0][0]
print(f"last fit: {(max(r[1] for r in fit) - t0) / 1e6:.2f} ms, {len(fit)} kernels; main queue = {sweep_q}")


def list_queue(kernels):
    prev_end, run, out = None, None, []
    for s, e, n, q in kernels:
        gap = 0 if prev_end is None else max(0, s - prev_end)
        if run is not None and run[2] == n and gap < 5_000:
            run[1] = e; run[3] += 1; run[4] += e - s
        else:
            if run is not None:
                out.append(run)
            run = [s, e, n, 1, e - s, gap]
        prev_end = max(prev_end or e, e)
    out.append(run)
    for s, e, n, c, busy, gap in out:
        if busy / 1e3 >= min_us or gap / 1e3 >= 50:
            print(f"{(s - t0) / 1e6:9.3f} ms  {busy / 1e3:9.1f} us  x{c:<3d} gap {gap / 1e3:7.1f} us  {n}")


list_queue([r for r in fit if r[3] == sweep_q])
if "all" in sys.argv[3:]:
    for q in sorted({r[3] for r in fit} - {sweep_q}):
        print(f"--- queue {q}")
        list_queue([r for r in fit if r[3] == q])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, n, q in fit:
    tot[q][n] += (e - s) / 1e6
for q, d in tot.items():
    print(f"queue {q}{' (main)' if q == sweep_q else ''}: {sum(d.values()):.1f} ms of kernel time")
    for n, v in sorted(d.items(), key=lambda kv: -kv[1])[:16]:
        print(f"     {v:8.2f} ms  {n}")
