#!/usr/bin/env python3
"""All queues of the LAST fit of a rocprofv3 --kernel-trace CSV inside a time window [t0, t1) ms (relative to the fit's
first kernel): runs of one kernel per queue, merged.   python tools/queue_window.py <kernel_trace.csv> t0 t1"""
import csv
import re
import sys

path, w0, w1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])


def short(n):
    m = re.search(r"(k_[a-zA-Z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else re.sub(r"\(.*", "", n)[-40:]


rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_widen") or r[2].startswith("k_gram")]
i0 = starts[-1] if starts else 0
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200_000:
    i0 -= 1
fit = rows[i0:]
t0 = fit[0][0]
runs = {}
out = []
for s, e, n, q in fit:
    if (e - t0) / 1e6 < w0 or (s - t0) / 1e6 >= w1:
        continue
    run = runs.get(q)
    if run is not None and run[2] == n and s - run[1] < 20_000:
        run[1] = e; run[4] += 1; run[5] += e - s
    else:
        run = runs[q] = [s, e, n, q, 1, e - s]
        out.append(run)
for s, e, n, q, c, busy in sorted(out):
    print(f"q{q}  {(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} ms  busy {busy / 1e3:9.1f} us  x{c:<4d} {n}")
