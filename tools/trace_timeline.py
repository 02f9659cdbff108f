#!/usr/bin/env python3
"""Per-queue timeline of the last `win` ms of a rocprofv3 --kernel-trace CSV: runs of the same kernel merged into one
line (start, span, busy, launches).   python tools/trace_timeline.py <kernel_trace.csv> <win_ms> [min_span_us]"""
import csv
import sys

path, win = sys.argv[1], float(sys.argv[2]) * 1e6
min_span = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0.0
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
t1 = max(r[1] for r in rows)
lo = t1 - win
rows = [r for r in rows if r[1] >= lo]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:46]


queues = sorted({r[3] for r in rows})
for q in queues:
    print(f"=== queue {q}")
    cur = None
    for s, e, n, qq in rows:
        if qq != q:
            continue
        n = short(n)
        if cur and cur[2] == n and s - cur[1] < 30e3:
            cur[1] = e; cur[3] += e - s; cur[4] += 1
        else:
            if cur and cur[1] - cur[0] >= min_span:
                print(f"  {(cur[0] - lo) / 1e6:8.3f} ms  span {(cur[1] - cur[0]) / 1e3:8.1f} us  busy {cur[3] / 1e3:8.1f} us  x{cur[4]:<4d} {cur[2]}")
            cur = [s, e, n, e - s, 1]
    if cur and cur[1] - cur[0] >= min_span:
        print(f"  {(cur[0] - lo) / 1e6:8.3f} ms  span {(cur[1] - cur[0]) / 1e3:8.1f} us  busy {cur[3] / 1e3:8.1f} us  x{cur[4]:<4d} {cur[2]}")
