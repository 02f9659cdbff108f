#!/usr/bin/env python3
"""GPU idle-gap report from a rocprofv3 --kernel-trace CSV: union of kernel intervals, total busy time, and the
largest gaps with the kernels on either side.   python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
last_name = rows[0][2]
for s, e, name, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, last_name, name))
        cur_s, cur_e = s, e
        last_name = name
    elif e > cur_e:
        cur_e = e
        last_name = name
busy += cur_e - cur_s
print(f"span {(t1 - t0) / 1e6:.1f} ms, busy (union) {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms, kernels {len(rows)}")
big = [g for g in gaps if g[0] / 1e3 >= min_gap]
print(f"gaps >= {min_gap:.0f} us: {len(big)}, total {sum(g[0] for g in big) / 1e6:.1f} ms; "
      f"smaller gaps: {len(gaps) - len(big)}, total {sum(g[0] for g in gaps if g[0] / 1e3 < min_gap) / 1e6:.1f} ms")
for g in sorted(big, reverse=True)[:40]:
    print(f"  {g[0] / 1e3:9.0f} us at {g[1] / 1e6:9.2f} ms   after {g[2]}   before {g[3]}")

# per-queue busy time and the top kernels per queue inside a window (last `win` ms of the trace)
if len(sys.argv) > 3:
    win = float(sys.argv[3]) * 1e6
    lo = t1 - win
    per_q = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if e < lo:
                continue
            q = r.get("Queue_Id", "?")
            d = per_q.setdefault(q, {})
            n = r["Kernel_Name"].split("(")[0][-48:]
            d[n] = d.get(n, 0) + (e - max(s, lo))
    for q, d in per_q.items():
        tot = sum(d.values())
        print(f"queue {q}: kernel time {tot / 1e6:.1f} ms in the last {win / 1e6:.0f} ms")
        for n, v in sorted(d.items(), key=lambda kv: -kv[1])[:14]:
            print(f"     {v / 1e6:8.2f} ms  {n}")
