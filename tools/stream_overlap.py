#!/usr/bin/env python3
"""Per-stream occupancy of the last fit in a rocprofv3 --kernel-trace CSV of bench.py: how long each stream has a
kernel in flight, how long both do, and the intervals in which only one (or none) does, with the kernels around them.
    python tools/stream_overlap.py <kernel_trace.csv> [min_interval_us]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]))
rows.sort()
# a fit starts with the Gram matrix
starts = [i for i, r in enumerate(rows) if "k_gram" in r[2]]
rows = rows[starts[-1]:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
by = defaultdict(list)
for s, e, n, q in rows:
    by[q].append((s, e, n))
order = sorted(by, key=lambda q: -sum(e - s for s, e, _ in by[q]))
print(f"last fit: {(t1 - t0) / 1e6:.1f} ms, {len(rows)} kernels, streams {[(q, len(by[q])) for q in order]}")


def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


U = {q: union([(s, e) for s, e, _ in by[q]]) for q in order}
for q in order:
    print(f"  stream {q}: busy {sum(e - s for s, e in U[q]) / 1e6:.1f} ms, kernel time {sum(e - s for s, e, _ in by[q]) / 1e6:.1f} ms")
# sweep line over the two busiest streams
a, b = order[0], order[1]
ev = []
for q, tag in ((a, 0), (b, 1)):
    for s, e in U[q]:
        ev.append((s, 1, tag)); ev.append((e, -1, tag))
ev.sort()
state = [0, 0]
last = t0
tot = defaultdict(int)
segs = []
for t, d, tag in ev:
    key = (state[0] > 0, state[1] > 0)
    if t > last:
        tot[key] += t - last
        segs.append((last, t, key))
    state[tag] += d
    last = t
names = {(True, True): "both", (True, False): f"only {a}", (False, True): f"only {b}", (False, False): "none"}
for k, v in tot.items():
    print(f"  {names[k]}: {v / 1e6:.1f} ms")


def around(t, q):
    prev = [n for s, e, n in by[q] if e <= t]
    nxt = [n for s, e, n in by[q] if s >= t]
    return (prev[-1][:40] if prev else "-"), (nxt[0][:40] if nxt else "-")


print(f"intervals >= {min_us:.0f} us in which a stream is idle:")
merged = []
for s, e, k in segs:
    if k == (True, True):
        continue
    if merged and merged[-1][2] == k and s - merged[-1][1] < 20000:
        merged[-1][1] = e
    else:
        merged.append([s, e, k])
for s, e, k in merged:
    if (e - s) / 1e3 >= min_us:
        idle = b if k[0] and not k[1] else a if k[1] and not k[0] else "both"
        q = b if idle == b else a
        p, n = around(s, q)
        print(f"  at {(s - t0) / 1e6:7.2f} ms for {(e - s) / 1e3:7.0f} us: idle {idle}; its last kernel {p} | next {n}")
