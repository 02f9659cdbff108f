#!/usr/bin/env python3
"""Random search over LITCODER_AMD_STREAM_ORDER (the order in which the engine's streams get their hardware queues): one
process per order (host-to-host fits first, then resident ones: bench.py's situation), median fit times of both.
    python tools/stream_order_search.py [n_random] [seed]"""
import os
import random
import subprocess
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
here = os.path.dirname(os.path.abspath(__file__))
items = ["0", "1", "2", "3", "4", "5", "6", "7", "u"]
orders = ["", "0,1,7,2,3,4,5,6,u", "0,1,7,6,4,2,5,3,u", "0,1,7,6,4,2,u,3,5", "0,1,7,3,2,4,5,6,u", "7,0,1,6,5,4,2,3,u"]
for _ in range(n):
    p = items[:]
    rng.shuffle(p)
    orders.append(",".join(p))
rows = []
for o in orders:
    env = dict(os.environ, LITCODER_AMD_STREAM_ORDER=o)
    out = subprocess.run([sys.executable, os.path.join(here, "host_then_resident.py"), "host"], env=env, capture_output=True, text=True).stdout
    vals = [float(t.split()[1]) for t in out.replace("\n", " ").split(";") if t.strip() and t.split()[0] in ("host", "resident")]
    if len(vals) == 2:
        rows.append((vals[0], vals[1], o))
        print(f"order '{o}': host {vals[0]:.1f} ms, resident {vals[1]:.1f} ms", flush=True)
    else:
        print(f"order '{o}': failed: {out[-200:]}", flush=True)
print("best by host + resident / 2:")
for h, r, o in sorted(rows, key=lambda t: t[0] + 0.5 * t[1])[:6]:
    print(f"  host {h:.1f}  resident {r:.1f}  '{o}'")
