#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <dir or csv> [rows]  (name cut to 70 chars)."""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    found = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))
    if not found:
        sys.exit(f"no *kernel_stats.csv under {path}")
    path = found[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
with open(path, newline="") as fh:
    rows = list(csv.DictReader(fh))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{path}: {len(rows)} kernels, {total / 1e6:.2f} ms in all")
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.3f} ms {float(r['AverageNs']) / 1e3:9.1f} us "
          f"{float(r['Percentage']):6.2f} %")
