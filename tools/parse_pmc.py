#!/usr/bin/env python3
"""Summarise rocprofv3 PMC CSVs for one kernel: per-launch FETCH_SIZE / WRITE_SIZE in bytes with the
gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB,
and FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled.

    python tools/parse_pmc.py <dir with *_counter_collection.csv ...> <kernel substring> <out.json>
"""
import csv
import glob
import json
import os
import sys


def main():
    roots, needle, out = sys.argv[1:-2], sys.argv[-2], sys.argv[-1]
    sums, counts = {}, {}
    for root in roots:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if needle not in row.get("Kernel_Name", ""):
                        continue
                    name = row["Counter_Name"]
                    sums[name] = sums.get(name, 0.0) + float(row["Counter_Value"])
                    counts[name] = counts.get(name, 0) + 1
    res = {"kernel": needle, "launches": counts, "raw_mean": {k: sums[k] / counts[k] for k in sums}}
    fetch = res["raw_mean"].get("FETCH_SIZE")
    write = res["raw_mean"].get("WRITE_SIZE")
    if fetch is not None:
        res["fetch_bytes_per_launch"] = fetch * 1024 * 2        # KiB, x2 for 16-B/lane streaming reads on gfx950
    if write is not None:
        res["write_bytes_per_launch"] = write * 1024
    if fetch is not None and write is not None:
        res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    res["note"] = ("FETCH_SIZE/WRITE_SIZE come from the L2's memory-side request counters; Infinity-Cache hits are "
                   "counted, so this is traffic leaving L2, an upper bound on HBM bytes")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
