#!/usr/bin/env python3
"""VERDICT r4 item 3, measured instead of costed: does ONE launch of the dominant kernel over the stacked rows of an inner
fold's two operators -- the factorised alphas' hat matrices (4 x 480 rows = 8 M-tiles) and the shared series terms (4 x 480
rows = 8 M-tiles) -- beat the two launches the fit issues, by fetching every 256-column panel of Y[tr] once for both?

The main loop of k_sweep_f16x3 is the same in every mode; what a stacked launch changes is the tile population per XCD
round (16 M-tiles of a column panel instead of 8) and with it the traffic that leaves L2.  So the question is answered with
the kernel's PEARSON mode (a reduction epilogue that reads targets and writes a few partials per column, like the score and
series-moments epilogues; no new kernel needed):

    arm "separate": two launches, 8 M-tiles each, over the same B image      (what the fit does)
    arm "stacked":  one launch, 16 M-tiles                                    (what item 3 proposes)

at the cfg2 shape (K = 1920, V = 80 000), interleaved in ONE process (cdna_hip_programming.md rule 24), HIP events on the
launch stream.  Run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE` / `--pmc WRITE_SIZE` and feed the
CSVs to this script with `parse <dir> ...` for the bytes beyond L2 and the clock of either arm (the launches are told apart
by their grid size).

    python tools/stack_experiment.py [rounds]
    python tools/stack_experiment.py parse <rocprof dir> [<rocprof dir> ...]
"""
import csv
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(dirs):
    acc = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "k_sweep_f16x3<false, false, false, false, true>" not in row["Kernel_Name"]:
                    continue
                g = int(row["Grid_Size"]) // 512
                e = acc.setdefault(g, {})
                e.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    e.setdefault("ns", []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for g, e in sorted(acc.items()):
        out = {"workgroups": g, "launches": len(next(iter(e.values())))}
        if "FETCH_SIZE" in e:
            out["fetch_GB"] = 2 * 1024 * np.mean(e["FETCH_SIZE"]) / 1e9           # KiB; x2: gfx950 wide-read correction
        if "WRITE_SIZE" in e:
            out["write_GB"] = 1024 * np.mean(e["WRITE_SIZE"]) / 1e9
        if "GRBM_GUI_ACTIVE" in e:
            out["ms_in_pmc_pass"] = np.mean(e["ns"]) / 1e6
            out["clock_GHz"] = np.sum(e["GRBM_GUI_ACTIVE"]) / 8 / np.sum(e["ns"])
        print(out)


if len(sys.argv) > 1 and sys.argv[1] == "parse":
    parse(sys.argv[2:])
    sys.exit(0)

import torch  # noqa: E402
from litcoder_core_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = ops.device()
g = torch.Generator(device=dev)
g.manual_seed(3)
K, V, M = 1920, 80000, 2048                      # rows per operator: 4 x 480 padded to whole 256-row tiles
Vt = ops.pad_to(V, 256)
Y = torch.randn((K, V), generator=g, device=dev, dtype=torch.float32)
Yte = torch.randn((2 * M, Vt), generator=g, device=dev, dtype=torch.float32)       # "targets" of the reduction epilogue
cs, _ = ops.col_scales_f16(Y, K, V)
cs_inv = torch.ones(Vt, dtype=torch.float32, device=dev)
cs_inv[:V] = cs[V:]
Yt = torch.empty(Vt * K * 2, dtype=torch.float16, device=dev)
ops.split_cols_f16(Y, V, torch.arange(K, dtype=torch.int32, device=dev), K, cs, Yt)
A = torch.randn((2 * M, K), generator=g, device=dev, dtype=torch.float32) * 0.02
At = torch.empty(2 * M * K * 2, dtype=torch.float16, device=dev)
rs = torch.empty(2 * M, dtype=torch.float32, device=dev)
ops.split_rows_f16(A, 2 * M, K, At, rs)            # tiles 0-7: operator 1, tiles 8-15: operator 2 (tile-aligned images)
At2, rs2 = At[M * K * 2:], rs[M:]
r1 = torch.empty(Vt, dtype=torch.float64, device=dev)
tiles = [0, Vt // 256]


def separate():
    ops.gemm_grouped_f16x3_pearson(At, rs, M, Yt, cs_inv, Vt, K, tiles, Yte, None, None, r1)
    ops.gemm_grouped_f16x3_pearson(At2, rs2, M, Yt, cs_inv, Vt, K, tiles, Yte[M:], None, None, r1)


def stacked():
    ops.gemm_grouped_f16x3_pearson(At, rs, 2 * M, Yt, cs_inv, Vt, K, tiles, Yte, None, None, r1)


for fn in (separate, stacked):
    fn()
torch.cuda.synchronize()
times = {"separate": [], "stacked": []}
for _ in range(rounds):
    for name, fn in (("separate", separate), ("stacked", stacked)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 5)
fl = 2.0 * 2 * M * K * V
for name, t in times.items():
    t = np.asarray(t)
    print(f"{name:9s}: median {np.median(t):.3f} ms, min {t.min():.3f}, max {t.max():.3f} over {rounds} interleaved rounds "
          f"(includes the small k_pearson_from_parts launches: 2 vs 1) -> {fl / np.median(t) / 1e9:.0f} TFLOP/s algorithmic")
print(f"stacked / separate (medians): {np.median(times['stacked']) / np.median(times['separate']):.4f}")
