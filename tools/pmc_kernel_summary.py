#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --kernel-trace --pmc passes (counters only): launches, mean duration, the clock the chip
held under the kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration), the matrix pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x cycles)) and the bytes leaving L2 per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md).  Launches of the (instantiation, grid size) that holds most of the kernel's time only:
the fit's full-width ones.

    python tools/pmc_kernel_summary.py <dir> [<dir> ...] -- <kernel substring> [<kernel substring> ...]
"""
import csv
import glob
import json
import os
import sys


def main():
    cut = sys.argv.index("--")
    roots, needles = sys.argv[1:cut], sys.argv[cut + 1:]
    rows = []
    for root in roots:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                rows += [(root, r) for r in csv.DictReader(f)]
    out = {}
    for needle in needles:
        sel = [(root, r) for root, r in rows if needle in r.get("Kernel_Name", "")]
        if not sel:
            out[needle] = None
            continue
        # the (instantiation, grid) group that holds most of the kernel's time: the full-width launches of the fit.  (Until the
        # round's last refresh: the largest grid -- which became the undecided voxels' first panel of a fit, five inner folds
        # in one launch over a capacity of V / 2 columns, nearly all of its workgroups leaving at once.)
        weight = {}
        for root, r in sel:
            if r["Counter_Name"] == sel[0][1]["Counter_Name"]:
                key = (r["Kernel_Name"], int(r["Grid_Size"]))
                weight[key] = weight.get(key, 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        kname, full = max(weight, key=weight.get)
        sel = [(root, r) for root, r in sel if r["Kernel_Name"] == kname]
        disp = {}
        for root, r in sel:
            if int(r["Grid_Size"]) != full:
                continue
            d = disp.setdefault((root, r["Dispatch_Id"]), {"ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
        ent = {"kernel": sel[0][1]["Kernel_Name"][:110], "grid_threads": full, "workgroup": sel[0][1].get("Workgroup_Size"),
               "lds_bytes": sel[0][1].get("LDS_Block_Size"), "vgprs": sel[0][1].get("VGPR_Count")}
        for name in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE"):
            ds = [d for d in disp.values() if name in d]
            if ds:
                ent[name] = {"launches": len(ds), "mean": sum(d[name] for d in ds) / len(ds), "mean_ms": 1e-6 * sum(d["ns"] for d in ds) / len(ds)}
        clk = [d for d in disp.values() if "GRBM_GUI_ACTIVE" in d]
        if clk:
            cyc = sum(d["GRBM_GUI_ACTIVE"] for d in clk) / 8.0
            ns = sum(d["ns"] for d in clk)
            busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in clk)
            ent.update(clock_ghz=cyc / ns, mfma_pipe_busy=busy / (1024.0 * cyc) if busy else None,
                       busy_x_clock_ghz=(busy / (1024.0 * cyc)) * cyc / ns if busy else None,
                       avg_launch_ms=1e-6 * ns / len(clk))
        if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
            ent["bytes_leaving_l2_per_launch"] = ent["FETCH_SIZE"]["mean"] * 2048 + ent["WRITE_SIZE"]["mean"] * 1024
        out[needle] = ent
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
