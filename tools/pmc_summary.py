#!/usr/bin/env python3
"""Condense a tools/pmc.sh output directory: per kernel, duration statistics from the kernel trace and the mean of every PMC
counter per launch; derived: effective clock (GRBM_GUI_ACTIVE / 8 / duration), MFMA pipe busy share, HBM-side bytes
(FETCH_SIZE doubled on gfx950 per MI355X_MICROARCH.md §HBM; both in KiB)."""
import csv
import glob
import json
import os
import statistics
import sys


def main():
    src = sys.argv[1]
    kern = {}
    for tr in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(tr)):
            kern.setdefault(r["Kernel_Name"], {"dur": []})["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for cc in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(cc)):
            k = kern.setdefault(r["Kernel_Name"], {"dur": []})
            k.setdefault("pmc", {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            k["dispatch"] = {x: int(r[x]) for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    out = []
    for name, k in kern.items():
        d = k["dur"]
        if not d or sum(d) < 1e5:
            continue
        rec = {"kernel": name[:100], "calls": len(d), "mean_us": sum(d) / len(d) / 1e3, "median_us": statistics.median(d) / 1e3, "min_us": min(d) / 1e3}
        p = {c: sum(v) / len(v) for c, v in k.get("pmc", {}).items()}
        rec["pmc_per_launch"] = p
        rec["dispatch"] = k.get("dispatch")
        der = {}
        if "GRBM_GUI_ACTIVE" in p:
            der["clock_GHz"] = p["GRBM_GUI_ACTIVE"] / 8 / (rec["median_us"] * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in p:   # summed over all SIMDs? the counter is per SE aggregated: report the ratio to 4 SIMD x 256 CU x cycles
                der["mfma_busy_share"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / (p["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
        if "SQ_WAVE_CYCLES" in p:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
                if c in p:
                    der[c + "/WAVE_CYCLES"] = p[c] / p["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in p:
            der["hbm_read_bytes"] = p["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in p:
            der["hbm_write_bytes"] = p["WRITE_SIZE"] * 1024
        rec["derived"] = der
        out.append(rec)
    out.sort(key=lambda r: -r["mean_us"] * r["calls"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
