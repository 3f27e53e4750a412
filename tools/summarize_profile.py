#!/usr/bin/env python3
"""Condense a tools/gpu_round.sh output directory (gpurun_out/<tag>/) into profiles/.

For every workload profiled it keeps
  profiles/<tag>_<WL>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_<WL>.json               dominant kernel: calls, mean / median / min duration,
                                         PMC counters per launch, HBM traffic per launch
and refreshes profiles/traffic.json, which bench.py reads for roofline.traffic.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in
KiB and come from separate --pmc passes (they do not fit one pass); on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced stream (16 B per lane; the MFMA kernel's LDS-DMA loads
are such a stream), so the read side is doubled.  WRITE_SIZE is exact only for 16-B-per-lane
stores; the narrower epilogue stores of these kernels are uncalibrated, which the summary notes.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    traffic_path = os.path.join(prof, "traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    for tdir in sorted(glob.glob(os.path.join(src, "prof_*_trace"))):
        wl = os.path.basename(tdir)[5:-6]
        stats = find(tdir, "*kernel_stats.csv")
        trace = find(tdir, "*kernel_trace.csv")
        if not stats:
            continue
        shutil.copy(stats, os.path.join(prof, f"{tag}_{wl}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(stats)))
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        name = top["Name"]
        out = {"workload": wl, "kernel": name, "calls": int(top["Calls"]), "mean_us": float(top["AverageNs"]) / 1e3,
               "min_us": float(top["MinNs"]) / 1e3, "max_us": float(top["MaxNs"]) / 1e3,
               "command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps {10 if wl == 'c3T' else 200} --warmup 5 --no-extra --no-cpu"}
        if trace:
            d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(trace))
                 if r["Kernel_Name"] == name]
            if d:
                out["median_us"] = statistics.median(d) / 1e3
                # the first launches run on cold caches and a ramping clock: also report the steady tail
                tail = d[len(d) // 2:]
                out["steady_mean_us"] = sum(tail) / len(tail) / 1e3
        pmc = {}
        meta = None
        for pdir in sorted(glob.glob(os.path.join(src, f"prof_{wl}_pmc_*"))):
            if not os.path.isdir(pdir):
                continue
            cc = find(pdir, "*counter_collection.csv")
            if not cc:
                continue
            acc = {}
            for r in csv.DictReader(open(cc)):
                if r["Kernel_Name"] != name:
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                meta = {k: int(r[k]) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                               "Accum_VGPR_Count", "SGPR_Count")}
            for k, v in acc.items():
                pmc[k] = sum(v) / len(v)
        out["pmc_per_launch"] = pmc
        out["dispatch"] = meta
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            rd = pmc["FETCH_SIZE"] * 1024 * 2
            wr = pmc["WRITE_SIZE"] * 1024
            out["hbm_read_bytes_per_launch"] = rd
            out["hbm_write_bytes_per_launch"] = wr
            out["hbm_bytes_per_launch"] = rd + wr
            out["traffic_note"] = ("FETCH_SIZE KiB x1024 x2 (gfx950 half-count of 16 B/lane streams) + WRITE_SIZE KiB x1024; "
                                   "WRITE_SIZE is uncalibrated for the epilogue's sub-16-B-per-lane stores")
            traffic[wl] = {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr, "profile": f"profiles/{tag}_{wl}.json"}
        if "GRBM_GUI_ACTIVE" in pmc and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc:
            cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            out["kernel_cycles"] = cyc
            out["mfma_pipe_util"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc  # 1024 SIMDs
        json.dump(out, open(os.path.join(prof, f"{tag}_{wl}.json"), "w"), indent=1)
        print(json.dumps(out))
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    for f in ("bench.json", "pytest_gpu.log", "smoke.log"):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(prof, f"{tag}_{f}"))


if __name__ == "__main__":
    main()
