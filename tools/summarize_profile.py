#!/usr/bin/env python3
"""Condense a tools/gpu_round.sh output directory (gpurun_out/<tag>/) into profiles/.

For every workload profiled it keeps
  profiles/<tag>_<WL>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_<WL>.json               dominant kernel: calls, mean / median / min duration,
                                         PMC counters per launch, HBM traffic per launch
and refreshes profiles/kernels.json, which bench.py reads for what its roofline blocks quote from counters (vector instructions
per MAC, VALU busy share, clock, MFMA pipe share, HBM-side traffic) — each entry together with the kernel symbol, the engine's
kernel id and step form, the git HEAD and the source hashes it was measured on (qublas_amd/profmeta.py: a block quotes an entry only
while those still match).

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
    kernels_path = os.path.join(prof, "kernels.json")
    kernels = json.load(open(kernels_path)) if os.path.exists(kernels_path) else {}
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
        steps = 10 if wl in ("c3T", "c3Td", "c5TF", "c5B", "long_k", "w32T") else 200
        out = {"workload": wl, "kernel": name, "calls": int(top["Calls"]), "mean_us": float(top["AverageNs"]) / 1e3,
               "min_us": float(top["MinNs"]) / 1e3, "max_us": float(top["MaxNs"]) / 1e3,
               "command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps {steps} --warmup 5 --no-extra --no-cpu"}
        # what the profiled run says about itself (bench.py's `profile_key`: engine kernel, step form, MACs, source hashes, HEAD)
        key = None
        log = os.path.join(src, f"prof_{wl}_trace.log")
        if os.path.exists(log):
            for ln in open(log, errors="replace"):
                if ln.startswith('{"metric"'):
                    try:
                        j = json.loads(ln)
                        key = j.get("profile_key")
                        out["bench_ms_per_step_events"] = j.get("ms_per_step_events")
                    except Exception:
                        pass
        out["profile_key"] = key
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
        entry = {"profile": f"profiles/{tag}_{wl}.json", "kernel_symbol": name, "kernel_us": out.get("steady_mean_us", out["mean_us"])}
        if "hbm_bytes_per_launch" in out:
            entry["hbm"] = {"bytes_per_launch": out["hbm_bytes_per_launch"], "read": out["hbm_read_bytes_per_launch"], "write": out["hbm_write_bytes_per_launch"]}
        if "GRBM_GUI_ACTIVE" in pmc:
            cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            out["kernel_cycles"] = cyc
            entry["cycles"] = cyc
            # clock the chip held under this kernel: cycles of a counter pass over the kernel's duration in the trace pass
            entry["clock_ghz"] = cyc / (out.get("median_us", out["mean_us"]) * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc:
                out["mfma_pipe_util"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc  # 1024 SIMDs
                if pmc.get("SQ_INSTS_MFMA", 0) > 0:
                    entry["mfma"] = {"pipe_busy": out["mfma_pipe_util"], "insts_mfma": pmc["SQ_INSTS_MFMA"]}
            if "SQ_INSTS_VALU" in pmc and key and key.get("macs"):
                v = {"insts_valu_wave64": pmc["SQ_INSTS_VALU"], "instr_per_mac": pmc["SQ_INSTS_VALU"] * 64.0 / key["macs"]}
                if "SQ_ACTIVE_INST_VALU" in pmc:
                    v["active_inst_valu"] = pmc["SQ_ACTIVE_INST_VALU"]
                    v["valu_busy"] = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)   # issue cycles (4 per wave64 instruction) over SIMD cycles
                out["valu"] = v
                entry["valu"] = v
        if key:
            entry.update({k: key[k] for k in ("engine_kernel", "engine_reason", "macs", "sources", "head")})
            kernels[wl] = entry
        json.dump(out, open(os.path.join(prof, f"{tag}_{wl}.json"), "w"), indent=1)
        print(json.dumps(out))
    json.dump(kernels, open(kernels_path, "w"), indent=1)
    for f in ("bench.json", "pytest_gpu.log", "smoke.log"):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(prof, f"{tag}_{f}"))


if __name__ == "__main__":
    main()
