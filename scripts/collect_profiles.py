#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of scripts/profile.sh (gpurun_out/prof/<tag>_{kt,sq,fetch,write}) into the
tracked summaries under profiles/: <round>_<tag>_kernel_stats.csv (the --stats table, our kernels first) and
<round>_<tag>_pmc.json (mean counter value per dispatch and per wave, plus derived figures).
Usage: scripts/collect_profiles.py r01 room mario ..."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "gpurun_out", "prof")


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    for tag in tags:
        newest = lambda files: sorted(files, key=os.path.getmtime)[-1:]  # a directory may hold older runs
        stats = newest(glob.glob(os.path.join(PROF, f"{tag}_kt", "**", "*kernel_stats.csv"), recursive=True))
        if stats:
            shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{rnd}_{tag}_kernel_stats.csv"))
        durations = {}
        for f in stats[:1]:
            for r in csv.DictReader(open(f)):
                durations[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                               "pct": float(r["Percentage"])}
        out = collections.defaultdict(dict)
        for part in ("sq", "fetch", "write"):
            for f in newest(glob.glob(os.path.join(PROF, f"{tag}_{part}", "**", "*counter_collection.csv"), recursive=True)):
                acc = collections.defaultdict(lambda: collections.defaultdict(list))
                meta = {}
                for r in csv.DictReader(open(f)):
                    k = short(r["Kernel_Name"])
                    if "k_bounce" not in k and "k_intersect" not in k and "k_walk" not in k:
                        continue
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[k] = {"VGPR": r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), "SGPR": r.get("SGPR_Count"),
                               "LDS": r.get("LDS_Block_Size"), "Scratch": r.get("Scratch_Size"),
                               "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")}
                for k, v in acc.items():
                    out[k].setdefault("meta", meta[k])
                    for c, x in v.items():
                        out[k][c] = {"dispatches": len(x), "mean_per_dispatch": sum(x) / len(x)}
        for k, v in out.items():
            if k in durations:
                v["duration"] = durations[k]
            waves = v.get("SQ_WAVES", {}).get("mean_per_dispatch")
            d = {}
            if waves:
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
                    if c in v:
                        d[c + "_per_wave"] = v[c]["mean_per_dispatch"] / waves
            if "SQ_ACTIVE_INST_VALU" in v and "duration" in v:
                # SQ_ACTIVE_INST_VALU counts 4-cycle issue slots summed over the chip's 1024 SIMDs
                cycles = v["duration"]["avg_ns"] * 2.4
                d["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] * 4 / (1024 * cycles)
            if "FETCH_SIZE" in v:
                # rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
                d["hbm_read_bytes_per_dispatch"] = v["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
            if "WRITE_SIZE" in v:
                d["hbm_write_bytes_per_dispatch"] = v["WRITE_SIZE"]["mean_per_dispatch"] * 1024
            v["derived"] = d
        with open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_pmc.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print(tag, {k: v.get("derived") for k, v in out.items()})


if __name__ == "__main__":
    main()
