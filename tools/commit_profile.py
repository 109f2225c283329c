#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_all.sh) into the summaries kept under profiles/:
  profiles/<tag>_kernel_stats.json, profiles/<tag>_pmc.json, profiles/<tag>_rocprofv3_kernel_stats.csv,
  profiles/<tag>_bench_under_rocprof.json and profiles/hbm_traffic.json (what bench.py labels its PMC fields with).
Usage: tools/commit_profile.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocprof_summary as rs  # noqa: E402

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_{tag}")
stats = rs.stats(os.path.join(src, "stats"))
pmc = rs.pmc(sorted(glob.glob(os.path.join(src, "pmc_*/"))))

# effective clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's own duration in that pass
clock = defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_GRBM*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fa::" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dt = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if dt > 0:
                clock[rs.short(r["Kernel_Name"])].append(float(r["Counter_Value"]) / 8 / dt)
for k, v in clock.items():
    pmc.setdefault(k, {})["clock_ghz"] = sorted(v)[len(v) // 2]

json.dump(stats, open(f"profiles/{tag}_kernel_stats.json", "w"), indent=1)
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True):
    shutil.copy(f, f"profiles/{tag}_rocprofv3_kernel_stats.csv")
b = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(b):
    shutil.copy(b, f"profiles/{tag}_bench_under_rocprof.json")


def pick(prefix):
    for k in pmc:
        if k.startswith(prefix):
            return pmc[k]
    return {}


names = {"decode_kernel": "decode_fast_kernel<9>", "hash_kernel": "hash_kernel", "verify_kernel": "verify_kernel<9, false, false"}
out = {"hbm_bytes_per_launch": {}, "valu_instructions_per_launch": {}, "clock_ghz": {},
       "_from": f"profiles/{tag}_pmc.json",
       "_note": "per launch of 2^20 Falcon-512 items. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; FETCH_SIZE doubled "
                "per MI355X_MICROARCH.md HBM section: calibrated on the verify kernel, an upper bound for the other two, DESIGN.md "
                "section 6). valu_instructions = SQ_INSTS_VALU (wave instructions). clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch time."}
for short, full in names.items():
    p = pick(full)
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        out["hbm_bytes_per_launch"][short] = int((2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024)
    if "SQ_INSTS_VALU" in p:
        out["valu_instructions_per_launch"][short] = p["SQ_INSTS_VALU"]
    if "clock_ghz" in p:
        out["clock_ghz"][short] = p["clock_ghz"]
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
