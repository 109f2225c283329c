#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small JSON/markdown summaries kept under
profiles/. Usage:
  tools/rocprof_summary.py stats <dir>            -> per-kernel count / avg / total from *_kernel_trace.csv
  tools/rocprof_summary.py pmc <dir> [<dir> ...]  -> per-kernel mean of every counter in *_counter_collection.csv
Only kernels of this library (namespace fa::) are listed unless --all is given."""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def _trace_rows(d):
    """Kernel-dispatch rows from either output format of rocprofv3 (csv or the rocpd sqlite db)."""
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            yield r
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        cur = sqlite3.connect(f).cursor()
        for r in cur.execute("select name, start, end, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                             "workgroup_x, grid_x from kernels"):
            yield {"Kernel_Name": r[0], "Start_Timestamp": r[1], "End_Timestamp": r[2], "VGPR_Count": r[3],
                   "Accum_VGPR_Count": r[4], "SGPR_Count": r[5], "LDS_Block_Size": r[6], "Scratch_Size": r[7],
                   "Workgroup_Size_X": r[8], "Grid_Size_X": r[9]}


def _counter_rows(d):
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            yield r
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        cur = sqlite3.connect(f).cursor()
        try:
            rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
        except sqlite3.Error:
            continue
        for r in rows:
            yield {"Dispatch_Id": str(r[0]), "Kernel_Name": r[1], "Counter_Name": r[2], "Counter_Value": r[3]}


def short(name):
    name = name.replace("void ", "")
    for k in ("decode_fast_kernel", "decode_kernel", "hash_kernel", "verify_kernel", "polymul_kernel", "fused_kernel"):
        if "fa::" + k in name:
            tmpl = name[name.find(k) + len(k):]
            args = tmpl[tmpl.find("<") + 1:tmpl.find(">")] if tmpl.startswith("<") else ""
            return f"{k}<{args}>" if args else k
    return name[:60]


def stats(d, all_kernels=False):
    rows = defaultdict(list)
    meta = {}
    for r in _trace_rows(d):
        if not all_kernels and "fa::" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        rows[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                   "lds_bytes": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]),
                   "workgroup": int(r["Workgroup_Size_X"]), "grid": int(r["Grid_Size_X"])}
    out = {}
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        v_sorted = sorted(v)
        out[k] = {"calls": len(v), "avg_us": sum(v) / len(v), "median_us": v_sorted[len(v) // 2], "min_us": v_sorted[0],
                  "max_us": v_sorted[-1], "total_us": sum(v), **meta[k]}
    return out


def pmc(dirs, all_kernels=False):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        per_dispatch = defaultdict(float)
        names = {}
        for r in _counter_rows(d):
            if not all_kernels and "fa::" not in r["Kernel_Name"]:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (disp, cname), val in per_dispatch.items():
            acc[names[disp]][cname].append(val)
    return {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"dispatches": len(next(iter(cs.values())))}
            for k, cs in acc.items()}


if __name__ == "__main__":
    all_k = "--all" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--all"]
    if args[0] == "stats":
        print(json.dumps(stats(args[1], all_k), indent=1))
    else:
        print(json.dumps(pmc(args[1:], all_k), indent=1))
