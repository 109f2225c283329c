#!/bin/bash
# usage: [FA_AB_DRIVER=tools/prof_prepared.py] tools/ab_cycles.sh "<driver args>" <lib.so> [<lib.so> ...]   (default driver: tools/prof_run.py)
# A/B of library builds in GPU CYCLES (GRBM_GUI_ACTIVE / 8 per dispatch) and issued vector instructions per kernel: the
# clock moves by several per cent between runs and boxes, cycles do not. One rocprofv3 --pmc pass per library and round.
set -uo pipefail
args="$1"; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/abc
i=0
for lib in "$@"; do
  i=$((i+1)); out="gpurun_out/abc/r$i"; rm -rf "$out"
  FALCON_AMD_LIB="$PWD/$lib" timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU -d "$out" -o pmc --output-format csv -- python3 ${FA_AB_DRIVER:-tools/prof_run.py} $args > "$out.log" 2>&1 || { echo "pass failed for $lib"; tail -3 "$out.log"; exit 1; }
  python3 - "$lib" "$out" <<'PY'
import csv, glob, sys, collections
lib, out = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void fa::", "").replace("fa::", "")
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
parts = []
for k in sorted(rows):
    if not any(t in k for t in ("verify_kernel", "hash_kernel", "decode_fast", "polymul")):
        continue
    c = sorted(rows[k].get("GRBM_GUI_ACTIVE", [0]))
    v = sorted(rows[k].get("SQ_INSTS_VALU", [0]))
    parts.append(f"{k[:28]}: {c[len(c)//2]/8/1e6:.3f} Mcyc, {v[len(v)//2]/1e6:.1f} Minstr")
print(lib, "|", " | ".join(parts))
PY
done
