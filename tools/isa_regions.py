#!/usr/bin/env python3
"""Instruction counts per region of a kernel in the device assembly of a -DFA_MARKS build (falcon_amd/csrc/verify_kernel.hpp
FA_MARK). usage: tools/isa_regions.py <file.s> <kernel name substring>
Static counts in program order: a region runs from its mark to the next mark; the scheduler may move instructions across
marks, so the split is approximate, the total is not."""
import collections
import re
import sys

src, want = sys.argv[1], sys.argv[2]
text = open(src).read()
m = re.search(r"^(\S*" + re.escape(want) + r"\S*):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M)
if not m:
    sys.exit("kernel not found")
region = "prologue"
counts = collections.OrderedDict()
for line in m.group(2).splitlines():
    t = line.strip()
    k = re.match(r"; FA_MARK (\w+)", t)
    if k:
        region = k.group(1)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    counts.setdefault(region, collections.Counter())[kind] += 1
tot = collections.Counter()
for r, c in counts.items():
    print(f"{r:12s} " + " ".join(f"{k}={c[k]:4d}" for k in ("valu", "salu", "lds", "vmem", "other")))
    tot.update(c)
print(f"{'total':12s} " + " ".join(f"{k}={tot[k]:4d}" for k in ("valu", "salu", "lds", "vmem", "other")))
