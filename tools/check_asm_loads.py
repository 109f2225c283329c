#!/usr/bin/env python3
"""Build check for the loads written as instructions (falcon_amd/csrc/verify_kernel.hpp prefetch_s2). The compiler does
not know that their target registers are the targets of loads in flight, so between such a load (marked FA_ASM_LOAD in
the device assembly) and the wait that follows it (marked FA_ASM_LOAD_WAIT) nothing may name the register: a copy made
there would read it too early, a reuse would be overwritten when the load lands.

The check follows the control flow: the device assembly of every function is cut into basic blocks (labels, s_branch,
s_cbranch_*, s_endpgm), the set of registers with a load in flight is propagated along the edges to a fixed point, and
every instruction that names a register of the set at that point is reported.
usage: tools/check_asm_loads.py <file.s> [...]   exit status 1 on findings or if no marked load was seen."""
import re
import sys


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def functions(path):
    """(name, [(line number, text)]) for every function of the file."""
    name, body = None, []
    for n, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.strip()
        m = re.match(r"^([A-Za-z_$][\w$.]*):", line)
        if m:
            if name and body:
                yield name, body
            name, body = m.group(1), []
            continue
        if line.startswith(".Lfunc_end"):
            if name and body:
                yield name, body
            name, body = None, []
            continue
        if name is not None and line:
            body.append((n, line))
    if name and body:
        yield name, body


def check_function(path, name, body):
    if not any("FA_ASM_LOAD" in t for _, t in body):
        return 0, []
    # basic blocks
    blocks, labels, cur = [], {}, []
    for n, t in body:
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            labels[m.group(1)] = len(blocks)
            continue
        if t.startswith((";;", ".", "//")) or (t.startswith(";") and "FA_ASM" not in t):
            continue
        cur.append((n, t))
        if re.match(r"^(s_branch|s_cbranch_\w+|s_endpgm)\b", t):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1][1] if b else ""
        m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)", last)
        if last.startswith("s_endpgm"):
            succ.append([])
        elif m and m.group(1) == "s_branch":
            succ.append([labels[m.group(2)]])
        elif m:
            succ.append([labels[m.group(2)]] + ([i + 1] if i + 1 < len(blocks) else []))
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])

    def run(i, pending, report):
        pending, loads = dict(pending), 0
        for n, t in blocks[i]:
            code = t.split(";")[0]
            if "FA_ASM_LOAD_WAIT" in t:
                pending = {}
            elif "FA_ASM_LOAD" in t:
                m = re.match(r"global_load_ushort v(\d+),(.*)", code)
                if not m:
                    report.append(f"{path}:{n}: marked load not understood: {t}")
                    continue
                loads += 1
                r = int(m.group(1))
                if r in pending and pending[r] != n:
                    report.append(f"{path}:{n}: {name}: v{r} is loaded again while its load of line {pending[r]} is in flight")
                for a in sorted(regs_of(m.group(2)) & set(pending)):
                    report.append(f"{path}:{n}: {name}: the address uses v{a}, whose load (line {pending[a]}) is in flight")
                pending[r] = n
            else:
                for a in sorted(regs_of(code) & set(pending)):
                    report.append(f"{path}:{n}: {name}: v{a} named while its load (line {pending[a]}) is in flight: {t}")
        return pending, loads

    ins = [dict() for _ in blocks]
    changed, rounds = True, 0
    while changed and rounds < 1000:
        changed, rounds = False, rounds + 1
        for i in range(len(blocks)):
            out, _ = run(i, ins[i], [])
            for s in succ[i]:
                for r, n in out.items():
                    if r not in ins[s]:
                        ins[s][r] = n
                        changed = True
    report, loads = [], 0
    for i in range(len(blocks)):
        _, k = run(i, ins[i], report)
        loads += k
    return loads, sorted(set(report))


def main():
    total, bad = 0, []
    for p in sys.argv[1:]:
        for name, body in functions(p):
            n, b = check_function(p, name, body)
            total += n
            bad += b
    for b in bad[:40]:
        print(b)
    print(f"check_asm_loads: {total} marked loads, {len(bad)} findings")
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
