#!/usr/bin/env python3
"""Safety check of the TYPED kernels' hand-written vector-memory operations (uwt_kernels.h: load_group_typed): the compiler does
not track loads issued from asm statements, so between such a load and the s_waitcnt that releases it NO instruction may read
or write the load's destination registers.  Scans a kernel of a .s file: for every vector load, every instruction up to the
first s_waitcnt vmcnt(N) that covers it (in-order retirement: a wait for vmcnt <= N releases a load once at most N younger
loads have been issued) must not mention its destination registers.
usage: check_asm_loads.py <file.s> <mangled-name-substring>"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3uwt") and ":" in l and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.split(";")[0].rstrip() for l in lines[start:end + 1]]
ins = [(i, l.strip()) for i, l in enumerate(body) if l.startswith("\t") and not l.strip().startswith(".") and l.strip()]


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def mentioned(text):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", text):
        out |= regs(tok)
    return out


bad = 0
loads = 0
for k, (i, text) in enumerate(ins):
    # the hand-written ones: the typed plane loads, the byte gathers, and the intensities' dword load in front of a typed load
    mine = re.match(r"(tbuffer_load|global_load_ubyte|buffer_load_ubyte)", text) or (
        re.match(r"global_load_dword v\d+, v\d+, s\[", text) and any(t.startswith("tbuffer_load") for _, t in ins[k + 1:k + 4]))
    if not mine:
        continue
    dst = regs(text.split()[1].rstrip(","))
    loads += 1
    younger = 0
    released = False
    for j, t in ins[k + 1:]:
        if re.match(r"(tbuffer_load|global_load|buffer_load|global_store|buffer_store|global_atomic)", t):
            younger += 1
            continue
        m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
        if m and int(m.group(1)) <= younger:
            released = True
            break
        if re.match(r"s_(c?branch|endpgm|setpc)", t):
            # control flow: follow neither edge further than the straight line (the kernels' loops wait at their heads)
            continue
        if mentioned(t) & dst:
            print("line %d: `%s` touches %s of the load at line %d `%s` before it is released" % (j, t, sorted(mentioned(t) & dst), i, text[:60]))
            bad += 1
    if not released:
        pass   # (the load is released by a wait further along a path this straight-line scan does not follow)
print("%d vector loads checked, %d violations" % (loads, bad))
sys.exit(1 if bad else 0)
