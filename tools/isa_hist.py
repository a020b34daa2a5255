#!/usr/bin/env python3
"""Instruction histogram of one kernel's main loop in uw-slam_amd/csrc/uwt_capi.gfx950.s (`make -C uw-slam_amd/csrc asm`).
usage: isa_hist.py <mangled-name-substring> [--loop | --f64loop]   (--loop: only the largest backward-branch loop body; --f64loop: the innermost loop with the f64 sums)"""
import collections
import re
import sys

import os
path = os.environ.get("ISA_FILE", "uw-slam_amd/csrc/uwt_capi.gfx950.s")
key = sys.argv[1]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3uwt") and ":" in l and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
if "--loop" in sys.argv or "--f64loop" in sys.argv:
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if "--f64loop" in sys.argv:   # the innermost loop that holds the f64 sums (kernels with other large loops around)
                n = sum("v_fmac_f64" in x for x in body[span[0]:span[1]])
                if n >= 100 and (best is None or span[1] - span[0] < best[1] - best[0]):
                    best = span
            elif best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    body = body[best[0]:best[1] + 1]
    print("loop: %d lines" % len(body))
hist = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
    if m and not l.strip().startswith((".", ";")):
        hist[m.group(1)] += 1
def cls(op):
    if op.startswith("v_pk_"): return "valu packed"
    if op.endswith("_f64") or "f64" in op: return "valu f64/cvt64"
    if op.startswith("v_cvt"): return "valu cvt"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"): return "valu cmp/sel"
    if op.startswith("v_"): return "valu other"
    if op.startswith("s_"): return "scalar"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"
byc = collections.Counter()
for op, n in hist.items():
    byc[cls(op)] += n
for c, n in byc.most_common():
    print("%-16s %5d" % (c, n))
print("valu total", sum(n for c, n in byc.items() if c.startswith("valu")))
for op, n in hist.most_common(60):
    print("  %-28s %4d" % (op, n))
