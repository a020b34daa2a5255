#!/usr/bin/env python3
"""Approximate VGPR liveness along the main loop of a kernel in an assembly file (which region sets the register count).
usage: liveness.py <file.s> <mangled-name-substring> [step]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3uwt") and ":" in l and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        sp = (labels[m.group(1)], i)
        if best is None or sp[1] - sp[0] > best[1] - best[0]:
            best = sp
loop = [l for l in body[best[0]:best[1] + 1] if re.match(r"\s+[a-z]", l)]

def regs(t):
    r = set()
    for m in re.finditer(r"\bv(\d+)\b", t):
        r.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return r

n = len(loop)
reads, writes = [], []
for l in loop:
    op, _, rest = l.strip().partition(" ")
    ops = rest.split(",")
    dst = regs(ops[0]) if not op.startswith(("global_store", "ds_write", "s_", "v_cmp", "buffer_store", "ds_add")) else set()
    src = regs(",".join(ops[1:])) if dst else regs(rest)
    if "fmac" in op or "mac_" in op:
        src |= dst
    if op.startswith("global_load") or op.startswith("buffer_load"):
        src = regs(",".join(ops[1:]))
    reads.append(src); writes.append(dst)
# backward liveness over the loop treated as a cycle (two passes)
live = set()
out = [None] * n
for _ in range(2):
    for i in range(n - 1, -1, -1):
        live = (live - writes[i]) | reads[i]
        out[i] = len(live | writes[i])
print("loop instructions", n, "max live", max(out))
for i in range(0, n, step):
    print("%4d %4d  %s" % (i, out[i], loop[i].strip()[:90]))
