#!/usr/bin/env python3
"""VGPRs that are live THROUGH a kernel's densest f64 loop without being touched inside it (what a caller's state costs the loop).
usage: live_through.py <file.s> <mangled-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3uwt') and ':' in l and key in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end + 1]
labels = {re.match(r'^(\.LBB\d+_\d+):', l).group(1): i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
loops = []
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def regs(t):
    r = set()
    for m in re.finditer(r'\bv(\d+)\b', t): r.add(int(m.group(1)))
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', t): r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return r
def used(seg):
    s = set()
    for l in seg:
        if re.match(r'\s+[a-z]', l): s |= regs(l.split(';')[0])
    return s
a, b = max(loops, key=lambda ab: sum('v_fmac_f64' in l for l in body[ab[0]:ab[1]]) / (ab[1] - ab[0] + 1))
inner, allr, before, after = used(body[a:b + 1]), used(body), used(body[:a]), used(body[b + 1:])
lt = sorted(r for r in allr - inner if r in before and r in after)
print('densest loop: lines %d..%d, %d registers used inside, %d in the kernel, %d live through untouched' % (a, b, len(inner), len(allr), len(lt)))
for l in body[:a]:
    t = l.split(';')[0]
    m = re.match(r'\s+(\w+)\s+(v\d+|v\[\d+:\d+\])', t)
    if m and regs(m.group(2)) & set(lt): print('   ', t.strip()[:100])
