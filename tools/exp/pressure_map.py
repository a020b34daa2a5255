#!/usr/bin/env python3
"""Highest VGPR index in use per stretch of a kernel's assembly (straight-line reading: where does the register count come from?).
usage: pressure_map.py <file.s> <mangled-name-substring> [stretch=60]"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN3uwt') and ':' in l and key in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end + 1]
def regs(t):
    r = set()
    for m in re.finditer(r'\bv(\d+)\b', t): r.add(int(m.group(1)))
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', t): r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return r
for a in range(0, len(body), step):
    seg = body[a:a + step]
    used = set()
    for l in seg:
        if re.match(r'\s+[a-z]', l): used |= regs(l.split(';')[0])
    marks = [l.split(':')[0] for l in seg if re.match(r'^\.LBB', l)]
    loops = [l for l in seg if 'Loop Header' in l]
    print('%5d  max v%-4d distinct %-4d %s %s' % (a, max(used) if used else -1, len(used), 'LOOP-HEADER' if loops else '', ' '.join(marks[:4])))
