#!/usr/bin/env python3
"""Per-call kernel timeline of the single-pair call from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/exp/latency.py 50 ; python tools/exp/latency_trace.py <dir>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# calls end with k_finish
calls, cur = [], []
for r in rows:
    if "k_iterate" in r[0] or "k_finish" in r[0]:
        cur.append(r)
        if "k_finish" in r[0]:
            calls.append(cur)
            cur = []
by_len = {}
for c in calls:
    by_len.setdefault(len(c), []).append(c)
for n, cs in sorted(by_len.items()):
    cs = cs[len(cs) // 2:]     # the later half: warm
    span = np.array([c[-1][2] - c[0][1] for c in cs]) / 1e3
    busy = np.array([sum(e - s for _, s, e in c) for c in cs]) / 1e3
    gaps = np.array([np.mean([c[i + 1][1] - c[i][2] for i in range(len(c) - 1)]) for c in cs]) / 1e3
    period = np.array([c[i + 1][1] - c[i][1] for c in cs for i in range(len(c) - 1)]) / 1e3
    nxt = np.array([cs[i + 1][0][1] - cs[i][-1][2] for i in range(len(cs) - 1) if cs[i + 1][0][1] > cs[i][-1][2]]) / 1e3
    print("%d launches per call, %d calls: span %.1f us, kernels %.1f us, mean gap %.2f us, launch period median %.2f us, between calls (last end -> next first start) median %.1f us"
          % (n, len(cs), np.median(span), np.median(busy), np.median(gaps), np.median(period), np.median(nxt) if len(nxt) else -1))
    durs = np.array([[e - s for _, s, e in c] for c in cs]) / 1e3
    print("   kernel durations by position (us):", " ".join("%.1f" % x for x in np.median(durs, 0)))
