#!/usr/bin/env python3
"""Per-level launch table of the alignment loop from a rocprofv3 kernel trace of bench.py at its defaults:
kernel, pyramid level, grid (blocks x pairs), launches, average duration, algorithmic GB/s (10 B per pixel-iteration).
The level of a k_residual launch follows from its place in the schedule: behind each k_coarse launch (the coarsest level)
come `iters` launches per level, coarse to fine.

usage: per_level_table.py <..._kernel_trace.csv> [pairs=1024] [w=640] [h=480] [levels=4] [iters=10] > table.md
"""
import collections
import csv
import sys


def main(path, pairs=1024, w=640, h=480, levels=4, iters=10):
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    acc = collections.OrderedDict()
    lvl, left = None, 0
    for r in rows:
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        gx, gy = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])
        if "k_coarse" in name:
            lvl, left = levels - 1, 0
            key = ("k_coarse (one launch = %d evaluations)" % iters, levels - 1, "%d x 1" % gx, iters)
        elif "k_residual<" in name and lvl is not None:
            twin = name.rstrip(">(uwt::ResidualArgs)").endswith("true")
            if left == 0:
                lvl, left = lvl - 1, iters
            left -= 1
            key = ("k_residual" + (" (compute-only twin)" if twin else ""), lvl, "%d x %d" % (gx, gy), 1)
        else:
            continue
        a = acc.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += dur
    print("| kernel | level | grid (blocks x pairs) | launches in the trace | avg µs per launch | avg µs per evaluation | algorithmic GB/s |")
    print("|---|---|---|---|---|---|---|")
    for (name, l, grid, evals), (n, total) in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        if l < 0:
            continue
        per_eval = total / n / evals
        gbs = 10.0 * pairs * (w >> l) * (h >> l) / (per_eval * 1e-6) / 1e9
        print("| `%s` | %d | %s | %d | %.1f | %.1f | %.0f |" % (name, l, grid, n, total / n, per_eval, gbs))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], *[int(x) for x in a[1:]])
