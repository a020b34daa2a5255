#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes for one kernel: per-dispatch counter sums over the dispatches with the largest grid
whose kernel name contains <substr>, with per-wave and per-(wave x pixel-per-lane) figures.

usage: sq_summary.py <dir with */*counter_collection.csv> <substr> <pixels_per_dispatch> [out.csv]

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md, constants table).
"""
import collections
import csv
import glob
import sys


def main(src, substr, pixels, dst=None):
    rows = []
    for f in sorted(glob.glob(src + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if substr in r["Kernel_Name"]:
                rows.append((f, r))
    if not rows:
        raise SystemExit("no dispatch of a kernel containing %r" % substr)
    grid = max(int(r["Grid_Size"]) for _, r in rows)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for f, r in rows:
        if int(r["Grid_Size"]) != grid:
            continue
        acc[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
        dur[(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    waves = grid / 64.0
    px_per_lane = float(pixels) / grid
    out = ["# kernel *%s*, grid %d threads (%d waves), %.1f pixels per lane; mean dispatch %.1f us over %d dispatches"
           % (substr, grid, waves, px_per_lane, sum(dur.values()) / len(dur), len(dur)),
           "counter,mean_per_dispatch,per_wave,per_wave_pixel"]
    for name in sorted(acc):
        vals = list(acc[name].values())
        m = sum(vals) / len(vals)
        out.append("%s,%.6g,%.2f,%.3f" % (name, m, m / waves, m / waves / px_per_lane))
    text = "\n".join(out) + "\n"
    sys.stdout.write(text)
    if dst:
        open(dst, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
