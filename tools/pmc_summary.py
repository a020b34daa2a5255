#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv files) per kernel and grid size.

usage: pmc_summary.py <dir with */*counter_collection.csv> <out.csv>

One row per (kernel, grid, counter): mean counter value per dispatch (FETCH_SIZE / WRITE_SIZE are in KiB on gfx950; the
read counter needs the x2 correction of MI355X_MICROARCH.md §HBM) and the number of dispatches averaged.
"""
import collections
import csv
import glob
import sys


def main(src, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))  # (kernel, grid, counter) -> dispatch -> value
    for f in sorted(glob.glob(src + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if len(name) > 60:
                name = name[:24] + ".." + name[-34:]
            key = (name, int(r["Grid_Size"]), r["Counter_Name"])
            acc[key][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])  # a counter is reported once per XCD / SE: sum
    with open(dst, "w") as o:
        o.write("kernel,grid,counter,mean_per_dispatch,dispatches\n")
        for (name, grid, ctr), per in sorted(acc.items()):
            vals = list(per.values())
            o.write('"%s",%d,%s,%.3f,%d\n' % (name, grid, ctr, sum(vals) / len(vals), len(vals)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
