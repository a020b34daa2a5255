#!/usr/bin/env python3
"""Kernel-resource table of libuwt_hip.so's device code: runs the library's `make asm` recipe (hipcc -S with
-Rpass-analysis=kernel-resource-usage) and prints one markdown row per kernel — VGPRs, SGPRs, scratch, LDS, waves/SIMD —
so that an occupancy regression shows up in a diff of profiles/rNN/kernel_resources.md.

usage: python tools/kernel_resources.py [-o profiles/r04/kernel_resources.md] [--filter k_residual]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--output")
    ap.add_argument("--filter", default="")
    ap.add_argument("--remarks", help="read remarks from this file instead of compiling")
    a = ap.parse_args()
    if a.remarks:
        text = open(a.remarks).read()
    else:
        text = ""
        for unit in ("uwt_capi", "uwt_launch_residual", "uwt_launch_general", "uwt_launch_flow"):   # one object per kernel family
            r = subprocess.run(["make", "-C", os.path.join(ROOT, "uw-slam_amd", "csrc"), "asm", "UNIT=" + unit], capture_output=True, text=True)
            text += r.stdout + r.stderr
            if r.returncode:
                sys.stderr.write(text[-4000:])
                return 1
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["pretty"] = re.sub(r"^void uwt::", "", n).split("(")[0] if n else r["name"]
    rows = [r for r in rows if a.filter in r["pretty"]]
    rows.sort(key=lambda r: r["pretty"])
    lines = ["| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/block | waves/SIMD |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %s | %s | %s | %s | %s | %s |" % (r["pretty"], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                               r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
    txt = "\n".join(lines) + "\n"
    if a.output:
        with open(a.output, "w") as f:
            f.write("# Kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n\n"
                    "Template arguments of `k_residual`: `<ARITH (0 OpenCV, 1 legacy), VEC, DEPTH, UNIT_FACTORS, DUMP, AccT, SQUARE, "
                    "SAMPLER, WEIGHTS, COMPUTE_ONLY, LOADS (bit 0: non-temporal plane loads, bit 1: typed plane loads)>`; of `k_iterate`: "
                    "`<ARITH, VEC, DEPTH, PLAIN, COMPUTE_ONLY, PASS>`; of `k_coarse` / `k_coarse_w4`: `<ARITH, DEPTH, PLAIN, ...>`.\n\n" + txt)
    else:
        sys.stdout.write(txt)
    return 0


if __name__ == "__main__":
    sys.exit(main())
