#!/usr/bin/env python3
"""Make an instrumented COPY of the reference's src/Tracker.cpp: the same file with six UW_REF_DUMP_* lines inserted
inside Tracker::EstimatePose (src/Tracker.cpp:362-597) and one #include at the top.  The reference checkout itself is
not modified and nothing of it is stored in this repository: the insertion points are found by short patterns.

    python tools/ref_dump/instrument.py /path/to/uw-slam  [out_dir]

writes <out_dir>/Tracker_refdump.cpp (default out_dir: <reference>/build_refdump/).  Build it INSTEAD of src/Tracker.cpp
together with tools/ref_dump/ref_dump.cpp (README.md).
"""
import os
import re
import sys

# (pattern inside EstimatePose, line to insert AFTER the matching line)
HOOKS = [
    (r"^\s*error\s*=\s*errorMat\.at<float>\(0\s*,\s*0\)\s*;", "UW_REF_DUMP_EVAL(lvl, k, num_valid, Residuals, error);"),
    (r"^\s*if\s*\(\s*error\s*>=\s*last_error\b.*\{\s*$", "UW_REF_DUMP_EXIT();"),
    (r"^\s*deltaMat\s*=\s*A\.inv\(\)\s*\*\s*b\s*;", "UW_REF_DUMP_SOLVE(A, b, deltaMat);"),
    (r"^\s*current_pose\s*=\s*current_pose\s*\*\s*SE3::exp\(deltaVector\)\s*;", "UW_REF_DUMP_POSE(current_pose);"),
    (r"^\s*_previous_frame->rigid_transformation_\s*=\s*current_pose\s*;", "UW_REF_DUMP_FINAL(current_pose);"),
]
BEGIN = r"^\s*void\s+Tracker::EstimatePose\s*\("
END = r"^\s*Mat\s+Tracker::AddPatchPointsFeatures\s*\("     # the next definition in the file


def instrument(text):
    lines = text.split("\n")
    b = next((i for i, l in enumerate(lines) if re.search(BEGIN, l)), None)
    if b is None:
        raise SystemExit("Tracker::EstimatePose not found")
    e = next((i for i in range(b + 1, len(lines)) if re.search(END, lines[i])), len(lines))
    out = lines[:b]
    found = [0] * len(HOOKS)
    for i in range(b, e):
        out.append(lines[i])
        if lines[i].lstrip().startswith("//"):
            continue
        for h, (pat, ins) in enumerate(HOOKS):
            if re.search(pat, lines[i]):
                indent = re.match(r"\s*", lines[i]).group(0)
                out.append(indent + ins + "   // tools/ref_dump")
                found[h] += 1
    out += lines[e:]
    missing = [HOOKS[h][0] for h, n in enumerate(found) if n != 1]
    if missing:
        raise SystemExit("insertion points not found exactly once: %r" % missing)
    return '#include "ref_dump_hooks.h"   // tools/ref_dump\n' + "\n".join(out)


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    ref = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ref, "build_refdump")
    src = os.path.join(ref, "src", "Tracker.cpp")
    with open(src) as f:
        text = f.read()
    os.makedirs(out_dir, exist_ok=True)
    dst = os.path.join(out_dir, "Tracker_refdump.cpp")
    with open(dst, "w") as f:
        f.write(instrument(text))
    print("wrote", dst, "(%d hooks)" % len(HOOKS))


if __name__ == "__main__":
    main()
