#!/usr/bin/env python3
"""Write the inputs of the reference-side dump (tools/ref_dump/README.md): every committed fixture
tests/golden/pair_*.npz that runs under the reference's own EstimatePose constants (5 levels, 4 -> 1, <= 50 iterations,
early exit — src/Tracker.cpp:364-372; an unmodified reference can run nothing else) becomes a directory

    <out>/<case>/ref.png  tgt.png  [depth.png, 16 bit]        and a line of <out>/cases.txt:
    <case> <w> <h> <fx> <fy> <cx> <cy> <has_depth>

PNG so that the reference reads them with the imread calls it uses on datasets (src/System.cpp:228, 243).

    python tools/ref_dump/export_inputs.py [out_dir]          (default: tools/ref_dump/inputs)
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def reference_constants(d):
    """True when the fixture was generated without overriding the reference's EstimatePose constants."""
    return len(d["over_keys"]) == 0


def main():
    from PIL import Image
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "ref_dump", "inputs")
    os.makedirs(out, exist_ok=True)
    lines = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "pair_*.npz"))):
        d = np.load(path)
        if not reference_constants(d):
            continue
        name = os.path.splitext(os.path.basename(path))[0]
        cdir = os.path.join(out, name)
        os.makedirs(cdir, exist_ok=True)
        Image.fromarray(d["ref"]).save(os.path.join(cdir, "ref.png"))
        Image.fromarray(d["tgt"]).save(os.path.join(cdir, "tgt.png"))
        has_depth = "depth" in d.files
        if has_depth:
            Image.fromarray(d["depth"]).save(os.path.join(cdir, "depth.png"))
        h, w = d["ref"].shape
        fx, fy, cx, cy = [float(v) for v in d["intr"]]
        lines.append("%s %d %d %r %r %r %r %d" % (name, w, h, fx, fy, cx, cy, int(has_depth)))
    with open(os.path.join(out, "cases.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote %d case(s) to %s" % (len(lines), out))


if __name__ == "__main__":
    main()
