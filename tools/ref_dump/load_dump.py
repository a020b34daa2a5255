#!/usr/bin/env python3
"""Turn the output directory of the reference-side ref_dump run into tests/golden/ref_<case>.npz, the files
tests/test_ref_vectors.py consumes (it skips while none exists).

    python tools/ref_dump/load_dump.py <out_dir of ref_dump> [golden_dir]

Per case: the per-evaluation records of Tracker::EstimatePose (level, iter, n_valid, sum_r2, error, exited, and — where the
evaluation was followed by an update — A, b, delta, pose, and delta_unfolded = the same solve through two statements), the
final pose, the test pose WarpFunction was called with, the fold probe (foldprobe = r20 r21 r22 x y z lo hi out), and the stage arrays per level (img, tgt, dep, gx, gy, pts, warp,
unpx / unpy = WarpFunction at the two axis permutations, whose column 2 is the unprojected X * z / Y * z).
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAGE_TYPES = {"u8": np.uint8, "u16": np.uint16, "i16": np.int16, "f32": np.float32}


def hexf(tokens):
    return np.array([float.fromhex(t) for t in tokens], np.float64).astype(np.float32)


def parse_dump(path):
    """dict: case name -> dict of arrays (see module docstring)."""
    cases, cur = {}, None
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "case":
                cur = dict(rows=[], final=None, testpose=None, foldprobe=None)
                cases[t[1]] = cur
            elif t[0] == "testpose":
                cur["testpose"] = hexf(t[1:8])
            elif t[0] == "foldprobe":
                assert t[1] == "row" and t[5] == "pt" and t[9] == "lo" and t[11] == "hi" and t[13] == "out", line[:80]
                cur["foldprobe"] = hexf(t[2:5] + t[6:9] + [t[10], t[12], t[14]])   # r20 r21 r22 x y z lo hi out
            elif t[0] == "eval":
                cur["rows"].append(dict(level=int(t[1]), iter=int(t[2]), n_valid=int(t[3]), sum_r2=int(round(float(t[4]))),
                                        error=hexf(t[5:6])[0], exited=0, A=np.zeros(36, np.float32), b=np.zeros(6, np.float32),
                                        delta=np.zeros(6, np.float32), pose=np.zeros(7, np.float32), updated=0,
                                        delta_unfolded=np.zeros(6, np.float32)))
            elif t[0] == "exit":
                cur["rows"][-1]["exited"] = 1
            elif t[0] == "solve":
                assert t[1] == "A" and t[38] == "b" and t[45] == "delta", line[:80]
                r = cur["rows"][-1]
                r["A"], r["b"], r["delta"], r["updated"] = hexf(t[2:38]), hexf(t[39:45]), hexf(t[46:52]), 1
            elif t[0] == "solve2":
                assert t[1] == "delta", line[:80]
                cur["rows"][-1]["delta_unfolded"] = hexf(t[2:8])
            elif t[0] == "pose":
                cur["rows"][-1]["pose"] = hexf(t[1:8])
            elif t[0] == "final":
                cur["final"] = hexf(t[1:8])
    out = {}
    for name, c in cases.items():
        rows = c["rows"]
        d = {k: np.array([r[k] for r in rows]) for k in ("level", "iter", "n_valid", "sum_r2", "error", "exited", "updated")}
        for k in ("A", "b", "delta", "pose", "delta_unfolded"):
            d[k] = np.stack([r[k] for r in rows]) if rows else np.zeros((0,), np.float32)
        d["A"] = d["A"].reshape(-1, 6, 6) if rows else d["A"]
        d["final"] = c["final"]
        d["testpose"] = c["testpose"]
        d["foldprobe"] = c["foldprobe"]
        out[name] = d
    return out


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = sys.argv[1]
    golden = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden")
    for name, d in parse_dump(os.path.join(src, "dump.txt")).items():
        for path in sorted(glob.glob(os.path.join(src, name + "_*.*"))):
            stem, ext = os.path.splitext(os.path.basename(path))
            if ext[1:] in STAGE_TYPES:
                d["stage_" + stem[len(name) + 1:]] = np.fromfile(path, STAGE_TYPES[ext[1:]])
        dst = os.path.join(golden, "ref_" + name + ".npz")
        np.savez_compressed(dst, **{k: v for k, v in d.items() if v is not None})
        print("wrote", dst, "(%d evaluations)" % len(d["level"]))


if __name__ == "__main__":
    main()
