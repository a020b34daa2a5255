#!/usr/bin/env python3
"""How many poses move when the SE(3) exponential takes this host's libm sinf / cosf (what Sophus calls, so3.hpp:538-558) instead of
the correctly rounded sine / cosine the oracle and the HIP library compute (oracle/uwt_oracle.h S5)?  CPU oracle only.
usage: trig_survey.py [pairs = 1000] [width = 160] [height = 96]"""
import importlib
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402

synth = importlib.import_module("uw-slam_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 160
h = int(sys.argv[3]) if len(sys.argv) > 3 else 96
f = 525.0 * w / 640.0
intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
SCHED = (("fixed 4 x 10", dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)), ("reference (4..1, early exit)", {}))


def one(s):
    ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=20000 + s, max_t=0.02, max_deg=1.5)
    out = []
    for _, over in SCHED:
        a = O.align_pair(O.default_params(w, h, *intr, trig=O.TRIG_ROUNDED, **over), ref, tgt, want_trace=True)
        b = O.align_pair(O.default_params(w, h, *intr, trig=O.TRIG_LIBM, **over), ref, tgt)
        d = float(np.abs(a[1].astype(np.float64) - b[1].astype(np.float64)).max())
        rot = max((float(2 * np.linalg.norm(t["delta"][3:])) for t in a[2] if not t["exited"]), default=0.0)
        out.append((not np.array_equal(a[1], b[1]), d, len(a[2]), rot))
    return out


with ThreadPoolExecutor(8) as ex:   # (the arithmetic travels with each call: threads with different settings do not interfere)
    res = list(ex.map(one, range(n)))
import platform
print("%d pairs of %d x %d, %s" % (n, w, h, " ".join(platform.libc_ver())))
for k, (name, _) in enumerate(SCHED):
    moved = sum(r[k][0] for r in res)
    evals = sum(r[k][2] for r in res)
    print("%-28s %d of %d poses differ (largest component difference %.3g); %d exponentials; largest rotation step |omega| %.3g rad"
          % (name + ":", moved, n, max(r[k][1] for r in res), evals, max(r[k][3] for r in res) / 2))
