#!/usr/bin/env python3
"""One pair per call, identity weights, 640 x 480 with depth: the fixed 4 x 10 schedule, the same on level 0 alone / level 1 alone /
level 2 alone (10 evaluations each: what a fine-level evaluation costs at each size) and the reference schedule.  usage: latency_identity.py [reps]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h = 640, 480
intr = (525.0, 525.0, 319.5, 239.5)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=3, z=1.0, with_depth=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = [("fixed 4x10 (levels 3..0)", dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)),
         ("level 0 alone x10", dict(n_levels=4, first_level=0, last_level=0, max_iters=10, early_exit=0)),
         ("level 1 alone x10", dict(n_levels=4, first_level=1, last_level=1, max_iters=10, early_exit=0)),
         ("level 2 alone x10", dict(n_levels=4, first_level=2, last_level=2, max_iters=10, early_exit=0)),
         ("level 3 alone x10 (one block)", dict(n_levels=4, first_level=3, last_level=3, max_iters=10, early_exit=0)),
         ("level 0 alone x1", dict(n_levels=4, first_level=0, last_level=0, max_iters=1, early_exit=0)),
         ("reference schedule", dict())]
for name, over in cases:
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, has_depth=1, **over))
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]))
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)
    for _ in range(10):
        ctx.estimate_pose_batch([0], [1])
    t0 = time.perf_counter()
    for _ in range(reps):
        p, s = ctx.estimate_pose_batch([0], [1])
    dt = (time.perf_counter() - t0) / reps
    print("%-30s %8.1f us per call, %2d evaluations" % (name, dt * 1e6, s[0]["iterations"]), flush=True)
    ctx.close()
