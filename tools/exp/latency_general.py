#!/usr/bin/env python3
"""Single-pair latency of the drop-in call on the general path (robust weights / bilinear sampler) next to the identity path:
640x480 with depth, fixed 4 x 10 schedule and the reference schedule."""
import importlib, sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h = 640, 480
intr = (525.0, 525.0, 319.5, 239.5)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=3, z=1.0, with_depth=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if kv}   # uwt_tuning fields, k=v[,k=v]
for sched, over in (("fixed 4x10", dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0, has_depth=1)),
                    ("reference", dict(has_depth=1))):
    for name, gen in (("identity", {}), ("huber", dict(weights=2)), ("tukey", dict(weights=1)), ("bilinear+huber", dict(weights=2, sampler=1))):
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over, **gen), tuning=tuning or None)
        ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]))
        ctx.build_pyramids(0, 2)
        ctx.apply_gradient(0, 2)
        for _ in range(10):
            ctx.estimate_pose_batch([0], [1])
        t0 = time.perf_counter()
        for _ in range(reps):
            p, s = ctx.estimate_pose_batch([0], [1])
        dt = (time.perf_counter() - t0) / reps
        print("%-11s %-15s %.3f ms per call, %d evaluations" % (sched, name, dt * 1e3, s[0]["iterations"]), flush=True)
        ctx.close()
