#!/usr/bin/env python3
"""Stage kernels alone (nothing else in flight): uwt_build_pyramids + uwt_apply_gradient over a resident batch of 640x480
frames, wall time per call and GB/s on the algorithmic bytes (Scharr: 1 B read + 4 B written per pixel per level).
usage: scharr_timing.py [path/to/lib.so]   (default: the in-tree library).  Run under rocprofv3 --kernel-trace --stats for
the per-kernel durations."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
synth = importlib.import_module("uw-slam_amd.synth")
w, h, n = 640, 480, 256
intr = (525.0, 525.0, 319.5, 239.5)
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=n, max_pairs=1, has_depth=1, n_levels=4, first_level=3, last_level=0))
ref, _, dep, _, _ = synth.render_pair(w, h, *intr, seed=1, with_depth=True)
ctx.upload_frames(0, np.stack([ref] * n), np.stack([dep] * n))
def timed(f, reps=20):
    f(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
tp = timed(lambda: ctx.build_pyramids(0, n))
tg = timed(lambda: ctx.apply_gradient(0, n))
px = sum(ctx.level_info(l).w * ctx.level_info(l).h for l in range(ctx.params.n_levels)) * n
print("%s: %d frames  build_pyramids %.3f ms  apply_gradient %.3f ms = %.0f GB/s on 5 B/px"
      % (os.path.basename(capi.LIB_PATH), n, tp, tg, 5.0 * px / tg / 1e6))
