#!/usr/bin/env python3
"""Timings of the rows next to the path: ObtainCandidatePoints (one frame, a batch of frames) and trajectory accumulation
(sequential kernel vs prefix scan), 640x480."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h, n = 640, 480, 64
intr = (525.0, 525.0, 319.5, 239.5)
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=n, max_pairs=1, has_depth=1))
ref, _, dep, _, _ = synth.render_pair(w, h, *intr, seed=1, with_depth=True)
ctx.upload_frames(0, np.stack([ref] * n), np.stack([dep] * n))
ctx.build_pyramids(0, n)
ctx.apply_gradient(0, n)
def timed(f, reps=20):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3
t1 = timed(lambda: ctx.obtain_candidate_points(0, 0, 20.0, cap=100000))
tb = timed(lambda: ctx.obtain_candidate_points_batch(0, n, 0, 20.0, cap=100000), 5)
tc = timed(lambda: ctx.obtain_candidate_points_batch(0, n, 0, 20.0, cap=256), 5)
print("ObtainCandidatePoints level 0: one frame %.3f ms per call (host round trip included); %d frames in one call %.3f ms = %.3f ms per frame "
      "with the points copied to pageable host memory, %.3f ms = %.4f ms per frame with 256 points per frame copied"
      % (t1, n, tb, tb / n, tc, tc / n))
rng = np.random.default_rng(0)
for m in (1000, 20000):
    q = rng.normal(0, 0.02, (m, 4)).astype(np.float32); q[:, 3] = 1.0
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([q, rng.normal(0, 0.01, (m, 3)).astype(np.float32)], axis=1)
    ts = timed(lambda: ctx.accumulate_trajectory(poses), 5)
    tp = timed(lambda: ctx.accumulate_trajectory(poses, scan=True), 5)
    print("trajectory of %d poses: sequential %.3f ms, scan %.3f ms" % (m, ts, tp))
