#!/usr/bin/env python3
"""The pre-processing stages alone on a resident batch (640x480, 4 levels, 2048 frames = the default step's): wall time per
call and the bytes they move per second.  pyramids: u8 of every frame + u16 of every frame here (the step builds the depth
pyramids of the 1024 reference frames only); gradients: all four levels of 1024 frames."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
w, h, F = 640, 480, 2048
ctx = capi.Context(capi.default_params(w, h, 525.0, 525.0, 319.5, 239.5, n_levels=4, first_level=3, last_level=0, max_frames=F, max_pairs=F // 2, has_depth=1))
rng = np.random.default_rng(1)
blk = rng.integers(0, 256, (128, h, w)).astype(np.uint8)
dblk = rng.integers(0, 65536, (128, h, w)).astype(np.uint16)
for s in range(0, F, 128):
    ctx.upload_frames(s, blk, dblk)
ctx.set_deferred(True)
px = sum((w >> l) * (h >> l) for l in range(4))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
t = timeit(lambda: ctx.build_pyramids(0, F))
byt = F * (w * h * 3 + (px - w * h) * 3)          # u8 + u16: level 0 read, levels 1..3 written
print("pyramids (u8 + u16) of %d frames: %.3f ms, %.2f TB/s" % (F, t * 1e3, byt / t / 1e12))
t = timeit(lambda: ctx.apply_gradient(0, F // 2))
byt = (F // 2) * px * 5
print("gradients of %d frames, 4 levels:  %.3f ms, %.2f TB/s" % (F // 2, t * 1e3, byt / t / 1e12))
