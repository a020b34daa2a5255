#!/usr/bin/env python3
"""Where the fixed cost of a one-pair call goes (640 x 480, depth, level 0 alone, ONE evaluation = k_iterate + k_finish): the Python
binding, the C call with prebuilt arrays, and a C call that does nothing on the GPU (uwt_get_params)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h = 640, 480
intr = (525.0, 525.0, 319.5, 239.5)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=3, z=1.0, with_depth=True)
reps = 2000
for name, over in (("1 evaluation (level 0)", dict(n_levels=4, first_level=0, last_level=0, max_iters=1, early_exit=0)),
                   ("10 evaluations (level 0)", dict(n_levels=4, first_level=0, last_level=0, max_iters=10, early_exit=0)),
                   ("reference schedule", dict())):
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, has_depth=1, **over))
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]))
    ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
    for _ in range(20): ctx.estimate_pose_batch([0], [1])
    t0 = time.perf_counter()
    for _ in range(reps): ctx.estimate_pose_batch([0], [1])
    t_py = (time.perf_counter() - t0) / reps
    r = np.array([0], np.int32); t = np.array([1], np.int32); poses = np.empty((1, 7), np.float32); st = (capi.Stats * 1)()
    L = capi.lib(); H = ctx._h
    rp, tp, pp = r.ctypes.data_as(C.POINTER(C.c_int32)), t.ctypes.data_as(C.POINTER(C.c_int32)), poses.ctypes.data_as(C.POINTER(C.c_float))
    t0 = time.perf_counter()
    for _ in range(reps): L.uwt_estimate_pose_batch(H, 1, rp, tp, pp, st)
    t_c = (time.perf_counter() - t0) / reps
    p = capi.Params()
    t0 = time.perf_counter()
    for _ in range(reps): L.uwt_get_params(H, C.byref(p))
    t_n = (time.perf_counter() - t0) / reps
    print("%-26s binding %.1f us, C call %.1f us, empty C call %.2f us" % (name, t_py * 1e6, t_c * 1e6, t_n * 1e6), flush=True)
    ctx.close()
