#!/usr/bin/env python3
"""Per-frame latency of the live loop (System::AddFrame + System::Tracking, src/System.cpp:193-251): a new 640x480 frame
(grey + depth) arrives in host memory, its pyramid is built, the previous frame's gradients are taken, the pair is aligned.
(a) the stage calls of the Tracker mirror, one synchronous call each; (b) asynchronous upload from page-locked memory +
uwt_track_batch_host_async over the two slots + one wait."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h = 640, 480
intr = (525.0, 525.0, 319.5, 239.5)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=3, z=1.0, with_depth=True)
frames = [ref, tgt]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, over in (("fixed 4x10", dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0, has_depth=1)),
                   ("reference", dict(has_depth=1))):
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
    ctx.upload_frames(0, ref[None], dep[None]); ctx.build_pyramids(0, 1)
    def frame_a(i):
        cur, prev = i % 2, (i + 1) % 2
        ctx.upload_frames(cur, frames[cur][None], dep[None])
        ctx.build_pyramids(cur, 1)
        ctx.apply_gradient(prev, 1)
        return ctx.estimate_pose_batch([prev], [cur])
    for deferred in (False, True):
        ctx.set_deferred(deferred)
        for i in range(1, 11): frame_a(i)
        t0 = time.perf_counter()
        for i in range(1, reps + 1): p, s = frame_a(i)
        dt = (time.perf_counter() - t0) / reps
        print("%-11s (a) stage calls%s: %.3f ms per frame (%d evaluations)" % (name, ", deferred" if deferred else "          ", dt * 1e3, s[0]["iterations"]), flush=True)
    ctx.set_deferred(False)
    # the parts
    parts = {}
    for lab, fn in (("upload", lambda: ctx.upload_frames(0, ref[None], dep[None])), ("pyramid", lambda: ctx.build_pyramids(0, 1)),
                    ("gradient", lambda: ctx.apply_gradient(0, 1)), ("estimate", lambda: ctx.estimate_pose_batch([1], [0]))):
        for _ in range(5): fn()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        parts[lab] = (time.perf_counter() - t0) / reps * 1e3
    print("            parts: " + ", ".join("%s %.3f" % kv for kv in parts.items()), flush=True)
    g = [capi.pinned_empty((1, h, w), np.uint8) for _ in range(2)]
    d = capi.pinned_empty((1, h, w), np.uint16); d[0] = dep
    g[0][0] = ref; g[1][0] = tgt
    hp = capi.pinned_empty((1, 7), np.float32)
    def frame_b(i):
        cur, prev = i % 2, (i + 1) % 2
        ctx.upload_frames_async(cur, g[cur], d)
        t = ctx.track_batch_host_async(0, 2, [prev], [cur], hp)
        ctx.wait_ticket(t)
    for i in range(1, 11): frame_b(i)
    t0 = time.perf_counter()
    for i in range(1, reps + 1): frame_b(i)
    dt = (time.perf_counter() - t0) / reps
    print("%-11s (b) async + one wait: %.3f ms per frame" % (name, dt * 1e3), flush=True)
    ctx.close()
