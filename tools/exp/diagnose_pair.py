#!/usr/bin/env python3
"""Find the seeded pairs of a parity-survey configuration whose GPU pose is not the oracle's bit for bit, and say where the two
part: the oracle's trace is replayed — at every evaluation's input pose the GPU's sums (the dump-capable stage entry points) are
compared with the oracle's A and b after the f32 rounding.
usage: diagnose_pair.py [--n 150 --w 320 --h 240 --depth 1 --weights 2 --sampler 0 --arith opencv --seed0 1000]"""
import importlib, os, sys, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
ap = argparse.ArgumentParser()
for k, v in dict(n=150, w=320, h=240, depth=1, weights=2, sampler=0, seed0=1000).items():
    ap.add_argument("--" + k, type=int, default=v)
ap.add_argument("--arith", default="opencv")
a = ap.parse_args()
AR = {"opencv": 0, "legacy": 1}[a.arith]
capi.DEFAULT_ARITH = AR; O.DEFAULT_ARITH = AR
w, h = a.w, a.h
f = 525.0 * w / 640.0
intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
if a.depth: over["has_depth"] = 1
if a.weights: over["weights"] = a.weights
if a.sampler: over["sampler"] = a.sampler
n = a.n
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
po = O.default_params(w, h, *intr, **over)
pairs = [synth.render_pair(w, h, *intr, seed=a.seed0 + s, z=1.0 + 0.2 * ((s % 5) - 2) / 2, with_depth=bool(a.depth)) for s in range(n)]
frames, depths = [], []
for ref, tgt, dep, _, _ in pairs:
    frames += [ref, tgt]
    if a.depth: depths += [dep, dep]
ctx.upload_frames(0, np.stack(frames), np.stack(depths) if a.depth else None)
ctx.build_pyramids(0, 2 * n); ctx.apply_gradient(0, 2 * n)
poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1)
bad = []
for i in range(n):
    ref, tgt, dep, _, _ = pairs[i]
    st, pose, tr = O.align_pair(po, ref, tgt, dep if a.depth else None, want_trace=True)
    if not np.array_equal(poses[i].view(np.uint32), pose.view(np.uint32)):
        bad.append((i, pose, tr))
print("pairs not bit-identical:", [b[0] for b in bad])
for i, pose, tr in bad:
    print("pair", i, "seed", a.seed0 + i, "gpu", poses[i], "cpu", pose, "ulps", (poses[i].view(np.int32) - pose.view(np.int32)))
    # single-pair call too
    p1, _ = ctx.estimate_pose_batch([2 * i], [2 * i + 1])
    print("  one pair per call identical to the batch:", np.array_equal(p1[0].view(np.uint32), poses[i].view(np.uint32)),
          " to the oracle:", np.array_equal(p1[0].view(np.uint32), pose.view(np.uint32)))
    pose_in = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    last_level = None
    fn = ctx.residual_jacobian_weighted if (a.weights or a.sampler) else (lambda r_, t_, l_, p_: ctx.residual_jacobian(r_, t_, l_, p_, dump=False))
    for t in tr:
        lvl = t["level"]
        if last_level is not None and lvl != last_level:
            pose_in = O.se3_handoff(pose_in, 0)     # the hand-off between levels (src/Tracker.cpp:580-590)
            pose_in = pose_in[0] if isinstance(pose_in, tuple) else pose_in
        last_level = lvl
        g = fn(2 * i, 2 * i + 1, lvl, pose_in)
        A32 = g["A"].astype(np.float32); b32 = (-g["jtr"]).astype(np.float32)
        if not (a.weights or a.sampler):
            b32 = (-(po.gain * g["jtr"])).astype(np.float32)
        dA = int(np.abs(A32.view(np.int32).astype(np.int64) - np.asarray(t["A"], np.float32).reshape(6, 6).view(np.int32)).max())
        db = int(np.abs(b32.view(np.int32).astype(np.int64) - np.asarray(t["b"], np.float32).view(np.int32)).max())
        if dA or db or g["n_valid"] != t["n_valid"]:
            print("  level %d iteration %d: A differs by %d ulps, b by %d ulps, n_valid %d / %d" % (lvl, t["iter"], dA, db, g["n_valid"], t["n_valid"]))
            if dA:
                k = np.argmax(np.abs(A32.view(np.int32).astype(np.int64) - np.asarray(t["A"], np.float32).reshape(6, 6).view(np.int32)))
                print("    A[%d] gpu f64 %.17g -> %r, cpu %r" % (k, g["A"].ravel()[k], A32.ravel()[k], np.asarray(t["A"], np.float32).ravel()[k]))
            if db:
                k = int(np.argmax(np.abs(b32.view(np.int32).astype(np.int64) - np.asarray(t["b"], np.float32).view(np.int32))))
                print("    b[%d] gpu f64 %.17g -> %r, cpu %r" % (k, -g["jtr"][k], b32[k], np.asarray(t["b"], np.float32)[k]))
        pose_in = np.asarray(t["pose"], np.float32)
ctx.close()
