#!/usr/bin/env python3
"""One seed of tests/test_gpu_parity.py::test_fuzz_whole_alignment_bit_identical replayed: where do the GPU's pose and the
oracle's part?  The oracle's trace is followed; at every evaluation's input pose the GPU's sums (stage entry points) are compared
with the oracle's A and b after the f32 rounding.   usage: diagnose_fuzz.py <seed> [opencv|legacy]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
seed = int(sys.argv[1])
AR = {"opencv": 0, "legacy": 1}[sys.argv[2] if len(sys.argv) > 2 else "opencv"]
capi.DEFAULT_ARITH = AR; O.DEFAULT_ARITH = AR
rng = np.random.default_rng(91000 + seed)      # the test's draw, line for line
w = int(rng.choice([64, 96, 112, 160, 208])); h = int(rng.choice([32, 48, 64, 96]))
n_levels = int(rng.integers(2, 5))
while (w >> (n_levels - 1)) < 8 or (h >> (n_levels - 1)) < 4:
    n_levels -= 1
first = int(rng.integers(0, n_levels)); last = int(rng.integers(0, first + 1))
fx = float(np.float32(rng.uniform(0.7, 1.3) * w))
fy = fx if rng.random() < 0.5 else float(np.float32(fx * rng.uniform(0.95, 1.05)))
intr = (fx, fy, float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
depth = bool(rng.random() < 0.5)
over = dict(n_levels=n_levels, first_level=first, last_level=last, has_depth=int(depth),
            max_iters=int(rng.integers(1, 12)), early_exit=int(rng.random() < 0.5),
            gain=float(np.float32(rng.choice([1.0, 10.0, 50.0]))), epsilon=float(np.float32(10.0 ** rng.uniform(-5, -2))),
            handoff_scale_t=int(rng.random() < 0.8))
if rng.random() < 0.3:
    over.update(z_factor=float(np.float32(rng.uniform(0.002, 1))), angle_factor=float(np.float32(rng.uniform(0.5, 2))))
r = rng.random()
if r < 0.2: over.update(weights=1)
elif r < 0.3: over.update(weights=2)
elif r < 0.4: over.update(sampler=1)
print(w, h, intr, over)
n = 3
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
p = O.default_params(w, h, *intr, **over)
frames, depths, cpu = [], [], []
for s in range(n):
    ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=91000 + 10 * seed + s, with_depth=depth,
                                            max_t=float(rng.uniform(0.002, 0.03)), max_deg=float(rng.uniform(0.1, 1.5)))
    frames += [ref, tgt]; depths += [dep, dep]
    cpu.append(O.align_pair(p, ref, tgt, dep if depth else None, want_trace=True))
ctx.upload_frames(0, np.stack(frames), np.stack(depths) if depth else None)
ctx.build_pyramids(0, 2 * n); ctx.apply_gradient(0, 2 * n)
poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1)
general = bool(over.get("weights") or over.get("sampler"))
for i in range(n):
    st, pose, tr = cpu[i]
    same = np.array_equal(poses[i].view(np.uint32), pose.view(np.uint32))
    print("pair", i, "bit-identical" if same else "DIFFERS: ulps %s" % (poses[i].view(np.int32).astype(np.int64) - pose.view(np.int32)))
    if same:
        continue
    pose_in = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    last_level = None
    for t in tr:
        lvl = t["level"]
        if last_level is not None and lvl != last_level:
            pose_in = O.se3_handoff(pose_in, over["handoff_scale_t"])
        last_level = lvl
        g = ctx.residual_jacobian_weighted(2 * i, 2 * i + 1, lvl, pose_in) if general else ctx.residual_jacobian(2 * i, 2 * i + 1, lvl, pose_in, dump=False)
        A32 = g["A"].astype(np.float32)
        b32 = (-g["jtr"]).astype(np.float32) if general else (-(np.float64(p.gain) * g["jtr"])).astype(np.float32)
        A_ref = np.asarray(t["A"], np.float32).reshape(6, 6); b_ref = np.asarray(t["b"], np.float32)
        dA = np.abs(A32.view(np.int32).astype(np.int64) - A_ref.view(np.int32)); db = np.abs(b32.view(np.int32).astype(np.int64) - b_ref.view(np.int32))
        if dA.max() or db.max() or g["n_valid"] != t["n_valid"]:
            print("  level %d iteration %d: A max %d ulps, b max %d ulps, n_valid %d / %d" % (lvl, t["iter"], dA.max(), db.max(), g["n_valid"], t["n_valid"]))
            if dA.max():
                k = int(np.argmax(dA)); print("    A[%d]: gpu f64 %.17g -> %r ; cpu %r" % (k, g["A"].ravel()[k], A32.ravel()[k], A_ref.ravel()[k]))
            if db.max():
                k = int(np.argmax(db)); v = -g["jtr"][k] if general else -(np.float64(p.gain) * g["jtr"][k]); print("    b[%d]: gpu f64 %.17g -> %r ; cpu %r" % (k, v, b32[k], b_ref[k]))
        pose_in = np.asarray(t["pose"], np.float32)
ctx.close()
