#!/usr/bin/env python3
"""Where does one event of tools/exp/stateful_fuzz.py part from the oracle?  Rebuilds the pair as that script renders it and replays
the oracle's trace: at every evaluation's input pose the GPU's f64 sums (the stage entry point) are rounded to f32 and compared with the
oracle's A and b.  usage: diagnose_case.py <geometry index> <fuzz seed> <distinct pair> "<params dict as printed>" """
import importlib, os, sys, ast
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
GEOM = [(160, 96, 4), (208, 112, 5), (112, 80, 5), (320, 240, 4)]
g, seed, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
over = ast.literal_eval(sys.argv[4])
w, h, nl = GEOM[g]
fx = float(np.float32(0.8 * w))
fy = fx if g % 2 == 0 else float(np.float32(fx * 0.996))
intr = (fx, fy, float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
base = dict(n_levels=nl, has_depth=1)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=seed * 1000 + 50 * g + s, with_depth=True, max_t=0.004 + 0.003 * (s % 5), max_deg=0.2 + 0.2 * (s % 4))
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **base, **over))
po = O.default_params(w, h, *intr, **base, **over)
ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]))
ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
poses, stats = ctx.estimate_pose_batch([0], [1])
st, pose, tr = O.align_pair(po, ref, tgt, dep, want_trace=True)
print("gpu", poses[0], "\ncpu", pose, "\nulps", poses[0].view(np.int32) - pose.view(np.int32), "status", stats[0]["status"], st)
plain = not (over.get("weights") or over.get("sampler"))
fn = (lambda l_, p_: ctx.residual_jacobian(0, 1, l_, p_, dump=False)) if plain else (lambda l_, p_: ctx.residual_jacobian_weighted(0, 1, l_, p_))
pose_in = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
last = None
for t in tr:
    lvl = t["level"]
    if last is not None and lvl != last:
        q = O.se3_handoff(pose_in, int(over.get("handoff_scale_t", 0)))
        pose_in = q[0] if isinstance(q, tuple) else q
    last = lvl
    G = fn(lvl, pose_in)
    A32 = G["A"].astype(np.float32)
    b32 = (-(po.gain * G["jtr"])).astype(np.float32) if plain else (-G["jtr"]).astype(np.float32)
    tA = np.asarray(t["A"], np.float32).reshape(6, 6); tb = np.asarray(t["b"], np.float32)
    dA = np.abs(A32.view(np.int32).astype(np.int64) - tA.view(np.int32)); db = np.abs(b32.view(np.int32).astype(np.int64) - tb.view(np.int32))
    if dA.max() or db.max() or G["n_valid"] != t["n_valid"]:
        print("level %d iteration %d: A differs by %d ulps, b by %d ulps, n_valid %d / %d" % (lvl, t["iter"], dA.max(), db.max(), G["n_valid"], t["n_valid"]))
        for name, d, g64, g32, c32 in (("A", dA.ravel(), G["A"].ravel(), A32.ravel(), tA.ravel()), ("b", db, -(po.gain if plain else 1.0) * G["jtr"], b32, tb)):
            for k in np.nonzero(d)[0]:
                lo, hi = sorted((float(g32[k]), float(c32[k])))
                mid = 0.5 * (lo + hi)
                print("  %s[%d]: gpu f64 sum %.17g -> %r, oracle %r; the midpoint of the two floats is %.17g: the sum is %.3g of an f32 ulp from it"
                      % (name, k, g64[k], g32[k], c32[k], mid, abs(g64[k] - mid) / (hi - lo)))
    pose_in = np.asarray(t["pose"], np.float32)
ctx.close()
