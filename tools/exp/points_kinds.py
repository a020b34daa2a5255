"""Which kind of table row makes uwt_estimate_pose_points part from the oracle?  One pair, level 0, one table kind at a time."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
w, h, nl = 160, 96, 4
intr = (128.0, 128.0, 79.5, 47.5)
rng = np.random.default_rng(3)
over = dict(n_levels=nl, first_level=0, last_level=0, max_iters=3, early_exit=0)
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
po = O.default_params(w, h, *intr, **over)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=77, with_depth=False, max_t=0.01, max_deg=0.5)
ctx.upload_frames(0, np.stack([ref, tgt])); ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
def table(n, integer=True, inb=True, z=(1.0,), wq=(1.0,)):
    t = np.empty((n, 4), np.float32)
    lo, hi = (2, -2) if inb else (-3, 3)
    t[:, 0] = rng.uniform(lo, w + hi, n); t[:, 1] = rng.uniform(lo, h + hi, n)
    if integer: t[:, :2] = np.floor(t[:, :2])
    t[:, 2] = rng.choice(z, n); t[:, 3] = rng.choice(wq, n)
    return t
for name, kw in [("integer, in bounds, z 1, w 1", {}), ("z in {0.3, 1, 2.5}", dict(z=(0.3, 1.0, 2.5))), ("fractional positions", dict(integer=False)),
                 ("w in {1, 0.5}", dict(wq=(1.0, 0.5))), ("w in {1, 0}", dict(wq=(1.0, 0.0))), ("z in {1, 0}", dict(z=(1.0, 0.0))), ("z in {1, -0.5}", dict(z=(1.0, -0.5))),
                 ("out of bounds, integer", dict(inb=False)), ("out of bounds, fractional", dict(inb=False, integer=False))]:
    for n in (64, 1000):
        t = table(n, **kw)
        pose, st = ctx.estimate_pose_points(0, 1, {0: t})
        cs, cp, tr = O.align_pair_points(po, ref, tgt, {0: t}, want_trace=True)
        print("%-30s n %5d: status %d/%d iterations %d/%d n_valid gpu %d cpu %s  pose equal %s" % (name, n, st["status"], cs, st["iterations"], len(tr), st["n_valid"], [q["n_valid"] for q in tr], np.array_equal(pose.view(np.uint32), cp.view(np.uint32))))
