"""Pose-parity survey: GPU (C ABI) vs CPU oracle on N seeded synthetic pairs.  Prints the distribution of
rotation / translation differences and the count of bit-identical poses."""
import importlib, os, sys, time, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--w", type=int, default=160)
ap.add_argument("--h", type=int, default=96)
ap.add_argument("--mode", default="fixed")
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--seed0", type=int, default=1000)
ap.add_argument("--single", type=int, default=0, help="1: one pair per synchronous call (the chained k_iterate flow)")
ap.add_argument("--weights", type=int, default=0, help="0 identity, 1 Tukey (reference medians), 2 Huber")
ap.add_argument("--sampler", type=int, default=0, help="0 nearest, 1 bilinear")
ap.add_argument("--arith", default="opencv", choices=["opencv", "legacy"], help="arithmetic set of both sides")
ap.add_argument("--intrinsics", default="", help="fx,fy,cx,cy (default: square pixels, 525 * w / 640)")
ap.add_argument("--tuning", default="", help="uwt_tuning fields of the GPU context, k=v[,k=v] (launch shapes; never results)")
a = ap.parse_args()
ARITH = {"opencv": 0, "legacy": 1}[a.arith]
capi.DEFAULT_ARITH = ARITH
O.DEFAULT_ARITH = ARITH
w, h = a.w, a.h
f = 525.0 * w / 640.0
intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
if a.intrinsics:
    intr = tuple(float(v) for v in a.intrinsics.split(","))
if a.mode == "fixed":
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
elif a.mode == "fixed5":     # all five levels, a few iterations each
    over = dict(n_levels=5, first_level=4, last_level=0, max_iters=4, early_exit=0)
else:
    over = dict()
if a.depth:
    over["has_depth"] = 1
if a.weights:
    over["weights"] = a.weights
if a.sampler:
    over["sampler"] = a.sampler
n = a.n
tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tuning.split(",") if kv}
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over), tuning=tuning or None)
po = O.default_params(w, h, *intr, **over)
frames, depths, cpu = [], [], []
t0 = time.time()
pairs = [synth.render_pair(w, h, *intr, seed=a.seed0 + s, z=1.0 + 0.2 * ((s % 5) - 2) / 2, with_depth=bool(a.depth)) for s in range(n)]
for ref, tgt, dep, _, _ in pairs:
    frames += [ref, tgt]
    if a.depth:
        depths += [dep, dep]
def _cpu(i):                                    # the C oracle releases the GIL: every host core
    ref, tgt, dep, _, _ = pairs[i]
    st, pose, tr = O.align_pair(po, ref, tgt, dep if a.depth else None, want_trace=True)
    return st, pose, len(tr)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(min(32, os.cpu_count() or 4)) as ex:
    cpu = list(ex.map(_cpu, range(n)))
t_cpu = time.time() - t0
ctx.upload_frames(0, np.stack(frames), np.stack(depths) if a.depth else None)
ctx.build_pyramids(0, 2 * n); ctx.apply_gradient(0, 2 * n)
if a.single:
    poses, stats = [], []
    for i in range(n):
        p1, s1 = ctx.estimate_pose_batch([2 * i], [2 * i + 1])
        poses.append(p1[0].copy()); stats.append(s1[0])
    poses = np.stack(poses)
else:
    poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1)
def rot_angle(qa, qb):
    qa, qb = qa.astype(np.float64), qb.astype(np.float64)
    wv = abs(float(np.dot(qa, qb)))
    v = qb[3] * qa[:3] - qa[3] * qb[:3] - np.cross(qa[:3], qb[:3])
    return 2.0 * np.arctan2(np.linalg.norm(v), wv)
dr = np.array([rot_angle(poses[i][:4], cpu[i][1][:4]) for i in range(n)])
dt = np.array([np.linalg.norm(poses[i][4:].astype(np.float64) - cpu[i][1][4:]) for i in range(n)])
bit = sum(np.array_equal(poses[i].view(np.uint32), cpu[i][1].view(np.uint32)) for i in range(n))
it_eq = sum(stats[i]["iterations"] == cpu[i][2] for i in range(n) if cpu[i][0] == 0)   # the count is defined for status 0 only
st_eq = sum(stats[i]["status"] == cpu[i][0] for i in range(n))
n_ok = sum(c[0] == 0 for c in cpu)
print("arith %s%s%s mode %s%s weights=%d sampler=%d %dx%d depth=%d n=%d: bit-identical %d, status equal %d, status 0: %d, iterations equal %d" % (a.arith, " fx!=fy" if intr[0] != intr[1] else "", " [%s]" % a.tuning if a.tuning else "", a.mode, " (one pair per call)" if a.single else " (one batch)", a.weights, a.sampler, w, h, a.depth, n, bit, st_eq, n_ok, it_eq))
print("  rot  diff: median %.2e  p90 %.2e  max %.2e  (>1e-4: %d)" % (np.median(dr), np.percentile(dr, 90), dr.max(), (dr > 1e-4).sum()))
print("  trans diff: median %.2e  p90 %.2e  max %.2e  (>1e-4: %d)" % (np.median(dt), np.percentile(dt, 90), dt.max(), (dt > 1e-4).sum()))
print("  |t| median %.2e ; cpu time/pair %.3fs" % (np.median([np.linalg.norm(c[1][4:]) for c in cpu]), t_cpu / n))
