"""Several host threads, one context each ("one ctx per host thread per GPU", include/uwt.h), all on the same device at once, each
with its own geometry, solver constants, arithmetic set and tuning, aligning random batches for a while: every result against the
oracle's (computed beforehand, single-threaded).  Hunts state shared between contexts (statics, symbols, scratch).
python tools/exp/thread_fuzz.py [threads] [steps] [seed]"""
import importlib, os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
GEOM = [(160, 96, 4), (208, 112, 5), (112, 80, 5), (320, 240, 4), (256, 192, 7), (640, 480, 4)]
U, NP = 8, 32
jobs = []
rng = np.random.default_rng(seed)
for t in range(T):
    w, h, nl = GEOM[t % len(GEOM)]
    fx = float(np.float32(0.8 * w)); fy = fx if t % 2 == 0 else float(np.float32(fx * 0.996))
    intr = (fx, fy, float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
    first = int(rng.integers(1, nl))
    over = dict(n_levels=nl, has_depth=int(t % 3 != 0), first_level=first, last_level=int(rng.integers(0, first)), max_iters=int(rng.integers(2, 9)),
                early_exit=int(rng.random() < 0.5), arith=int(t % 3 == 1), weights=[0, 0, 2, 1][t % 4], sampler=int(t % 5 == 4 and t % 4 != 3))
    pairs = [synth.render_pair(w, h, *intr, seed=seed * 100 + 10 * t + s, with_depth=bool(over["has_depth"]), max_t=0.004 + 0.002 * s, max_deg=0.3) for s in range(U)]
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, p[0], p[1], p[2] if over["has_depth"] else None) for p in pairs]
    jobs.append((w, h, intr, over, pairs, want))
results = [None] * T
barrier = threading.Barrier(T)


def work(t):
    try:
        _work(t)
    except BaseException:
        barrier.abort()
        results[t] = (0, 1)
        raise


def _work(t):
    w, h, intr, over, pairs, want = jobs[t]
    r = np.random.default_rng(seed * 1000 + t)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * NP, max_pairs=NP, **over))
    frames = np.stack([pairs[i % U][k] for i in range(NP) for k in (0, 1)])
    depth = np.stack([pairs[i % U][2] for i in range(NP) for k in (0, 1)]) if over["has_depth"] else None
    ctx.upload_frames(0, frames, depth); ctx.build_pyramids(0, 2 * NP); ctx.apply_gradient(0, 2 * NP)
    hp = capi.pinned_empty((NP, 7), np.float32); hs = capi.pinned_empty((NP, 4), np.int32)
    barrier.wait()
    n_al = n_bad = 0
    for step in range(steps):
        if r.random() < 0.2:
            ctx.set_tuning(split=int(r.integers(1, 5)), split_min_px=int(r.choice([1, 1 << 24])), chained=int(r.integers(-1, 2)), tail_update=int(r.integers(0, 3)),
                           coarse=int(r.random() < 0.7), typed_loads=int(r.random() < 0.7), speculation=int(r.random() < 0.7))
        n = int(r.choice([1, 2, 5, 16, NP]))
        sel = r.choice(NP, n, replace=False)
        if r.random() < 0.6:
            poses, stats = ctx.estimate_pose_batch(sel * 2, sel * 2 + 1)
            st = [s["status"] for s in stats]
        else:
            tk = ctx.track_batch_host_async(0, 2 * NP, sel * 2, sel * 2 + 1, hp, hs)
            ctx.wait_ticket(tk)
            poses = hp[:n].copy(); st = [int(hs[i, 0]) for i in range(n)]
        for k, i in enumerate(sel):
            cs, cp, _ = want[int(i) % U]
            n_al += 1
            if st[k] != cs or not np.array_equal(poses[k].view(np.uint32), cp.view(np.uint32)):
                n_bad += 1
                if n_bad <= 3:
                    print("DIFFERS thread %d (%dx%d %s) step %d pair %d: status %d/%d\n  got  %s\n  want %s" % (t, w, h, over, step, int(i), st[k], cs, poses[k], cp), flush=True)
    ctx.close()
    results[t] = (n_al, n_bad)


t0 = time.time()
th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
for x in th: x.start()
for x in th: x.join()
tot = sum(r[0] for r in results); bad = sum(r[1] for r in results)
print("thread fuzz seed %d: %d threads x %d steps, %d alignments, %d differ from the oracle, %.0f s" % (seed, T, steps, tot, bad, time.time() - t0))
sys.exit(1 if bad else 0)
