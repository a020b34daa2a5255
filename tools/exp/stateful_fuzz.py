"""Stateful fuzz of ONE long-lived context: between alignments the solver constants (uwt_update_params), the launch shapes
(uwt_set_tuning), the batch size and the call form (synchronous batch / whole per-frame path with host outputs) change at
random; every result is compared with the oracle run under the same constants — status, pose bits, and the iteration count
where the status is 0.  What it hunts: state that survives a change (scale buffers, speculation budget, coarse-level plans,
per-level vector widths, split plans).  python tools/exp/stateful_fuzz.py [steps] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
GEOM = [(160, 96, 4), (208, 112, 5), (112, 80, 5), (320, 240, 4), (384, 256, 8), (256, 192, 7), (163, 99, 4), (245, 157, 5), (333, 251, 5)] if os.environ.get('FUZZ_DEEP') \
    else [(160, 96, 4), (208, 112, 5), (112, 80, 5), (320, 240, 4), (163, 99, 4), (245, 157, 5)]   # (the last ones: odd sizes, round 6)
U, NP = 12, 64          # distinct pairs, resident pairs
total = bad = 0
t0 = time.time()
pool = ThreadPoolExecutor(min(32, os.cpu_count() or 4))
for g, (w, h, nl) in enumerate(GEOM):
    fx = float(np.float32(0.8 * w))
    fy = fx if g % 2 == 0 else float(np.float32(fx * 0.996))
    intr = (fx, fy, float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
    base = dict(n_levels=nl, has_depth=1)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * NP, max_pairs=NP, first_level=nl - 1, last_level=0, **base))
    pairs = [synth.render_pair(w, h, *intr, seed=seed * 1000 + 50 * g + s, with_depth=True, max_t=0.004 + 0.003 * (s % 5),
                               max_deg=0.2 + 0.2 * (s % 4)) for s in range(U)]
    frames = np.empty((2 * NP, h, w), np.uint8); depth = np.empty((2 * NP, h, w), np.uint16)
    for i in range(NP):
        ref, tgt, dep, _, _ = pairs[i % U]
        frames[2 * i], frames[2 * i + 1], depth[2 * i], depth[2 * i + 1] = ref, tgt, dep, dep
    pf = capi.pinned_empty(frames.shape, np.uint8); pf[...] = frames
    pd = capi.pinned_empty(depth.shape, np.uint16); pd[...] = depth
    ctx.upload_frames(0, frames, depth)
    ctx.build_pyramids(0, 2 * NP); ctx.apply_gradient(0, 2 * NP)
    h_poses = capi.pinned_empty((NP, 7), np.float32); h_stats = capi.pinned_empty((NP, 4), np.int32)
    for step in range(steps):
        first = int(rng.integers(0, nl)); last = int(rng.integers(0, first + 1))
        over = dict(first_level=first, last_level=last, max_iters=int(rng.integers(1, 11)) if rng.random() < 0.93 else int(rng.choice([17, 25, 60])), early_exit=int(rng.random() < 0.5),
                    gain=float(np.float32(rng.choice([1.0, 10.0, 50.0]))), epsilon=float(np.float32(10.0 ** rng.uniform(-5, -2))),
                    handoff_scale_t=int(rng.random() < 0.8), arith=int(rng.random() < 0.3), weights=0, sampler=0,
                    z_factor=1.0, angle_factor=1.0)
        if rng.random() < 0.3:
            over.update(z_factor=float(np.float32(rng.uniform(0.002, 1))), angle_factor=float(np.float32(rng.uniform(0.5, 2))))
        r = rng.random()
        if r < 0.15: over.update(weights=1)
        elif r < 0.3: over.update(weights=2)
        elif r < 0.4: over.update(sampler=1)
        elif r < 0.45: over.update(sampler=1, weights=2)
        tune = dict(split=int(rng.integers(1, 5)), split_min=int(rng.choice([1, 2, 8])), split_min_px=int(rng.choice([1, 1 << 20, 1 << 24])),
                    stream_bytes=int(rng.choice([0, 1 << 20, 200 << 20])), tail_update=int(rng.integers(0, 3)),
                    target_blocks=int(rng.choice([0, 16, 64, 1024])), coarse=int(rng.random() < 0.7),
                    coarse_batch_px=int(rng.choice([0, 1000, 6144, 100000])), coarse_weighted=int(rng.random() < 0.7),
                    overlap_gradients=int(rng.random() < 0.5), first_poll=int(rng.integers(1, 6)), chained=int(rng.integers(-1, 2)),
                    speculation=int(rng.random() < 0.7), fused_stages=int(rng.random() < 0.5), pyramid_batch=int(rng.random() < 0.5),
                    typed_loads=int(rng.random() < 0.7))
        if rng.random() < 0.6:
            ctx.set_tuning(**tune)
        ctx.update_params(**over)
        n = int(rng.choice([1, 1, 2, 3, 5, 8, 17, 33, 64]))
        sel = rng.choice(NP, n, replace=bool(rng.random() < 0.3) or n > NP)
        po = O.default_params(w, h, *intr, **base, **over)
        need = sorted(set(int(i) % U for i in sel))
        cpu = dict(zip(need, pool.map(lambda u: (lambda q: (q[0], q[1], len(q[2])))(O.align_pair(po, pairs[u][0], pairs[u][1], pairs[u][2], want_trace=True)), need)))
        form = rng.random()
        if form < 0.6:
            poses, stats = ctx.estimate_pose_batch(sel * 2, sel * 2 + 1)
            st = [(s["status"], s["iterations"]) for s in stats]
        else:      # the per-frame path: upload (sometimes), pyramids, gradients, alignment, host outputs
            if rng.random() < 0.5:
                ctx.upload_frames_async(0, pf, pd)
            tk = ctx.track_batch_host_async(0, 2 * NP, sel * 2, sel * 2 + 1, h_poses, h_stats, grad_refs_only=bool(rng.random() < 0.5))
            ctx.wait_ticket(tk)
            poses = h_poses[:n].copy(); st = [(int(h_stats[i, 0]), int(h_stats[i, 1])) for i in range(n)]
        for k, i in enumerate(sel):
            cs, cp, ci = cpu[int(i) % U]
            total += 1
            ok = st[k][0] == cs and np.array_equal(poses[k].view(np.uint32), cp.view(np.uint32)) and (cs != 0 or st[k][1] == ci)
            if not ok:
                bad += 1
                if bad <= 20:
                    print("DIFFERS geom %dx%d step %d pair %d (of %d, form %s): status %d/%d iterations %d/%d pose %s / %s\n  params %s\n  tuning %s" % (
                        w, h, step, int(i), n, "batch" if form < 0.6 else "frame path", st[k][0], cs, st[k][1], ci, poses[k], cp, over,
                        {f[0]: getattr(ctx.get_tuning(), f[0]) for f in capi.Tuning._fields_ if f[0] != "reserved"}), flush=True)
    ctx.close()
    print("geometry %dx%d x %d levels (fx %s fy): %d steps done, %d alignments so far, %d differ, %.0f s" % (w, h, nl, "==" if fx == fy else "!=", steps, total, bad, time.time() - t0), flush=True)
print("stateful fuzz seed %d: %d alignments over %d geometries x %d steps, %d differ from the oracle" % (seed, total, len(GEOM), steps, bad))
sys.exit(1 if bad else 0)
