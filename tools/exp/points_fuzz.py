"""Fuzz of the explicit point-table alignment (uwt_estimate_pose_points: what EstimatePose / EstimatePoseFeatures iterate over when
a sparse producer filled Frame::candidatePoints_, src/Tracker.cpp:401, 669) against the oracle: random tables (sub-sampled dense
tables, candidate points, random positions in and out of the frame, zero and negative depths, odd w columns, empty tables, tables
longer than one block's share), random solver constants (robust weights and the bilinear sampler included), both arithmetic sets; status, pose bits and — where the status is 0 —
the iteration count.  python tools/exp/points_fuzz.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
GEOM = [(160, 96, 4), (208, 112, 5), (112, 80, 5), (320, 240, 4)]
total = bad = 0
t0 = time.time()
for g, (w, h, nl) in enumerate(GEOM):
    fx = float(np.float32(0.8 * w))
    fy = fx if g % 2 == 0 else float(np.float32(fx * 0.996))
    intr = (fx, fy, float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
    for depth in (0, 1):
        base = dict(n_levels=nl, has_depth=depth)
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, first_level=nl - 1, last_level=0, **base))
        for case in range(cases):
            ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=seed * 100000 + 1000 * g + case, with_depth=bool(depth),
                                                    max_t=float(rng.uniform(0.002, 0.02)), max_deg=float(rng.uniform(0.1, 1.0)))
            ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]) if depth else None)
            ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
            first = int(rng.integers(0, nl)); last = int(rng.integers(0, first + 1))
            over = dict(first_level=first, last_level=last, max_iters=int(rng.integers(1, 11)), early_exit=int(rng.random() < 0.5),
                        gain=float(np.float32(rng.choice([1.0, 10.0, 50.0]))), epsilon=float(np.float32(10.0 ** rng.uniform(-5, -2))),
                        handoff_scale_t=int(rng.random() < 0.8), arith=int(rng.random() < 0.3), z_factor=1.0, angle_factor=1.0)
            if rng.random() < 0.4:
                over.update(z_factor=float(np.float32(rng.choice([0.002, rng.uniform(0.002, 1)]))), angle_factor=float(np.float32(rng.uniform(0.5, 2))))
            r = rng.random()     # the general path over a table: robust weights, bilinear sampler
            over.update(weights=0, sampler=0)
            if r < 0.15: over.update(weights=1)
            elif r < 0.3: over.update(weights=2)
            elif r < 0.4: over.update(sampler=1)
            elif r < 0.5: over.update(sampler=1, weights=2)
            ctx.update_params(**over)
            po = O.default_params(w, h, *intr, **base, **over)
            tables = {}
            dl = dep if depth else None
            dps = [dl]
            for l in range(1, nl):
                dps.append(O.halve_u16(dps[-1]) if depth else None)
            for l in range(last, first + 1):
                lw, lh = w >> l, h >> l
                kind = rng.random()
                dense = O.dense_points(dps[l], lw, lh, l)
                if kind < 0.35:      # a random subset of the dense table, in table order
                    keep = rng.random(dense.shape[0]) < rng.uniform(0.02, 1.0)
                    t = dense[keep]
                elif kind < 0.5:     # the producer's own: high-gradient candidates
                    t = ctx.obtain_candidate_points(0, l, float(rng.choice([0.0, 20.0, 60.0])))[0]
                elif kind < 0.6 and l != first:
                    t = np.zeros((0, 4), np.float32)
                else:                # arbitrary rows: positions in and out of the frame, any depth sign, odd w
                    n = int(rng.choice([1, 3, 63, 64, 65, 1000, 9000, 20000]))
                    t = np.empty((n, 4), np.float32)
                    t[:, 0] = rng.uniform(-3, lw + 3, n); t[:, 1] = rng.uniform(-3, lh + 3, n)
                    t[:, 2] = rng.choice([0.0, -0.5, 0.3, 1.0, 2.5], n, p=[0.05, 0.05, 0.2, 0.5, 0.2]) * rng.uniform(0.5, 1.5, n)
                    t[:, 3] = rng.choice([1.0, 0.0, 0.5], n, p=[0.9, 0.05, 0.05])
                    if rng.random() < 0.5:
                        t[:, :2] = np.floor(t[:, :2])
                tables[l] = np.ascontiguousarray(t, np.float32)
            pose, st = ctx.estimate_pose_points(0, 1, tables)
            cs, cp, tr = O.align_pair_points(po, ref, tgt, tables, ref_depth=dl, want_trace=True)
            total += 1
            same = np.array_equal(pose.view(np.uint32), cp.view(np.uint32)) or (np.isnan(pose).all() and np.isnan(cp).all())  # a diverged pair: NaN on both sides (x86 and gfx950 differ in the NaN's sign bit)
            ok = st["status"] == cs and same and (cs != 0 or st["iterations"] == len(tr))
            if not ok:
                bad += 1
                if bad <= 15:
                    print("DIFFERS %dx%d depth %d case %d: status %d/%d iterations %d/%d\n  gpu %s\n  cpu %s\n  params %s\n  tables %s" % (
                        w, h, depth, case, st["status"], cs, st["iterations"], len(tr), pose, cp, over, {l: t.shape[0] for l, t in tables.items()}), flush=True)
        ctx.close()
    print("geometry %dx%d x %d levels: %d cases so far, %d differ, %.0f s" % (w, h, nl, total, bad, time.time() - t0), flush=True)
print("points fuzz seed %d: %d alignments over point tables, %d differ from the oracle" % (seed, total, bad))
sys.exit(1 if bad else 0)
