"""Fuzz of the small entry points next to the path, against the oracle, bit for bit, at the edges of their domains:
  SE3 exp / mul / matrix / hand-off (angles from 1e-12 to beyond 2 pi, unnormalised and tiny quaternions),
  solve_delta (well- and ill-conditioned, exactly singular, rank-deficient, huge and tiny scales),
  robust weights (constant vectors, one element, all-negative, huge residuals, half-integers),
  trajectory accumulation (random lengths, scales, axis permutation, start poses),
  sparse point producers (gradient magnitude, candidates, patches, patch growth) on flat / saturated / striped / noisy frames,
  ingest (random calibrations and sizes: new camera matrix, maps, undistorted frame, ROI).
python tools/exp/aux_fuzz.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
counts, bad = {}, {}
shown = 0


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        nb = np.isnan(a) & np.isnan(b)      # NaN payloads / signs differ between x86 and gfx950
        return np.array_equal(np.where(nb, 0, a.view(np.uint32) if a.dtype == np.float32 else a), np.where(nb, 0, b.view(np.uint32) if b.dtype == np.float32 else b))
    return np.array_equal(a, b)


def check(what, ok, **kw):
    global shown
    counts[what] = counts.get(what, 0) + 1
    if not ok:
        bad[what] = bad.get(what, 0) + 1
        if shown < 25:
            shown += 1
            print("DIFFERS:", what, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in kw.items()}, flush=True)


t0 = time.time()
for arith in (0, 1):
    O.set_arith(arith)
    ctx = capi.Context(capi.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=3, first_level=2, last_level=0, arith=arith))
    # ---- SE3
    for c in range(cases * 4):
        k = rng.random()
        xi = rng.normal(0, 1, 6).astype(np.float32)
        if k < 0.2: xi[3:] *= np.float32(10.0 ** rng.uniform(-12, -3))
        elif k < 0.4: xi[3:] = (xi[3:] / np.linalg.norm(xi[3:]) * np.float32(np.pi * rng.choice([1, 2, 0.5]) + rng.normal(0, 1e-4))).astype(np.float32)
        elif k < 0.6: xi *= np.float32(10.0 ** rng.uniform(-2, 1.5))
        elif k < 0.65: xi[3:] = 0
        a, b = ctx.se3_exp(xi), O.se3_exp(xi)
        check("se3_exp", same(a, b), xi=xi, gpu=a, cpu=b)
        q = rng.normal(0, 1, 7).astype(np.float32)
        if rng.random() < 0.5: q[:4] /= np.linalg.norm(q[:4])
        if rng.random() < 0.1: q[:4] *= np.float32(10.0 ** rng.uniform(-20, 3))
        check("se3_mul", same(ctx.se3_mul(b, q), O.se3_mul(b, q)), a=b, b=q)
        check("se3_matrix", same(ctx.se3_matrix(q), O.se3_matrix(q)), q=q)
        for s in (0, 1):
            try:
                g = ctx.se3_handoff(q, s)
            except capi.UwtError:
                g = None
            try:
                o = O.se3_handoff(q, s)
            except ValueError:
                o = None
            check("se3_handoff", (g is None) == (o is None) and (g is None or same(g, o)), q=q, s=s, gpu=g, cpu=o)
    # ---- solve
    for c in range(cases * 4):
        k = rng.random()
        M = rng.normal(0, 1, (6, 6))
        A = (M @ M.T)
        if k < 0.2: A = A * 10.0 ** rng.uniform(-12, 12)
        elif k < 0.35: A[:, 3] = A[:, 1]; A[3, :] = A[1, :]          # exactly singular (symmetric)
        elif k < 0.45: A = np.zeros((6, 6)); A[:3, :3] = M[:3, :3] @ M[:3, :3].T
        elif k < 0.55: A = M                                          # not symmetric
        elif k < 0.65: A = A + np.diag(10.0 ** rng.uniform(-8, 8, 6))
        elif k < 0.7: A = np.diag(rng.choice([0.0, 1.0, 1e-30, 1e30], 6))
        A = A.astype(np.float32)
        b = (rng.normal(0, 1, 6) * 10.0 ** rng.uniform(-6, 6)).astype(np.float32)
        dg, Ag, okg = ctx.solve_delta(A, b)
        o = O.solve_delta(A, b)
        check("solve_delta", same(dg, o), A=A, b=b, gpu=dg, cpu=o)
    # ---- robust weights
    for c in range(cases * 2):
        n = int(rng.choice([1, 2, 3, 7, 64, 255, 256, 257, 1000, 5000, 70000]))
        k = rng.random()
        if k < 0.15: r = np.full(n, rng.integers(-255, 256), np.float32)
        elif k < 0.3: r = -np.abs(rng.normal(0, 30, n)).astype(np.float32)
        elif k < 0.45: r = (rng.integers(-255, 256, n) + 0.5).astype(np.float32)
        elif k < 0.6: r = (rng.normal(0, 1, n) * 10.0 ** rng.uniform(0, 4)).astype(np.float32)
        else: r = rng.integers(-255, 256, n).astype(np.float32)
        wg, med, mad = ctx.robust_weights(r, kind=1)
        check("robust_weights", same(wg, O.tukey_weights(r)) and np.float32(med) == np.float32(O.median_mat(r)) and np.float32(mad) == np.float32(O.mad(r)),
              n=n, k=round(k, 2), med=(med, O.median_mat(r)), mad=(mad, O.mad(r)))
    # ---- trajectory
    for c in range(cases // 4 + 1):
        n = int(rng.choice([0, 1, 2, 3, 63, 64, 65, 300, 2000]))
        q = rng.normal(0, 0.05, (n, 4)).astype(np.float32); q[:, 3] = 1
        if rng.random() < 0.7 and n: q /= np.linalg.norm(q, axis=1, keepdims=True)
        poses = np.concatenate([q, rng.normal(0, 0.05, (n, 3)).astype(np.float32)], axis=1)
        kw = dict(t_scale=float(np.float32(rng.choice([1.0, 40.0, 0.5]))), reference_axes=bool(rng.random() < 0.5))
        if rng.random() < 0.5: kw["start"] = O.se3_exp(rng.normal(0, 0.5, 6).astype(np.float32))
        check("trajectory", same(ctx.accumulate_trajectory(poses, **kw), O.accumulate_trajectory(poses, **kw)), n=n, kw={k: (v if not hasattr(v, "tolist") else v.tolist()) for k, v in kw.items()})
    ctx.close()
# ---- sparse point producers (independent of the arithmetic set): flat, saturated, noisy and tiny frames, any threshold and cap
for c in range(max(8, cases // 10)):
    w = int(rng.choice([16, 32, 48, 80, 160, 208, 320])); h = int(rng.choice([16, 32, 48, 96, 240]))
    nl = int(rng.integers(1, 4))
    depth = bool(rng.random() < 0.5)
    ctx = capi.Context(capi.default_params(w, h, float(w), float(w), w / 2 - 0.5, h / 2 - 0.5, n_levels=nl, first_level=nl - 1, last_level=0,
                                           has_depth=int(depth), max_frames=3, max_pairs=1))
    k = rng.random()
    if k < 0.2: img = np.full((h, w), rng.integers(0, 256), np.uint8)
    elif k < 0.4: img = (rng.random((h, w)) < 0.5).astype(np.uint8) * 255
    elif k < 0.6: img = np.tile(np.arange(w, dtype=np.uint8) * 3, (h, 1))
    else: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    frames = np.stack([img, np.roll(img, 1, 1), img.T.copy().reshape(-1)[: h * w].reshape(h, w)])
    dep = rng.integers(0, 40000, frames.shape).astype(np.uint16); dep[rng.random(frames.shape) < 0.3] = 0
    ctx.upload_frames(0, frames, dep if depth else None); ctx.build_pyramids(0, 3); ctx.apply_gradient(0, 3)
    for slot in range(3):
        a_img, dp = frames[slot], (dep[slot] if depth else None)
        for lvl in range(nl):
            if lvl:
                a_img = O.halve_u8(a_img); dp = O.halve_u16(dp) if depth else None
            gx, gy = O.scharr3(a_img)
            mag = O.gradient_mag(gx, gy)
            ok = np.array_equal(ctx.gradient_magnitude(slot, lvl), mag)
            for thr in (float(rng.choice([0.0, 20.0, 300.0, -5.0])), float(rng.uniform(0, 80))):
                cap = None if rng.random() < 0.7 else int(rng.integers(0, 50))
                a, na = ctx.obtain_candidate_points(slot, lvl, thr, cap=cap)
                b, nb = O.candidate_points(mag, dp, thr)
                ok = ok and na == nb and np.array_equal(a.view(np.uint32), (b if cap is None else b[:cap]).view(np.uint32))
            check("candidates", ok, size=(w, h), lvl=lvl, slot=slot, kind=round(k, 2), depth=depth)
        n = int(rng.choice([0, 1, 5, 60, 230]))
        kp = rng.uniform(0, [w - 0.01, h - 0.01], (n, 2)).astype(np.float32)
        if n and rng.random() < 0.5: kp[0] = (0, 0); kp[-1] = (w - 1, h - 1)
        a, na = ctx.obtain_patch_points(slot, kp)
        b, nb = O.patch_points(kp, dep[slot] if depth else None, w, h)
        check("patch_points", na == nb and np.array_equal(a.view(np.uint32), b.view(np.uint32)), size=(w, h), n=n, depth=depth)
    for lvl in range(nl):
        L = ctx.level_info(lvl)
        n = int(rng.choice([0, 1, 7, 300, 1000]))
        pts = np.column_stack([rng.uniform(-2, L.w + 2, n), rng.uniform(-2, L.h + 2, n), rng.uniform(0.5, 2.0, n), np.ones(n)]).astype(np.float32)
        ps = int(rng.choice([1, 3, 5]))
        got, cnt = ctx.add_patch_points(lvl, pts, patch_size=ps)
        want, n_want = O.add_patch_points(pts, L.w, L.h, patch_size=ps)
        check("add_patch_points", cnt == n_want and np.array_equal(got.view(np.uint32), want.view(np.uint32)), lvl=lvl, n=n, ps=ps)
    ctx.close()
# ---- ingest (independent of the arithmetic set)
for c in range(max(4, cases // 25)):
    in_w, in_h = int(rng.choice([320, 376, 640, 752])), int(rng.choice([240, 256, 480]))
    out_w, out_h = in_w - int(rng.choice([0, 16, 24])), in_h - int(rng.choice([0, 8]))
    f = rng.uniform(0.5, 1.0) * in_w
    K = [float(f), float(f * rng.uniform(0.98, 1.02)), float(in_w / 2 + rng.uniform(-10, 10)), float(in_h / 2 + rng.uniform(-10, 10))]
    D = [float(rng.uniform(-0.35, 0.1)), float(rng.uniform(-0.05, 0.1)), float(rng.normal(0, 1e-3)), float(rng.normal(0, 1e-3))]
    try:
        ing = capi.Ingest(K, D, in_w, in_h, out_w, out_h)
    except capi.UwtError as e:
        check("ingest", False, K=K, D=D, err=str(e)); continue
    nk = O.optimal_new_camera_matrix(K, D, in_w, in_h, out_w, out_h)
    ok = same(np.asarray(ing.newK, np.float32), nk.astype(np.float32))
    m1, m2 = ing.maps()
    o1, o2 = O.init_undistort_maps(K, D, nk, out_w, out_h)
    ok = ok and np.array_equal(m1, o1) and np.array_equal(m2, o2)
    raw = rng.integers(1, 256, (in_h, in_w), dtype=np.uint8)
    und = ing.undistort(raw)
    ok = ok and np.array_equal(und, O.remap_linear(raw, o1, o2)) and np.array_equal(ing.calculate_roi(raw), O.calculate_roi(und))
    check("ingest", ok, K=K, D=D, size=(in_w, in_h, out_w, out_h), maps=int((m1 != o1).sum()), frac=int((m2 != o2).sum()))
    ing.close()
print("aux fuzz seed %d (%.0f s): " % (seed, time.time() - t0) + ", ".join("%s %d/%d differ" % (k, bad.get(k, 0), v) for k, v in counts.items()))
sys.exit(1 if bad else 0)
