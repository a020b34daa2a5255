"""Fuzz of the per-stage entry points at poses an alignment never starts from but can diverge to: rotations up to radians,
translations that put part of the scene behind the camera (z2 <= 0: the reference clamps 1 / z2 at 0, src/Tracker.cpp:452), poses
that push every pixel out of the frame.  uwt_residual_jacobian (dump form and the production sums), uwt_residual_jacobian_weighted
(robust weights / bilinear sampler) and uwt_warp against the oracle, bit for bit: valid masks, residuals, Jacobian rows, weights,
A and b.  python tools/exp/stage_fuzz.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
total = bad = 0
t0 = time.time()


def differs(what, **kw):
    global bad
    bad += 1
    if bad <= 12:
        print("DIFFERS:", what, kw, flush=True)


for case in range(cases):
    w = int(rng.choice([32, 48, 80, 96, 112, 160, 208, 320]))
    h = int(rng.choice([16, 32, 48, 64, 96, 240]))
    if rng.random() < 0.5:   # any size is a frame size (round 6): odd, ROI-like, grids smaller than their images
        w, h = int(rng.integers(17, 330)), int(rng.integers(17, 250))
    n_levels = int(rng.integers(1, 4))
    fx = float(np.float32(rng.uniform(0.5, 1.5) * w))
    fy = fx if rng.random() < 0.5 else float(np.float32(fx * rng.uniform(0.9, 1.1)))
    intr = (fx, fy, float(np.float32(w / 2 + rng.uniform(-5, 5))), float(np.float32(h / 2 + rng.uniform(-5, 5))))
    depth = bool(rng.random() < 0.5)
    arith = int(rng.random() < 0.3)
    O.set_arith(arith)            # the oracle's per-stage functions compute in the process-wide set
    over = dict(n_levels=n_levels, first_level=n_levels - 1, last_level=0, has_depth=int(depth), arith=arith)
    if rng.random() < 0.4:
        over.update(z_factor=float(np.float32(rng.uniform(0.001, 2))), angle_factor=float(np.float32(rng.uniform(0.1, 3))))
    general = rng.random() < 0.4
    if general:
        over.update([dict(weights=1), dict(weights=2), dict(sampler=1), dict(sampler=1, weights=2)][int(rng.integers(0, 4))])
    ref = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref[: h // 3] = 255
    ref[:, : w // 5] = 0
    tgt = np.roll(ref, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(0, 1))
    dep = None
    if depth:
        dep = rng.integers(0, 40000, (h, w)).astype(np.uint16)
        dep[rng.random((h, w)) < 0.2] = 0
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]) if depth else None)
    ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
    p = O.default_params(w, h, *intr, **over)
    a_img, b_img, dp = ref, tgt, dep
    for lvl in range(n_levels):
        if lvl:
            a_img, b_img = O.resize_half_u8(a_img), O.resize_half_u8(b_img)
            dp = O.resize_half_u16(dp) if depth else None
        L = O.level_intrinsics(p, lvl)
        gx, gy = O.scharr3(a_img)
        pts = O.dense_points(dp, L.w, L.h, lvl)
        for rep in range(3):
            xi = (rng.normal(0, 1, 6) * 10.0 ** rng.uniform(-3, 0.7)).astype(np.float32)
            kind = rng.random()
            if kind < 0.25:
                xi[2] = np.float32(-rng.uniform(0.5, 3.0))      # the camera moves through the scene: z2 <= 0 for part of it
            elif kind < 0.35:
                xi[:3] = 0; xi[3:] = (rng.normal(0, 1, 3) * 2).astype(np.float32)   # a pure, large rotation
            pose = O.se3_exp(xi)
            if not np.isfinite(pose).all():
                continue
            total += 1
            wp = O.warp(pts, pose, L)
            gw = ctx.warp(lvl, pts, pose)
            if not np.array_equal(np.asarray(gw).view(np.uint32), wp.view(np.uint32)):
                nanboth = np.isnan(gw) & np.isnan(wp)
                if not np.array_equal(np.where(nanboth, 0, np.asarray(gw).view(np.uint32)), np.where(nanboth, 0, wp.view(np.uint32))):
                    differs("warp", case=case, lvl=lvl, xi=xi.tolist(), over=over)
            valid = np.zeros(L.w * L.h, np.uint8)
            if not general:
                J, r, idx = O.residual_jacobian(a_img, b_img, gx, gy, pts, wp, L, p.z_factor, p.angle_factor)
                valid[idx] = 1
                out = ctx.residual_jacobian(0, 1, lvl, pose)
                fast = ctx.residual_jacobian(0, 1, lvl, pose, dump=False)
                ok = np.array_equal(out["valid"], valid) and np.array_equal(out["r"][idx], r) and \
                    np.array_equal(out["J"][idx].view(np.uint32), J.view(np.uint32)) and out["n_valid"] == len(idx) and \
                    out["sum_r2"] == int((r.astype(np.int64) ** 2).sum()) and fast["n_valid"] == len(idx) and fast["sum_r2"] == out["sum_r2"]
                if ok and len(idx):
                    A_ref, b_ref = O.normal_equations(J, r, None, 1.0)
                    for o in (out, fast):
                        ok = ok and np.array_equal(o["A"].astype(np.float32).view(np.uint32), A_ref.view(np.uint32)) and \
                            np.array_equal((-o["jtr"]).astype(np.float32).view(np.uint32), b_ref.view(np.uint32))
                if not ok:
                    differs("residual_jacobian", case=case, lvl=lvl, xi=xi.tolist(), over=over, size=(w, h), n_valid=(out["n_valid"], fast["n_valid"], len(idx)),
                            masks=int((out["valid"] != valid).sum()))
            else:
                J, r, idx = O.residual_jacobian_ex(a_img, b_img, gx, gy, pts, wp, L, p.z_factor, p.angle_factor, sampler=over.get("sampler", 0))
                valid[idx] = 1
                out = ctx.residual_jacobian_weighted(0, 1, lvl, pose)
                wts = {0: None, 1: O.tukey_weights, 2: O.huber_weights}[over.get("weights", 0)]
                W = wts(r) if (wts and len(r)) else None
                ok = np.array_equal(out["valid"], valid) and np.array_equal(out["r"][idx].view(np.uint32), r.view(np.uint32)) and \
                    np.array_equal(out["J"][idx].view(np.uint32), J.view(np.uint32)) and out["n_valid"] == len(idx)
                if ok and W is not None:
                    ok = np.array_equal(out["w"][idx].view(np.uint32), W.view(np.uint32))
                if ok and len(idx):
                    A_ref, b_ref = O.normal_equations(J, r, W, p.gain)
                    ok = np.array_equal(out["A"].astype(np.float32).view(np.uint32), A_ref.view(np.uint32)) and \
                        np.array_equal((-out["jtr"]).astype(np.float32).view(np.uint32), b_ref.view(np.uint32))
                if not ok:
                    # where it parts: the inputs (planes of this level read back) or the evaluation (a second call)
                    planes = dict(ref=int((ctx.get_plane(0, lvl, capi.PLANE_IMAGE) != a_img).sum()), tgt=int((ctx.get_plane(1, lvl, capi.PLANE_IMAGE) != b_img).sum()),
                                  gx=int((ctx.get_plane(0, lvl, capi.PLANE_GRADX) != gx).sum()), gy=int((ctx.get_plane(0, lvl, capi.PLANE_GRADY) != gy).sum()))
                    if depth:
                        planes["depth"] = int((ctx.get_plane(0, lvl, capi.PLANE_DEPTH) != dp).sum())
                    again = ctx.residual_jacobian_weighted(0, 1, lvl, pose)
                    differs("residual_jacobian_weighted", case=case, lvl=lvl, xi=xi.tolist(), over=over, size=(w, h), n_valid=(out["n_valid"], len(idx)),
                            masks=int((out["valid"] != valid).sum()), planes_differ=planes, second_call_n_valid=again["n_valid"],
                            second_call_masks=int((again["valid"] != valid).sum()))
    ctx.close()
print("stage fuzz seed %d: %d evaluations at extreme poses over %d contexts, %d differ from the oracle, %.0f s" % (seed, total, cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
