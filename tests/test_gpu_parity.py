"""GPU parity tests: the HIP path, called through the C ABI (libuwt_hip.so), against the CPU oracle on the same
seeded inputs and against tests/golden/*.npz.

Bars: bit-exact for integer / byte / index work (pyramid, gradients, validity masks, residuals, Σr², counts) and for
the per-pixel float terms (same pinned op order); accumulators within 2e-6 relative of the oracle's f64 sums;
poses within 1e-4 rad / 1e-4 m (north_star tolerance).
"""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4    # rad
TRANS_TOL = 1e-4  # m


@pytest.fixture(scope="module")
def capi():
    m = importlib.import_module("uw-slam_amd.capi")
    m.lib()  # raises if libuwt_hip.so is missing: no fallback
    return m


def rot_angle(qa, qb):
    """angle of R_a R_b^T from unit quaternions (x y z w): q_rel = qa * conj(qb)"""
    qa, qb = qa.astype(np.float64), qb.astype(np.float64)
    w = abs(float(np.dot(qa, qb)))
    v = qb[3] * qa[:3] - qa[3] * qb[:3] - np.cross(qa[:3], qb[:3])
    return 2.0 * np.arctan2(np.linalg.norm(v), w)


def assert_pose_close(p_gpu, p_cpu):
    assert rot_angle(p_gpu[:4], p_cpu[:4]) <= ROT_TOL, (p_gpu, p_cpu)
    assert np.linalg.norm(p_gpu[4:].astype(np.float64) - p_cpu[4:].astype(np.float64)) <= TRANS_TOL, (p_gpu, p_cpu)


def make_ctx(capi, w, h, intr, max_frames=2, max_pairs=1, **over):
    return capi.Context(capi.default_params(w, h, *intr, max_frames=max_frames, max_pairs=max_pairs, **over))


FUZZ_SEEDS = int(os.environ.get("UWT_FUZZ_SEEDS", "12"))  # raise for a longer hunt
SMALL = (64.0, 64.0, 31.5, 23.5)
MID = (131.25, 131.25, 79.5, 47.5)
TUM = (525.0, 525.0, 319.5, 239.5)


# ------------------------------------------------------------------ stages

@pytest.mark.parametrize("shape", [(12, 16), (48, 64), (30, 38), (480, 640), (2, 2)])
def test_halve_bit_exact(capi, O, shape):
    rng = np.random.default_rng(shape[0])
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    dep = rng.integers(0, 65536, shape).astype(np.uint16)
    assert np.array_equal(ctx.halve_u8(img), O.halve_u8(img))
    assert np.array_equal(ctx.halve_u16(dep), O.halve_u16(dep))
    sat = np.full(shape, 255, np.uint8)
    assert np.array_equal(ctx.halve_u8(sat), sat[::2, ::2])
    assert np.array_equal(ctx.halve_u16(np.full(shape, 65535, np.uint16)), np.full((shape[0] // 2, shape[1] // 2), 65535))


@pytest.mark.parametrize("shape", [(12, 16), (5, 5), (1, 7), (9, 1), (17, 65), (30, 40), (480, 640), (96, 130),
                                   (1, 8), (2, 4), (3, 132), (33, 260), (64, 4), (65, 128)])   # vector path: one row, ragged tiles
def test_scharr_bit_exact(capi, O, shape):
    rng = np.random.default_rng(shape[1])
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    gx, gy = ctx.scharr3(img)
    ox, oy = O.scharr3(img)
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy)
    ext = np.zeros(shape, np.uint8)
    ext[:, shape[1] // 2:] = 255  # extreme step: |g| = 48*255, no int16 saturation
    gx, gy = ctx.scharr3(ext)
    ox, oy = O.scharr3(ext)
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy)


def test_stage_kernels_on_random_shapes(capi, O):
    """Scharr and the 2x2 mean on 60 random shapes (the vector tiles' widths — multiples of 4 — and odd widths, one to a few
    hundred rows, ragged last tiles), every output compared with the oracle's."""
    rng = np.random.default_rng(77)
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    for k in range(60):
        h = int(rng.integers(1, 200))
        w = int(rng.integers(1, 90)) * 4 if k % 3 else int(rng.integers(1, 300))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        gx, gy = ctx.scharr3(img)
        ox, oy = O.scharr3(img)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy), (h, w)
        if h >= 2 and w >= 2:
            ev = img[: h // 2 * 2, : w // 2 * 2]
            assert np.array_equal(ctx.halve_u8(ev), O.halve_u8(ev)), (h, w)
            d16 = rng.integers(0, 65536, ev.shape).astype(np.uint16)
            assert np.array_equal(ctx.halve_u16(d16), O.halve_u16(d16)), (h, w)


def test_golden_stage_vectors(capi, golden):
    g = golden("stages.npz")
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    assert np.array_equal(ctx.halve_u8(g["img"]), g["half"])
    assert np.array_equal(ctx.halve_u16(g["dep"]), g["dep_half"])
    gx, gy = ctx.scharr3(g["img"])
    assert np.array_equal(gx, g["gx"]) and np.array_equal(gy, g["gy"])


def test_batched_pyramids_and_gradients_in_slots(capi, O, synth):
    w, h = 160, 96
    ctx = make_ctx(capi, w, h, MID, max_frames=5, max_pairs=2, has_depth=1)
    rng = np.random.default_rng(0)
    frames = np.stack([synth.texture(w, h, seed=s) for s in range(5)])
    depth = rng.integers(0, 65536, frames.shape).astype(np.uint16)
    ctx.upload_frames(0, frames, depth)
    ctx.build_pyramids(0, 5)
    ctx.apply_gradient(0, 5)
    for s in range(5):
        im, dp = frames[s], depth[s]
        for l in range(5):
            if l:
                im, dp = O.halve_u8(im), O.halve_u16(dp)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_IMAGE), im)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_DEPTH), dp)
            gx, gy = O.scharr3(im)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_GRADX), gx)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_GRADY), gy)
    # strided single-frame upload (cv::Mat::step) lands the same bytes
    big = np.zeros((h, w + 24), np.uint8)
    big[:, :w] = frames[3]
    bigd = np.zeros((h, w + 8), np.uint16)
    bigd[:, :w] = depth[3]
    ctx.set_frame(1, big[:, :w], bigd[:, :w])
    assert np.array_equal(ctx.get_plane(1, 0, capi.PLANE_IMAGE), frames[3])
    assert np.array_equal(ctx.get_plane(1, 0, capi.PLANE_DEPTH), depth[3])
    # ... and so does a view into the corner of a parent (the span of its rows ends with the parent's last byte)
    ctx.set_frame(2, np.pad(frames[4], ((9, 0), (40, 0)))[9:, 40:], np.pad(depth[4], ((9, 0), (40, 0)))[9:, 40:])
    assert np.array_equal(ctx.get_plane(2, 0, capi.PLANE_IMAGE), frames[4])
    assert np.array_equal(ctx.get_plane(2, 0, capi.PLANE_DEPTH), depth[4])


@pytest.mark.parametrize("shape", [(160, 96, 5), (640, 480, 5), (640, 480, 4), (72, 56, 3), (200, 120, 4), (1280, 960, 5), (192, 128, 7),
                                   (70, 50, 2), (36, 20, 3)])
@pytest.mark.parametrize("fused", [True, False])
def test_one_launch_pyramid_and_gradient_forms(capi, O, synth, monkeypatch, shape, fused):
    """A frame or a few take their whole pyramid in one launch per plane (k_pyramid_all) and the gradients of every level in
    one launch (k_scharr3_levels); larger sets take a launch per level (uwt_tuning::fused_stages = 0 forces that form).  Same
    integers either way: tile borders (sizes that are no multiple of 64 / 128), levels whose width is no multiple of 4
    (scalar gradient tile), 2 to 7 levels, frame ranges and the slot-list form used by the tracker."""
    w, h, levels = shape
    f = 525.0 * w / 640.0
    ctx = capi.Context(capi.default_params(w, h, f, f, w / 2 - 0.5, h / 2 - 0.5, n_levels=levels, first_level=levels - 1,
                                           last_level=0, max_frames=4, max_pairs=2, has_depth=1), tuning=dict(fused_stages=int(fused)))
    rng = np.random.default_rng(w + levels)
    frames = np.stack([synth.texture(w, h, seed=300 + s) for s in range(4)])
    frames[3] = rng.integers(0, 256, (h, w)).astype(np.uint8)       # white noise: every rounding case of the 2x2 mean
    depth = rng.integers(0, 65536, frames.shape).astype(np.uint16)
    ctx.upload_frames(0, frames, depth)
    ctx.build_pyramids(1, 3)          # a range that does not start at slot 0
    ctx.apply_gradient(1, 3)
    for s in range(1, 4):
        im, dp = frames[s], depth[s]
        for l in range(levels):
            if l:
                im, dp = O.halve_u8(im), O.halve_u16(dp)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_IMAGE), im), (s, l)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_DEPTH), dp), (s, l)
            gx, gy = O.scharr3(im)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_GRADX), gx), (s, l)
            assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_GRADY), gy), (s, l)
    ctx.close()


@pytest.mark.parametrize("shape", [(640, 480, 4), (320, 240, 5), (128, 64, 4), (144, 72, 4), (1280, 960, 6), (160, 96, 3),
                                   (48, 24, 4), (16, 8, 4)])   # (the last two: level widths 6 and 2, odd frame strides)
def test_batch_pyramids_in_one_pass(capi, O, monkeypatch, shape):
    """More than a few frames take levels 1..3 of their pyramids in one pass over level 0 (k_pyramid_batch: level-0 width a
    multiple of 16, height of 8, four levels or more; the levels beyond and every other shape by the per-level chain):
    white-noise frames (every rounding case of the 2x2 mean), u8 and u16, a range that does not start at slot 0, tiles that
    end inside the 128 x 64 block — every level equal to the oracle's chain of halvings, with and without the batch form."""
    w, h, levels = shape
    f = 525.0 * w / 640.0
    n = 11
    rng = np.random.default_rng(w * 7 + levels)
    frames = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    depth = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
    for switch in ("0", "1"):
        ctx = capi.Context(capi.default_params(w, h, f, f, w / 2 - 0.5, h / 2 - 0.5, n_levels=levels, first_level=levels - 1,
                                               last_level=0, max_frames=n, max_pairs=2, has_depth=1),
                           tuning=dict(pyramid_batch=int(switch == "0")))
        ctx.upload_frames(0, frames, depth)
        ctx.build_pyramids(1, n - 1)
        for s in (1, 5, n - 1):
            im, dp = frames[s], depth[s]
            for l in range(1, levels):
                im, dp = O.halve_u8(im), O.halve_u16(dp)
                assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_IMAGE), im), (switch, s, l)
                assert np.array_equal(ctx.get_plane(s, l, capi.PLANE_DEPTH), dp), (switch, s, l)
        ctx.close()


def test_level_info_matches_oracle(capi, O):
    ctx = make_ctx(capi, 640, 480, TUM)
    p = O.default_params(640, 480, *TUM)
    for l in range(5):
        a, b = ctx.level_info(l), O.level_intrinsics(p, l)
        for f in ("w", "h", "fx", "fy", "cx", "cy", "invfx", "invfy"):
            assert getattr(a, f) == getattr(b, f)


# ------------------------------------------------------------------ SE(3), solve

def test_se3_ops_match_oracle_and_golden(capi, O, golden):
    g = golden("se3.npz")
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    e = g["exp"]
    for i, xi in enumerate(g["xi"]):
        assert np.array_equal(ctx.se3_exp(xi), e[i])
        assert np.array_equal(ctx.se3_mul(e[i], e[(i + 1) % len(e)]), g["mul"][i])
        assert np.array_equal(ctx.se3_handoff(e[i], 0), g["handoff"][i])
        assert np.array_equal(ctx.se3_handoff(e[i], 1), g["handoff_t"][i])
        assert np.array_equal(ctx.se3_matrix(e[i]), g["mat"][i])
    with pytest.raises(capi.UwtError):
        ctx.se3_handoff(np.zeros(7, np.float32))
    rng = np.random.default_rng(77)
    for xi in rng.normal(0, 0.3, (50, 6)).astype(np.float32):
        assert np.array_equal(ctx.se3_exp(xi), O.se3_exp(xi))


def test_solve_delta_matches_oracle_and_golden(capi, O, golden):
    g = golden("inv6.npz")
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    for A, inv, ok, b, d in zip(g["A"], g["inv"], g["ok"], g["b"], g["delta"]):
        dd, Ai, good = ctx.solve_delta(A, b)
        assert good == bool(ok)
        assert np.array_equal(Ai, inv) and np.array_equal(dd, d)
    # singular ⇒ zero inverse ⇒ δ = 0, no error raised (cv::Mat::inv semantics)
    dd, Ai, good = ctx.solve_delta(np.zeros((6, 6), np.float32), np.ones(6, np.float32))
    assert not good and not Ai.any() and not dd.any()


# ------------------------------------------------------------------ warp + per-pixel terms

def test_warp_table_bit_exact(capi, O):
    ctx = make_ctx(capi, 160, 96, MID)
    p = O.default_params(160, 96, *MID)
    rng = np.random.default_rng(3)
    for lvl in (0, 2, 4):
        L = O.level_intrinsics(p, lvl)
        dep = rng.integers(0, 40000, (L.h, L.w)).astype(np.uint16)
        dep[rng.random(dep.shape) < 0.1] = 0
        pts = O.dense_points(dep, L.w, L.h, lvl)
        pose = O.se3_exp(rng.normal(0, 0.02, 6).astype(np.float32))
        a = ctx.warp(lvl, pts, pose)
        b = O.warp(pts, pose, L)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # bitwise, NaNs included


@pytest.mark.one_arith
def test_warp_folds_the_rigid_product_as_the_oracle_does(capi, O):
    # the one-point vector of tests/test_oracle.py that separates "s0 += s1 + s2 + s3" from the left-to-right fold
    from test_oracle import gemm_fold_vector
    pose, pts, L, lo, hi = gemm_fold_vector(O)
    ctx = make_ctx(capi, 64, 64, (1.0, 1.0, 0.0, 0.0))
    a = ctx.warp(0, pts, pose)
    assert a[0][2] == lo and a[0][2] != hi
    assert np.array_equal(a.view(np.uint32), O.warp(pts, pose, L).view(np.uint32))
    # and the probe tools/ref_dump builds for a generic pose (what a reference build is asked)
    from test_ref_vectors import fold_probe
    pose = O.se3_exp(np.array([0.01, -0.02, 0.015, 0.004, -0.003, 0.002], np.float32))
    fp = fold_probe(O.se3_matrix(pose).reshape(4, 4)[2, :3])
    a = ctx.warp(0, np.array([[fp[0], fp[1], fp[2], 0.0]], np.float32), pose)
    assert a[0][2] == fp[3]


def _load_pair(ctx, ref, tgt, depth=None):
    d = None if depth is None else np.stack([depth, depth])
    ctx.upload_frames(0, np.stack([ref, tgt]), d)
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)


@pytest.mark.parametrize("case", ["nodepth", "depth", "factors"])
def test_residual_jacobian_per_pixel_bit_exact(capi, O, synth, case):
    w, h = 160, 96
    over = {}
    if case == "depth":
        over["has_depth"] = 1
    if case == "factors":
        over.update(z_factor=0.002, angle_factor=0.5)
    ctx = make_ctx(capi, w, h, MID, **over)
    ref, tgt, depth, _, _ = synth.render_pair(w, h, *MID, seed=31, max_t=0.03, max_deg=2.0, with_depth=(case == "depth"), z=1.2)
    _load_pair(ctx, ref, tgt, depth)
    p = O.default_params(w, h, *MID, **over)
    rng = np.random.default_rng(8)
    a_img, b_img, dp = ref, tgt, depth
    for lvl in range(5):
        if lvl:
            a_img, b_img = O.halve_u8(a_img), O.halve_u8(b_img)
            if dp is not None:
                dp = O.halve_u16(dp)
        L = O.level_intrinsics(p, lvl)
        gx, gy = O.scharr3(a_img)
        pts = O.dense_points(dp, L.w, L.h, lvl)
        # poses that push part of the image out of bounds so the border tests are exercised
        pose = O.se3_exp((rng.normal(0, 1, 6) * [0.05, 0.05, 0.02, 0.01, 0.01, 0.03]).astype(np.float32))
        wp = O.warp(pts, pose, L)
        J, r, idx = O.residual_jacobian(a_img, b_img, gx, gy, pts, wp, L, p.z_factor, p.angle_factor)
        out = ctx.residual_jacobian(0, 1, lvl, pose)
        valid = np.zeros(L.w * L.h, np.uint8)
        valid[idx] = 1
        assert np.array_equal(out["valid"], valid)
        assert 0 < len(idx) < L.w * L.h or lvl == 4
        assert np.array_equal(out["r"][idx], r)
        assert np.array_equal(out["J"][idx].view(np.uint32), J.view(np.uint32))
        assert out["n_valid"] == len(idx)
        assert out["sum_r2"] == int((r.astype(np.int64) ** 2).sum())
        Jd = J.astype(np.float64)
        A_ref = Jd.T @ Jd
        jtr_ref = Jd.T @ r.astype(np.float64)
        scale = np.sqrt(np.outer(np.diag(A_ref), np.diag(A_ref)))
        assert np.abs(out["A"] - A_ref).max() <= 2e-6 * scale.max() and (np.abs(out["A"] - A_ref) <= 2e-6 * scale + 1e-30).all()
        mag = np.abs(Jd).T @ np.abs(r.astype(np.float64))
        assert (np.abs(out["jtr"] - jtr_ref) <= 2e-6 * mag + 1e-30).all()


def test_residual_all_invalid_and_nan_safe(capi, O, synth):
    w, h = 64, 48
    ctx = make_ctx(capi, w, h, SMALL, n_levels=3, first_level=2, last_level=0)
    ref, tgt = synth.shifted_pair(w, h, seed=2)
    _load_pair(ctx, ref, tgt)
    far = np.array([0, 0, 0, 1, 50.0, 0, 0], np.float32)  # everything projects off-image
    out = ctx.residual_jacobian(0, 1, 0, far)
    assert out["n_valid"] == 0 and out["sum_r2"] == 0 and not out["A"].any() and not out["valid"].any()
    behind = np.array([0, 0, 0, 1, 0, 0, -1.0], np.float32)  # z2 == 0 exactly for the z = 1 plane
    out = ctx.residual_jacobian(0, 1, 0, behind)
    assert out["n_valid"] == 0


# ------------------------------------------------------------------ LS mirror

def test_ls_accumulate_matches_oracle_ls(capi, O, golden):
    g = golden("ls.npz")
    ctx = make_ctx(capi, 64, 48, SMALL, n_levels=3, first_level=2, last_level=0)
    A, b, err, n = ctx.ls_accumulate(g["J"], g["r"], g["w"], divide=True)
    assert n == 16
    # round 6: every accumulator's f32 chain runs in the reference's order (k_ls_sequential) — the oracle's LS bit for bit
    assert np.array_equal(A, g["A_scalar"]) and np.array_equal(b, g["b_scalar"]) and err == float(g["err_scalar"])
    rng = np.random.default_rng(12)
    for npts in (5000, 4999, 1, 24200):
        J = rng.normal(0, 30, (npts, 6)).astype(np.float32)
        r = rng.integers(-255, 256, npts).astype(np.float32)
        w = rng.uniform(0, 1, npts).astype(np.float32)
        for div in (False, True):
            A, b, err, n = ctx.ls_accumulate(J, r, w, divide=div)
            ls = O.ls_new()
            for i in range(npts):
                O.ls_update(ls, J[i], r[i], w[i])
            A0, b0, e0, n0 = O.ls_finish(ls, divide=div)
            assert n == n0 == npts
            assert np.array_equal(A.view(np.uint32), A0.view(np.uint32)) and np.array_equal(b.view(np.uint32), b0.view(np.uint32))
            assert np.float32(err) == np.float32(e0) and np.array_equal(A, A.T)
        if npts % 4 == 0:
            for quirk in (True, False):
                A4, b4, e4, n4 = ctx.ls_accumulate_sse(J, r, w, divide=False, count_quirk=quirk)
                ls = O.ls_new()
                for k in range(0, npts, 4):
                    O.ls_update4(ls, J[k:k + 4].T, r[k:k + 4], w[k:k + 4], quirk_plus6=quirk)
                A0, b0, e0, n0 = O.ls_finish(ls, divide=False)
                assert n4 == n0 and np.array_equal(A4.view(np.uint32), A0.view(np.uint32)) and np.array_equal(b4.view(np.uint32), b0.view(np.uint32))
                assert np.float32(e4) == np.float32(e0)
    Jd, wd, rd = J.astype(np.float64), w.astype(np.float64), r.astype(np.float64)
    Aex = (Jd * wd[:, None]).T @ Jd
    A, b, err, n = ctx.ls_accumulate(J, r, w, divide=False)
    assert np.abs(A - Aex).max() <= 2e-4 * np.abs(Aex).max()          # and the f32 chain is what it is against the exact sum
    A, b, err, n = ctx.ls_accumulate(np.zeros((0, 6), np.float32), np.zeros(0, np.float32))
    assert n == 0 and not A.any() and not b.any() and err == 0.0   # LS::initialize state
    # LS::updateSSE form incl. the "+= 6 per 4 points" quirk (src/LeastSquares.cpp:201)
    A4, b4, e4, n4 = ctx.ls_accumulate_sse(g["J"], g["r"], g["w"], divide=False, count_quirk=True)
    assert n4 == 24 == int(g["n_sse"])
    assert np.array_equal(A4, g["A_sse"]) and np.array_equal(b4, g["b_sse"]) and e4 == float(g["err_sse"])
    assert ctx.ls_accumulate_sse(g["J"], g["r"], g["w"], count_quirk=False)[3] == 16
    with pytest.raises(capi.UwtError):
        ctx.ls_accumulate_sse(g["J"][:5], g["r"][:5], g["w"][:5])


# ------------------------------------------------------------------ whole alignment

GOLDEN_PAIRS = ["pair_64x48_ref", "pair_64x48_fixed", "pair_160x96_ref5", "pair_160x96_fixed", "pair_160x96_depth",
                "pair_160x96_features"]


def _golden_over(g):
    over = {}
    for k, v in zip(g["over_keys"], g["over_vals"]):
        over[str(k)] = float(v) if str(k) in ("gain", "z_factor", "angle_factor", "epsilon") else int(v)
    return over


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_alignment_matches_golden_trace(capi, golden, name):
    g = golden(name)
    h, w = g["ref"].shape
    over = _golden_over(g)
    depth = g["depth"] if "depth" in g else None
    if depth is not None:
        over["has_depth"] = 1
    ctx = make_ctx(capi, w, h, [float(v) for v in g["intr"]], **over)
    _load_pair(ctx, g["ref"], g["tgt"], depth)
    poses, stats = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
    assert stats[0]["status"] == int(g["status"]) == 0
    assert stats[0]["iterations"] == len(g["trace_level"])            # same termination decisions
    # the last evaluation sees a pose that differs from the oracle's by rounding; nearest-neighbour sampling may
    # flip a handful of pixels, so its count / error are compared loosely (the pose bar is what counts)
    nv = int(g["trace_n_valid"][-1])
    assert abs(stats[0]["n_valid"] - nv) <= max(2, nv // 500)
    assert stats[0]["error"] == pytest.approx(float(g["trace_error"][-1]), rel=2e-2)
    assert_pose_close(poses[0], g["pose"])
    # first iteration of the coarsest level: accumulators vs the golden A, b (identity pose ⇒ same pixel set)
    lvl = int(g["trace_level"][0])
    out = ctx.residual_jacobian(0, 1, lvl, np.array([0, 0, 0, 1, 0, 0, 0], np.float32), dump=False)
    assert out["n_valid"] == int(g["trace_n_valid"][0]) and out["sum_r2"] == int(g["trace_sum_r2"][0])
    A0 = g["trace_A"][0].astype(np.float64)
    sc = np.sqrt(np.outer(np.diag(A0), np.diag(A0)))
    assert (np.abs(out["A"] - A0) <= 3e-6 * sc + 1e-30).all()


@pytest.mark.parametrize("mode", ["reference", "fixed"])
def test_alignment_matches_oracle_live_seeds(capi, O, synth, mode):
    w, h = 160, 96
    over = dict() if mode == "reference" else dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
    n = 12
    ctx = make_ctx(capi, w, h, MID, max_frames=2 * n, max_pairs=n, **over)
    p = O.default_params(w, h, *MID, **over)
    frames, cpu = [], []
    for s in range(n):
        ref, tgt, _, _, _ = synth.render_pair(w, h, *MID, seed=100 + s, max_t=0.02, max_deg=1.0)
        frames += [ref, tgt]
        st, pose, tr = O.align_pair(p, ref, tgt, want_trace=True)
        cpu.append((st, pose, len(tr)))
    ctx.upload_frames(0, np.stack(frames))
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)
    poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1, raise_on_pair_failure=True)
    for i in range(n):
        assert stats[i]["status"] == cpu[i][0] == 0
        assert stats[i]["iterations"] == cpu[i][2]
        assert_pose_close(poses[i], cpu[i][1])


def test_identical_frames_identity_and_status_codes(capi, synth):
    w, h = 160, 96
    img = synth.texture(w, h, seed=9)
    ctx = make_ctx(capi, w, h, MID)
    _load_pair(ctx, img, img)
    poses, stats = ctx.estimate_pose_batch([0], [1])
    assert np.allclose(poses[0], [0, 0, 0, 1, 0, 0, 0], atol=1e-7)
    assert stats[0]["iterations"] == 8 and stats[0]["error"] == 0.0   # exit at k = 1 on every level (4..1)
    # all depths invalid ⇒ no valid points ⇒ status code instead of the reference's cv::Exception
    ctx = make_ctx(capi, w, h, MID, has_depth=1)
    _load_pair(ctx, img, img, np.zeros((h, w), np.uint16))
    poses, stats = ctx.estimate_pose_batch([0], [1])
    assert stats[0]["status"] == capi.ERR_NO_VALID_POINTS
    with pytest.raises(capi.UwtError):
        ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
    with pytest.raises(capi.UwtError):
        ctx.estimate_pose_batch([0], [7])          # slot out of range
    with pytest.raises(capi.UwtError) as e:
        ctx.estimate_pose_batch([0, 0], [1, 1])    # exceeds max_pairs
    assert e.value.status == capi.ERR_CAPACITY
    capi.Context(capi.default_params(100, 96, *MID)).close()   # any size is a frame size (round 6: 100 / 50 / 25 / 12 / 6 wide) ...
    with pytest.raises(capi.UwtError):
        capi.Context(capi.default_params(12, 96, *MID))   # ... as long as the coarsest level has a point grid: 12 >> 4 = 0


# ------------------------------------------------------------------ full size (BASELINE configs): properties

def test_full_size_640x480_properties(capi, O, synth):
    """640x480, 4 levels, 10 iterations/level (the bench workload): determinism, batch-size invariance, oracle
    parity on one pair, identity on identical frames."""
    w, h = 640, 480
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0, has_depth=1)
    n = 6
    ctx = make_ctx(capi, w, h, TUM, max_frames=2 * n, max_pairs=n, **over)
    frames, depths = [], []
    for s in range(n - 1):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *TUM, seed=500 + s, z=1.0 + 0.05 * s, with_depth=True)
        frames += [ref, tgt]
        depths += [dep, dep]
    frames += [frames[0], frames[0]]
    depths += [depths[0], depths[0]]
    ctx.upload_frames(0, np.stack(frames), np.stack(depths))
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)
    ref_s, tgt_s = np.arange(n) * 2, np.arange(n) * 2 + 1
    p1, s1 = ctx.estimate_pose_batch(ref_s, tgt_s, raise_on_pair_failure=True)
    p2, _ = ctx.estimate_pose_batch(ref_s, tgt_s, raise_on_pair_failure=True)
    assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32))                 # run-to-run bit identical
    p3, _ = ctx.estimate_pose_batch(ref_s[2:3], tgt_s[2:3], raise_on_pair_failure=True)
    assert np.array_equal(p3[0].view(np.uint32), p1[2].view(np.uint32))           # alone == inside a batch
    assert np.allclose(p1[n - 1], [0, 0, 0, 1, 0, 0, 0], atol=1e-7)               # identical frames
    assert all(s["iterations"] == 40 for s in s1)
    po = O.default_params(w, h, *TUM, **over)
    st, pose_cpu, _ = O.align_pair(po, frames[2], frames[3], depths[2])
    assert st == 0
    assert_pose_close(p1[1], pose_cpu)


def test_full_size_1280x960_5_levels(capi, O, synth):
    w, h = 1280, 960
    intr = (1050.0, 1050.0, 639.5, 479.5)
    over = dict(n_levels=5, first_level=4, last_level=0, max_iters=10, early_exit=0)
    ctx = make_ctx(capi, w, h, intr, max_frames=2, max_pairs=1, **over)
    ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=900)
    _load_pair(ctx, ref, tgt)
    poses, stats = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
    assert stats[0]["iterations"] == 50
    st, pose_cpu, _ = O.align_pair(O.default_params(w, h, *intr, **over), ref, tgt)
    assert st == 0
    assert_pose_close(poses[0], pose_cpu)


def test_track_batch_async_device_outputs_match_sync_path(capi, O, synth):
    """uwt_track_batch_async (pyramids + reference-only gradients + alignment, poses/stats to device memory) gives
    bit-identical poses to the step-by-step synchronous entry points and to the oracle."""
    import torch
    w, h, n = 160, 96, 80
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0)
    ctx = make_ctx(capi, w, h, MID, max_frames=2 * n, max_pairs=n, **over)
    frames = []
    for s in range(n):
        ref, tgt, _, _, _ = synth.render_pair(w, h, *MID, seed=3000 + s)
        frames += [ref, tgt]
    frames = np.stack(frames)
    ref_s, tgt_s = np.arange(n) * 2, np.arange(n) * 2 + 1
    ctx.upload_frames(0, frames)
    d_poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    d_stats = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # torch's fill kernels run on torch's stream, not on the context's streams
    ctx.track_batch_async(0, 2 * n, ref_s, tgt_s, d_poses.data_ptr(), d_stats.data_ptr(), grad_refs_only=True)
    ctx.sync()
    a = d_poses.cpu().numpy()
    st = d_stats.cpu().numpy()
    assert (st[:, 0] == 0).all() and (st[:, 1] == 20).all()
    # target slots never had gradients computed in this mode; reference slots did
    gx, _ = O.scharr3(frames[0])
    assert np.array_equal(ctx.get_plane(0, 0, capi.PLANE_GRADX), gx)
    ctx2 = make_ctx(capi, w, h, MID, max_frames=2 * n, max_pairs=n, **over)
    ctx2.upload_frames(0, frames)
    ctx2.build_pyramids(0, 2 * n)
    ctx2.apply_gradient(0, 2 * n)
    b, _ = ctx2.estimate_pose_batch(ref_s, tgt_s, raise_on_pair_failure=True)
    p = O.default_params(w, h, *MID, **over)
    cpu = np.stack([O.align_pair(p, frames[2 * i], frames[2 * i + 1])[1] for i in range(n)])
    bad_a = [i for i in range(n) if not np.array_equal(a[i], cpu[i])]
    bad_b = [i for i in range(n) if not np.array_equal(b[i], cpu[i])]
    assert bad_a == [] and bad_b == [], "async!=cpu %s ; sync!=cpu %s ; a==b %s" % (bad_a, bad_b, np.array_equal(a, b))


def test_track_batch_async_depth_planes_for_reference_slots_only(capi, O, synth):
    """grad_refs_only: the depth pyramid (levels 1..) and the gradients are built for the pairs' reference slots only —
    the target's are never read (src/Tracker.cpp:407-408, 1266-1272) — and the poses equal the oracle's; with the flag off
    every slot of the range gets them."""
    import torch
    w, h, n = 160, 96, 6
    over = dict(n_levels=3, first_level=2, last_level=0, max_iters=4, early_exit=0, has_depth=1)
    p = O.default_params(w, h, *MID, **over)
    frames, depths, cpu = [], [], []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *MID, seed=3300 + s, with_depth=True)
        frames += [ref, tgt]
        depths += [dep, (dep // 2 + 7).astype(np.uint16)]  # a different plane in the target slot: it must not matter
        cpu.append(O.align_pair(p, ref, tgt, dep)[1])
    ref_s, tgt_s = np.arange(n) * 2, np.arange(n) * 2 + 1
    for refs_only in (True, False):
        ctx = make_ctx(capi, w, h, MID, max_frames=2 * n, max_pairs=n, **over)
        ctx.upload_frames(0, np.stack(frames), np.zeros((2 * n, h, w), np.uint16))
        ctx.build_pyramids(0, 2 * n)       # every depth level of every slot is zero now
        ctx.upload_frames(0, np.stack(frames), np.stack(depths))
        d_poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.track_batch_async(0, 2 * n, ref_s, tgt_s, d_poses.data_ptr(), None, grad_refs_only=refs_only)
        ctx.sync()
        assert np.array_equal(d_poses.cpu().numpy(), np.stack(cpu))
        assert np.array_equal(ctx.get_plane(2, 1, capi.PLANE_DEPTH), O.halve_u16(depths[2]))
        tgt_l1 = ctx.get_plane(3, 1, capi.PLANE_DEPTH)
        if refs_only:
            assert not tgt_l1.any()   # still the zeros of the first build
        else:
            assert np.array_equal(tgt_l1, O.halve_u16(depths[3]))
        assert np.array_equal(ctx.get_plane(3, 1, capi.PLANE_IMAGE), O.halve_u8(frames[3]))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 33])
@pytest.mark.parametrize("cfg", ["fixed_4lvl", "fixed_1lvl_1it", "fixed_top_only", "reference"])
def test_batch_sizes_and_schedules_match_oracle(capi, O, synth, n, cfg):
    """Odd / tiny batches through the fixed and the reference (early-exit, polled) schedules, every pose checked."""
    w, h = 64, 48
    over = dict(fixed_4lvl=dict(n_levels=3, first_level=2, last_level=0, max_iters=4, early_exit=0),
                fixed_1lvl_1it=dict(n_levels=3, first_level=0, last_level=0, max_iters=1, early_exit=0),
                fixed_top_only=dict(n_levels=3, first_level=2, last_level=2, max_iters=3, early_exit=0),
                reference=dict(n_levels=3, first_level=2, last_level=1, max_iters=50, early_exit=1))[cfg]
    ctx = make_ctx(capi, w, h, SMALL, max_frames=2 * n, max_pairs=n, **over)
    p = O.default_params(w, h, *SMALL, **over)
    frames, cpu = [], []
    for s in range(n):
        ref, tgt, _, _, _ = synth.render_pair(w, h, *SMALL, seed=7000 + s, max_t=0.02, max_deg=1.0)
        frames += [ref, tgt]
        cpu.append(O.align_pair(p, ref, tgt, want_trace=True))
    ctx.upload_frames(0, np.stack(frames))
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)
    poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1, raise_on_pair_failure=True)
    for i in range(n):
        assert cpu[i][0] == 0 and stats[i]["iterations"] == len(cpu[i][2])
        assert np.array_equal(poses[i], cpu[i][1]), (i, poses[i], cpu[i][1])


@pytest.mark.parametrize("general", [dict(), dict(weights=2), dict(weights=1), dict(sampler=1), dict(sampler=1, weights=2)],
                         ids=["identity", "huber", "tukey", "bilinear", "bilinear_huber"])
def test_streamed_plane_loads_change_no_bit(capi, O, synth, monkeypatch, general):
    """A batch whose planes exceed the caches runs the STREAM twins of the accumulation kernels (non-temporal plane loads,
    load_group).  uwt_tuning::stream_bytes = 0 makes every level of a small batch take them; a huge value none: same poses, the
    oracle's, bit for bit.  (The batch is split over two streams: the launch path the twins are dispatched from.)"""
    w, h, n = 160, 96, 12
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0, has_depth=1, **general)
    p = O.default_params(w, h, *MID, **over)
    frames, depths, cpu = [], [], []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *MID, seed=7300 + s, with_depth=True)
        frames += [ref, tgt]
        depths += [dep, dep]
        cpu.append(O.align_pair(p, ref, tgt, dep)[1])
    got = {}
    for mb in ("0", "1000000"):
        ctx = make_ctx(capi, w, h, MID, max_frames=2 * n, max_pairs=n, **over)
        ctx.set_tuning(split_min_px=1, split_min=2, stream_bytes=int(mb) << 20)   # two parts on two streams even at this size
        ctx.upload_frames(0, np.stack(frames), np.stack(depths))
        ctx.build_pyramids(0, 2 * n)
        ctx.apply_gradient(0, 2 * n)
        got[mb], _ = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1, raise_on_pair_failure=True)
        ctx.close()
    assert np.array_equal(got["0"].view(np.uint32), got["1000000"].view(np.uint32))
    for i in range(n):
        assert np.array_equal(got["0"][i].view(np.uint32), cpu[i].view(np.uint32)), i


@pytest.mark.parametrize("sched", ["fixed", "reference"])
@pytest.mark.parametrize("weights", [1, 2], ids=["tukey", "huber"])
def test_coarse_weighted_kernel_and_the_launches_agree(capi, O, synth, monkeypatch, weights, sched):
    """Coarse levels of the robust-weight path run in k_coarse_weighted (one block per pair: histogram, scale, weighted sums and
    update in LDS, a level's iterations in one launch); uwt_tuning::coarse_weighted = 0 keeps them on the scale / accumulation / update
    launches.  Same poses, same iteration counts, the oracle's — one pair per call and a batch, fixed and early-exit schedules."""
    w, h, n = 160, 96, 5
    over = dict(has_depth=1, weights=weights)
    if sched == "fixed":
        over.update(n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0)
    p = O.default_params(w, h, *MID, **over)
    frames, depths, cpu = [], [], []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *MID, seed=7400 + s, with_depth=True)
        frames += [ref, tgt]
        depths += [dep, dep]
        cpu.append(O.align_pair(p, ref, tgt, dep, want_trace=True))
    got = {}
    for mode in ("coarse", "launches"):
        ctx = make_ctx(capi, w, h, MID, max_frames=2 * n, max_pairs=n, **over)
        ctx.set_tuning(coarse_weighted=int(mode == "coarse"))
        ctx.upload_frames(0, np.stack(frames), np.stack(depths))
        ctx.build_pyramids(0, 2 * n)
        ctx.apply_gradient(0, 2 * n)
        batch, bstats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1, raise_on_pair_failure=True)
        single, sstats = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
        got[mode] = (batch.copy(), [s["iterations"] for s in bstats], single[0].copy(), sstats[0]["iterations"])
        ctx.close()
    for mode in got:
        batch, its, single, sit = got[mode]
        for i in range(n):
            assert np.array_equal(batch[i].view(np.uint32), cpu[i][1].view(np.uint32)), (mode, i)
            assert its[i] == len(cpu[i][2]), (mode, i)
        assert np.array_equal(single.view(np.uint32), cpu[0][1].view(np.uint32)) and sit == len(cpu[0][2]), mode


def test_failing_pair_does_not_disturb_its_batch(capi, O, synth):
    """One pair with no valid depth gets UWT_ERR_NO_VALID_POINTS; the other pairs of the batch are untouched."""
    w, h, n = 64, 48, 6
    over = dict(n_levels=3, first_level=2, last_level=0, max_iters=4, early_exit=0, has_depth=1)
    ctx = make_ctx(capi, w, h, SMALL, max_frames=2 * n, max_pairs=n, **over)
    p = O.default_params(w, h, *SMALL, **over)
    frames, depths, cpu = [], [], []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *SMALL, seed=7100 + s, with_depth=True)
        if s in (1, 4):
            dep = np.zeros_like(dep)
        frames += [ref, tgt]
        depths += [dep, dep]
        cpu.append(O.align_pair(p, ref, tgt, dep))
    ctx.upload_frames(0, np.stack(frames), np.stack(depths))
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)
    poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1)
    for i in range(n):
        assert stats[i]["status"] == cpu[i][0] == (capi.ERR_NO_VALID_POINTS if i in (1, 4) else 0)
        if i not in (1, 4):
            assert np.array_equal(poses[i], cpu[i][1])
        else:
            assert np.array_equal(poses[i], np.array([0, 0, 0, 1, 0, 0, 0], np.float32)) and stats[i]["iterations"] == 1


@pytest.mark.parametrize("depth", [False, True])
def test_non_square_intrinsics(capi, O, synth, depth):
    """fx != fy (EUROC-like) takes the general Jw formulation instead of the fx == fy specialisation."""
    w, h, n = 160, 96, 4
    intr = (458.654 * w / 752, 457.296 * w / 752, 367.215 * w / 752 - 8, 248.375 * w / 752 - 4)
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0, has_depth=int(depth))
    ctx = make_ctx(capi, w, h, intr, max_frames=2 * n, max_pairs=n, **over)
    p = O.default_params(w, h, *intr, **over)
    frames, depths, cpu = [], [], []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=7200 + s, with_depth=depth)
        frames += [ref, tgt]
        depths += [dep, dep]
        cpu.append(O.align_pair(p, ref, tgt, dep if depth else None))
    ctx.upload_frames(0, np.stack(frames), np.stack(depths) if depth else None)
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)
    poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1, raise_on_pair_failure=True)
    for i in range(n):
        assert cpu[i][0] == 0 and np.array_equal(poses[i], cpu[i][1])
    out = ctx.residual_jacobian(0, 1, 1, cpu[0][1])
    assert out["n_valid"] > 0


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_fuzz_per_pixel_terms_bit_exact(capi, O, seed):
    """Random sizes (vector and scalar level widths), intrinsics (square or not), depth on/off, factors, images with hard
    edges and saturated regions, large and small motions: valid masks, residuals and Jacobian rows stay bit-identical."""
    rng = np.random.default_rng(90000 + seed)
    w = int(rng.choice([32, 48, 80, 96, 112, 160, 208]))
    h = int(rng.choice([16, 32, 48, 64, 96]))
    n_levels = 3 if rng.random() < 0.5 else 2
    fx = float(np.float32(rng.uniform(0.5, 1.5) * w))
    fy = fx if rng.random() < 0.5 else float(np.float32(fx * rng.uniform(0.9, 1.1)))
    intr = (fx, fy, float(np.float32(w / 2 + rng.uniform(-5, 5))), float(np.float32(h / 2 + rng.uniform(-5, 5))))
    depth = bool(rng.random() < 0.5)
    over = dict(n_levels=n_levels, first_level=n_levels - 1, last_level=0, has_depth=int(depth))
    if rng.random() < 0.4:
        over.update(z_factor=float(np.float32(rng.uniform(0.001, 2))), angle_factor=float(np.float32(rng.uniform(0.1, 3))))
    ref = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref[: h // 3] = 255
    ref[:, : w // 5] = 0
    tgt = np.roll(ref, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(0, 1))
    dep = None
    if depth:
        dep = rng.integers(0, 40000, (h, w)).astype(np.uint16)
        dep[rng.random((h, w)) < 0.2] = 0
    ctx = make_ctx(capi, w, h, intr, **over)
    _load_pair(ctx, ref, tgt, dep)
    p = O.default_params(w, h, *intr, **over)
    a_img, b_img, dp = ref, tgt, dep
    for lvl in range(n_levels):
        if lvl:
            a_img, b_img = O.halve_u8(a_img), O.halve_u8(b_img)
            dp = O.halve_u16(dp) if depth else None
        L = O.level_intrinsics(p, lvl)
        gx, gy = O.scharr3(a_img)
        pts = O.dense_points(dp, L.w, L.h, lvl)
        scale = 10.0 ** rng.uniform(-3, -0.5)
        pose = O.se3_exp((rng.normal(0, 1, 6) * scale).astype(np.float32))
        wp = O.warp(pts, pose, L)
        J, r, idx = O.residual_jacobian(a_img, b_img, gx, gy, pts, wp, L, p.z_factor, p.angle_factor)
        out = ctx.residual_jacobian(0, 1, lvl, pose)
        valid = np.zeros(L.w * L.h, np.uint8)
        valid[idx] = 1
        assert np.array_equal(out["valid"], valid)
        assert np.array_equal(out["r"][idx], r)
        assert np.array_equal(out["J"][idx].view(np.uint32), J.view(np.uint32))
        assert out["n_valid"] == len(idx) and out["sum_r2"] == int((r.astype(np.int64) ** 2).sum())
        if len(idx):
            A_ref, b_ref = O.normal_equations(J, r, None, 1.0)
            assert np.array_equal(out["A"].astype(np.float32), A_ref)
            assert np.array_equal((-out["jtr"]).astype(np.float32), b_ref)


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_fuzz_whole_alignment_bit_identical(capi, O, synth, seed):
    """Random solver configurations (level range, iteration cap, early exit, gain, epsilon, factors, weights, sampler,
    depth, square or non-square intrinsics, vector and scalar level widths): status, iteration count, final error and
    pose bits equal the oracle's for every pair of a small batch."""
    rng = np.random.default_rng(91000 + seed)
    w = int(rng.choice([64, 96, 112, 160, 208]))
    h = int(rng.choice([32, 48, 64, 96]))
    n_levels = int(rng.integers(2, 5))
    while (w >> (n_levels - 1)) < 8 or (h >> (n_levels - 1)) < 4:
        n_levels -= 1
    first = int(rng.integers(0, n_levels))
    last = int(rng.integers(0, first + 1))
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
    if r < 0.2:
        over.update(weights=1)
    elif r < 0.3:
        over.update(weights=2)
    elif r < 0.4:
        over.update(sampler=1)
    n = 3
    ctx = make_ctx(capi, w, h, intr, max_frames=2 * n, max_pairs=n, **over)
    p = O.default_params(w, h, *intr, **over)
    frames, depths, cpu = [], [], []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=91000 + 10 * seed + s, with_depth=depth,
                                                max_t=float(rng.uniform(0.002, 0.03)), max_deg=float(rng.uniform(0.1, 1.5)))
        frames += [ref, tgt]
        depths += [dep, dep]
        cpu.append(O.align_pair(p, ref, tgt, dep if depth else None, want_trace=True))
    ctx.upload_frames(0, np.stack(frames), np.stack(depths) if depth else None)
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)
    poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1)
    for i in range(n):
        assert stats[i]["status"] == cpu[i][0], (over, i)
        if cpu[i][0] == 0:
            assert stats[i]["iterations"] == len(cpu[i][2]), (over, i)
            assert np.array_equal(poses[i], cpu[i][1]), (over, i, poses[i], cpu[i][1])


def test_large_batch_takes_the_overlapped_gradient_path(capi, O, synth):
    """From 768 pairs up uwt_track_batch_async computes the finer levels' gradients on a side stream beside the coarse
    iterations.  800 pairs (8 distinct, tiled, with depth): every copy bit-identical to the oracle, twice in a row."""
    import torch
    w, h, n, u = 64, 48, 800, 8
    over = dict(n_levels=3, first_level=2, last_level=0, max_iters=4, early_exit=0, has_depth=1)
    p = O.default_params(w, h, *SMALL, **over)
    refs, tgts, deps, cpu = [], [], [], []
    for s in range(u):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *SMALL, seed=8800 + s, with_depth=True, max_t=0.02, max_deg=1.0)
        refs.append(ref); tgts.append(tgt); deps.append(dep)
        cpu.append(O.align_pair(p, ref, tgt, dep)[1])
    idx = np.arange(n) % u
    frames = np.empty((2 * n, h, w), np.uint8)
    frames[0::2] = np.stack(refs)[idx]
    frames[1::2] = np.stack(tgts)[idx]
    depth = np.empty((2 * n, h, w), np.uint16)
    depth[0::2] = np.stack(deps)[idx]
    depth[1::2] = depth[0::2]
    ctx = make_ctx(capi, w, h, SMALL, max_frames=2 * n, max_pairs=n, **over)
    ctx.upload_frames(0, frames, depth)
    d_poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    expect = np.stack(cpu)[idx]
    for _ in range(2):
        ctx.track_batch_async(0, 2 * n, np.arange(n) * 2, np.arange(n) * 2 + 1, d_poses.data_ptr())
        ctx.sync()
        assert np.array_equal(d_poses.cpu().numpy(), expect)
        d_poses.zero_()
        torch.cuda.synchronize()
