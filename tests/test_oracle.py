"""CPU tests of the oracle (oracle/uwt_oracle.c): independent cross-checks against numpy/scipy math,
known-answer tests derived from the reference's semantics, and regression against tests/golden/*.npz.

The reference holds no golden vectors for this path — "parity unpinned"; these tests pin the
restatement to independent mathematics instead.
"""
import os

import numpy as np
import pytest
from scipy import linalg, ndimage


def hat6(xi):
    u, w = xi[:3], xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


# ---------------------------------------------------------------- SE(3) (sophus/se3.hpp, so3.hpp)

def test_se3_exp_matches_expm(O, golden):
    g = golden("se3.npz")
    for xi, e in zip(g["xi"], g["exp"]):
        got = O.se3_exp(xi)
        assert np.array_equal(got, e)  # regression, bit-exact
        T = O.se3_matrix(got).astype(np.float64)
        ref = linalg.expm(hat6(xi.astype(np.float64)))
        assert np.allclose(T, ref, atol=2e-6 * max(1.0, np.abs(ref).max()))
        assert abs(np.linalg.norm(got[:4]) - 1) < 1e-6


def test_se3_exp_small_angle_branch(O):
    # so3.hpp:548-553 Taylor branch; se3.hpp:733-735 V = R
    xi = np.array([0.3, -0.2, 0.1, 1e-7, -2e-7, 3e-7], np.float32)
    p = O.se3_exp(xi)
    assert np.allclose(p[4:], xi[:3], atol=1e-6)
    assert np.allclose(p[:3], 0.5 * xi[3:], atol=1e-12)
    assert p[3] == np.float32(1.0)
    ident = O.se3_exp(np.zeros(6, np.float32))
    assert np.array_equal(ident, np.array([0, 0, 0, 1, 0, 0, 0], np.float32))


def test_se3_mul_matches_matrix_product(O, golden):
    g = golden("se3.npz")
    e = g["exp"]
    for i in range(len(e)):
        a, b = e[i], e[(i + 1) % len(e)]
        got = O.se3_mul(a, b)
        assert np.array_equal(got, g["mul"][i])
        ref = O.se3_matrix(a).astype(np.float64) @ O.se3_matrix(b).astype(np.float64)
        assert np.allclose(O.se3_matrix(got), ref, atol=5e-6 * max(1.0, np.abs(ref).max()))


def test_se3_handoff(O, golden):
    # Tracker.cpp:580-590: q.xyz *= 2 then normalise; t unchanged (EstimatePose) or doubled (EstimatePoseFeatures :856)
    g = golden("se3.npz")
    for e, h0, h1 in zip(g["exp"], g["handoff"], g["handoff_t"]):
        a = O.se3_handoff(e, 0)
        b = O.se3_handoff(e, 1)
        assert np.array_equal(a, h0) and np.array_equal(b, h1)
        q = np.array([2 * e[0], 2 * e[1], 2 * e[2], e[3]], np.float64)
        assert np.allclose(a[:4], q / np.linalg.norm(q), atol=1e-6)
        assert np.array_equal(a[4:], e[4:])
        assert np.array_equal(b[4:], 2 * e[4:])
    with pytest.raises(ValueError):
        O.se3_handoff(np.zeros(7, np.float32))


# ---------------------------------------------------------------- 6x6 LU inverse (cv::Mat::inv, Tracker.cpp:564)

def test_inv6_matches_numpy_and_singular_is_zero(O, golden):
    g = golden("inv6.npz")
    for A, inv, ok, b, d in zip(g["A"], g["inv"], g["ok"], g["b"], g["delta"]):
        X, good = O.inv6(A)
        assert good == bool(ok)
        assert np.array_equal(X, inv)
        assert np.array_equal(O.solve_delta(A, b), d)
        if good:
            ref = np.linalg.inv(A.astype(np.float64))
            # column/row scaled comparison (A is badly scaled, like the tracker's JᵀJ)
            s = np.sqrt(np.diag(A).astype(np.float64))
            assert np.allclose(X * np.outer(s, s), ref * np.outer(s, s), atol=5e-3)
            assert np.allclose(d, ref @ b.astype(np.float64), rtol=2e-2, atol=1e-6 * np.abs(ref @ b).max())
        else:
            assert not X.any() and not d.any()  # singular ⇒ zeros ⇒ δ = 0
    X, good = O.inv6(np.eye(6, dtype=np.float32) * 2)
    assert good and np.array_equal(X, np.eye(6, dtype=np.float32) * 0.5)


# ---------------------------------------------------------------- pyramid / gradients (System.cpp:246-251, Tracker.cpp:1133-1142)

def test_halve_is_rounded_2x2_mean(O, golden):
    g = golden("stages.npz")
    img = g["img"]
    ref = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(O.halve_u8(img), ref)
    assert np.array_equal(O.halve_u8(img), g["half"])
    dep = g["dep"]
    refd = (dep[0::2, 0::2].astype(np.int64) + dep[0::2, 1::2] + dep[1::2, 0::2] + dep[1::2, 1::2] + 2) >> 2
    assert np.array_equal(O.halve_u16(dep), refd)
    assert np.array_equal(O.halve_u16(dep), g["dep_half"])


def test_scharr3_matches_scipy_correlate(O, golden):
    g = golden("stages.npz")
    img = g["img"]
    kx = 3 * np.array([[-3, 0, 3], [-10, 0, 10], [-3, 0, 3]])
    gx_ref = ndimage.correlate(img.astype(np.int32), kx, mode="mirror")  # scipy 'mirror' == OpenCV BORDER_REFLECT_101
    gy_ref = ndimage.correlate(img.astype(np.int32), kx.T, mode="mirror")
    gx, gy = O.scharr3(img)
    assert np.array_equal(gx, gx_ref) and np.array_equal(gy, gy_ref)
    assert np.array_equal(gx, g["gx"]) and np.array_equal(gy, g["gy"])
    # hand-computable: horizontal ramp I = 10x ⇒ gx = 3·(3+10+3)·20 = 960 in the interior, gy = 0
    ramp = np.tile((10 * np.arange(5)).astype(np.uint8), (5, 1))
    gx, gy = O.scharr3(ramp)
    assert (gx[:, 1:4] == 960).all() and (gx[:, 0] == 0).all() and (gx[:, 4] == 0).all() and not gy.any()
    assert np.array_equal(O.gradient_mag(g["gx"], g["gy"]), g["mag"])
    assert O.gradient_mag(np.array([[3]], np.int16), np.array([[2]], np.int16))[0, 0] == 2  # 2.5 → even


# ---------------------------------------------------------------- intrinsics / points / warp / per-point terms

def test_level_intrinsics(O):
    # Tracker.cpp:313-331
    p = O.default_params(640, 480, 525.0, 525.0, 319.5, 239.5)
    for l in range(5):
        L = O.level_intrinsics(p, l)
        assert (L.w, L.h) == (640 >> l, 480 >> l)
        assert L.fx == np.float32(525.0 / 2 ** l)
        assert L.cx == np.float32((319.5 + 0.5) / 2 ** l - 0.5)
        assert L.invfx == np.float32(1) / np.float32(L.fx)
    with pytest.raises(ValueError):
        O.level_intrinsics(p, 5)


def test_dense_points(O):
    # Tracker.cpp:1259-1310; depth read as signed short, invalid ⇒ [0 0 1 0]
    dep = np.array([[0, 5000], [40000, 1]], np.uint16)
    pts = O.dense_points(dep, 2, 2, 1)
    f = np.float32(np.float64(np.float32(0.0002)) / 2.0)
    assert np.array_equal(pts[0], [0, 0, 1, 0])
    assert np.array_equal(pts[1], np.array([1, 0, np.float32(5000) * f, 1], np.float32))
    assert np.array_equal(pts[2], [0, 0, 1, 0])  # 40000 as int16 is negative
    assert np.array_equal(pts[3], np.array([1, 1, np.float32(1) * f, 1], np.float32))
    pts = O.dense_points(None, 3, 2, 0)
    assert np.array_equal(pts[:, 2], np.ones(6)) and np.array_equal(pts[4], [1, 1, 1, 1])


def test_warp_identity_and_float64_model(O):
    p = O.default_params(64, 48, 64.0, 64.0, 31.5, 23.5)
    L = O.level_intrinsics(p, 0)
    pts = O.dense_points(None, 64, 48, 0)
    w = O.warp(pts, np.array([0, 0, 0, 1, 0, 0, 0], np.float32), L)
    assert np.allclose(w[:, :2], pts[:, :2], atol=1e-4) and np.array_equal(w[:, 2:], pts[:, 2:])
    pose = O.se3_exp(np.array([0.02, -0.01, 0.03, 0.01, -0.02, 0.015], np.float32))
    T = O.se3_matrix(pose).astype(np.float64)
    X = (pts[:, 0] - L.cx) / L.fx
    Y = (pts[:, 1] - L.cy) / L.fy
    P = T @ np.stack([X, Y, np.ones_like(X), np.ones_like(X)])
    u = P[0] * L.fx / P[2] + L.cx
    v = P[1] * L.fy / P[2] + L.cy
    w = O.warp(pts, pose, L)
    assert np.allclose(w[:, 0], u, atol=2e-4) and np.allclose(w[:, 1], v, atol=2e-4)
    assert np.allclose(w[:, 2], P[2], atol=1e-6)
    # invalid point (w = 0) is zeroed (Tracker.cpp:1466-1467) and later rejected by x2 > 0
    bad = np.array([[0, 0, 1, 0]], np.float32)
    wb = O.warp(bad, pose, L)
    assert wb[0, 0] == 0 and wb[0, 1] == 0 and wb[0, 3] == 0


def test_residual_jacobian_against_float64_formulas(O, synth):
    w, h = 64, 48
    ref, tgt, _, _, _ = synth.render_pair(w, h, 64.0, 64.0, 31.5, 23.5, seed=3, max_t=0.02, max_deg=1.0)
    p = O.default_params(w, h, 64.0, 64.0, 31.5, 23.5)
    L = O.level_intrinsics(p, 0)
    gx, gy = O.scharr3(ref)
    pts = O.dense_points(None, w, h, 0)
    pose = O.se3_exp(np.array([0.01, 0.02, -0.01, 0.004, -0.003, 0.01], np.float32))
    wp = O.warp(pts, pose, L)
    J, r, idx = O.residual_jacobian(ref, tgt, gx, gy, pts, wp, L, 0.5, 2.0)
    x2, y2, z2 = wp[:, 0].astype(np.float64), wp[:, 1].astype(np.float64), wp[:, 2].astype(np.float64)
    valid = (y2 > 0) & (y2 < h) & (x2 > 0) & (x2 < w) & (z2 != 0)
    assert np.array_equal(np.nonzero(valid)[0], idx)
    x2, y2, iz = x2[idx], y2[idx], 1.0 / z2[idx]
    ix2 = np.minimum(np.floor(x2 + 0.5).astype(int), w - 1)  # round-half-away for positives; S7 clamp
    iy2 = np.minimum(np.floor(y2 + 0.5).astype(int), h - 1)
    ix1, iy1 = pts[idx, 0].astype(int), pts[idx, 1].astype(int)
    assert np.array_equal(r, tgt[iy2, ix2].astype(np.float32) - ref[iy1, ix1].astype(np.float32))
    g0, g1 = gx[iy1, ix1].astype(np.float64), gy[iy1, ix1].astype(np.float64)
    fx, fy, zf, af = L.fx, L.fy, 0.5, 2.0
    Jw0 = np.stack([fx * iz, 0 * iz, -fx * x2 * iz * iz * zf, -fx * x2 * y2 * iz * iz * af,
                    fx * (1 + x2 * x2 * iz * iz) * af, -fx * y2 * iz * af], 1)
    Jw1 = np.stack([0 * iz, fy * iz, -fy * y2 * iz * iz * zf, -fy * (1 + y2 * y2 * iz * iz) * af,
                    fy * x2 * y2 * iz * iz * af, fy * x2 * iz * af], 1)
    Jref = g0[:, None] * Jw0 + g1[:, None] * Jw1
    mag = np.abs(g0[:, None] * Jw0) + np.abs(g1[:, None] * Jw1)  # the two terms may cancel
    assert (np.abs(J - Jref) <= 1e-6 * mag + 1e-9).all()


def test_normal_equations_and_error(O):
    rng = np.random.default_rng(5)
    J = rng.normal(0, 100, (500, 6)).astype(np.float32)
    r = rng.integers(-255, 256, 500).astype(np.float32)
    A, b = O.normal_equations(J, r, None, 50.0)
    Jd = J.astype(np.float64)
    assert np.array_equal(A, (Jd.T @ Jd).astype(np.float32)) or np.allclose(A, Jd.T @ Jd, rtol=1e-7)
    assert np.allclose(b, -(Jd.T @ (50.0 * r.astype(np.float64))), rtol=1e-7)
    assert np.array_equal(A, A.T)
    e, s = O.error(r)
    assert s == int((r.astype(np.int64) ** 2).sum())
    assert e == np.float32(np.float64(np.float32(1.0 / 500)) * s)
    # weighted form uses w on J and on r (Tracker.cpp:554-561) ⇒ effective w²
    w = rng.uniform(0, 1, 500).astype(np.float32)
    Aw, bw = O.normal_equations(J, r, w, 50.0)
    Jw = (w[:, None] * J).astype(np.float64)
    assert np.allclose(Aw, Jw.T @ Jw, rtol=1e-6)
    assert np.allclose(bw, -(Jw.T @ ((r * np.float32(50.0)) * w).astype(np.float64)), rtol=1e-6)


def test_tukey_weights_reference_quirks(O):
    # Tracker.cpp:1571-1654: medians from a histogram of u8-SATURATED values (negatives → 0)
    r = np.array([-30, -3, -1, 0, 1, 2, 3, 4, 200], np.float32)
    assert O.median_mat(r) == 1.0  # saturated: [0,0,0,0,1,2,3,4,200]; m = 4; first bin with count > 4 is 1
    mad = O.mad(r)
    dev = np.abs(r - 1.0)
    assert mad == np.float32(1.4826) * np.float32(O.median_mat(dev))
    w = O.tukey_weights(r)
    x = r * np.float32(1.0 / mad)
    exp = np.where(np.abs(x) <= 4.6851, (1 - x * x / 4.6851 ** 2) ** 2, 0)
    assert np.allclose(w, exp, atol=1e-6)
    assert O.tukey_weights(np.zeros(8, np.float32)).tolist() == [1.0] * 8  # MAD == 0 ⇒ MAD := 1


# ---------------------------------------------------------------- LS (LeastSquares.cpp)

def test_ls_closed_forms_and_sse_equivalence(O, golden):
    g = golden("ls.npz")
    J, r, w = g["J"], g["r"], g["w"]
    ls = O.ls_new()
    O.ls_update(ls, J[0], r[0], w[0])
    A, b, e, n = O.ls_finish(ls, divide=False)
    assert np.array_equal(A, np.outer(J[0], J[0]) * w[0])       # single row ⇒ A = (J Jᵀ)·w
    assert np.array_equal(b, -(J[0] * (r[0] * w[0])))           # b = −w·r·J (LeastSquares.cpp:206)
    assert e == r[0] * r[0] * w[0] and n == 1
    ls = O.ls_new()
    for i in range(16):
        O.ls_update(ls, J[i], r[i], w[i])
    A1, b1, e1, n1 = O.ls_finish(ls, True)
    assert np.array_equal(A1, g["A_scalar"]) and np.array_equal(b1, g["b_scalar"]) and n1 == 16
    Jd, wd, rd = J.astype(np.float64), w.astype(np.float64), r.astype(np.float64)
    assert np.allclose(A1 * 16, (Jd * wd[:, None]).T @ Jd, rtol=1e-5)
    assert np.allclose(b1 * 16, -(Jd * (wd * rd)[:, None]).sum(0), rtol=1e-4, atol=1e-2)
    ls = O.ls_new()
    for i in range(0, 16, 4):
        O.ls_update4(ls, J[i:i + 4].T.copy(), r[i:i + 4], w[i:i + 4], True)
    A4, b4, e4, n4 = O.ls_finish(ls, False)
    assert np.array_equal(A4, g["A_sse"]) and np.array_equal(b4, g["b_sse"])
    assert n4 == 24                                             # quirk C-6: += 6 per 4 points (LeastSquares.cpp:201)
    assert np.allclose(A4, A1 * 16, rtol=1e-5) and np.allclose(b4, b1 * 16, rtol=1e-4, atol=1e-2)
    assert np.isclose(e4, e1 * 16, rtol=1e-5) and np.array_equal(A4, A4.T)
    ls = O.ls_new()
    O.ls_update4(ls, J[:4].T.copy(), r[:4], w[:4], False)
    assert O.ls_finish(ls, False)[3] == 4


# ---------------------------------------------------------------- EstimatePose known answers + golden regression

def test_identical_frames_give_identity_and_exit_at_k1(O, synth):
    # error 0 at k=0 → one (zero) update; at k=1 error 0 >= last_error 0 → exit (Tracker.cpp:508)
    img = synth.texture(160, 96, seed=9)
    p = O.default_params(160, 96, 131.25, 131.25, 79.5, 47.5)
    st, pose, tr = O.align_pair(p, img, img, want_trace=True)
    assert st == 0
    assert np.allclose(pose, [0, 0, 0, 1, 0, 0, 0], atol=1e-7)
    assert [t["level"] for t in tr] == [4, 4, 3, 3, 2, 2, 1, 1]
    assert all(t["sum_r2"] == 0 and t["error"] == 0 for t in tr)
    assert [t["exited"] for t in tr] == [0, 1] * 4
    assert not tr[0]["b"].any() and not tr[0]["delta"].any()


def test_shift_is_recovered_in_sign_and_scale(O, synth):
    # z=1 plane, pure +x image shift of 2 px at level 0 (0.25 px at level 3): t_x should come out positive,
    # of the order 2/fx, with the other components much smaller.
    w, h, fx = 160, 96, 131.25
    ref, tgt = synth.shifted_pair(w, h, seed=21, dx=2)
    p = O.default_params(w, h, fx, fx, 79.5, 47.5, n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
    st, pose, _ = O.align_pair(p, ref, tgt)
    assert st == 0
    assert 0.3 * 2 / fx < pose[4] < 2.0 * 2 / fx
    assert abs(pose[5]) < 0.6 * pose[4]   # (0.54 under the OpenCV set, 0.4 under the legacy one: the fixed schedule amplifies 1-ulp differences)


def test_no_valid_points_status(O, synth):
    img = synth.texture(64, 48, seed=2)
    p = O.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=3, first_level=2, last_level=0, has_depth=1)
    st, _, _ = O.align_pair(p, img, img, ref_depth=np.zeros((48, 64), np.uint16))
    assert st == 2  # reference: cv::Exception on the empty Mat product (Tracker.cpp:501)


GOLDEN_PAIRS = ["pair_64x48_ref", "pair_64x48_fixed", "pair_160x96_ref5", "pair_160x96_fixed", "pair_160x96_depth",
                "pair_160x96_features"]


def params_from_golden(O, g):
    h, w = g["ref"].shape
    fx, fy, cx, cy = [float(v) for v in g["intr"]]
    over = {}
    for k, v in zip(g["over_keys"], g["over_vals"]):
        over[str(k)] = float(v) if str(k) in ("gain", "z_factor", "angle_factor", "epsilon") else int(v)
    p = O.default_params(w, h, fx, fy, cx, cy, **over)
    if "depth" in g:
        p.has_depth = 1
    return p


@pytest.mark.parametrize("name", GOLDEN_PAIRS)
def test_golden_pair_regression(O, golden, name):
    g = golden(name)
    p = params_from_golden(O, g)
    st, pose, tr = O.align_pair(p, g["ref"], g["tgt"], g["depth"] if "depth" in g else None, want_trace=True)
    assert st == int(g["status"])
    assert np.array_equal(pose, g["pose"])
    assert len(tr) == len(g["trace_level"])
    for i, t in enumerate(tr):
        assert t["level"] == g["trace_level"][i] and t["iter"] == g["trace_iter"][i]
        assert t["n_valid"] == g["trace_n_valid"][i] and t["sum_r2"] == g["trace_sum_r2"][i]
        assert t["exited"] == g["trace_exited"][i]
        assert np.array_equal(t["A"], g["trace_A"][i]) and np.array_equal(t["b"], g["trace_b"][i])
        assert np.array_equal(t["pose"], g["trace_pose"][i])


def test_golden_trace_is_self_consistent_with_float64(O, golden):
    """Brute-force float64 recomputation of A, b, error for one iteration of a golden trace."""
    g = golden("pair_160x96_fixed.npz")
    p = params_from_golden(O, g)
    ref, tgt = g["ref"], g["tgt"]
    lvl, it = int(g["trace_level"][3]), 3
    assert int(g["trace_iter"][it]) == 3 and lvl == 3
    pose_in = g["trace_pose"][it - 1]
    a, b = ref, tgt
    for _ in range(lvl):
        a, b = O.halve_u8(a), O.halve_u8(b)
    gx, gy = O.scharr3(a)
    L = O.level_intrinsics(p, lvl)
    pts = O.dense_points(None, L.w, L.h, lvl)
    wp = O.warp(pts, pose_in, L)
    J, r, _ = O.residual_jacobian(a, b, gx, gy, pts, wp, L)
    assert len(r) == g["trace_n_valid"][it]
    assert int((r.astype(np.int64) ** 2).sum()) == g["trace_sum_r2"][it]
    Jd = J.astype(np.float64)
    assert np.allclose(g["trace_A"][it], Jd.T @ Jd, rtol=1e-6)
    assert np.allclose(g["trace_b"][it], -(Jd.T @ (50.0 * r.astype(np.float64))), rtol=1e-6, atol=1e-3)


@pytest.mark.one_arith
def test_arithmetic_set_sensitivity(O, synth):
    """Distance between the two arithmetic sets (OpenCV's generic paths — double-accumulated 4-/2-term gemm products, the
    folded unprojection, A.inv()*b as a solve — against the legacy f32 FMA chains / inverse-then-multiply): the same
    algorithm, last bits per pixel term apart.  In the reference's own schedule (levels 4 -> 1, early exit: one or two updates
    per level) the poses stay within 2e-5 m of each other — inside the 1e-4 parity tolerance; in the fixed 4 x 10 schedule
    the non-contractive iteration amplifies the same differences to 1e-3 m.  This is why parity is only
    meaningful against a bit-pinned oracle, and why the pinned set has to be OpenCV's.  (tools/exp/arith_distance.py prints
    the table for 320x240 and 640x480, with and without depth.)"""
    w, h, f = 320, 240, 262.5
    fixed = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
    dist = {"fixed": [], "reference": []}
    for s in (4000, 4007, 4011):
        ref, tgt, _, _, _ = synth.render_pair(w, h, f, f, 159.5, 119.5, seed=s)
        for name, over in (("fixed", fixed), ("reference", {})):
            a = O.align_pair(O.default_params(w, h, f, f, 159.5, 119.5, arith=O.ARITH_OPENCV, **over), ref, tgt)[1]
            b = O.align_pair(O.default_params(w, h, f, f, 159.5, 119.5, arith=O.ARITH_LEGACY, **over), ref, tgt)[1]
            c = O.align_pair(O.default_params(w, h, f, f, 159.5, 119.5, arith=O.ARITH_OPENCV, **over), ref, tgt)[1]
            assert np.array_equal(a, c)                     # the switch does not leak between calls
            dist[name].append(float(np.linalg.norm(a[4:].astype(np.float64) - b[4:].astype(np.float64))))
    assert all(0 < d < 1e-2 for d in dist["fixed"]), dist
    assert max(dist["fixed"]) > 1e-5, dist                # the fixed schedule does amplify
    assert all(d < 1e-4 for d in dist["reference"]), dist   # the reference's schedule does not


def test_oracle_params_reject_unknown_fields(O):
    with pytest.raises(AttributeError):
        O.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, small_products_f64=1)


# ---------------------------------------------------------------- the fold of GEMMSingleMul's partial sums (G1)

def gemm_fold_vector(O):
    """One point whose warped z separates the two folds of the 4-term rigid product (src/Tracker.cpp:1450).
    Row 2 of the rigid matrix of q = (0, 1/4, 1/4, w) is (-w/2, 1/8, 7/8), exactly.  With z = 8 m / 7, m = (2^24 + 13) 2^-24
    — a midpoint of two floats whose lower neighbour is even — the term s2 = (7/8) z is m itself; x and y put s0 = 0.49 and
    s1 = 0.29 units of m's last double place beside it.  "s0 += s1 + s2 + s3" adds s1 to m first (m again), then s0 (m again):
    the tie rounds to the even float.  ((s0 + s1) + s2) + s3 adds the two small terms first (0.78 units), m moves up one
    place and the float rounds up."""
    from fractions import Fraction
    pose = np.array([0, .25, .25, np.float32(np.sqrt(3) / 2), 0, 0, 0], np.float32)
    M = 2 ** 24 + 13
    z = np.float32(Fraction(M, 7) / 2 ** 21)
    assert Fraction(float(z)) * 7 / 8 == Fraction(M, 2 ** 24)
    pts = np.array([[-2.0 ** -52, 2.0 ** -51, z, 1.0]], np.float32)
    L = O.Level()
    L.w = L.h = 64
    L.fx = L.fy = L.invfx = L.invfy = 1.0
    L.cx = L.cy = 0.0
    lo = np.float32(Fraction(M - 1, 2 ** 24))
    return pose, pts, L, lo, np.nextafter(lo, np.float32(2))


@pytest.mark.one_arith
def test_gemm_fold_follows_the_published_statement(O):
    pose, pts, L, lo, hi = gemm_fold_vector(O)
    T = O.se3_matrix(pose).reshape(4, 4)
    assert tuple(T[2, 1:]) == (0.125, 0.875, 0.0)
    assert O.warp(pts, pose, L)[0][2] == lo            # default: s0 + ((s1 + s2) + s3)
    prev = O.set_gemm_fold(1)
    try:
        assert O.warp(pts, pose, L)[0][2] == hi        # ((s0 + s1) + s2) + s3
    finally:
        O.set_gemm_fold(prev)
    assert O.warp(pts, pose, L)[0][2] == lo


# ---------------------------------------------------------------- frame sizes the resize chain does not divide
# (System.cpp:148-191 crops to a data-dependent ROI; :246-251 halves with cv::resize(.., 0.5, 0.5); Tracker.cpp:312-313 sizes
#  the point grid with ">> lvl")

def _np_resize_half(img):
    """Independent restatement of resizeAreaFast for scale 2 x 2: whole cells (a+b+c+d+2)>>2; the partial last column of whole
    rows and EVERY cell of a partial last row: mean of the pixels that exist, rounded half to even."""
    img = np.asarray(img)
    sh, sw = img.shape
    dh, dw = int(np.rint(sh * 0.5)), int(np.rint(sw * 0.5))    # np.rint: half to even, like cvRound
    a = img.astype(np.int64)
    out = np.zeros((dh, dw), np.int64)
    fh, fw = sh // 2, sw // 2
    c = a[:2 * fh, :2 * fw]
    out[:fh, :fw] = (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) >> 2
    if dw > fw:   # partial last column: two pixels of column sw - 1
        col = a[:2 * fh, sw - 1]
        out[:fh, fw] = np.rint((col[0::2] + col[1::2]) / 2.0)
    if dh > fh:   # partial last row: every cell by the generic tail
        row = a[sh - 1]
        out[fh, :fw] = np.rint((row[0:2 * fw:2] + row[1:2 * fw:2]) / 2.0)
        if dw > fw:
            out[fh, fw] = row[sw - 1]
    return out.astype(img.dtype)


@pytest.mark.one_arith
def test_half_size_is_cvround(O):
    # saturate_cast<int>(ssize * 0.5) = cvRound: half to even
    assert [O.half_size(n) for n in (733, 735, 725, 465, 471, 479, 640, 1, 2, 3, 5, 7)] == [366, 368, 362, 232, 236, 240, 320, 0, 1, 2, 2, 4]


@pytest.mark.one_arith
def test_resize_half_any_size(O):
    rng = np.random.default_rng(77)
    for (h, w) in ((480, 640), (471, 733), (479, 735), (465, 725), (7, 9), (5, 6), (6, 5), (3, 3), (2, 7), (7, 2), (9, 11), (10, 13)):
        im = rng.integers(0, 256, (h, w)).astype(np.uint8)
        d16 = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        assert np.array_equal(O.resize_half_u8(im), _np_resize_half(im)), (h, w)
        assert np.array_equal(O.resize_half_u16(d16), _np_resize_half(d16)), (h, w)
        if h % 2 == 0 and w % 2 == 0:
            assert np.array_equal(O.resize_half_u8(im), O.halve_u8(im))
            assert np.array_equal(O.resize_half_u16(d16), O.halve_u16(d16))
    # the two roundings apart: a whole cell rounds half up, a partial cell half to even
    im = np.array([[1, 2, 2], [1, 2, 3], [2, 3, 0]], np.uint8)        # 3 x 3 -> 2 x 2
    assert O.resize_half_u8(im).tolist() == [[2, 2], [2, 0]]           # (6+2)>>2 = 2; rint(2.5) = 2; rint(2.5) = 2; the corner itself


@pytest.mark.one_arith
def test_level_geometry_of_odd_sizes(O):
    for (w, h, n) in ((733, 471, 5), (735, 479, 5), (725, 465, 5), (752, 480, 5), (640, 480, 4), (163, 99, 4)):
        p = O.default_params(w, h, 400.0, 400.0, w / 2, h / 2, n_levels=n, first_level=n - 1)
        iw, ih = w, h
        for l in range(n):
            L = O.level_intrinsics(p, l)
            assert (L.w, L.h) == (w >> l, h >> l)                      # Tracker.cpp:312-313
            assert (L.iw, L.ih) == (iw, ih)                            # the resize chain
            assert L.iw >= L.w and L.ih >= L.h                         # the grid never leaves the image
            iw, ih = int(np.rint(iw * 0.5)), int(np.rint(ih * 0.5))
    L3 = O.level_intrinsics(O.default_params(733, 471, 1, 1, 0, 0), 3)
    assert (L3.w, L3.h, L3.iw, L3.ih) == (91, 58, 92, 59)


def test_odd_size_alignment_equals_the_cropped_one_above_level_0(O, synth):
    """161 x 97 halves to 80 x 48 with the last column and row DROPPED (161 * 0.5 = 80.5 -> 80, 97 * 0.5 = 48.5 -> 48: both
    to even), so every level >= 1 — image and grid — is that of the 160 x 96 crop: an alignment that stops at level 1 must
    give the crop's pose bit for bit.  (Ties the odd-size path to the sizes the goldens pin.)"""
    w, h, f = 161, 97, 131.25
    ref, tgt, dep, _, _ = synth.render_pair(w, h, f, f, 79.5, 47.5, seed=31, with_depth=True)
    for depth in (False, True):
        over = dict(n_levels=4, first_level=3, last_level=1, max_iters=6, early_exit=0, has_depth=int(depth))
        st, pose, tr = O.align_pair(O.default_params(w, h, f, f, 79.5, 47.5, **over), ref, tgt, dep if depth else None, want_trace=True)
        st2, pose2, tr2 = O.align_pair(O.default_params(160, 96, f, f, 79.5, 47.5, **over), ref[:96, :160], tgt[:96, :160],
                                       dep[:96, :160] if depth else None, want_trace=True)
        assert st == 0 and st2 == 0
        assert np.array_equal(pose, pose2)
        assert [t["n_valid"] for t in tr] == [t["n_valid"] for t in tr2]


def test_odd_size_alignment_recovers_the_motion(O, synth):
    """Whole alignments at sizes with partial cells (163 -> 82, 99 -> 50), dropped columns (165 -> 82) and grids smaller than
    their images: the pose lands where the even-sized alignments of the same scene land.  (The fixed 4 x 10 schedule is not
    contractive — DESIGN.md §2 — so crops of one scene, even-sized ones too (164 x 100, 168 x 104), end up to 2.5e-3 m apart; the
    bound below is that spread, not an accuracy claim.)"""
    f = 131.25
    big_ref, big_tgt, big_dep, R, t = synth.render_pair(168, 104, f, f, 83.5, 51.5, seed=37, with_depth=True)
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
    base = O.align_pair(O.default_params(160, 96, f, f, 83.5, 51.5, **over), big_ref[:96, :160], big_tgt[:96, :160])[1]
    for (w, h) in ((163, 99), (165, 101), (167, 103), (161, 97), (166, 98)):
        p = O.default_params(w, h, f, f, 83.5, 51.5, **over)
        st, pose, tr = O.align_pair(p, big_ref[:h, :w], big_tgt[:h, :w], want_trace=True)
        assert st == 0
        assert np.linalg.norm(pose[4:] - base[4:]) < 5e-3 and np.linalg.norm(pose[:3] - base[:3]) < 2.5e-3, (w, h, pose, base)
        for tt in tr:   # never more valid points than the level's grid holds
            L = O.level_intrinsics(p, tt["level"])
            assert 0 < tt["n_valid"] <= L.w * L.h
        pd = O.default_params(w, h, f, f, 83.5, 51.5, has_depth=1, **over)
        assert O.align_pair(pd, big_ref[:h, :w], big_tgt[:h, :w], big_dep[:h, :w])[0] == 0


def test_two_threads_with_different_arithmetic_sets(O, synth):
    """The arithmetic set travels with the call (uwo_params::arith): two threads aligning under different sets at the same
    time get what each gets alone.  (Rounds 1-5 kept the set in a process-wide variable that every call overwrote.)"""
    import threading
    w, h, f = 160, 96, 131.25
    ref, tgt, _, _, _ = synth.render_pair(w, h, f, f, 79.5, 47.5, seed=41)
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
    alone = {a: O.align_pair(O.default_params(w, h, f, f, 79.5, 47.5, arith=a, **over), ref, tgt)[1] for a in (0, 1)}
    assert not np.array_equal(alone[0], alone[1])
    out = {0: [], 1: []}

    def work(a):
        for _ in range(12):
            out[a].append(O.align_pair(O.default_params(w, h, f, f, 79.5, 47.5, arith=a, **over), ref, tgt)[1])

    th = [threading.Thread(target=work, args=(a,)) for a in (0, 1)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    for a in (0, 1):
        assert len(out[a]) == 12 and all(np.array_equal(p, alone[a]) for p in out[a])


@pytest.mark.one_arith
def test_sine_of_the_exponential_rounded_or_libm(O, synth):
    """S5: Sophus calls libm's sinf / cosf (so3.hpp:538-558), whose last bit is the libm build's; the oracle and the HIP library
    compute the correctly rounded (float)sin((double)x).  uwo_params::trig = TRIG_LIBM evaluates this host's sinf / cosf instead:
    at x = 0x1.d12ed2p-12 — the smallest float where glibc 2.35's sinf is not the rounded value (tools/trig/trig_sweep.c,
    exhaustive over [0, 0.5]) — the exponential's quaternion moves by one ulp, by no more anywhere, and whole alignments,
    whose Gauss-Newton rotation steps stay below 1e-2 rad (842 such floats of 4e7 there), keep their poses."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.restype = ctypes.c_float
    libm.sinf.argtypes = [ctypes.c_float]
    x = float.fromhex("0x1.d12ed2p-12")
    xi = np.array([0.1, 0.2, 0.3, 2 * x, 0, 0], np.float32)
    a = O.se3_exp(xi)
    prev = O.set_trig(O.TRIG_LIBM)
    try:
        b = O.se3_exp(xi)
    finally:
        O.set_trig(prev)
    rounded = np.float32(np.sin(np.float64(np.float32(x))))
    host = np.float32(libm.sinf(x))
    assert a[0] == np.float32(rounded / np.float32(2 * x)) * np.float32(2 * x)          # imag * omega_x, imag = sin(theta / 2) / theta
    if host != rounded:                                                                  # (a libm whose sinf is correctly rounded here: nothing to see)
        assert abs(int(a[:1].view(np.uint32)[0]) - int(b[:1].view(np.uint32)[0])) == 1 and np.array_equal(a[1:], b[1:])
    else:
        assert np.array_equal(a, b)
    w, h, f = 160, 96, 131.25
    for s in range(6):
        ref, tgt, _, _, _ = synth.render_pair(w, h, f, f, 79.5, 47.5, seed=9000 + s)
        for over in (dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0), {}):
            p0 = O.align_pair(O.default_params(w, h, f, f, 79.5, 47.5, trig=O.TRIG_ROUNDED, **over), ref, tgt)[1]
            p1 = O.align_pair(O.default_params(w, h, f, f, 79.5, 47.5, trig=O.TRIG_LIBM, **over), ref, tgt)[1]
            assert np.abs(p0.astype(np.float64) - p1.astype(np.float64)).max() <= 1e-6
