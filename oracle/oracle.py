"""ctypes binding of the CPU oracle (oracle/uwt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (uw-slam_amd/) never imports this module.
PARITY UNPINNED — see oracle/uwt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# UWT_ORACLE_SAN=1: the AddressSanitizer + UBSan build of the same sources (oracle/Makefile SAN=1; the sanitizer runtimes must be
# preloaded into the interpreter: tools/sanitize_cpu.sh).  Test infrastructure may read the environment; the product never does.
_SAN = os.environ.get("UWT_ORACLE_SAN", "") == "1"
_LIB_PATH = os.path.join(_HERE, "libuwt_oracle_san.so" if _SAN else "libuwt_oracle.so")

MAX_LEVELS = 8
ARITH_OPENCV, ARITH_LEGACY = 0, 1   # uwo_params.arith (oracle/uwt_oracle.h: G1..G4 / S1, S3, S4)
ARITH_NAMES = {ARITH_OPENCV: "opencv", ARITH_LEGACY: "legacy"}
TRIG_ROUNDED, TRIG_LIBM = 0, 1     # uwo_params.trig (S5): (float)sin((double)x) / this host's sinf, cosf


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("n_levels", C.c_int32), ("first_level", C.c_int32), ("last_level", C.c_int32),
        ("max_iters", C.c_int32),
        ("epsilon", C.c_float), ("gain", C.c_float), ("z_factor", C.c_float), ("angle_factor", C.c_float),
        ("depth_scale", C.c_float), ("initial_error", C.c_float),
        ("early_exit", C.c_int32), ("has_depth", C.c_int32), ("handoff_scale_t", C.c_int32),
        ("weights", C.c_int32), ("sampler", C.c_int32), ("arith", C.c_int32), ("gemm_fold", C.c_int32),
        ("trig", C.c_int32),
    ]


class Level(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("invfx", C.c_float), ("invfy", C.c_float),
                ("iw", C.c_int32), ("ih", C.c_int32)]   # w, h: the point grid (size >> lvl); iw, ih: the level's image (resize chain)


class Trace(C.Structure):
    _fields_ = [("level", C.c_int32), ("iter", C.c_int32), ("n_valid", C.c_int32), ("exited", C.c_int32),
                ("sum_r2", C.c_int64), ("error", C.c_float),
                ("A", C.c_float * 36), ("b", C.c_float * 6), ("delta", C.c_float * 6), ("pose", C.c_float * 7)]


class LS(C.Structure):
    _fields_ = [("A", C.c_float * 36), ("b", C.c_float * 6), ("error", C.c_float),
                ("num_constraints", C.c_int32), ("sse", C.c_float * 112)]


def build(force=False):
    """Compile oracle/libuwt_oracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "uwt_oracle.c")
    hdr = os.path.join(_HERE, "uwt_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)   # (a preloaded sanitizer runtime is for the interpreter, not for make / gcc)
    subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_LIB_PATH)] + (["SAN=1"] if _SAN else []), stdout=subprocess.DEVNULL, env=env)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.uwo_error.restype = C.c_float
        _lib.uwo_median_mat.restype = C.c_float
        _lib.uwo_mad.restype = C.c_float
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# what default_params() fills in when the caller names no set: OpenCV's, like the HIP library's uwt_default_params (no
# environment variable is read); tests/conftest.py switches it per test, child processes of tests set it from their arguments
DEFAULT_ARITH = ARITH_OPENCV


def default_params(width, height, fx, fy, cx, cy, **over):
    p = Params()
    lib().uwo_default_params(C.byref(p), width, height, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy))
    p.arith = DEFAULT_ARITH
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def level_intrinsics(p, lvl):
    L = Level()
    st = lib().uwo_level_intrinsics(C.byref(p), lvl, C.byref(L))
    if st:
        raise ValueError("bad level")
    return L


def halve_u8(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().uwo_halve_u8(_p(img, C.c_uint8), w, h, _p(out, C.c_uint8))
    return out


def halve_u16(img):
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint16)
    lib().uwo_halve_u16(_p(img, C.c_uint16), w, h, _p(out, C.c_uint16))
    return out


def half_size(n):
    """cvRound(n * 0.5): the output size of cv::resize(.., Size(), 0.5, 0.5) (half to even: 733 -> 366, 735 -> 368)."""
    return int(lib().uwo_half_size(int(n)))


def resize_half_u8(img):
    """cv::resize(img, Size(), 0.5, 0.5) on any size (partial last column / row included)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((half_size(h), half_size(w)), np.uint8)
    lib().uwo_resize_half_u8(_p(img, C.c_uint8), w, h, _p(out, C.c_uint8))
    return out


def resize_half_u16(img):
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    out = np.empty((half_size(h), half_size(w)), np.uint16)
    lib().uwo_resize_half_u16(_p(img, C.c_uint16), w, h, _p(out, C.c_uint16))
    return out


def pyramid(img, n_levels):
    """images_[0..n_levels) / depths_[0..n_levels) of System::AddFrame (System.cpp:246-251) for a u8 or u16 level-0 image."""
    img = np.ascontiguousarray(img)
    f = resize_half_u16 if img.dtype == np.uint16 else resize_half_u8
    out = [img]
    for _ in range(1, n_levels):
        out.append(f(out[-1]))
    return out


def scharr3(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    gx = np.empty((h, w), np.int16)
    gy = np.empty((h, w), np.int16)
    lib().uwo_scharr3(_p(img, C.c_uint8), w, h, _p(gx, C.c_int16), _p(gy, C.c_int16))
    return gx, gy


def gradient_mag(gx, gy):
    gx = np.ascontiguousarray(gx, np.int16)
    gy = np.ascontiguousarray(gy, np.int16)
    out = np.empty(gx.shape, np.uint8)
    lib().uwo_gradient_mag(_p(gx, C.c_int16), _p(gy, C.c_int16), gx.size, _p(out, C.c_uint8))
    return out


def dense_points(depth, w, h, lvl, depth_scale=0.0002):
    """The w x h point grid; a depth image wider than the grid (an odd-sized level) is read with its own row length."""
    pts = np.empty((w * h, 4), np.float32)
    stride = w
    if depth is not None:
        depth = np.ascontiguousarray(depth, np.uint16)
        assert depth.ndim == 1 or (depth.shape[1] >= w and depth.shape[0] >= h), (depth.shape, w, h)
        if depth.ndim == 2:
            stride = depth.shape[1]
        dp = _p(depth, C.c_uint16)
    else:
        dp = None
    lib().uwo_dense_points_ex(dp, stride, w, h, lvl, C.c_float(depth_scale), _p(pts, C.c_float))
    return pts


def se3_exp(xi):
    xi = np.ascontiguousarray(xi, np.float32)
    out = np.empty(7, np.float32)
    lib().uwo_se3_exp(_p(xi, C.c_float), _p(out, C.c_float))
    return out


def se3_mul(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty(7, np.float32)
    lib().uwo_se3_mul(_p(a, C.c_float), _p(b, C.c_float), _p(out, C.c_float))
    return out


def se3_matrix(pose):
    pose = np.ascontiguousarray(pose, np.float32)
    out = np.empty(16, np.float32)
    lib().uwo_se3_matrix(_p(pose, C.c_float), _p(out, C.c_float))
    return out.reshape(4, 4)


def se3_handoff(pose, scale_t=0):
    pose = np.array(pose, np.float32)
    st = lib().uwo_se3_handoff(_p(pose, C.c_float), int(scale_t))
    if st:
        raise ValueError("quaternion close to zero")
    return pose


def warp(pts, pose, L):
    pts = np.ascontiguousarray(pts, np.float32)
    pose = np.ascontiguousarray(pose, np.float32)
    out = np.empty_like(pts)
    lib().uwo_warp(_p(pts, C.c_float), pts.shape[0], _p(pose, C.c_float), C.byref(L), _p(out, C.c_float))
    return out


def residual_jacobian(img1, img2, gx1, gy1, pts, warped, L, z_factor=1.0, angle_factor=1.0):
    img1 = np.ascontiguousarray(img1, np.uint8)
    img2 = np.ascontiguousarray(img2, np.uint8)
    gx1 = np.ascontiguousarray(gx1, np.int16)
    gy1 = np.ascontiguousarray(gy1, np.int16)
    pts = np.ascontiguousarray(pts, np.float32)
    warped = np.ascontiguousarray(warped, np.float32)
    n = pts.shape[0]
    J = np.empty((n, 6), np.float32)
    r = np.empty(n, np.float32)
    idx = np.empty(n, np.int32)
    nv = lib().uwo_residual_jacobian(_p(img1, C.c_uint8), _p(img2, C.c_uint8), _p(gx1, C.c_int16), _p(gy1, C.c_int16),
                                     _p(pts, C.c_float), _p(warped, C.c_float), n, C.byref(L),
                                     C.c_float(z_factor), C.c_float(angle_factor),
                                     _p(J, C.c_float), _p(r, C.c_float), _p(idx, C.c_int32))
    return J[:nv].copy(), r[:nv].copy(), idx[:nv].copy()


def error(r, w=None):
    r = np.ascontiguousarray(r, np.float32)
    s = C.c_int64(0)
    wp = None
    if w is not None:
        w = np.ascontiguousarray(w, np.float32)
        wp = _p(w, C.c_float)
    e = lib().uwo_error(_p(r, C.c_float), wp, r.size, C.byref(s))
    return float(e), int(s.value)


def normal_equations(J, r, w=None, gain=50.0):
    J = np.ascontiguousarray(J, np.float32)
    r = np.ascontiguousarray(r, np.float32)
    A = np.empty(36, np.float32)
    b = np.empty(6, np.float32)
    wp = None
    if w is not None:
        w = np.ascontiguousarray(w, np.float32)
        wp = _p(w, C.c_float)
    lib().uwo_normal_equations(_p(J, C.c_float), _p(r, C.c_float), wp, r.size, C.c_float(gain),
                               _p(A, C.c_float), _p(b, C.c_float))
    return A.reshape(6, 6), b


def inv6(A):
    A = np.ascontiguousarray(A, np.float32).reshape(36)
    X = np.empty(36, np.float32)
    ok = lib().uwo_inv6(_p(A, C.c_float), _p(X, C.c_float))
    return X.reshape(6, 6), bool(ok)


def solve_delta(A, b):
    A = np.ascontiguousarray(A, np.float32).reshape(36)
    b = np.ascontiguousarray(b, np.float32)
    d = np.empty(6, np.float32)
    lib().uwo_solve_delta(_p(A, C.c_float), _p(b, C.c_float), _p(d, C.c_float))
    return d


def set_arith(arith):
    """Arithmetic set of the per-stage functions (warp, residual_jacobian, normal_equations, error, solve_delta);
    returns the previous one.  align_pair* set it from Params.arith."""
    return lib().uwo_set_arith(int(arith))


def set_gemm_fold(fold):
    """Fold of GEMMSingleMul's four partial sums in the 4-term rigid product and the short 1-wide sums: 0 (default, the
    source's "s0 += s1 + s2 + s3": s0 + ((s1 + s2) + s3)) or 1 (((s0 + s1) + s2) + s3, diagnosis only: the HIP kernels
    implement fold 0).  Returns the previous one."""
    return lib().uwo_set_gemm_fold(int(fold))


def set_trig(trig):
    """Sine / cosine of se3_exp for the calling thread (TRIG_ROUNDED / TRIG_LIBM); returns the previous one."""
    return lib().uwo_set_trig(int(trig))


def solve6(A, b):
    """cv::solve(A, b, x, DECOMP_LU): (x, ok)."""
    A = np.ascontiguousarray(A, np.float32).reshape(36)
    b = np.ascontiguousarray(b, np.float32).reshape(6)
    out = np.empty(6, np.float32)
    ok = lib().uwo_solve6(_p(A, C.c_float), _p(b, C.c_float), _p(out, C.c_float))
    return out, bool(ok)


def median_mat(v):
    v = np.ascontiguousarray(v, np.float32)
    return float(lib().uwo_median_mat(_p(v, C.c_float), v.size))


def mad(v):
    v = np.ascontiguousarray(v, np.float32)
    return float(lib().uwo_mad(_p(v, C.c_float), v.size))


def tukey_weights(r):
    r = np.ascontiguousarray(r, np.float32)
    w = np.empty_like(r)
    lib().uwo_tukey_weights(_p(r, C.c_float), r.size, _p(w, C.c_float))
    return w


def trace_to_dict(t):
    return dict(level=t.level, iter=t.iter, n_valid=t.n_valid, exited=t.exited, sum_r2=int(t.sum_r2),
                error=float(t.error), A=np.array(t.A, np.float32).reshape(6, 6), b=np.array(t.b, np.float32),
                delta=np.array(t.delta, np.float32), pose=np.array(t.pose, np.float32))


def align_pair(p, ref_gray, tgt_gray, ref_depth=None, want_trace=False):
    """Pyramid + gradients + EstimatePose on one pair of level-0 frames. Returns (status, pose[7], trace list)."""
    ref_gray = np.ascontiguousarray(ref_gray, np.uint8)
    tgt_gray = np.ascontiguousarray(tgt_gray, np.uint8)
    pose = np.empty(7, np.float32)
    dp = None
    if ref_depth is not None:
        ref_depth = np.ascontiguousarray(ref_depth, np.uint16)
        dp = _p(ref_depth, C.c_uint16)
    cap = (p.first_level - p.last_level + 1) * p.max_iters if want_trace else 0
    tr = (Trace * max(cap, 1))()
    n = C.c_int32(cap)
    st = lib().uwo_align_pair(C.byref(p), _p(ref_gray, C.c_uint8), _p(tgt_gray, C.c_uint8), dp, None,
                              _p(pose, C.c_float), tr if want_trace else None, C.byref(n) if want_trace else None)
    traces = [trace_to_dict(tr[i]) for i in range(n.value)] if want_trace else []
    return st, pose, traces


def ls_new():
    ls = LS()
    lib().uwo_ls_initialize(C.byref(ls))
    return ls


def ls_update(ls, J, res, weight):
    J = np.ascontiguousarray(J, np.float32)
    lib().uwo_ls_update(C.byref(ls), _p(J, C.c_float), C.c_float(res), C.c_float(weight))


def ls_update4(ls, J6x4, res4, w4, quirk_plus6=True):
    J = np.ascontiguousarray(J6x4, np.float32)
    r = np.ascontiguousarray(res4, np.float32)
    w = np.ascontiguousarray(w4, np.float32)
    lib().uwo_ls_update4(C.byref(ls), _p(J, C.c_float), _p(r, C.c_float), _p(w, C.c_float), int(bool(quirk_plus6)))


def ls_finish(ls, divide=True):
    (lib().uwo_ls_finish if divide else lib().uwo_ls_finish_no_divide)(C.byref(ls))
    return (np.array(ls.A, np.float32).reshape(6, 6), np.array(ls.b, np.float32), float(ls.error),
            int(ls.num_constraints))


def accumulate_trajectory(poses, start=None, t_scale=1.0, reference_axes=False):
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
    start = np.array([0, 0, 0, 1, 0, 0, 0], np.float32) if start is None else np.ascontiguousarray(start, np.float32)
    out = np.empty_like(poses)
    lib().uwo_accumulate_trajectory(_p(poses, C.c_float), poses.shape[0], _p(start, C.c_float), C.c_float(t_scale),
                                    int(bool(reference_axes)), _p(out, C.c_float))
    return out


def align_pair_points(p, ref_gray, tgt_gray, tables, ref_depth=None, want_trace=False):
    """tables: {level: n x 4 float32 array} for every iterated level."""
    ref_gray = np.ascontiguousarray(ref_gray, np.uint8)
    tgt_gray = np.ascontiguousarray(tgt_gray, np.uint8)
    arrs = {}
    ptrs = (C.POINTER(C.c_float) * MAX_LEVELS)()
    counts = (C.c_int32 * MAX_LEVELS)()
    for l, t in tables.items():
        arrs[l] = np.ascontiguousarray(t, np.float32).reshape(-1, 4)
        counts[l] = arrs[l].shape[0]
        if arrs[l].shape[0]:
            ptrs[l] = _p(arrs[l], C.c_float)
    pose = np.empty(7, np.float32)
    dp = None
    if ref_depth is not None:
        ref_depth = np.ascontiguousarray(ref_depth, np.uint16)
        dp = _p(ref_depth, C.c_uint16)
    cap = (p.first_level - p.last_level + 1) * p.max_iters if want_trace else 0
    tr = (Trace * max(cap, 1))()
    n = C.c_int32(cap)
    st = lib().uwo_align_pair_points(C.byref(p), _p(ref_gray, C.c_uint8), _p(tgt_gray, C.c_uint8), dp, ptrs, counts,
                                     _p(pose, C.c_float), tr if want_trace else None, C.byref(n) if want_trace else None)
    return st, pose, [trace_to_dict(tr[i]) for i in range(n.value)] if want_trace else []


def patch_points(kp, depth0, w, h, cap=200 * 144):
    kp = np.ascontiguousarray(kp, np.float32).reshape(-1, 2)
    pts = np.empty((cap, 4), np.float32)
    dp = None
    if depth0 is not None:
        depth0 = np.ascontiguousarray(depth0, np.uint16)
        dp = _p(depth0, C.c_uint16)
    n = lib().uwo_patch_points(_p(kp, C.c_float), kp.shape[0], dp, w, h, _p(pts, C.c_float), cap)
    return pts[:min(n, cap)].copy(), n


def add_patch_points(pts, w, h, patch_size=5, cap=None):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
    cap = pts.shape[0] * patch_size * patch_size if cap is None else cap
    out = np.empty((max(cap, 1), 4), np.float32)
    n = lib().uwo_add_patch_points(_p(pts, C.c_float), pts.shape[0], w, h, patch_size, _p(out, C.c_float), cap)
    return out[:min(n, cap)].copy(), n


def candidate_points(mag, depth=None, threshold=20.0, grid=None):
    """grid = (w, h): the level's point grid where it is smaller than the image (mag.shape)."""
    mag = np.ascontiguousarray(mag, np.uint8)
    ih, iw = mag.shape
    w, h = grid if grid is not None else (iw, ih)
    pts = np.empty((w * h, 4), np.float32)
    dp = None
    if depth is not None:
        depth = np.ascontiguousarray(depth, np.uint16)
        dp = _p(depth, C.c_uint16)
    n = lib().uwo_candidate_points_ex(_p(mag, C.c_uint8), dp, iw, ih, w, h, C.c_double(threshold), _p(pts, C.c_float), w * h)
    return pts[:n].copy(), n


def bilinear_u8(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    lib().uwo_bilinear_u8.restype = C.c_float
    return float(lib().uwo_bilinear_u8(_p(img, C.c_uint8), img.shape[1], img.shape[0], C.c_float(x), C.c_float(y)))


def huber_weights(r):
    r = np.ascontiguousarray(r, np.float32)
    w = np.empty_like(r)
    lib().uwo_huber_weights(_p(r, C.c_float), r.size, _p(w, C.c_float))
    return w


def residual_jacobian_ex(img1, img2, gx1, gy1, pts, warped, L, z_factor=1.0, angle_factor=1.0, sampler=0):
    img1 = np.ascontiguousarray(img1, np.uint8)
    img2 = np.ascontiguousarray(img2, np.uint8)
    gx1 = np.ascontiguousarray(gx1, np.int16)
    gy1 = np.ascontiguousarray(gy1, np.int16)
    pts = np.ascontiguousarray(pts, np.float32)
    warped = np.ascontiguousarray(warped, np.float32)
    n = pts.shape[0]
    J = np.empty((n, 6), np.float32)
    r = np.empty(n, np.float32)
    idx = np.empty(n, np.int32)
    nv = lib().uwo_residual_jacobian_ex(_p(img1, C.c_uint8), _p(img2, C.c_uint8), _p(gx1, C.c_int16), _p(gy1, C.c_int16),
                                        _p(pts, C.c_float), _p(warped, C.c_float), n, C.byref(L),
                                        C.c_float(z_factor), C.c_float(angle_factor), int(sampler),
                                        _p(J, C.c_float), _p(r, C.c_float), _p(idx, C.c_int32))
    return J[:nv].copy(), r[:nv].copy(), idx[:nv].copy()


def optimal_new_camera_matrix(K4, dist4, in_w, in_h, new_w, new_h, alpha=1.0):
    K = np.ascontiguousarray(K4, np.float32)
    d = np.ascontiguousarray(dist4, np.float32)
    out = np.empty(4, np.float64)
    lib().uwo_optimal_new_camera_matrix(_p(K, C.c_float), _p(d, C.c_float), in_w, in_h, C.c_double(alpha), new_w, new_h,
                                        _p(out, C.c_double))
    return out


def init_undistort_maps(K4, dist4, newK4, w, h):
    K = np.ascontiguousarray(K4, np.float32)
    d = np.ascontiguousarray(dist4, np.float32)
    nk = np.ascontiguousarray(newK4, np.float64)
    m1 = np.empty((h, w, 2), np.int16)
    m2 = np.empty((h, w), np.uint16)
    lib().uwo_init_undistort_maps(_p(K, C.c_float), _p(d, C.c_float), _p(nk, C.c_double), w, h, _p(m1, C.c_int16),
                                  _p(m2, C.c_uint16))
    return m1, m2


def remap_linear(src, m1, m2):
    src = np.ascontiguousarray(src, np.uint8)
    m1 = np.ascontiguousarray(m1, np.int16)
    m2 = np.ascontiguousarray(m2, np.uint16)
    dh, dw = m2.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().uwo_remap_linear(_p(src, C.c_uint8), src.shape[1], src.shape[0], _p(m1, C.c_int16), _p(m2, C.c_uint16), dw, dh,
                           _p(dst, C.c_uint8))
    return dst


def calculate_roi(und):
    und = np.ascontiguousarray(und, np.uint8)
    roi = np.empty(4, np.int32)
    lib().uwo_calculate_roi(_p(und, C.c_uint8), und.shape[1], und.shape[0], _p(roi, C.c_int32))
    return roi
