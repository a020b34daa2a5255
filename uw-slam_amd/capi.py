"""ctypes binding of libuwt_hip.so (the C ABI in include/uwt.h).  Plumbing only: every call lands in the HIP
library; there is no Python or CPU implementation of the path here.  Import fails loudly when the shared
library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuwt_hip.so")

OK, ERR_INVALID_ARG, ERR_NO_VALID_POINTS, ERR_HIP, ERR_NO_DEVICE, ERR_CAPACITY, ERR_PAIR_FAILED = range(7)
PLANE_IMAGE, PLANE_DEPTH, PLANE_GRADX, PLANE_GRADY = range(4)
MAX_LEVELS = 8
ARITH_OPENCV, ARITH_LEGACY = 0, 1   # uwt_params.arith (include/uwt.h: enum uwt_arith)


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("n_levels", C.c_int32), ("first_level", C.c_int32), ("last_level", C.c_int32), ("max_iters", C.c_int32),
        ("epsilon", C.c_float), ("gain", C.c_float), ("z_factor", C.c_float), ("angle_factor", C.c_float),
        ("depth_scale", C.c_float), ("initial_error", C.c_float),
        ("early_exit", C.c_int32), ("has_depth", C.c_int32), ("handoff_scale_t", C.c_int32),
        ("accumulate_f64", C.c_int32), ("sampler", C.c_int32), ("weights", C.c_int32), ("max_frames", C.c_int32), ("max_pairs", C.c_int32), ("device", C.c_int32),
        ("arith", C.c_int32),
    ]


class Tuning(C.Structure):
    """uwt_tuning: launch-shape switches of a context (never what is computed)."""
    _fields_ = [
        ("split", C.c_int32), ("split_min", C.c_int32), ("split_min_px", C.c_int64), ("stream_bytes", C.c_int64),
        ("tail_update", C.c_int32), ("target_blocks", C.c_int32), ("coarse", C.c_int32), ("coarse_batch_px", C.c_int32),
        ("coarse_weighted", C.c_int32), ("overlap_gradients", C.c_int32), ("first_poll", C.c_int32), ("chained", C.c_int32),
        ("speculation", C.c_int32), ("fused_stages", C.c_int32), ("pyramid_batch", C.c_int32), ("typed_loads", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class Level(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("invfx", C.c_float), ("invfy", C.c_float),
                ("img_w", C.c_int32), ("img_h", C.c_int32), ("pitch", C.c_int32)]   # w, h: point grid; img_*: the level's image


class Stats(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_int32), ("n_valid", C.c_int32), ("error", C.c_float)]


class Accum(C.Structure):
    _fields_ = [("A", C.c_double * 21), ("jtr", C.c_double * 6), ("sum_r2", C.c_int64),
                ("n_valid", C.c_int32), ("pad", C.c_int32)]


# every symbol include/uwt.h declares (tests check the library exports all of them)
SYMBOLS = [
    "uwt_abi_version", "uwt_source_id", "uwt_status_string", "uwt_last_error", "uwt_default_params", "uwt_create", "uwt_destroy",
    "uwt_level_info", "uwt_set_frame", "uwt_upload_frames", "uwt_upload_frames_async", "uwt_host_alloc", "uwt_host_free", "uwt_plane_device_ptr", "uwt_get_plane",
    "uwt_build_pyramids", "uwt_apply_gradient", "uwt_estimate_pose_batch", "uwt_track_batch_async", "uwt_track_batch_host_async", "uwt_wait_ticket", "uwt_sync", "uwt_set_deferred",
    "uwt_stream", "uwt_profile_enable", "uwt_profile_read", "uwt_profile_read_levels", "uwt_profile_clock", "uwt_halve_u8", "uwt_halve_u16", "uwt_scharr3",
    "uwt_half_size", "uwt_resize_half_u8", "uwt_resize_half_u16",
    "uwt_warp", "uwt_residual_jacobian", "uwt_ls_accumulate", "uwt_se3_exp", "uwt_se3_mul", "uwt_se3_matrix",
    "uwt_se3_handoff", "uwt_solve_delta", "uwt_accumulate_trajectory", "uwt_accumulate_trajectory_scan",
    "uwt_residual_jacobian_weighted", "uwt_estimate_pose_points", "uwt_gradient_magnitude",
    "uwt_obtain_candidate_points", "uwt_obtain_candidate_points_batch", "uwt_obtain_patch_points", "uwt_add_patch_points",
    "uwt_ingest_create", "uwt_ingest_destroy", "uwt_ingest_maps", "uwt_ingest_undistort", "uwt_ingest_calculate_roi",
    "uwt_ingest_frame", "uwt_update_params", "uwt_get_params", "uwt_ls_accumulate_sse", "uwt_robust_weights",
    "uwt_get_tuning", "uwt_set_tuning",
]

_lib = None


class UwtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("uwt status %d: %s" % (status, msg))
        self.status = status


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If libuwt_hip.so pulled in the
    system copy first, a later `import torch` would load a second HIP runtime into the process (two device states,
    "No HIP GPUs are available", unordered streams).  Loading torch's copy first makes both use one runtime; without
    torch installed the system runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Loads libuwt_hip.so; raises if it is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libuwt_hip.so not built at %s — run __graft_entry__.build()" % LIB_PATH)
        _share_torch_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _lib.uwt_status_string.restype = C.c_char_p
        _lib.uwt_last_error.restype = C.c_char_p
        _lib.uwt_source_id.restype = C.c_char_p
        _lib.uwt_source_id.argtypes = []
        _lib.uwt_last_error.argtypes = [C.c_void_p]
    return _lib


def source_id():
    """uwt_source_id(): sha256 of the sources and flags the loaded library was built from"""
    return lib().uwt_source_id().decode()


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


DEFAULT_ARITH = None   # None: the library's default (ARITH_OPENCV); the parity suite sets it to run every test under both sets


def default_params(width, height, fx, fy, cx, cy, **over):
    p = Params()
    st = lib().uwt_default_params(C.byref(p), width, height, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy))
    if st:
        raise UwtError(st, "uwt_default_params")
    if DEFAULT_ARITH is not None:
        p.arith = DEFAULT_ARITH
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class _Pinned:
    """Owner of one uwt_host_alloc block; the numpy view below keeps it alive."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        st = lib().uwt_host_alloc(C.c_size_t(nbytes), C.byref(self.ptr))
        if st:
            raise UwtError(st, "uwt_host_alloc(%d)" % nbytes)

    def __del__(self):
        try:
            lib().uwt_host_free(self.ptr)
        except Exception:
            pass


class _PinnedArray(np.ndarray):
    """ndarray view that keeps its page-locked block alive (views and slices inherit the reference)."""

    def __new__(cls, arr, owner):
        obj = arr.view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def pinned_empty(shape, dtype):
    """Page-locked numpy array (hipHostMalloc through the C ABI) for upload_frames_async."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    owner = _Pinned(nbytes)
    buf = (C.c_uint8 * nbytes).from_address(owner.ptr.value)
    arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
    return _PinnedArray(arr, owner)


class Context:
    """Owns one uwt_ctx (device buffers + stream) — the state behind a reference `Tracker` instance."""

    def __init__(self, params, tuning=None):
        """tuning: dict of uwt_tuning fields to change from their defaults (launch shapes only; the A/B tools and the tests
        that run one form against another use it)."""
        self.params = params
        self._h = C.c_void_p()
        st = lib().uwt_create(C.byref(params), C.byref(self._h))
        if st:
            raise UwtError(st, lib().uwt_status_string(st).decode())
        self.w, self.h = params.width, params.height
        if tuning:
            self.set_tuning(**tuning)

    def get_tuning(self):
        t = Tuning()
        self._chk(lib().uwt_get_tuning(self._h, C.byref(t)))
        return t

    def set_tuning(self, **over):
        """Change launch-shape switches of the live context (uwt_set_tuning)."""
        t = self.get_tuning()
        for k, v in over.items():
            if k == "reserved" or not hasattr(t, k):
                raise AttributeError(k)
            setattr(t, k, int(v))
        self._chk(lib().uwt_set_tuning(self._h, C.byref(t)))

    def close(self):
        if self._h:
            lib().uwt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, allow=()):
        if st and st not in allow:
            raise UwtError(st, lib().uwt_last_error(self._h).decode())
        return st

    def update_params(self, **over):
        """Change solver constants of the live context (uwt_update_params)."""
        p = Params()
        self._chk(lib().uwt_get_params(self._h, C.byref(p)))
        for k, v in over.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        self._chk(lib().uwt_update_params(self._h, C.byref(p)))
        self.params = p

    # -- frames
    def level_info(self, lvl):
        L = Level()
        self._chk(lib().uwt_level_info(self._h, lvl, C.byref(L)))
        return L

    def set_frame(self, slot, gray, depth=None):
        gray = np.asarray(gray)
        assert gray.dtype == np.uint8 and gray.shape == (self.h, self.w) and gray.strides[1] == 1
        dp, ds = None, 0
        if depth is not None:
            depth = np.asarray(depth)
            assert depth.dtype == np.uint16 and depth.shape == (self.h, self.w) and depth.strides[1] == 2
            dp, ds = _p(depth, C.c_uint16), depth.strides[0]
        self._chk(lib().uwt_set_frame(self._h, slot, _p(gray, C.c_uint8), C.c_size_t(gray.strides[0]), dp, C.c_size_t(ds)))

    def upload_frames(self, first_slot, gray, depth=None):
        gray = np.ascontiguousarray(gray, np.uint8)
        n = gray.shape[0]
        assert gray.shape[1:] == (self.h, self.w)
        dp = None
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
            assert depth.shape == gray.shape
            dp = _p(depth, C.c_uint16)
        self._chk(lib().uwt_upload_frames(self._h, first_slot, n, _p(gray, C.c_uint8), dp))

    def upload_frames_async(self, first_slot, gray, depth=None):
        """gray / depth: page-locked arrays (pinned_empty) that stay untouched until the next sync()."""
        assert gray.flags["C_CONTIGUOUS"] and gray.dtype == np.uint8
        if depth is not None:
            assert depth.flags["C_CONTIGUOUS"] and depth.dtype == np.uint16
        self._chk(lib().uwt_upload_frames_async(self._h, first_slot, gray.shape[0], _p(gray, C.c_uint8),
                                                _p(depth, C.c_uint16) if depth is not None else None))

    def plane_device_ptr(self, slot, lvl, plane):
        out = C.c_void_p()
        self._chk(lib().uwt_plane_device_ptr(self._h, slot, lvl, plane, C.byref(out)))
        return out.value

    def get_plane(self, slot, lvl, plane):
        L = self.level_info(lvl)
        dt = {PLANE_IMAGE: np.uint8, PLANE_DEPTH: np.uint16, PLANE_GRADX: np.int16, PLANE_GRADY: np.int16}[plane]
        out = np.empty((L.img_h, L.img_w), dt)   # the level's image (larger than its point grid at odd sizes)
        self._chk(lib().uwt_get_plane(self._h, slot, lvl, plane, out.ctypes.data_as(C.c_void_p)))
        return out

    def build_pyramids(self, first_slot, n):
        self._chk(lib().uwt_build_pyramids(self._h, first_slot, n))

    def apply_gradient(self, first_slot, n):
        self._chk(lib().uwt_apply_gradient(self._h, first_slot, n))

    # -- tracking
    def estimate_pose_batch(self, ref_slots, tgt_slots, raise_on_pair_failure=False):
        ref = np.ascontiguousarray(ref_slots, np.int32)
        tgt = np.ascontiguousarray(tgt_slots, np.int32)
        n = ref.size
        poses = np.empty((n, 7), np.float32)
        stats = (Stats * n)()
        st = lib().uwt_estimate_pose_batch(self._h, n, _p(ref, C.c_int32), _p(tgt, C.c_int32), _p(poses, C.c_float), stats)
        self._chk(st, allow=() if raise_on_pair_failure else (ERR_PAIR_FAILED,))
        return poses, [dict(status=s.status, iterations=s.iterations, n_valid=s.n_valid, error=s.error) for s in stats]

    def track_batch_async(self, first_slot, n_frames, ref_slots, tgt_slots, d_poses_ptr, d_stats_ptr=None, grad_refs_only=True):
        ref = np.ascontiguousarray(ref_slots, np.int32)
        tgt = np.ascontiguousarray(tgt_slots, np.int32)
        self._chk(lib().uwt_track_batch_async(self._h, first_slot, n_frames, int(grad_refs_only), ref.size, _p(ref, C.c_int32),
                                              _p(tgt, C.c_int32), C.c_void_p(d_poses_ptr),
                                              C.c_void_p(d_stats_ptr) if d_stats_ptr else None))

    def track_batch_host_async(self, first_slot, n_frames, ref_slots, tgt_slots, h_poses, h_stats=None, grad_refs_only=True):
        """h_poses: pinned float32 [n, 7]; h_stats: pinned int32 [n, 4] or None.  Returns the ticket for wait_ticket()."""
        ref = np.ascontiguousarray(ref_slots, np.int32)
        tgt = np.ascontiguousarray(tgt_slots, np.int32)
        t = C.c_int64()
        self._chk(lib().uwt_track_batch_host_async(self._h, first_slot, n_frames, int(grad_refs_only), ref.size, _p(ref, C.c_int32),
                                                   _p(tgt, C.c_int32), _p(h_poses, C.c_float),
                                                   h_stats.ctypes.data_as(C.c_void_p) if h_stats is not None else None,
                                                   C.byref(t)))
        return t.value

    def wait_ticket(self, ticket):
        self._chk(lib().uwt_wait_ticket(self._h, C.c_int64(ticket)))

    def sync(self):
        self._chk(lib().uwt_sync(self._h))

    def set_deferred(self, on=True):
        """build_pyramids / apply_gradient return once enqueued; the calls that deliver results wait as before."""
        self._chk(lib().uwt_set_deferred(self._h, int(on)))

    def stream(self):
        out = C.c_void_p()
        self._chk(lib().uwt_stream(self._h, C.byref(out)))
        return out.value

    def profile_enable(self, on=True):
        self._chk(lib().uwt_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, n, px = C.c_double(), C.c_int64(), C.c_int64()
        self._chk(lib().uwt_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(px)))
        return ms.value, n.value, px.value

    def profile_read_levels(self):
        """[(ms, launches)] by pyramid level for the residual launches since profiling was enabled."""
        n = self.params.n_levels
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self._chk(lib().uwt_profile_read_levels(self._h, ms, cnt, n))
        return [(ms[l], cnt[l]) for l in range(n)]

    def profile_clock(self):
        """Shader clock (GHz) inside the last profiled residual launch."""
        ghz = C.c_double()
        self._chk(lib().uwt_profile_clock(self._h, C.byref(ghz)))
        return ghz.value

    # -- per-stage entry points
    def halve_u8(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty((h // 2, w // 2), np.uint8)
        self._chk(lib().uwt_halve_u8(self._h, _p(img, C.c_uint8), w, h, _p(out, C.c_uint8)))
        return out

    def halve_u16(self, img):
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        out = np.empty((h // 2, w // 2), np.uint16)
        self._chk(lib().uwt_halve_u16(self._h, _p(img, C.c_uint16), w, h, _p(out, C.c_uint16)))
        return out

    def resize_half_u8(self, img):
        """cv::resize(img, Size(), 0.5, 0.5) on any size."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty((lib().uwt_half_size(h), lib().uwt_half_size(w)), np.uint8)
        self._chk(lib().uwt_resize_half_u8(self._h, _p(img, C.c_uint8), w, h, _p(out, C.c_uint8)))
        return out

    def resize_half_u16(self, img):
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        out = np.empty((lib().uwt_half_size(h), lib().uwt_half_size(w)), np.uint16)
        self._chk(lib().uwt_resize_half_u16(self._h, _p(img, C.c_uint16), w, h, _p(out, C.c_uint16)))
        return out

    def scharr3(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        gx = np.empty((h, w), np.int16)
        gy = np.empty((h, w), np.int16)
        self._chk(lib().uwt_scharr3(self._h, _p(img, C.c_uint8), w, h, _p(gx, C.c_int16), _p(gy, C.c_int16)))
        return gx, gy

    def warp(self, lvl, pts, pose):
        pts = np.ascontiguousarray(pts, np.float32)
        pose = np.ascontiguousarray(pose, np.float32)
        out = np.empty_like(pts)
        self._chk(lib().uwt_warp(self._h, lvl, _p(pts, C.c_float), pts.shape[0], _p(pose, C.c_float), _p(out, C.c_float)))
        return out

    def residual_jacobian(self, ref_slot, tgt_slot, lvl, pose, dump=True):
        pose = np.ascontiguousarray(pose, np.float32)
        L = self.level_info(lvl)
        n = L.w * L.h
        acc = Accum()
        J = np.empty((n, 6), np.float32) if dump else None
        r = np.empty(n, np.float32) if dump else None
        v = np.empty(n, np.uint8) if dump else None
        self._chk(lib().uwt_residual_jacobian(self._h, ref_slot, tgt_slot, lvl, _p(pose, C.c_float), C.byref(acc),
                                              _p(J, C.c_float) if dump else None, _p(r, C.c_float) if dump else None,
                                              _p(v, C.c_uint8) if dump else None))
        A = np.zeros((6, 6))
        s = 0
        for i in range(6):
            for j in range(i, 6):
                A[i, j] = A[j, i] = acc.A[s]
                s += 1
        return dict(A=A, jtr=np.array(acc.jtr), sum_r2=int(acc.sum_r2), n_valid=int(acc.n_valid), J=J, r=r, valid=v)

    def ls_accumulate(self, J, r, w=None, divide=False):
        J = np.ascontiguousarray(J, np.float32)
        r = np.ascontiguousarray(r, np.float32)
        A = np.empty(36, np.float32)
        b = np.empty(6, np.float32)
        err, cnt = C.c_float(), C.c_int32()
        wp = None
        if w is not None:
            w = np.ascontiguousarray(w, np.float32)
            wp = _p(w, C.c_float)
        self._chk(lib().uwt_ls_accumulate(self._h, _p(J, C.c_float), _p(r, C.c_float), wp, r.size, int(divide),
                                          _p(A, C.c_float), _p(b, C.c_float), C.byref(err), C.byref(cnt)))
        return A.reshape(6, 6), b, err.value, cnt.value

    def ls_accumulate_sse(self, J, r, w=None, divide=False, count_quirk=True):
        J = np.ascontiguousarray(J, np.float32)
        r = np.ascontiguousarray(r, np.float32)
        A = np.empty(36, np.float32)
        b = np.empty(6, np.float32)
        err, cnt = C.c_float(), C.c_int32()
        wp = None
        if w is not None:
            w = np.ascontiguousarray(w, np.float32)
            wp = _p(w, C.c_float)
        self._chk(lib().uwt_ls_accumulate_sse(self._h, _p(J, C.c_float), _p(r, C.c_float), wp, r.size, int(divide),
                                              int(count_quirk), _p(A, C.c_float), _p(b, C.c_float), C.byref(err), C.byref(cnt)))
        return A.reshape(6, 6), b, err.value, cnt.value

    def se3_exp(self, xi):
        xi = np.ascontiguousarray(xi, np.float32)
        out = np.empty(7, np.float32)
        self._chk(lib().uwt_se3_exp(self._h, _p(xi, C.c_float), _p(out, C.c_float)))
        return out

    def se3_mul(self, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.empty(7, np.float32)
        self._chk(lib().uwt_se3_mul(self._h, _p(a, C.c_float), _p(b, C.c_float), _p(out, C.c_float)))
        return out

    def se3_matrix(self, pose):
        pose = np.ascontiguousarray(pose, np.float32)
        out = np.empty(16, np.float32)
        self._chk(lib().uwt_se3_matrix(self._h, _p(pose, C.c_float), _p(out, C.c_float)))
        return out.reshape(4, 4)

    def se3_handoff(self, pose, scale_t=0):
        pose = np.array(pose, np.float32)
        self._chk(lib().uwt_se3_handoff(self._h, _p(pose, C.c_float), int(scale_t)))
        return pose

    def solve_delta(self, A, b):
        A = np.ascontiguousarray(A, np.float32).reshape(36)
        b = np.ascontiguousarray(b, np.float32)
        d = np.empty(6, np.float32)
        Ai = np.empty(36, np.float32)
        ok = C.c_int32()
        self._chk(lib().uwt_solve_delta(self._h, _p(A, C.c_float), _p(b, C.c_float), _p(d, C.c_float), _p(Ai, C.c_float),
                                        C.byref(ok)))
        return d, Ai.reshape(6, 6), bool(ok.value)

    def accumulate_trajectory(self, poses, start=None, t_scale=1.0, reference_axes=False, scan=False):
        """scan=True: the parallel prefix product (equal to the sequential form to float rounding, not bit for bit)."""
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        start = np.array([0, 0, 0, 1, 0, 0, 0], np.float32) if start is None else np.ascontiguousarray(start, np.float32)
        out = np.empty_like(poses)
        fn = lib().uwt_accumulate_trajectory_scan if scan else lib().uwt_accumulate_trajectory
        self._chk(fn(self._h, _p(poses, C.c_float), poses.shape[0], _p(start, C.c_float), C.c_float(t_scale),
                     int(bool(reference_axes)), _p(out, C.c_float)))
        return out

    def estimate_pose_points(self, ref_slot, tgt_slot, tables):
        """tables: {level: n x 4 float32 array}"""
        nl = self.params.n_levels
        arrs = [None] * nl
        ptrs = (C.POINTER(C.c_float) * MAX_LEVELS)()
        counts = (C.c_int32 * MAX_LEVELS)()
        for l, t in tables.items():
            arrs[l] = np.ascontiguousarray(t, np.float32).reshape(-1, 4)
            counts[l] = arrs[l].shape[0]
            if arrs[l].shape[0]:
                ptrs[l] = _p(arrs[l], C.c_float)
        pose = np.empty(7, np.float32)
        st = Stats()
        rc = lib().uwt_estimate_pose_points(self._h, ref_slot, tgt_slot, ptrs, counts, _p(pose, C.c_float), C.byref(st))
        self._chk(rc, allow=(ERR_PAIR_FAILED,))
        return pose, dict(status=st.status, iterations=st.iterations, n_valid=st.n_valid, error=st.error)

    def robust_weights(self, residuals, kind=1, want_weights=True):
        """MedianMat / MedianAbsoluteDeviation / IdentityWeights (kind 0) / TukeyFunctionWeights (kind 1) of an N x 1 vector.
        Returns (weights or None, median, MAD)."""
        r = np.ascontiguousarray(residuals, np.float32).reshape(-1)
        w = np.empty(r.size, np.float32) if want_weights else None
        med, mad = C.c_float(), C.c_float()
        self._chk(lib().uwt_robust_weights(self._h, _p(r, C.c_float), r.size, int(kind), _p(w, C.c_float) if want_weights else None,
                                           C.byref(med), C.byref(mad)))
        return w, med.value, mad.value

    def gradient_magnitude(self, slot, lvl):
        L = self.level_info(lvl)
        out = np.empty((L.img_h, L.img_w), np.uint8)   # gradient_[lvl]: the level's image
        self._chk(lib().uwt_gradient_magnitude(self._h, slot, lvl, _p(out, C.c_uint8)))
        return out

    def obtain_candidate_points(self, slot, lvl, threshold=20.0, cap=None):
        L = self.level_info(lvl)
        cap = L.w * L.h if cap is None else cap
        pts = np.empty((max(cap, 1), 4), np.float32)
        cnt = C.c_int32()
        self._chk(lib().uwt_obtain_candidate_points(self._h, slot, lvl, C.c_double(threshold), _p(pts, C.c_float), cap,
                                                    C.byref(cnt)))
        return pts[:min(cnt.value, cap)].copy(), cnt.value

    def obtain_candidate_points_batch(self, first_slot, n_frames, lvl, threshold=20.0, cap=None):
        """Returns (list of [count_f, 4] arrays, counts)."""
        L = self.level_info(lvl)
        cap = L.w * L.h if cap is None else cap
        pts = np.empty((n_frames, max(cap, 1), 4), np.float32)
        cnt = np.zeros(n_frames, np.int32)
        self._chk(lib().uwt_obtain_candidate_points_batch(self._h, first_slot, n_frames, lvl, C.c_double(threshold), _p(pts, C.c_float),
                                                          cap, _p(cnt, C.c_int32)))
        return [pts[f, :min(int(cnt[f]), cap)].copy() for f in range(n_frames)], cnt

    def obtain_patch_points(self, slot, keypoints, cap=200 * 144):
        kp = np.ascontiguousarray(keypoints, np.float32).reshape(-1, 2)
        pts = np.empty((max(cap, 1), 4), np.float32)
        cnt = C.c_int32()
        self._chk(lib().uwt_obtain_patch_points(self._h, slot, _p(kp, C.c_float), kp.shape[0], _p(pts, C.c_float), cap,
                                                C.byref(cnt)))
        return pts[:min(cnt.value, cap)].copy(), cnt.value

    def add_patch_points(self, lvl, pts, patch_size=5, cap=None):
        """Tracker::AddPatchPointsFeatures (src/Tracker.cpp:599-629).  Returns (table, full count)."""
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
        cap = pts.shape[0] * patch_size * patch_size if cap is None else cap
        out = np.empty((max(cap, 1), 4), np.float32)
        cnt = C.c_int32()
        self._chk(lib().uwt_add_patch_points(self._h, lvl, _p(pts, C.c_float), pts.shape[0], patch_size, _p(out, C.c_float), cap,
                                             C.byref(cnt)))
        return out[:min(cnt.value, cap)].copy(), cnt.value

    def residual_jacobian_weighted(self, ref_slot, tgt_slot, lvl, pose):
        pose = np.ascontiguousarray(pose, np.float32)
        L = self.level_info(lvl)
        n = L.w * L.h
        acc = Accum()
        err, inv_mad = C.c_double(), C.c_float()
        J = np.empty((n, 6), np.float32)
        r = np.empty(n, np.float32)
        v = np.empty(n, np.uint8)
        w = np.empty(n, np.float32)
        self._chk(lib().uwt_residual_jacobian_weighted(self._h, ref_slot, tgt_slot, lvl, _p(pose, C.c_float), C.byref(acc),
                                                       C.byref(err), C.byref(inv_mad), _p(J, C.c_float), _p(r, C.c_float),
                                                       _p(v, C.c_uint8), _p(w, C.c_float)))
        A = np.zeros((6, 6))
        s = 0
        for i in range(6):
            for j in range(i, 6):
                A[i, j] = A[j, i] = acc.A[s]
                s += 1
        return dict(A=A, jtr=np.array(acc.jtr), sum_r2=int(acc.sum_r2), n_valid=int(acc.n_valid), J=J, r=r, valid=v, w=w,
                    err_num=err.value, inv_mad=inv_mad.value)


class Ingest:
    """Frame ingest next to the path: undistortion maps + fused remap/crop into a tracker slot."""

    def __init__(self, K4, dist4, in_w, in_h, out_w, out_h, device=0):
        K = np.ascontiguousarray(K4, np.float32)
        d = np.ascontiguousarray(dist4, np.float32)
        self._h = C.c_void_p()
        nk = np.empty(4, np.float32)
        st = lib().uwt_ingest_create(_p(K, C.c_float), _p(d, C.c_float), in_w, in_h, out_w, out_h, device, C.byref(self._h),
                                     _p(nk, C.c_float))
        if st:
            raise UwtError(st, lib().uwt_status_string(st).decode())
        self.newK = nk
        self.in_w, self.in_h, self.out_w, self.out_h = in_w, in_h, out_w, out_h

    def close(self):
        if self._h:
            lib().uwt_ingest_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st:
            raise UwtError(st, lib().uwt_status_string(st).decode())

    def maps(self):
        m1 = np.empty((self.out_h, self.out_w, 2), np.int16)
        m2 = np.empty((self.out_h, self.out_w), np.uint16)
        self._chk(lib().uwt_ingest_maps(self._h, _p(m1, C.c_int16), _p(m2, C.c_uint16)))
        return m1, m2

    def undistort(self, raw):
        raw = np.asarray(raw)
        assert raw.dtype == np.uint8 and raw.shape == (self.in_h, self.in_w) and raw.strides[1] == 1
        out = np.empty((self.out_h, self.out_w), np.uint8)
        self._chk(lib().uwt_ingest_undistort(self._h, _p(raw, C.c_uint8), C.c_size_t(raw.strides[0]), _p(out, C.c_uint8)))
        return out

    def calculate_roi(self, raw_first):
        raw = np.asarray(raw_first)
        roi = np.empty(4, np.int32)
        self._chk(lib().uwt_ingest_calculate_roi(self._h, _p(raw, C.c_uint8), C.c_size_t(raw.strides[0]), _p(roi, C.c_int32)))
        return roi

    def frame(self, ctx, slot, raw, x0, y0):
        raw = np.asarray(raw)
        assert raw.dtype == np.uint8 and raw.shape == (self.in_h, self.in_w) and raw.strides[1] == 1
        self._chk(lib().uwt_ingest_frame(self._h, ctx._h, slot, _p(raw, C.c_uint8), C.c_size_t(raw.strides[0]), x0, y0))
