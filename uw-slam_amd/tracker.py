"""Host-side mirror of the reference's Tracker / LS / Frame surface for the direct-tracking path, over the C ABI.

Same method names, argument meaning and call order as include/Tracker.h:97-170, include/LeastSquares.h:31-45 and
the Frame fields of include/System.h:85-100 that the path touches, so that System::Tracking()'s sequence
(src/System.cpp:193-223) reads the same here:

    tracker = Tracker(depth_available); tracker.InitializePyramid(w, h, K)
    tracker.ApplyGradient(prev); tracker.ApplyGradient(cur)
    tracker.ObtainAllPoints(prev)
    tracker.EstimatePose(prev, cur)     # -> prev.rigid_transformation_

Everything numeric happens in libuwt_hip.so; this file only moves buffers.  (The compiled-language mirror for C++
callers is include/uw_tracker.hpp.)
"""
import numpy as np

from . import capi

PYRAMID_LEVELS = 5  # src/Options.cpp:26


class Frame:
    """include/System.h:63-103 — only the members the tracker reads or writes."""

    def __init__(self, image, depth=None, id_frame=0):
        self.idFrame_ = id_frame
        self.images_ = [np.ascontiguousarray(image, np.uint8)]
        self.depths_ = [np.ascontiguousarray(depth, np.uint16)] if depth is not None else []
        self.depth_available_ = depth is not None
        self.obtained_gradients_ = False
        self.obtained_candidatePoints_ = False
        self.rigid_transformation_ = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)  # qx qy qz qw tx ty tz
        self.keypoints_ = np.zeros((0, 2), np.float32)       # cv::KeyPoint::pt of Frame::keypoints_
        self.candidatePoints_ = {}                           # level -> N x 4 [x y z w] when a sparse producer ran
        self._slot = None


class Tracker:
    """include/Tracker.h:90-170.  Solver constants are the locals of Tracker::EstimatePose (src/Tracker.cpp:364-372)
    unless overridden through **params (the uwt_params fields)."""

    def __init__(self, _depth_available=False, max_frames=16, device=0, **params):
        self.depth_available_ = bool(_depth_available)
        self._max_frames = max(2, int(max_frames))   # an alignment binds two frames at once (uw::Tracker clamps the same way)
        self._device = device
        self._over = params
        self._ctx = None
        self._owner = [None] * self._max_frames          # slot -> Frame holding it
        self._last_use = [0] * self._max_frames
        self._clock = 0

    def InitializePyramid(self, _width, _height, _K):
        K = np.asarray(_K, np.float32)
        over = dict(n_levels=PYRAMID_LEVELS, max_frames=self._max_frames, max_pairs=max(1, self._max_frames // 2),
                    has_depth=int(self.depth_available_), device=self._device)
        over.update(self._over)
        p = capi.default_params(int(_width), int(_height), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), **over)
        self._ctx = capi.Context(p)
        if LS._default_ctx is None or not LS._default_ctx._h:
            LS.bind(self._ctx)                  # `LS ls;` inside the tracking loop folds on the tracker's context
        self._ctx.set_deferred(True)   # one wait per frame, in EstimatePose (pyramids and gradients are enqueued only)
        lv = [self._ctx.level_info(l) for l in range(p.n_levels)]
        self.w_ = [L.w for L in lv]
        self.h_ = [L.h for L in lv]
        self.fx_ = [L.fx for L in lv]
        self.fy_ = [L.fy for L in lv]
        self.cx_ = [L.cx for L in lv]
        self.cy_ = [L.cy for L in lv]
        self.invfx_ = [L.invfx for L in lv]
        self.invfy_ = [L.invfy for L in lv]

    def InitializeMasks(self):
        """src/Tracker.cpp:342-359 builds masks nothing reads; kept as a no-op for call-order compatibility."""

    def _bind(self, frame):
        """System::AddFrame's pyramid loop (src/System.cpp:246-251): upload level 0, build levels 1.. on the GPU."""
        self._clock += 1
        if frame._slot is not None and self._owner[frame._slot] is frame:
            self._last_use[frame._slot] = self._clock
            return frame._slot
        # least-recently-used slot; the frame that held it is told (it uploads again when next used, and its gradients
        # are gone) instead of silently reading another frame's planes
        free = [i for i, o in enumerate(self._owner) if o is None]
        slot = free[0] if free else min(range(self._max_frames), key=lambda i: self._last_use[i])
        old = self._owner[slot]
        if old is not None:
            old._slot = None
            old.obtained_gradients_ = False
        self._owner[slot] = frame
        self._last_use[slot] = self._clock
        frame._slot = slot
        frame.obtained_gradients_ = False
        self._ctx.set_frame(slot, frame.images_[0], frame.depths_[0] if self.depth_available_ else None)
        self._ctx.build_pyramids(slot, 1)
        return slot

    def ApplyGradient(self, _frame):
        slot = self._bind(_frame)
        self._ctx.apply_gradient(slot, 1)
        _frame.obtained_gradients_ = True

    def ObtainAllPoints(self, _frame):
        """src/Tracker.cpp:1259-1310.  The dense table is the pixel grid; the kernels derive (x, y, z, w) in
        registers instead of materialising N x 4 floats."""
        self._bind(_frame)
        _frame.obtained_candidatePoints_ = True

    def WarpFunction(self, _points2warp, _rigid_transformation, _lvl):
        return self._ctx.warp(int(_lvl), _points2warp, _rigid_transformation)

    def EstimatePose(self, _previous_frame, _current_frame):
        a, b = self._bind(_previous_frame), self._bind(_current_frame)
        if not _previous_frame.obtained_gradients_:  # reference: empty cv::Mat
            raise RuntimeError("ApplyGradient(previous_frame) must run before EstimatePose (or its slot was reused since)")
        poses, stats = self._ctx.estimate_pose_batch([a], [b], raise_on_pair_failure=True)
        _previous_frame.rigid_transformation_ = poses[0]
        return stats[0]

    def FastEstimatePose(self, _previous_frame, _current_frame):
        """include/Tracker.h:124 — the vectorised prototype's schedule (levels PYRAMID_LEVELS-1 .. 0, <= 50 iterations,
        gain 50; src/Tracker.cpp:877-885, 1082) over EstimatePose's per-point terms.  The reference body is broken
        (residual ignores the warp :933-944, Jw1 column 0 zeroed :986) and is not reproduced."""
        a, b = self._bind(_previous_frame), self._bind(_current_frame)
        if not _previous_frame.obtained_gradients_:
            raise RuntimeError("ApplyGradient(previous_frame) must run before FastEstimatePose (or its slot was reused since)")
        keys = ("first_level", "last_level", "max_iters", "gain", "epsilon", "z_factor", "angle_factor", "early_exit", "handoff_scale_t")
        saved = {k: getattr(self._ctx.params, k) for k in keys}
        self._ctx.update_params(first_level=self._ctx.params.n_levels - 1, last_level=0, max_iters=50, gain=50.0, epsilon=0.001,
                                z_factor=1.0, angle_factor=1.0, early_exit=1, handoff_scale_t=0)
        try:
            poses, stats = self._ctx.estimate_pose_batch([a], [b], raise_on_pair_failure=True)
        finally:
            self._ctx.update_params(**saved)
        _previous_frame.rigid_transformation_ = poses[0]
        return stats[0]

    def ObtainCandidatePoints(self, _frame, gradient_threshold=20.0):
        """src/Tracker.cpp:1314-1398 (GRADIENT_THRESHOLD, src/Options.cpp:27)."""
        slot = self._bind(_frame)
        for l in range(self._ctx.params.n_levels):
            _frame.candidatePoints_[l] = self._ctx.obtain_candidate_points(slot, l, gradient_threshold)[0]
        _frame.obtained_candidatePoints_ = True

    def ObtainPatchesPoints(self, _previous_frame):
        """src/Tracker.cpp:1178-1257."""
        slot = self._bind(_previous_frame)
        _previous_frame.candidatePoints_[0] = self._ctx.obtain_patch_points(slot, _previous_frame.keypoints_)[0]
        _previous_frame.obtained_candidatePoints_ = True

    def AddPatchPointsFeatures(self, candidatePoints, lvl, patch_size=5):
        """src/Tracker.cpp:599-629 (include/Tracker.h:126): the N x 4 table plus the patch cells around every point."""
        return self._ctx.add_patch_points(lvl, candidatePoints, patch_size)[0]

    def Mat2SE3(self, _input):
        """src/Tracker.cpp:1596-1605 (include/Tracker.h:178): 6 x 1 [w1 w2 w3 x1 x2 x3] -> SE3(SO3::exp(w), x) — the
        translation is taken as it is, not through V(w).  Returns qx qy qz qw tx ty tz."""
        v = np.asarray(_input, np.float32).reshape(6)
        pose = self._ctx.se3_exp(np.array([0, 0, 0, v[0], v[1], v[2]], np.float32))   # rotation part of exp; V(w) * 0 = 0
        pose[4:] = v[3:]
        return pose

    def EstimatePoseFeatures(self, _previous_frame, _current_frame):
        """src/Tracker.cpp:632-872 — the reference's live variant (constants :634-640, :834, :856)."""
        a, b = self._bind(_previous_frame), self._bind(_current_frame)
        if not _previous_frame.obtained_gradients_:
            raise RuntimeError("ApplyGradient(previous_frame) must run before EstimatePoseFeatures (or its slot was reused since)")
        saved = {k: getattr(self._ctx.params, k) for k in ("first_level", "last_level", "max_iters", "gain", "z_factor",
                                                            "angle_factor", "handoff_scale_t", "early_exit")}
        self._ctx.update_params(first_level=0, last_level=0, max_iters=10, gain=1.0, z_factor=0.002, angle_factor=1.0,
                                handoff_scale_t=1, early_exit=1)
        try:
            pose, st = self._ctx.estimate_pose_points(a, b, {0: _previous_frame.candidatePoints_[0]})
        finally:
            self._ctx.update_params(**saved)
        _previous_frame.rigid_transformation_ = pose
        return st

    def ObtainGradientXY(self, _inputImage):
        """include/Tracker.h:197 — (gradientX, gradientY) = 3 x Scharr of a u8 image, CV_16S (src/Tracker.cpp:1133-1134)."""
        return self._ctx.scharr3(_inputImage)

    def MedianMat(self, _input):
        """include/Tracker.h:206, src/Tracker.cpp:1571-1594."""
        return self._ctx.robust_weights(_input, kind=0, want_weights=False)[1]

    def MedianAbsoluteDeviation(self, x):
        """include/Tracker.h:216, src/Tracker.cpp:1607-1619."""
        return self._ctx.robust_weights(x, kind=0, want_weights=False)[2]

    def IdentityWeights(self, _num_residuals):
        """include/Tracker.h:224, src/Tracker.cpp:1621-1624."""
        return self._ctx.robust_weights(np.zeros(int(_num_residuals), np.float32), kind=0)[0]

    def TukeyFunctionWeights(self, _residuals):
        """include/Tracker.h:235, src/Tracker.cpp:1626-1654."""
        return self._ctx.robust_weights(_residuals, kind=1)[0]

    def GetFrameData(self, _frame, lvl, plane):
        return self._ctx.get_plane(self._bind(_frame), lvl, plane)


class LS:
    """include/LeastSquares.h:26-50 over the GPU reduction: rows are buffered by update() / updateSSE() and folded by
    finish*() — scalar rows through uwt_ls_accumulate, 4-wide rows through uwt_ls_accumulate_sse (the two forms associate
    their products differently, src/LeastSquares.cpp:151-153 vs :205), the two added as finishNoDivide adds the lane sums
    onto A, b, error (:39-139)."""

    _default_ctx = None                         # the context `LS()` folds on: LS.bind(ctx), else a small one of its own

    @classmethod
    def bind(cls, ctx):
        """The context a default-constructed LS uses (the reference writes `LS ls;`, src/Tracker.cpp:537)."""
        cls._default_ctx = ctx

    def __init__(self, ctx=None, count_quirk=True):
        self._ctx_arg = ctx
        self.count_quirk = count_quirk          # "num_constraints += 6" per updateSSE call (src/LeastSquares.cpp:201)
        self.initialize(0)

    @property
    def _ctx(self):
        if self._ctx_arg is not None:
            return self._ctx_arg
        if LS._default_ctx is not None and not LS._default_ctx._h:
            LS._default_ctx = None              # the context it was bound to has been closed
        if LS._default_ctx is None:             # nothing bound: a minimal context just for the reductions
            LS._default_ctx = capi.Context(capi.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=1, first_level=0,
                                                               last_level=0, max_frames=2, max_pairs=1))
        return LS._default_ctx

    def initialize(self, max_num_constraints):
        self._J, self._r, self._w = [], [], []
        self._J4, self._r4, self._w4 = [], [], []
        self.A = np.zeros((6, 6), np.float32)
        self.b = np.zeros(6, np.float32)
        self.error = 0.0
        self.num_constraints = 0

    def update(self, J, res, weight):
        self._J.append(np.asarray(J, np.float32).reshape(6))
        self._r.append(res)
        self._w.append(weight)

    def updateSSE(self, J1, J2, J3, J4, J5, J6, res, weight):
        """Four points at once; J1..J6 hold Jacobian component k of the four points (include/LeastSquares.h:42-43)."""
        Jc = np.stack([np.asarray(x, np.float32).reshape(4) for x in (J1, J2, J3, J4, J5, J6)])   # 6 x 4
        self._J4.append(Jc.T.copy())                                                              # 4 rows of 6
        self._r4.append(np.asarray(res, np.float32).reshape(4))
        self._w4.append(np.asarray(weight, np.float32).reshape(4))

    def _fold(self, divide):
        A = np.zeros((6, 6), np.float32); b = np.zeros(6, np.float32); err = np.float32(0); n = 0
        if self._r:
            A1, b1, e1, n1 = self._ctx.ls_accumulate(np.stack(self._J), np.array(self._r, np.float32), np.array(self._w, np.float32), False)
            A, b, err, n = A + A1, b + b1, np.float32(err + np.float32(e1)), n + n1
        if self._r4:
            A2, b2, e2, n2 = self._ctx.ls_accumulate_sse(np.concatenate(self._J4), np.concatenate(self._r4), np.concatenate(self._w4),
                                                         False, self.count_quirk)
            A, b, err, n = A + A2, b + b2, np.float32(err + np.float32(e2)), n + n2
        if divide:   # unconditional, as LS::finish (src/LeastSquares.cpp:141-146) and uw::LS::finish: an empty system gives NaN
            fn = np.float32(n)
            with np.errstate(divide="ignore", invalid="ignore"):
                A, b, err = A / fn, b / fn, np.float32(np.float32(err) / fn)
        self.A, self.b, self.error, self.num_constraints = A.astype(np.float32), b.astype(np.float32), float(err), n

    def finishNoDivide(self):
        self._fold(False)

    def finish(self):
        self._fold(True)
