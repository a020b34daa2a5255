"""The Python mirror of the reference's Tracker / LS / Frame surface (uw-slam_amd/tracker.py) driven like
System::Tracking() (src/System.cpp:193-223); results must equal the oracle's."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_system_tracking_sequences(O, synth):
    T = importlib.import_module("uw-slam_amd.tracker")
    capi = importlib.import_module("uw-slam_amd.capi")
    w, h = 160, 96
    f = 525.0 * w / 640.0
    intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
    K = np.array([[f, 0, intr[2]], [0, f, intr[3]], [0, 0, 1]], np.float32)
    ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=91)
    # dense variant (EstimatePose un-commented at src/System.cpp:220)
    tracker_ = T.Tracker(False)
    tracker_.InitializePyramid(w, h, K)
    tracker_.InitializeMasks()
    previous_frame_, current_frame_ = T.Frame(ref, id_frame=0), T.Frame(tgt, id_frame=1)
    tracker_.ApplyGradient(previous_frame_)
    tracker_.ApplyGradient(current_frame_)
    tracker_.ObtainAllPoints(previous_frame_)
    st = tracker_.EstimatePose(previous_frame_, current_frame_)
    so, pose_cpu, tr = O.align_pair(O.default_params(w, h, *intr), ref, tgt, want_trace=True)
    assert so == 0 and st["iterations"] == len(tr) and np.array_equal(previous_frame_.rigid_transformation_, pose_cpu)
    assert tracker_.w_ == [160, 80, 40, 20, 10] and tracker_.fx_[2] == np.float32(f / 4)
    gx, _ = O.scharr3(O.halve_u8(ref))
    assert np.array_equal(tracker_.GetFrameData(previous_frame_, 1, capi.PLANE_GRADX), gx)
    # the live variant: ObtainPatchesPoints + EstimatePoseFeatures (src/System.cpp:221-222)
    rng = np.random.default_rng(1)
    previous_frame_.keypoints_ = rng.uniform([6, 6], [w - 7, h - 7], (60, 2)).astype(np.float32)
    tracker_.ObtainPatchesPoints(previous_frame_)
    st = tracker_.EstimatePoseFeatures(previous_frame_, current_frame_)
    feat = dict(first_level=0, last_level=0, max_iters=10, gain=1.0, z_factor=0.002, handoff_scale_t=1)
    pts, _ = O.patch_points(previous_frame_.keypoints_, None, w, h)
    so, pose_f, tr = O.align_pair_points(O.default_params(w, h, *intr, **feat), ref, tgt, {0: pts}, want_trace=True)
    assert so == 0 and st["iterations"] == len(tr) and np.array_equal(previous_frame_.rigid_transformation_, pose_f)
    # the solver constants are restored afterwards: the dense call gives the dense answer again
    tracker_.EstimatePose(previous_frame_, current_frame_)
    assert np.array_equal(previous_frame_.rigid_transformation_, pose_cpu)
    # semi-dense producer
    tracker_.ObtainCandidatePoints(previous_frame_)
    mag = O.gradient_mag(*O.scharr3(ref))
    assert np.array_equal(previous_frame_.candidatePoints_[0], O.candidate_points(mag)[0])
    # LS mirror
    ls = T.LS()                              # default-constructed, as the reference writes it (src/Tracker.cpp:537)
    ls.initialize(4)
    J = rng.normal(0, 5, (4, 6)).astype(np.float32)
    for i in range(4):
        ls.update(J[i], float(i - 1), 0.5)
    ls.finish()
    o = O.ls_new()
    for i in range(4):
        O.ls_update(o, J[i], float(i - 1), 0.5)
    A, b, e, n = O.ls_finish(o, True)
    assert ls.num_constraints == n == 4 and np.allclose(ls.A, A, rtol=1e-5, atol=1e-5) and np.allclose(ls.b, b, rtol=1e-5, atol=1e-5)
    empty = T.LS()
    empty.finish()                           # LS::finish divides by num_constraints = 0 unconditionally (src/LeastSquares.cpp:141-146)
    assert empty.num_constraints == 0 and np.isnan(empty.A).all() and np.isnan(empty.b).all()
    # Tracker::AddPatchPointsFeatures (src/Tracker.cpp:599-629) and Tracker::Mat2SE3 (:1596-1605)
    tab = np.array([[3.4, 2.6, 0.7, 1.0], [0.2, 0.4, 1.5, 1.0], [w - 1.2, h - 0.6, 0.9, 1.0], [40.5, 20.5, 1.1, 1.0]], np.float32)
    for lvl in (0, 2):
        L = tracker_._ctx.level_info(lvl)
        got = tracker_.AddPatchPointsFeatures(tab, lvl)
        want, n = O.add_patch_points(tab, L.w, L.h)
        assert n == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32)), lvl
    assert np.array_equal(got[:4], tab) and len(tracker_.AddPatchPointsFeatures(tab[3:], 0)) == 25
    v = np.array([0.02, -0.01, 0.03, 0.5, -0.25, 0.125], np.float32)
    se3 = tracker_.Mat2SE3(v.reshape(6, 1))
    assert np.array_equal(se3[:4], O.se3_exp(np.array([0, 0, 0, 0.02, -0.01, 0.03], np.float32))[:4]) and np.array_equal(se3[4:], v[3:])
    with pytest.raises(RuntimeError):
        tracker_.EstimatePose(T.Frame(ref), current_frame_)       # ApplyGradient not called on the new frame
    # FastEstimatePose: EstimatePose's terms under the prototype's schedule (4 -> 0, <= 50 iterations, gain 50)
    st = tracker_.FastEstimatePose(previous_frame_, current_frame_)
    fp = dict(n_levels=5, first_level=4, last_level=0, max_iters=50, early_exit=1, gain=50.0)
    so, pose_fast, tr = O.align_pair(O.default_params(w, h, *intr, **fp), ref, tgt, want_trace=True)
    assert so == 0 and st["iterations"] == len(tr) and np.array_equal(previous_frame_.rigid_transformation_, pose_fast)
    tracker_.EstimatePose(previous_frame_, current_frame_)          # constants restored
    assert np.array_equal(previous_frame_.rigid_transformation_, pose_cpu)
    # LS::updateSSE next to LS::update in one system: the oracle's LS fed the same way
    ls.initialize(0)
    o = O.ls_new()
    J4 = rng.normal(0, 3, (6, 4)).astype(np.float32); r4 = rng.normal(0, 2, 4).astype(np.float32); w4 = rng.uniform(0.2, 1, 4).astype(np.float32)
    ls.updateSSE(*J4, r4, w4)
    O.ls_update4(o, J4, r4, w4, quirk_plus6=True)
    ls.update(J[0], 1.5, 0.25)
    O.ls_update(o, J[0], 1.5, 0.25)
    ls.finishNoDivide()
    A, b, e, n = O.ls_finish(o, False)
    assert ls.num_constraints == n == 7
    assert np.allclose(ls.A, A, rtol=1e-5, atol=1e-5) and np.allclose(ls.b, b, rtol=1e-5, atol=1e-5) and abs(ls.error - e) <= 1e-5 * abs(e)


def test_slot_reuse_tells_the_evicted_frame(O, synth):
    """More live frames than device slots: the least recently used frame loses its slot, knows it (slot None, gradient
    flag cleared), and transparently uploads again — never another frame's planes under its name."""
    T = importlib.import_module("uw-slam_amd.tracker")
    w, h = 160, 96
    f = 525.0 * w / 640.0
    intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
    K = np.array([[f, 0, intr[2]], [0, f, intr[3]], [0, 0, 1]], np.float32)
    pairs = [synth.render_pair(w, h, *intr, seed=200 + s)[:2] for s in range(3)]
    trk = T.Tracker(False, max_frames=4)
    trk.InitializePyramid(w, h, K)
    frames = [(T.Frame(a), T.Frame(b)) for a, b in pairs]
    want = [O.align_pair(O.default_params(w, h, *intr), a, b)[1] for a, b in pairs]
    for (fa, fb) in frames:                       # six frames through four slots
        trk.ApplyGradient(fa)
        trk.EstimatePose(fa, fb)
    assert frames[0][0]._slot is None and frames[0][1]._slot is None and not frames[0][0].obtained_gradients_
    assert frames[2][0]._slot is not None
    with pytest.raises(RuntimeError):
        trk.EstimatePose(frames[0][0], frames[0][1])            # re-bound, but its gradients were lost with the slot
    for k in (0, 1, 2, 0):
        fa, fb = frames[k]
        trk.ApplyGradient(fa)
        trk.EstimatePose(fa, fb)
        assert np.array_equal(fa.rigid_transformation_, want[k]), k
