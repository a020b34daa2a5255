"""frame ingest — getOptimalNewCameraMatrix / initUndistortRectifyMap restatement, fixed-point remap,
CalculateROI, fused remap+crop into a tracker slot."""
import importlib

ARITH_INDEPENDENT = True   # nothing here depends on the arithmetic set (tests/conftest.py): run once

import numpy as np
import pytest

EUROC_K = [458.654, 457.296, 367.215, 248.375]                        # calibration/calibrationEUROC.xml:18-22
EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]      # :25-29


def test_oracle_new_camera_matrix_and_maps_are_consistent(O):
    nk = O.optimal_new_camera_matrix(EUROC_K, EUROC_D, 752, 480, 736, 480)
    assert 250 < nk[0] < 458 and 250 < nk[1] < 458          # alpha = 1 keeps every source pixel => zoomed out
    m1, m2 = O.init_undistort_maps(EUROC_K, EUROC_D, nk, 736, 480)
    # float64 model of the map at a few pixels: distort the normalised ray and project with K
    k1, k2, p1, p2 = EUROC_D
    for (v, u) in [(240, 368), (100, 200), (400, 600), (10, 10)]:
        x, y = (u - nk[2]) / nk[0], (v - nk[3]) / nk[1]
        r2 = x * x + y * y
        kr = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        su, sv = EUROC_K[0] * xd + EUROC_K[2], EUROC_K[1] * yd + EUROC_K[3]
        fx, fy = m2[v, u] & 31, m2[v, u] >> 5
        assert abs((m1[v, u, 0] + fx / 32.0) - su) <= 1 / 32 and abs((m1[v, u, 1] + fy / 32.0) - sv) <= 1 / 32
    # the centre maps (almost) to itself; corners of the output sample outside the source => border 0
    img = np.full((480, 752), 200, np.uint8)
    und = O.remap_linear(img, m1, m2)
    assert und[240, 368] == 200 and und[0, 0] < 200 and und[479, 735] < 200 and (und == 0).sum() > 10000
    roi = O.calculate_roi(und)
    x, y, w, h = roi
    assert (und[y:y + h, x + w // 2] == 200).all() and (und[y + h // 2, x:x + w] == 200).all()
    assert und[y + h // 2, x - 6] == 0 or x - 6 < 0         # 5-pixel margin (System.cpp:180-183)


def test_oracle_remap_fixed_point_weights(O):
    src = np.array([[0, 100], [200, 40]], np.uint8)
    m1 = np.zeros((1, 3, 2), np.int16)
    m2 = np.array([[0, 16, 16 * 32 + 16]], np.uint16)        # (0,0), (0.5,0), (0.5,0.5)
    out = O.remap_linear(src, m1, m2)
    assert out.tolist() == [[0, 50, 85]]
    m1[0, 0] = (1, 1)                                        # right/bottom neighbours fall outside => contribute 0
    m2[0, 0] = 16 * 32 + 16
    assert O.remap_linear(src, m1, m2)[0, 0] == 10


@pytest.mark.gpu
def test_gpu_ingest_matches_oracle(O, synth):
    capi = importlib.import_module("uw-slam_amd.capi")
    ing = capi.Ingest(EUROC_K, EUROC_D, 752, 480, 736, 480)
    nk = O.optimal_new_camera_matrix(EUROC_K, EUROC_D, 752, 480, 736, 480)
    assert np.array_equal(ing.newK, nk.astype(np.float32))
    m1, m2 = ing.maps()
    o1, o2 = O.init_undistort_maps(EUROC_K, EUROC_D, nk, 736, 480)
    assert np.array_equal(m1, o1) and np.array_equal(m2, o2)
    raw = synth.texture(752, 480, seed=5)
    raw[raw == 0] = 1                                        # keep 0 as the "outside" marker CalculateROI looks for
    und = ing.undistort(raw)
    assert np.array_equal(und, O.remap_linear(raw, o1, o2))
    roi = ing.calculate_roi(raw)
    assert np.array_equal(roi, O.calculate_roi(und))
    # strided input (cv::Mat::step) gives the same frame
    big = np.zeros((480, 800), np.uint8)
    big[:, :752] = raw
    assert np.array_equal(ing.undistort(big[:, :752]), und)
    # fused remap + crop into a tracker slot: the ROI itself (any size is a frame size since round 6)
    x0, y0 = int(roi[0]), int(roi[1])
    cw, ch = int(roi[2]), int(roi[3])
    f = float(ing.newK[0])
    ctx = capi.Context(capi.default_params(cw, ch, f, float(ing.newK[1]), float(ing.newK[2]) - x0, float(ing.newK[3]) - y0,
                                           max_frames=2, max_pairs=1))
    ing.frame(ctx, 1, raw, x0, y0)
    assert np.array_equal(ctx.get_plane(1, 0, capi.PLANE_IMAGE), und[y0:y0 + ch, x0:x0 + cw])
    with pytest.raises(capi.UwtError):
        ing.frame(ctx, 1, raw, 736 - cw + 1, 0)              # window leaves the undistorted frame
    ing.close()


@pytest.mark.gpu
def test_euroc_pipeline_end_to_end_at_the_roi_size(O, synth):
    """The reference's own EUROC path (configs 1-2) from raw frame to pose, without the survey's "centre-crop to 640 x 480"
    deviation: remap with the rectification maps (src/System.cpp:233), System::CalculateROI on the first frame (:148-191) — a
    data-dependent, odd-sized window —, every frame cropped to it (:234), w_ / h_ = the ROI size (:186-190), the tracker initialised
    with that size and the UNSHIFTED new camera matrix (:105-123: the crop does not move cx, cy — reference quirk C-10), pyramids
    by cv::resize, EstimatePose.  GPU: uwt_ingest_frame straight into the tracker's slots at the ROI size; oracle: its restatement
    of each stage.  Poses bit for bit."""
    capi = importlib.import_module("uw-slam_amd.capi")
    ing = capi.Ingest(EUROC_K, EUROC_D, 752, 480, 736, 480)
    nk = O.optimal_new_camera_matrix(EUROC_K, EUROC_D, 752, 480, 736, 480)
    o1, o2 = O.init_undistort_maps(EUROC_K, EUROC_D, nk, 736, 480)
    f, cx, cy = 458.654, 367.215, 248.375
    frames = []
    for s in range(3):   # a raw (distorted-camera) frame and two moved views of it
        ref, tgt, _, _, _ = synth.render_pair(752, 480, f, f, cx, cy, seed=8100, max_t=0.004 * (s + 1), max_deg=0.2 * (s + 1))
        fr = ref if s == 0 else tgt
        fr = fr.copy()
        fr[fr == 0] = 1                                      # 0 is the "outside" marker CalculateROI looks for
        frames.append(fr)
    roi = ing.calculate_roi(frames[0])
    x0, y0, rw, rh = [int(v) for v in roi]
    assert (rw % 16, rh % 16) != (0, 0) and rw >= 600 and rh >= 300, roi      # a size no earlier round could track
    K = [float(np.float32(v)) for v in nk]                   # K_ = camera_model_->GetK(): the new camera matrix, not shifted by the crop
    over = dict(has_depth=0)                                 # EUROC is monocular; the reference schedule (levels 4 -> 1, early exit)
    ctx = capi.Context(capi.default_params(rw, rh, *K, max_frames=3, max_pairs=2, **over))
    und = []
    for i, fr in enumerate(frames):
        ing.frame(ctx, i, fr, x0, y0)
        und.append(O.remap_linear(fr, o1, o2)[y0:y0 + rh, x0:x0 + rw])
        assert np.array_equal(ctx.get_plane(i, 0, capi.PLANE_IMAGE), und[i]), i
    ctx.build_pyramids(0, 3)
    ctx.apply_gradient(0, 3)
    po = O.default_params(rw, rh, *K, **over)
    poses, stats = ctx.estimate_pose_batch([0, 1], [1, 2], raise_on_pair_failure=True)
    for i in range(2):
        st, pose_cpu, tr = O.align_pair(po, und[i], und[i + 1], None, want_trace=True)
        assert st == 0 and stats[i]["iterations"] == len(tr)
        assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (i, poses[i], pose_cpu)
    one, st1 = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)      # the drop-in call: one pair
    assert np.array_equal(one[0].view(np.uint32), poses[0].view(np.uint32))
    ctx.close()
    ing.close()
