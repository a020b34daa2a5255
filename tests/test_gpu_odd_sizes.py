"""GPU parity at the frame sizes the reference's own pipeline produces (round 6).  With a distorted camera System::CalculateROI
crops every frame to a DATA-DEPENDENT window — w_ = p2.x - p1.x, h_ = p2.y - p1.y (src/System.cpp:148-191, applied at
:232-236) — and the pyramid halves that with cv::resize(.., 0.5, 0.5) (:246-251): a level's image is cvRound(size / 2) of
the one above (half to even: 733 -> 366, 735 -> 368), with a partial last column / row where 2 x that exceeds the source,
while Tracker::InitializePyramid sizes the point grid with "size >> lvl" (src/Tracker.cpp:312-313) and the per-point loop
tests its bounds against the image (:450).  Every stage and every launch form, bit for bit against the oracle.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROI_SIZES = [(725, 465, 5), (733, 471, 5), (735, 479, 5)]                       # what a 752 x 480 EUROC crop looks like
SMALL_SIZES = [(163, 99, 4), (161, 97, 4), (165, 101, 4), (166, 98, 4), (167, 103, 4), (91, 57, 3), (154, 101, 4)]


@pytest.fixture(scope="module")
def capi():
    m = importlib.import_module("uw-slam_amd.capi")
    m.lib()
    return m


def _intr(w, h):
    f = 0.82 * w
    return (f, f * 0.997, w / 2 - 0.3, h / 2 + 0.2)       # fx != fy, like the reference's EUROC calibration


def _pairs(synth, w, h, intr, n, seed, depth):
    out = []
    for s in range(n):
        ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=seed + s, with_depth=depth, max_t=0.012, max_deg=0.6)
        out.append((ref, tgt, dep if depth else None))
    return out


def _load(ctx, pairs, n):
    frames = np.stack([f for i in range(n) for f in pairs[i % len(pairs)][:2]])
    depth = None
    if pairs[0][2] is not None:
        depth = np.stack([pairs[i % len(pairs)][2] for i in range(n) for _ in (0, 1)])
    ctx.upload_frames(0, frames, depth)
    ctx.build_pyramids(0, 2 * n)
    ctx.apply_gradient(0, 2 * n)


# ------------------------------------------------------------------ stages

@pytest.mark.one_arith
def test_resize_half_any_size_bit_exact(capi, O):
    ctx = capi.Context(capi.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=3, first_level=2, last_level=0))
    rng = np.random.default_rng(5)
    for (h, w) in ((480, 640), (471, 733), (479, 735), (465, 725), (7, 9), (5, 6), (6, 5), (3, 3), (2, 7), (7, 2), (9, 11), (10, 13),
                   (33, 130), (130, 33), (2, 2), (3, 2), (2, 3), (240, 367), (239, 368)):
        im = rng.integers(0, 256, (h, w)).astype(np.uint8)
        d16 = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        assert np.array_equal(ctx.resize_half_u8(im), O.resize_half_u8(im)), (h, w)
        assert np.array_equal(ctx.resize_half_u16(d16), O.resize_half_u16(d16)), (h, w)
    # ties of the partial cells go to even, whole cells round half up
    im = np.array([[1, 2, 2], [1, 2, 3], [2, 3, 0]], np.uint8)
    assert ctx.resize_half_u8(im).tolist() == [[2, 2], [2, 0]]
    with pytest.raises(capi.UwtError):
        ctx.resize_half_u8(np.zeros((1, 8), np.uint8))   # cvRound(0.5) = 0 rows
    ctx.close()


@pytest.mark.one_arith
@pytest.mark.parametrize("size", ROI_SIZES + SMALL_SIZES, ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("form", ["few", "batch"])
def test_level_geometry_pyramids_and_gradients(capi, O, size, form):
    """uwt_level_info (grid, image, pitch), every level of the image and depth pyramids and both gradient planes against the
    oracle's resize chain; a few frames (the one-launch forms where they apply) and a batch of them."""
    w, h, nl = size
    n = 3 if form == "few" else 11
    intr = _intr(w, h)
    over = dict(n_levels=nl, first_level=nl - 1, last_level=0, has_depth=1)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=n, max_pairs=1, **over))
    p = O.default_params(w, h, *intr, **over)
    for l in range(nl):
        a, b = ctx.level_info(l), O.level_intrinsics(p, l)
        for f in ("w", "h", "fx", "fy", "cx", "cy", "invfx", "invfy"):
            assert getattr(a, f) == getattr(b, f)
        assert (a.img_w, a.img_h) == (b.iw, b.ih) and a.pitch == (b.iw + 3) // 4 * 4
    rng = np.random.default_rng(w * 31 + h)
    frames = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    depth = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
    ctx.upload_frames(0, frames, depth)
    ctx.build_pyramids(0, n)
    ctx.apply_gradient(0, n)
    for slot in (0, n - 1):
        imgs, deps = O.pyramid(frames[slot], nl), O.pyramid(depth[slot], nl)
        for l in range(nl):
            assert np.array_equal(ctx.get_plane(slot, l, capi.PLANE_IMAGE), imgs[l]), (slot, l)
            assert np.array_equal(ctx.get_plane(slot, l, capi.PLANE_DEPTH), deps[l]), (slot, l)
            gx, gy = O.scharr3(imgs[l])
            assert np.array_equal(ctx.get_plane(slot, l, capi.PLANE_GRADX), gx), (slot, l)
            assert np.array_equal(ctx.get_plane(slot, l, capi.PLANE_GRADY), gy), (slot, l)
    # set_frame (a strided host image) lands in the pitched rows like the packed upload
    wide = np.zeros((h, w + 5), np.uint8)
    wide[:, :w] = frames[1]
    wd = np.zeros((h, w + 3), np.uint16)
    wd[:, :w] = depth[1]
    ctx.set_frame(0, wide[:, :w], wd[:, :w])
    assert np.array_equal(ctx.get_plane(0, 0, capi.PLANE_IMAGE), frames[1]) and np.array_equal(ctx.get_plane(0, 0, capi.PLANE_DEPTH), depth[1])
    # ... and so does a tightly packed one (one linear copy and a kernel that spreads the rows, like the batch uploads)
    ctx.set_frame(1, frames[2], depth[2])
    assert np.array_equal(ctx.get_plane(1, 0, capi.PLANE_IMAGE), frames[2]) and np.array_equal(ctx.get_plane(1, 0, capi.PLANE_DEPTH), depth[2])
    # a view into the corner of a parent image (the reference's images_[0] = distortion(ROI), src/System.cpp:235): the rows' span —
    # the bytes between the rows included — crosses in one copy and ends with the parent's last byte
    parent, parent_d = rng.integers(0, 256, (h + 7, w + 19)).astype(np.uint8), rng.integers(0, 65536, (h + 7, w + 19)).astype(np.uint16)
    ctx.set_frame(2, parent[7:, 19:], parent_d[7:, 19:])
    assert np.array_equal(ctx.get_plane(2, 0, capi.PLANE_IMAGE), parent[7:, 19:]) and np.array_equal(ctx.get_plane(2, 0, capi.PLANE_DEPTH), parent_d[7:, 19:])
    # a column out of a parent more than four times as wide: the 2-D copy
    wide5, wide5_d = rng.integers(0, 256, (h, 5 * w)).astype(np.uint8), rng.integers(0, 65536, (h, 5 * w)).astype(np.uint16)
    ctx.set_frame(2, wide5[:, w:2 * w], wide5_d[:, w:2 * w])
    assert np.array_equal(ctx.get_plane(2, 0, capi.PLANE_IMAGE), wide5[:, w:2 * w]) and np.array_equal(ctx.get_plane(2, 0, capi.PLANE_DEPTH), wide5_d[:, w:2 * w])
    # the asynchronous upload from page-locked memory: a first call (the staging area is created), a larger one (it grows)
    pg, pd = capi.pinned_empty((n, h, w), np.uint8), capi.pinned_empty((n, h, w), np.uint16)
    pg[:], pd[:] = frames[::-1], depth[::-1]
    ctx.upload_frames_async(0, pg[:1], pd[:1])
    ctx.upload_frames_async(1, pg[1:], pd[1:])
    ctx.sync()
    for slot in (0, 1, n - 1):
        assert np.array_equal(ctx.get_plane(slot, 0, capi.PLANE_IMAGE), frames[n - 1 - slot]), slot
        assert np.array_equal(ctx.get_plane(slot, 0, capi.PLANE_DEPTH), depth[n - 1 - slot]), slot
    ctx.close()


@pytest.mark.parametrize("size", [(733, 471, 5), (735, 479, 5), (163, 99, 4), (166, 98, 4)], ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("depth", [False, True], ids=["nodepth", "depth"])
def test_per_pixel_terms_bit_exact(capi, O, synth, size, depth):
    """Validity, residual and the six Jacobian entries of every grid point of every level (uint32 views), the valid count and the
    sums: the grid is (size >> lvl), the bounds and the sample clamp are the image's."""
    w, h, nl = size
    intr = _intr(w, h)
    over = dict(n_levels=nl, first_level=nl - 1, last_level=0, has_depth=int(depth))
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
    ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=w + h, max_t=0.03, max_deg=2.0, with_depth=depth, z=1.2)
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]) if depth else None)
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)
    p = O.default_params(w, h, *intr, **over)
    a_pyr, b_pyr = O.pyramid(ref, nl), O.pyramid(tgt, nl)
    d_pyr = O.pyramid(dep, nl) if depth else [None] * nl
    rng = np.random.default_rng(8)
    for lvl in range(nl):
        L = O.level_intrinsics(p, lvl)
        gx, gy = O.scharr3(a_pyr[lvl])
        pts = O.dense_points(d_pyr[lvl], L.w, L.h, lvl)
        pose = O.se3_exp((rng.normal(0, 1, 6) * [0.05, 0.05, 0.02, 0.01, 0.01, 0.03]).astype(np.float32))
        wp = O.warp(pts, pose, L)
        J, r, idx = O.residual_jacobian(a_pyr[lvl], b_pyr[lvl], gx, gy, pts, wp, L, p.z_factor, p.angle_factor)
        out = ctx.residual_jacobian(0, 1, lvl, pose)
        valid = np.zeros(L.w * L.h, np.uint8)
        valid[idx] = 1
        assert out["valid"].shape == (L.w * L.h,)
        assert np.array_equal(out["valid"], valid), lvl
        assert np.array_equal(out["r"][idx], r)
        assert np.array_equal(out["J"][idx].view(np.uint32), J.view(np.uint32))
        assert out["n_valid"] == len(idx) and out["sum_r2"] == int((r.astype(np.int64) ** 2).sum())
        # the same evaluation through the production instantiation (no dumps): the same sums
        acc = ctx.residual_jacobian(0, 1, lvl, pose, dump=False)
        assert acc["n_valid"] == len(idx) and acc["sum_r2"] == out["sum_r2"]
        Jd = J.astype(np.float64)
        A_ref = Jd.T @ Jd
        scale = np.sqrt(np.outer(np.diag(A_ref), np.diag(A_ref)))
        assert (np.abs(acc["A"] - A_ref) <= 2e-6 * scale + 1e-30).all()
    ctx.close()


# ------------------------------------------------------------------ whole alignments

@pytest.mark.parametrize("size", ROI_SIZES, ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("sched", ["reference", "fixed"])
def test_roi_sized_alignments_every_launch_form(capi, O, synth, size, sched):
    """One pair per call (the drop-in use), a few (the chained flow) and a batch (per-evaluation launches, one-block coarse
    levels, two streams), the reference's early-exit schedule and a fixed one over all levels, monocular like EUROC."""
    w, h, nl = size
    intr = _intr(w, h)
    over = dict(has_depth=0)
    if sched == "fixed":
        over.update(n_levels=nl, first_level=nl - 1, last_level=0, max_iters=3, early_exit=0)
    distinct = 3
    pairs = _pairs(synth, w, h, intr, distinct, 6100 + w, False)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, r, t, None, want_trace=True) for r, t, _ in pairs]
    assert all(st == 0 for st, _, _ in want)
    n = 34
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    _load(ctx, pairs, n)
    for count in (1, 2, 5, n):
        ref = np.arange(count) * 2
        poses, stats = ctx.estimate_pose_batch(ref, ref + 1, raise_on_pair_failure=True)
        for i in range(count):
            st, pose_cpu, tr = want[i % distinct]
            assert stats[i]["iterations"] == len(tr), (count, i, stats[i], len(tr))
            assert stats[i]["n_valid"] == tr[-1]["n_valid"]
            assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (count, i, poses[i], pose_cpu)
    ctx.close()


@pytest.mark.one_arith
def test_first_upload_after_create_survives_the_create_time_clears(capi):
    """uwt_create clears the planes of pitched levels with hipMemset — asynchronous, on the NULL stream, which the context's
    non-blocking streams do not wait for.  Until uwt_create waited for it, the zeros could land on top of the first upload's rows
    (12-17 of 30 contexts of this shape lost their first depth frame; profiles/r06/EXPERIMENTS.md 14).  Large planes make the clears
    take milliseconds; one frame goes into the first or the last slot at once and is read back."""
    w, h = 725, 465
    rng = np.random.default_rng(3)
    for case in range(8):
        slots = int(rng.integers(400, 1000))
        g = rng.integers(1, 256, (1, h, w), dtype=np.uint8)
        d = rng.integers(1, 40000, (1, h, w)).astype(np.uint16)
        ctx = capi.Context(capi.default_params(w, h, 0.8 * w, 0.8 * w, w / 2, h / 2, max_frames=slots, max_pairs=1, n_levels=1, first_level=0,
                                               last_level=0, has_depth=1, weights=case % 3))
        s = slots - 1 if case % 2 else 0
        ctx.upload_frames(s, g, d)
        assert np.array_equal(ctx.get_plane(s, 0, capi.PLANE_IMAGE), g[0]), (case, slots)
        assert np.array_equal(ctx.get_plane(s, 0, capi.PLANE_DEPTH), d[0]), (case, slots)
        ctx.close()


@pytest.mark.one_arith
@pytest.mark.parametrize("size", [(1279, 959, 5), (1281, 963, 6)], ids=lambda s: "%dx%dx%d" % s)
def test_large_odd_sizes_alignments(capi, O, synth, size):
    """Config 3's scale (1280 x 960) one pixel to either side: level 0 of 1.2 M pixels in the sliced launches, rows of 1279 / 1281
    bytes (pitch 1280 / 1284), with depth; one pair and a batch of five."""
    w, h, nl = size
    intr = _intr(w, h)
    over = dict(n_levels=nl, first_level=nl - 1, last_level=0, max_iters=2, early_exit=0, has_depth=1)
    pairs = _pairs(synth, w, h, intr, 2, 6900 + w, True)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, *pr, want_trace=True) for pr in pairs]
    n = 5
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    _load(ctx, pairs, n)
    for count in (1, n):
        ref = np.arange(count) * 2
        poses, stats = ctx.estimate_pose_batch(ref, ref + 1)
        for i in range(count):
            st, pose_cpu, tr = want[i % 2]
            assert stats[i]["status"] == st and stats[i]["iterations"] == len(tr) and stats[i]["n_valid"] == tr[-1]["n_valid"]
            assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (count, i, poses[i], pose_cpu)
    ctx.close()


@pytest.mark.parametrize("size", SMALL_SIZES, ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("depth", [False, True], ids=["nodepth", "depth"])
def test_small_odd_sizes_alignments(capi, O, synth, size, depth):
    """Both roundings of the resize chain, grids smaller than their images, dropped and partial columns and rows, with and
    without a depth plane; one pair, a few, a batch; fixed and early-exit schedules."""
    w, h, nl = size
    intr = _intr(w, h)
    distinct = 3
    pairs = _pairs(synth, w, h, intr, distinct, 6300 + w, depth)
    n = 12
    for sched in ("fixed", "early"):
        over = dict(n_levels=nl, first_level=nl - 1, last_level=0, has_depth=int(depth))
        over.update(dict(max_iters=6, early_exit=0) if sched == "fixed" else dict(max_iters=30, early_exit=1))
        po = O.default_params(w, h, *intr, **over)
        want = [O.align_pair(po, *pr, want_trace=True) for pr in pairs]
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
        _load(ctx, pairs, n)
        for count in (1, 3, n):
            ref = np.arange(count) * 2
            poses, stats = ctx.estimate_pose_batch(ref, ref + 1)
            for i in range(count):
                st, pose_cpu, tr = want[i % distinct]
                assert stats[i]["status"] == st
                if st == 0:
                    assert stats[i]["iterations"] == len(tr), (sched, count, i)
                    assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (sched, count, i, poses[i], pose_cpu)
        ctx.close()


@pytest.mark.parametrize("size", [(733, 471, 5), (165, 101, 4)], ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("weights,sampler", [(1, 0), (2, 0), (0, 1), (2, 1)], ids=["tukey", "huber", "bilinear", "bilinear_huber"])
def test_odd_sizes_on_the_general_path(capi, O, synth, size, weights, sampler):
    """Robust weights (the scale pass and the weighted sums, per-evaluation launches and the one-block coarse levels) and the
    bilinear sampler at odd sizes, with depth."""
    w, h, nl = size
    intr = _intr(w, h)
    over = dict(n_levels=nl, first_level=nl - 1, last_level=0, max_iters=3, early_exit=0, has_depth=1, weights=weights, sampler=sampler)
    pairs = _pairs(synth, w, h, intr, 2, 6500 + w, True)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, *pr, want_trace=True) for pr in pairs]
    n = 6
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    _load(ctx, pairs, n)
    for count in (1, n):
        ref = np.arange(count) * 2
        poses, stats = ctx.estimate_pose_batch(ref, ref + 1, raise_on_pair_failure=True)
        for i in range(count):
            st, pose_cpu, tr = want[i % 2]
            assert st == 0 and stats[i]["iterations"] == len(tr)
            if sampler == 0:
                assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (count, i, poses[i], pose_cpu)
            else:   # float residuals: sums agree to rounding, poses to the north-star tolerance
                assert np.linalg.norm(poses[i][4:] - pose_cpu[4:]) <= 1e-4 and np.linalg.norm(poses[i][:4] - pose_cpu[:4]) <= 1e-4
    ctx.close()


def test_asynchronous_batch_and_reference_slot_gradients_at_an_odd_size(capi, O, synth):
    """uwt_track_batch_async (pyramids + gradients of the reference slots only + alignment, nothing waits) at 733 x 471 x 5."""
    import torch
    w, h, nl, n = 733, 471, 5, 9
    intr = _intr(w, h)
    over = dict(n_levels=nl, first_level=nl - 1, last_level=0, max_iters=3, early_exit=0, has_depth=1)
    pairs = _pairs(synth, w, h, intr, 3, 6700, True)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, *p)[1] for p in pairs]
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    frames = np.stack([f for i in range(n) for f in pairs[i % 3][:2]])
    depth = np.stack([pairs[i % 3][2] for i in range(n) for _ in (0, 1)])
    ctx.upload_frames(0, frames, depth)
    buf = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    ref = np.arange(n, dtype=np.int32) * 2
    ctx.track_batch_async(0, 2 * n, ref, ref + 1, buf.data_ptr(), grad_refs_only=True)
    ctx.sync()
    poses = buf.cpu().numpy()
    for i in range(n):
        assert np.array_equal(poses[i].view(np.uint32), want[i % 3].view(np.uint32)), i
    ctx.close()


@pytest.mark.one_arith
def test_candidate_points_and_magnitude_at_an_odd_size(capi, O, synth):
    """gradient_ (the level's image), its mean over the whole image, the candidates over the point grid (src/Tracker.cpp:1314-1362)."""
    w, h, nl = 163, 99, 4
    intr = _intr(w, h)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, n_levels=nl, first_level=nl - 1, last_level=0))
    ref = synth.texture(w, h, 71)
    ctx.upload_frames(0, np.stack([ref, ref]))
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)
    p = O.default_params(w, h, *intr, n_levels=nl)
    imgs = O.pyramid(ref, nl)
    for l in range(nl):
        L = O.level_intrinsics(p, l)
        gx, gy = O.scharr3(imgs[l])
        mag = O.gradient_mag(gx, gy)
        assert np.array_equal(ctx.gradient_magnitude(0, l), mag), l
        pts, n = O.candidate_points(mag, None, 20.0, grid=(L.w, L.h))
        got, m = ctx.obtain_candidate_points(0, l, 20.0)
        assert m == n and np.array_equal(got, pts), l
    ctx.close()


def test_sizes_with_an_empty_grid_are_refused(capi):
    with pytest.raises(capi.UwtError):
        capi.Context(capi.default_params(7, 40, 10.0, 10.0, 3.0, 20.0, n_levels=4, first_level=3, last_level=0))   # 7 >> 3 = 0
    capi.Context(capi.default_params(9, 40, 10.0, 10.0, 4.0, 20.0, n_levels=4, first_level=3, last_level=0)).close()


@pytest.mark.parametrize("depth", [False, True], ids=["nodepth", "depth"])
def test_point_tables_and_patches_at_an_odd_size(capi, O, synth, depth):
    """The reference's LIVE flow (src/System.cpp:193-223: key points -> ObtainPatchesPoints -> EstimatePoseFeatures) and explicit
    per-level tables (candidates on every level -> EstimatePose) on a 163 x 99 frame: patch points read depth through pitched rows,
    table rows index the level's IMAGE (a candidate of the grid's last column sits one short of the image's last), the candidates
    walk the grid."""
    w, h, nl = 163, 99, 4
    intr = _intr(w, h)
    ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=6900, with_depth=True, max_t=0.01, max_deg=0.5)
    feat = dict(n_levels=nl, first_level=0, last_level=0, max_iters=10, early_exit=1, gain=1.0, z_factor=0.002, handoff_scale_t=1, has_depth=int(depth))
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **feat))
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]) if depth else None)
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)
    rng = np.random.default_rng(19)
    kp = rng.uniform([6, 6], [w - 7, h - 7], (120, 2)).astype(np.float32)
    kp[0] = (w - 1.5, h - 1.5)                                   # a patch that leaves the frame on two sides
    pts, n = ctx.obtain_patch_points(0, kp)
    pts_cpu, n_cpu = O.patch_points(kp, dep if depth else None, w, h)
    assert n == n_cpu and np.array_equal(pts, pts_cpu)
    pose, st = ctx.estimate_pose_points(0, 1, {0: pts})
    so, pose_cpu, tr = O.align_pair_points(O.default_params(w, h, *intr, **feat), ref, tgt, {0: pts_cpu}, ref_depth=dep if depth else None, want_trace=True)
    assert so == st["status"] == 0 and st["iterations"] == len(tr) and np.array_equal(pose.view(np.uint32), pose_cpu.view(np.uint32))
    ctx.close()
    # candidates of every level as tables for EstimatePose (levels 3 -> 1)
    over = dict(n_levels=nl, first_level=nl - 1, last_level=1, max_iters=5, early_exit=0, has_depth=int(depth))
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]) if depth else None)
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)
    p = O.default_params(w, h, *intr, **over)
    imgs = O.pyramid(ref, nl)
    tables, tables_cpu = {}, {}
    for l in range(1, nl):
        L = O.level_intrinsics(p, l)
        mag = O.gradient_mag(*O.scharr3(imgs[l]))
        want, nw = O.candidate_points(mag, None, 5.0, grid=(L.w, L.h))     # z = 1 tables (the depth-less producer): x, y, 1, 1
        got, m = ctx.obtain_candidate_points(0, l, 5.0) if not depth else (want, nw)
        assert m == nw and np.array_equal(got, want), l
        tables[l], tables_cpu[l] = got, want
        assert nw > 20
    pose, st = ctx.estimate_pose_points(0, 1, tables)
    so, pose_cpu, tr = O.align_pair_points(p, ref, tgt, tables_cpu, ref_depth=dep if depth else None, want_trace=True)
    assert so == st["status"] == 0 and st["iterations"] == len(tr) and np.array_equal(pose.view(np.uint32), pose_cpu.view(np.uint32))
    ctx.close()
