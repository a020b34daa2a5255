"""GPU parity tests at the reference's own EUROC geometry (round 5): calibration/calibrationEUROC.xml:7-20 — 752 x 480 in,
736 x 480 out, fx = 458.654 != fy = 457.296 — with PYRAMID_LEVELS = 5 (src/Options.cpp:26).  Level widths 736 / 368 / 184 /
92 / 46 and 752 / 376 / 188 / 94 / 47: the coarsest ones are not multiples of four, so those levels go pixel by pixel
(VEC = 1) while every finer level keeps its four-pixel groups, and fx != fy takes the general Jacobian form in every launch
form (per-evaluation launches, the chained flow, the one-block coarse levels, the one-launch alignment).  Bit-identical to
the oracle throughout.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EUROC = (458.654, 457.296, 367.215, 248.375)          # calibration/calibrationEUROC.xml:16-21


@pytest.fixture(scope="module")
def capi():
    m = importlib.import_module("uw-slam_amd.capi")
    m.lib()
    return m


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


@pytest.mark.parametrize("width", [736, 752])
@pytest.mark.parametrize("sched", ["reference", "fixed"])
def test_euroc_geometry_every_launch_form_matches_the_oracle(capi, O, synth, width, sched):
    """5 levels at 736 / 752 x 480 with the EUROC calibration: one pair per call (the drop-in use), a handful (the chained
    flow) and a batch (per-evaluation launches + one-block coarse levels), the reference's early-exit schedule and a fixed
    one over all five levels: poses and iteration counts are the oracle's."""
    w, h = width, 480
    intr = EUROC if width == 752 else (EUROC[0], EUROC[1], EUROC[2] - 8.0, EUROC[3])
    over = dict(has_depth=0)                                            # EUROC is monocular: z = 1
    if sched == "fixed":
        over.update(n_levels=5, first_level=4, last_level=0, max_iters=4, early_exit=0)
    distinct = 3
    pairs = _pairs(synth, w, h, intr, distinct, 5100 + width, False)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, r, t, None, want_trace=True) for r, t, _ in pairs]
    assert all(st == 0 for st, _, _ in want)
    n = 20
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    lv = [ctx.level_info(l) for l in range(5)]
    assert [L.w % 4 for L in lv] == ([0, 0, 0, 0, 2] if width == 736 else [0, 0, 0, 2, 3])
    _load(ctx, pairs, n)
    for count in (1, 2, 5, n):
        ref = np.arange(count) * 2
        poses, stats = ctx.estimate_pose_batch(ref, ref + 1, raise_on_pair_failure=True)
        for i in range(count):
            st, pose_cpu, tr = want[i % distinct]
            assert stats[i]["iterations"] == len(tr), (count, i, stats[i], len(tr))
            assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (count, i, poses[i], pose_cpu)
    ctx.close()


@pytest.mark.parametrize("weights", [1, 2], ids=["tukey", "huber"])
def test_euroc_geometry_robust_weights(capi, O, synth, weights):
    """The same geometry on the general path: the scale pass and the weighted sums pixel by pixel on the 46-wide level, in
    groups of four below it, the 92 x 60 level in one block per pair (k_coarse_weighted, general Jacobian form)."""
    w, h = 736, 480
    intr = (EUROC[0], EUROC[1], EUROC[2] - 8.0, EUROC[3])
    over = dict(has_depth=0, weights=weights, n_levels=5, first_level=4, last_level=1, max_iters=4, early_exit=0)
    pairs = _pairs(synth, w, h, intr, 2, 5300, False)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, r, t, None, want_trace=True) for r, t, _ in pairs]
    n = 6
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    _load(ctx, pairs, n)
    for count in (1, n):
        ref = np.arange(count) * 2
        poses, stats = ctx.estimate_pose_batch(ref, ref + 1, raise_on_pair_failure=True)
        for i in range(count):
            st, pose_cpu, tr = want[i % 2]
            assert st == 0 and stats[i]["iterations"] == len(tr)
            assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (count, i)
    ctx.close()


@pytest.mark.parametrize("case", ["square_factors", "nonsquare_depth", "nonsquare_factors_depth"])
def test_general_jacobian_form_in_the_flow_kernels(capi, O, synth, case):
    """fx != fy and / or z / angle factors other than 1 on the launch forms that used to be reserved for square pixels with
    unit factors: the chained flow (a few pairs), the one-block coarse levels of a batch, the split batch with the update in
    its tail.  640 x 480, 4 levels x 3 iterations."""
    w, h = 640, 480
    intr = (525.0, 525.0, 319.5, 239.5) if case == "square_factors" else (EUROC[0], EUROC[1], 319.5 - 2.285, 248.375)
    depth = "depth" in case
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=3, early_exit=0, has_depth=int(depth))
    if "factors" in case:
        over.update(z_factor=0.5, angle_factor=1.25)
    distinct = 2
    pairs = _pairs(synth, w, h, intr, distinct, 5500, depth)
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, *p) for p in pairs]
    n = 34
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
    _load(ctx, pairs, n)
    for count in (1, 4, n):                  # 34 pairs of 640x480 run as two halves on two streams
        ref = np.arange(count) * 2
        poses, stats = ctx.estimate_pose_batch(ref, ref + 1, raise_on_pair_failure=True)
        for i in range(count):
            st, pose_cpu, _ = want[i % distinct]
            assert st == 0 and stats[i]["iterations"] == 12
            assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (case, count, i)
    ctx.close()


def test_gradients_of_reference_slots_only_at_scalar_level_widths(capi, O, synth):
    """uwt_track_batch_async with grad_refs_only on a pyramid whose coarsest levels are not multiples of four wide: the scalar
    Scharr tile takes the pairs' reference slots as a list too (it used to need whole groups of four)."""
    import torch
    w, h, n = 184, 120, 9                                         # 184 / 92 / 46 / 23
    intr = (150.0, 149.0, 91.5, 59.5)
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=4, early_exit=0, has_depth=1)
    pairs = _pairs(synth, w, h, intr, 3, 5700, True)
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
    # the gradient planes of a target slot were not touched, those of the reference slots are the oracle's at every level
    im = pairs[0][0]
    for l in range(4):
        if l:
            im = O.halve_u8(im)
        gx, gy = O.scharr3(im)
        assert np.array_equal(ctx.get_plane(0, l, capi.PLANE_GRADX), gx) and np.array_equal(ctx.get_plane(0, l, capi.PLANE_GRADY), gy), l
    ctx.close()
