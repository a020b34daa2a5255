"""GPU tests at BASELINE.json's sizes, through the instantiations and launch paths the bench times:

* the production residual kernel (VEC = 4, packed pairs, SQUARE shortcut, f64 accumulation, no per-pixel dumps) checked
  per term: A, J^T r, sum r^2 and N of every level, bitwise after the single f32 rounding, against the oracle's sequential
  f64 sums — 640x480 / 4 levels and 1280x960 / 5 levels, depth plane present;
* BASELINE config 3 as stated: 1280x960, 5 levels, depth, 256 resident pairs;
* robust weights (Tukey, Huber) and the bilinear sampler at 640x480 with depth (BASELINE config 5's shape);
* the multi-process launch paths of bench.py on a one-GPU box: RCCL with a world of one, a refused --gpus 2, strong scaling.
"""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUM = (525.0, 525.0, 319.5, 239.5)
BIG = (1050.0, 1050.0, 639.5, 479.5)


@pytest.fixture(scope="module")
def capi():
    m = importlib.import_module("uw-slam_amd.capi")
    m.lib()  # raises if libuwt_hip.so is missing: no fallback
    return m


def _track_batch(ctx, n_frames, ref_s, tgt_s):
    """The bench's launch path: uwt_track_batch_async (pyramids, reference-only gradients, alignment) into device memory."""
    import torch
    n = len(ref_s)
    d_poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    d_stats = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # torch's fill kernels run on torch's stream, not on the context's
    ctx.track_batch_async(0, n_frames, ref_s, tgt_s, d_poses.data_ptr(), d_stats.data_ptr(), grad_refs_only=True)
    ctx.sync()
    st = d_stats.cpu().numpy()
    stats = [dict(status=int(r[0]), iterations=int(r[1]), n_valid=int(r[2])) for r in st]
    return d_poses.cpu().numpy(), stats


def _upload_pair(ctx, ref, tgt, dep):
    ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]))
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)


@pytest.mark.parametrize("shape", ["640x480x4", "1280x960x5"])
def test_production_instantiation_sums_bitwise_per_level(capi, O, synth, shape):
    w, h, levels = (640, 480, 4) if shape.startswith("640") else (1280, 960, 5)
    intr = TUM if w == 640 else BIG
    over = dict(n_levels=levels, first_level=levels - 1, last_level=0, max_iters=10, early_exit=0, has_depth=1)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
    ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=1200 + w, z=0.9, with_depth=True)
    _upload_pair(ctx, ref, tgt, dep)
    p = O.default_params(w, h, *intr, **over)
    rng = np.random.default_rng(w)
    a_img, b_img, dp = ref, tgt, dep
    for lvl in range(levels):
        if lvl:
            a_img, b_img, dp = O.halve_u8(a_img), O.halve_u8(b_img), O.halve_u16(dp)
        L = O.level_intrinsics(p, lvl)
        gx, gy = O.scharr3(a_img)
        pts = O.dense_points(dp, L.w, L.h, lvl)
        for k in range(2):   # identity (the first evaluation of every alignment) and a small random motion
            xi = np.zeros(6, np.float32) if k == 0 else (rng.normal(0, 1, 6) * [0.01, 0.01, 0.005, 0.003, 0.003, 0.005]).astype(np.float32)
            pose = O.se3_exp(xi)
            wp = O.warp(pts, pose, L)
            J, r, idx = O.residual_jacobian(a_img, b_img, gx, gy, pts, wp, L, p.z_factor, p.angle_factor)
            A_ref, b_ref = O.normal_equations(J, r, None, 1.0)       # sequential f64 sums, rounded to f32 once
            out = ctx.residual_jacobian(0, 1, lvl, pose, dump=False)  # the instantiation the alignment loop launches
            assert out["n_valid"] == len(idx) and 0 < len(idx) <= L.w * L.h
            assert k == 0 or lvl > 1 or len(idx) < L.w * L.h   # the moved pose leaves part of the finer levels outside
            assert out["sum_r2"] == int((r.astype(np.int64) ** 2).sum())
            assert np.array_equal(out["A"].astype(np.float32).view(np.uint32), A_ref.view(np.uint32)), (lvl, k)
            assert np.array_equal((-out["jtr"]).astype(np.float32).view(np.uint32), b_ref.view(np.uint32)), (lvl, k)
            # and the dump-capable twin (scalar Jacobian form, signed zeros kept) agrees with it per pixel and in the sums
            full = ctx.residual_jacobian(0, 1, lvl, pose, dump=True)
            valid = np.zeros(L.w * L.h, np.uint8)
            valid[idx] = 1
            assert np.array_equal(full["valid"], valid)
            assert np.array_equal(full["r"][idx], r)
            assert np.array_equal(full["J"][idx].view(np.uint32), J.view(np.uint32))
            assert np.array_equal(full["A"].astype(np.float32), A_ref) and full["n_valid"] == out["n_valid"]
    ctx.close()


def test_config3_1280x960_5_levels_256_resident_pairs_with_depth(capi, O, synth):
    """BASELINE config 3: synthetic 1280x960 random-texture pairs, 5 pyramid levels, batch = 256 resident."""
    w, h, n, distinct = 1280, 960, 256, 4
    over = dict(n_levels=5, first_level=4, last_level=0, max_iters=10, early_exit=0, has_depth=1)
    ctx = capi.Context(capi.default_params(w, h, *BIG, max_frames=2 * n, max_pairs=n, **over))
    pairs = [synth.render_pair(w, h, *BIG, seed=3000 + s, z=0.85 + 0.1 * s, with_depth=True)[:3] for s in range(distinct)]
    for i in range(n):     # slot 2i = reference, 2i + 1 = target; uploaded pair by pair (the host never holds the batch)
        ref, tgt, dep = pairs[i % distinct]
        ctx.upload_frames(2 * i, np.stack([ref, tgt]), np.stack([dep, dep]))
    ref_s = np.arange(n, dtype=np.int32) * 2
    poses, stats = _track_batch(ctx, 2 * n, ref_s, ref_s + 1)
    assert all(s["status"] == 0 and s["iterations"] == 50 for s in stats)
    for i in range(distinct, n):   # every tiled copy bit-identical to its original
        assert np.array_equal(poses[i].view(np.uint32), poses[i % distinct].view(np.uint32)), i
    po = O.default_params(w, h, *BIG, **over)
    for s in range(distinct):
        st, pose_cpu, _ = O.align_pair(po, pairs[s][0], pairs[s][1], pairs[s][2])
        assert st == 0
        assert np.array_equal(poses[s].view(np.uint32), pose_cpu.view(np.uint32)), (s, poses[s], pose_cpu)
    ctx.close()


@pytest.mark.parametrize("mode", ["tukey", "huber", "bilinear_huber"])
def test_robust_weights_at_640x480_with_depth(capi, O, synth, mode):
    """BASELINE config 5's shape (640x480 + robust weighting), synthetic stand-in for the TUM sequence."""
    w, h = 640, 480
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0, has_depth=1)
    over.update(dict(tukey=dict(weights=1), huber=dict(weights=2), bilinear_huber=dict(sampler=1, weights=2))[mode])
    n = 3
    ctx = capi.Context(capi.default_params(w, h, *TUM, max_frames=2 * n, max_pairs=n, **over))
    pairs = [synth.render_pair(w, h, *TUM, seed=4100 + s, z=0.9 + 0.1 * s, with_depth=True)[:3] for s in range(n)]
    for i, (ref, tgt, dep) in enumerate(pairs):
        ctx.upload_frames(2 * i, np.stack([ref, tgt]), np.stack([dep, dep]))
    ref_s = np.arange(n, dtype=np.int32) * 2
    poses, stats = _track_batch(ctx, 2 * n, ref_s, ref_s + 1)
    po = O.default_params(w, h, *TUM, **over)
    for s in range(n):
        st, pose_cpu, _ = O.align_pair(po, *pairs[s])
        assert st == 0 and stats[s]["status"] == 0
        assert np.array_equal(poses[s].view(np.uint32), pose_cpu.view(np.uint32)), (mode, s, poses[s], pose_cpu)
    ctx.close()


def test_per_stage_weighted_entry_on_a_level_larger_than_the_record_budget_assumed(capi, O, synth):
    """1920x1088: the create-time slicing is coarser than the per-stage dump's 8192 pixels per record, so the dump has to
    follow it (it used to need 255 records where 146 were allocated)."""
    w, h = 1920, 1088
    intr = (1500.0, 1500.0, 959.5, 543.5)
    over = dict(n_levels=1, first_level=0, last_level=0, max_iters=2, early_exit=0, weights=2)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over))
    ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=77)
    ctx.upload_frames(0, np.stack([ref, tgt]))
    ctx.build_pyramids(0, 2)
    ctx.apply_gradient(0, 2)
    p = O.default_params(w, h, *intr, **over)
    L = O.level_intrinsics(p, 0)
    gx, gy = O.scharr3(ref)
    pts = O.dense_points(None, w, h, 0)
    pose = O.se3_exp(np.array([0.004, -0.003, 0.002, 0.001, -0.002, 0.003], np.float32))
    wp = O.warp(pts, pose, L)
    J, r, idx = O.residual_jacobian_ex(ref, tgt, gx, gy, pts, wp, L, sampler=0)
    W = O.huber_weights(r)
    A_ref, b_ref = O.normal_equations(J, r, W, 50.0)
    out = ctx.residual_jacobian_weighted(0, 1, 0, pose)
    assert out["n_valid"] == len(idx)
    assert np.array_equal(out["A"].astype(np.float32), A_ref) and np.array_equal((-out["jtr"]).astype(np.float32), b_ref)
    ctx.close()


def test_speculative_launching_redoes_a_cut_short_alignment(capi, O, synth, monkeypatch):
    """One pair under the reference's early-exit schedule is launched without read-backs, a level's usual evaluations plus
    one; with the budget forced down to two evaluations per level nearly every level is cut short, the flag is raised and
    the careful second run must deliver the oracle's pose and iteration count — as the unforced call does."""
    w, h = 320, 240
    intr = (262.5, 262.5, 159.5, 119.5)
    ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=808, max_t=0.02, max_deg=1.0, with_depth=True)
    st, pose_cpu, tr = O.align_pair(O.default_params(w, h, *intr, has_depth=1), ref, tgt, dep, want_trace=True)
    assert st == 0 and len(tr) > 8          # more than two evaluations on some level
    out = []
    for forced in (False, True):
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, has_depth=1),
                           tuning=dict(first_poll=1) if forced else None)
        _upload_pair(ctx, ref, tgt, dep)
        poses, stats = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
        out.append((poses[0].copy(), stats[0]["iterations"]))
        ctx.close()
    for pose, iters in out:
        assert np.array_equal(pose.view(np.uint32), pose_cpu.view(np.uint32)) and iters == len(tr)


@pytest.mark.parametrize("mode", ["reference", "fixed"])
def test_two_pairs_with_slots_in_the_kernel_arguments(capi, O, synth, mode):
    """The synchronous call with one or two pairs hands the frame slots to k_iterate in its arguments and has the results
    written into page-locked host memory: two different pairs in scrambled slots (ref 3 / tgt 1, ref 0 / tgt 2), then the
    same context with the pairs swapped and with one pair only — every pose and count the oracle's."""
    w, h = 320, 240
    intr = (262.5, 262.5, 159.5, 119.5)
    over = dict(has_depth=1) if mode == "reference" else dict(has_depth=1, n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0)
    pairs = [synth.render_pair(w, h, *intr, seed=4100 + i, max_t=0.015, max_deg=0.8, with_depth=True)[:3] for i in range(2)]
    want = [O.align_pair(O.default_params(w, h, *intr, **over), r, t, d, want_trace=True) for r, t, d in pairs]
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=4, max_pairs=2, **over))
    slots = [(3, 1), (0, 2)]
    for (r, t, d), (rs, ts) in zip(pairs, slots):
        ctx.upload_frames(rs, r[None], d[None])
        ctx.upload_frames(ts, t[None], d[None])
    ctx.build_pyramids(0, 4)
    ctx.apply_gradient(0, 4)
    for order in ([0, 1], [1, 0], [1], [0]):
        poses, stats = ctx.estimate_pose_batch([slots[i][0] for i in order], [slots[i][1] for i in order], raise_on_pair_failure=True)
        for k, i in enumerate(order):
            st, pose_cpu, tr = want[i]
            assert st == 0
            assert np.array_equal(poses[k].view(np.uint32), pose_cpu.view(np.uint32)) and stats[k]["iterations"] == len(tr)
    ctx.close()


def test_deferred_stage_calls_wait_once_per_frame(capi, O, synth):
    """uwt_set_deferred(1): uwt_build_pyramids / uwt_apply_gradient only enqueue; a plane read right behind them and the
    alignment that follows deliver what the synchronous calls deliver — a sliding sequence of frames through two slots."""
    w, h = 320, 240
    intr = (262.5, 262.5, 159.5, 119.5)
    seq = [synth.render_pair(w, h, *intr, seed=7700 + i, max_t=0.01, max_deg=0.5, with_depth=True)[:3] for i in range(3)]
    frames = [seq[0][0], seq[0][1], seq[1][1], seq[2][1]]       # unrelated textures after the first pair: poses still defined
    dep = seq[0][2]
    po = O.default_params(w, h, *intr, has_depth=1)
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, has_depth=1))
    ctx.set_deferred(True)
    ctx.upload_frames(0, frames[0][None], dep[None])
    ctx.build_pyramids(0, 1)
    for i in range(1, len(frames)):
        cur, prev = i % 2, (i + 1) % 2
        ctx.upload_frames(cur, frames[i][None], dep[None])
        ctx.build_pyramids(cur, 1)
        ctx.apply_gradient(prev, 1)
        if i == 1:   # read straight behind the deferred calls
            gx, gy = O.scharr3(O.halve_u8(frames[0]))
            assert np.array_equal(ctx.get_plane(prev, 1, capi.PLANE_GRADX), gx)
            assert np.array_equal(ctx.get_plane(cur, 2, capi.PLANE_IMAGE), O.halve_u8(O.halve_u8(frames[1])))
        poses, stats = ctx.estimate_pose_batch([prev], [cur])
        st, pose_cpu, tr = O.align_pair(po, frames[i - 1], frames[i], dep, want_trace=True)
        assert stats[0]["status"] == st
        if st == 0:
            assert np.array_equal(poses[0].view(np.uint32), pose_cpu.view(np.uint32)) and stats[0]["iterations"] == len(tr)
    ctx.close()
    # a deferred call still in flight holds its slots: an asynchronous upload into them waits for it
    big = capi.Context(capi.default_params(1280, 960, 1050.0, 1050.0, 639.5, 479.5, max_frames=1, max_pairs=1, n_levels=5,
                                           first_level=4, last_level=0))
    big.set_deferred(True)
    old = synth.texture(1280, 960, seed=1)
    new = capi.pinned_empty((1, 960, 1280), np.uint8)
    new[0] = synth.texture(1280, 960, seed=2)
    for _ in range(3):
        big.upload_frames(0, old[None])
        big.build_pyramids(0, 1)
        big.apply_gradient(0, 1)                 # enqueued only
        big.upload_frames_async(0, new)          # must land behind the gradient kernels that read the old level 0
        big.sync()
        gx, _ = O.scharr3(old)
        assert np.array_equal(big.get_plane(0, 0, capi.PLANE_GRADX), gx)
        assert np.array_equal(big.get_plane(0, 0, capi.PLANE_IMAGE), new[0])
    big.close()


@pytest.mark.parametrize("mode", ["identity", "huber"])
def test_split_batch_on_two_streams_gives_the_same_poses(capi, O, synth, monkeypatch, mode):
    """Fixed-schedule batches of 16 pairs or more run as two parts on two streams (uwt_tuning::split = 1: one stream).  21 pairs (an
    odd count: parts of 10 and 11), pyramids and gradients through uwt_track_batch_async: same poses bit for bit either way,
    and the oracle's on the pairs checked."""
    w, h, n, distinct = 320, 240, 21, 5
    intr = (262.5, 262.5, 159.5, 119.5)
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0, has_depth=1)
    if mode == "huber":
        over["weights"] = 2
    pairs = [synth.render_pair(w, h, *intr, seed=6100 + s, max_t=0.012, max_deg=0.6, with_depth=True)[:3] for s in range(distinct)]
    results = []
    for split in ("2", "1"):      # (split_min_px: the test's batch is smaller than the size from which the split pays)
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over), tuning=dict(split_min_px=1, split=int(split)))
        for i in range(n):
            ref, tgt, dep = pairs[i % distinct]
            ctx.upload_frames(2 * i, np.stack([ref, tgt]), np.stack([dep, dep]))
        ref_s = np.arange(n, dtype=np.int32) * 2
        poses, stats = _track_batch(ctx, 2 * n, ref_s, ref_s + 1)
        assert all(s["status"] == 0 and s["iterations"] == 24 for s in stats)
        results.append(poses.copy())
        ctx.close()
    assert np.array_equal(results[0].view(np.uint32), results[1].view(np.uint32))
    po = O.default_params(w, h, *intr, **over)
    for i in (0, 9, 10, 11, 20):       # both parts and their seam
        st, pose_cpu, _ = O.align_pair(po, *pairs[i % distinct])
        assert st == 0 and np.array_equal(results[0][i].view(np.uint32), pose_cpu.view(np.uint32)), i


def test_split_batch_small_images_parts_at_different_levels(capi, synth, monkeypatch):
    """Small images make the two parts of a split batch drift apart by whole levels (launches of a few microseconds): the
    second part's records must not land where the first part's live (they are placed independently of the level).  160x96,
    80 pairs, split forced: poses bit-identical to the one-stream run, statuses clean — repeated, as the drift varies."""
    import torch
    w, h, n = 160, 96, 80
    f = 525.0 * w / 640.0
    intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
    over = dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0)
    frames = []
    for s in range(n):
        ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=3000 + s)
        frames += [ref, tgt]
    frames = np.stack(frames)
    ref_s = np.arange(n, dtype=np.int32) * 2
    out = {}
    for split in ("1", "2"):
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over), tuning=dict(split_min_px=1, split=int(split)))
        ctx.upload_frames(0, frames)
        runs = []
        for rep in range(1 if split == "1" else 6):
            poses, stats = _track_batch(ctx, 2 * n, ref_s, ref_s + 1)
            assert all(s["status"] == 0 and s["iterations"] == 20 for s in stats), (split, rep)
            runs.append(poses.copy())
        out[split] = runs
        ctx.close()
    for rep, poses in enumerate(out["2"]):
        assert np.array_equal(poses.view(np.uint32), out["1"][0].view(np.uint32)), rep


@pytest.mark.parametrize("mode", ["reference", "fixed"])
def test_coarse_levels_in_one_launch_match_the_per_launch_form(capi, O, synth, monkeypatch, mode):
    """The coarsest levels of a lone pair (those one block evaluates) run to their end in one launch, k_coarse — exit test
    and hand-offs on the device; uwt_tuning::coarse = 0 keeps a launch per evaluation.  Same poses and iteration counts, the
    oracle's; 5 levels of 320x240: three coarse levels, then k_iterate; a pair of unrelated frames included (many
    evaluations per level, the level limit of 50 in reach)."""
    w, h = 320, 240
    intr = (262.5, 262.5, 159.5, 119.5)
    over = dict(has_depth=1) if mode == "reference" else dict(has_depth=1, n_levels=5, first_level=4, last_level=0, max_iters=7, early_exit=0)
    pairs = [synth.render_pair(w, h, *intr, seed=8800 + i, max_t=0.02, max_deg=1.0, with_depth=True)[:3] for i in range(3)]
    pairs.append((pairs[0][0], pairs[1][1], pairs[0][2]))      # unrelated frames
    po = O.default_params(w, h, *intr, **over)
    want = [O.align_pair(po, r, t, d, want_trace=True) for r, t, d in pairs]
    for no_coarse in (False, True):
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, **over), tuning=dict(coarse=int(not no_coarse)))
        for (r, t, d), (st, pose_cpu, tr) in zip(pairs, want):
            _upload_pair(ctx, r, t, d)
            poses, stats = ctx.estimate_pose_batch([0], [1])
            assert stats[0]["status"] == st and stats[0]["iterations"] == len(tr), (no_coarse, stats[0], len(tr))
            if st == 0:
                assert np.array_equal(poses[0].view(np.uint32), pose_cpu.view(np.uint32)), no_coarse
        ctx.close()


# ------------------------------------------------------------------ launch paths of bench.py

def _run(cmd, extra_env=None, timeout=540):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


SMALL_BENCH = ["--width", "160", "--height", "96", "--steps", "2", "--warmup", "1", "--unique", "4", "--cpu-pairs", "0"]


def test_bench_refuses_more_gpus_than_visible():
    import torch
    have = torch.cuda.device_count()
    r = _run([sys.executable, "bench.py", "--gpus", str(have + 1), "--pairs", "8"] + SMALL_BENCH)
    assert r.returncode != 0 and not r.stdout.strip()
    assert b"GPU(s) visible" in r.stderr


def test_bench_strong_scaling_mode_single_gpu():
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--total-pairs", "24"] + SMALL_BENCH)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["config"]["total_pairs"] == 24
    # (at 160x96 a launch lasts a few microseconds: the ratio of two such timings scatters widely — 0.9 … 1.5 seen)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["limited_by"] == "valu" and 0 < d["roofline"]["valu"]["valu_issue_frac"] <= 2.0
    assert set(d["arith_sets"]) >= {"opencv", "legacy"} and d["arith_sets"]["legacy"]["value"] > 0
    assert d["roofline"]["valu"]["shader_clock_GHz"] > 0.5


def test_bench_rccl_world_of_one_gathers_on_the_context_stream():
    """Under a launcher (RANK set) the RCCL path runs even with one rank: nccl process group, all_gather and the
    permutation to global order enqueued on the context's stream, the gathered tensor checked on rank 0."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", "29611", "bench.py", "--gpus", "1", "--pairs", "12"] + SMALL_BENCH)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["sharding"]
    # the collective itself ran (a world of one no longer returns early) and rank 0 compared its block bit for bit
    assert d["config"]["gather"] == {"collective_ran": True, "world": 1, "rank0_block_bitwise_equal_to_its_own_poses": True}


def test_native_bench_rccl_allgather_on_the_one_device():
    """tools/uwt_bench --gpus 1 --rccl: the native multi-GPU mode's ncclCommInitAll + ncclAllGather on the context's stream,
    the gathered block compared with the device's own poses on the host."""
    exe = os.path.join(ROOT, "tools", "uwt_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    r = _run([exe, "--pairs", "20", "--unique", "5", "--width", "160", "--height", "96", "--levels", "3", "--steps", "3", "--warmup", "1",
              "--gpus", "1", "--rccl"], timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["gathered_blocks_match"] is True and d["tiled_pairs_identical"] and d["poses_finite"] and d["n_gpus"] == 1


def test_pose_gatherer_nccl_world_one_in_process_stream_order():
    """PoseGatherer with the nccl backend (world 1) on a side stream behind a kernel that writes the poses: with a process
    group the all_gather_into_tensor and the index_select really run (RCCL kernels on the caller's stream) — both the
    direct form and the one staged through the padded send block that ranks with uneven shards use — and the gathered
    tensor is the local one bit for bit, in a buffer of the gatherer's own."""
    code = r'''
import os, importlib, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29612", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
d = importlib.import_module("uw-slam_amd.dist")
calls = []
real = dist.all_gather_into_tensor
def spy(*a, **k):
    calls.append(1)
    return real(*a, **k)
dist.all_gather_into_tensor = spy
for stage in (False, True):
    g = d.PoseGatherer(37, dev, stage_send=stage)
    assert g.collective
    s = torch.cuda.Stream()
    for rep in range(3):
        with torch.cuda.stream(s):
            local = (torch.arange(37 * 7, dtype=torch.float32, device=dev).reshape(37, 7) * 2.0 + rep) / 3.0
            out = g.gather(local)
        s.synchronize()
        assert out.data_ptr() == g.out.data_ptr() != local.data_ptr()
        assert torch.equal(out.view(torch.int32), local.view(torch.int32)), (stage, rep)
assert len(calls) == 6, calls
dist.destroy_process_group()
print("ok")
'''
    r = _run([sys.executable, "-c", code])
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-2000:]



@pytest.mark.gpu
def test_gather_pipeline_consumer_on_its_own_stream_wait_and_release():
    """GatherPipeline.wait() / release() (uw-slam_amd/dist.py): a consumer that reads a step's gathered poses on a stream of its own
    — held up there by a long-running kernel — while two more steps are enqueued.  The step that reuses the gatherer's buffer (the
    next but one) must not overwrite it before the consumer's queued read has run: wait() orders the consumer behind the exchange,
    release() orders the buffer's next writer behind the consumer."""
    code = r'''
import os, importlib, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
d = importlib.import_module("uw-slam_amd.dist")
n = 64
work = torch.cuda.Stream()                      # stands for the context's stream
pipe = d.GatherPipeline(n, n, dev, work.cuda_stream)
consumer = torch.cuda.Stream()
def align_of(k):
    def align(buf):
        with torch.cuda.stream(work):
            buf.copy_(torch.full((n, 7), float(k), device=dev))
    return align
for trial in range(5):
    base = 10 * trial
    g = pipe.step(align_of(base))
    pipe.wait(consumer)
    with torch.cuda.stream(consumer):
        torch.cuda._sleep(400_000_000)          # ~0.2 s: the read below is still queued when the next two steps are enqueued
        seen = g.clone()
    pipe.release(consumer)
    pipe.step(align_of(base + 1))
    pipe.step(align_of(base + 2))               # reuses the gatherer buffer `g` lives in
    torch.cuda.synchronize()
    assert bool((seen == float(base)).all()), (trial, seen[0])
    assert bool((pipe.gathered == float(base + 2)).all())
dist.destroy_process_group()
print("ok")
'''
    r = _run([sys.executable, "-c", code])
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-2000:]


@pytest.mark.gpu
def test_update_launch_form_matches_the_tail_update():
    """Where a batch runs as two parts on two streams the Gauss-Newton update runs in the tail of the evaluation's own launch
    (tail_update_wave: the pair's last block folds the records and solves); elsewhere a k_gn_update launch follows every
    evaluation.  uwt_tuning::tail_update = 0 / 2 selects the launch form / the tail form everywhere.  Both forms
    add the records in the same order: a batch with several blocks per pair on every level — fixed schedule and the
    reference's early-exit schedule (whose polls count the pairs still iterating through the same code), identity and Huber
    weights — gives the oracle's poses bit for bit in either form.  (Child processes: a stalled ticket wait would otherwise
    take the whole run with it.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import CHILD_ARITH_HEADER
    code = (r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
''' % root) + CHILD_ARITH_HEADER + r'''
capi = importlib.import_module("uw-slam_amd.capi"); synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
import os
tuning = dict(tail_update=int(os.environ["TEST_TAIL_UPDATE"]))
if os.environ.get("TEST_FEW_LARGE"):
    tuning.update(chained=0)
if os.environ.get("TEST_FEW_LARGE"):   # two pairs of 640x480 off the chained flow: 150 records per pair at level 0 (five rounds of the tail's fold)
    w, h, intr, n = 640, 480, (525.0, 525.0, 319.5, 239.5), 2
else:
    w, h, intr, n = 320, 240, (262.5, 262.5, 159.5, 119.5), 24
pairs = [synth.render_pair(w, h, *intr, seed=900 + i, z=0.8 + 0.02 * i, with_depth=True) for i in range(n)]
frames = np.stack([f for p in pairs for f in (p[0], p[1])]); depth = np.stack([p[2] for p in pairs for _ in (0, 1)])
out = []
for over in (dict(n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0, has_depth=1),
             dict(n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0, has_depth=1, weights=2),
             dict(n_levels=5, first_level=4, last_level=1, max_iters=50, early_exit=1, has_depth=1)):
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over), tuning=tuning)
    ctx.upload_frames(0, frames, depth); ctx.build_pyramids(0, 2 * n); ctx.apply_gradient(0, 2 * n)
    for rep in range(3):
        poses, stats = ctx.estimate_pose_batch(np.arange(n) * 2, np.arange(n) * 2 + 1, raise_on_pair_failure=True)
        po = O.default_params(w, h, *intr, **over)
        for i, p in enumerate(pairs if rep == 0 else pairs[:4]):
            st, pose_cpu, tr = O.align_pair(po, p[0], p[1], p[2], want_trace=True)
            assert st == 0 and np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), (over, i, poses[i], pose_cpu)
            assert stats[i]["iterations"] == len(tr), (over, i)
    ctx.close()
print("ok")
'''
    for switch, extra in (("0", {}), ("2", {}), ("2", dict(TEST_FEW_LARGE="1"))):
        env = dict(os.environ, TEST_TAIL_UPDATE=switch, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=540)
        assert r.returncode == 0 and b"ok" in r.stdout, (switch, extra, r.stderr.decode()[-2000:])


@pytest.mark.gpu
def test_two_contexts_driven_from_two_host_threads_concurrently(capi, O, synth):
    """include/uwt.h: "one ctx per host thread per GPU".  Two contexts (different sizes and schedules), each driven by a host
    thread of its own at the same time — batches through uwt_track_batch_async, one pair per call, per-stage calls — give what
    each gives alone, bit for bit, and the oracle's poses."""
    import threading
    cfgs = [dict(w=320, h=240, intr=(262.5, 262.5, 159.5, 119.5), n=24, seed=9100,
                 over=dict(n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0, has_depth=1)),
            dict(w=160, h=96, intr=(131.25, 130.5, 79.5, 47.5), n=9, seed=9200,
                 over=dict(has_depth=0, weights=2))]                      # the reference's early-exit schedule, Huber weights
    work = []
    for c in cfgs:
        pairs = [synth.render_pair(c["w"], c["h"], *c["intr"], seed=c["seed"] + s, with_depth=bool(c["over"]["has_depth"]))[:3] for s in range(3)]
        po = O.default_params(c["w"], c["h"], *c["intr"], **c["over"])
        want = [O.align_pair(po, r, t, d if c["over"]["has_depth"] else None)[1] for r, t, d in pairs]
        ctx = capi.Context(capi.default_params(c["w"], c["h"], *c["intr"], max_frames=2 * c["n"], max_pairs=c["n"], **c["over"]))
        frames = np.stack([f for i in range(c["n"]) for f in pairs[i % 3][:2]])
        depth = np.stack([pairs[i % 3][2] for i in range(c["n"]) for _ in (0, 1)]) if c["over"]["has_depth"] else None
        ctx.upload_frames(0, frames, depth)
        work.append((c, ctx, want))

    def drive(c, ctx, want, out, reps):
        try:
            n = c["n"]
            ref = np.arange(n, dtype=np.int32) * 2
            for rep in range(reps):
                ctx.build_pyramids(0, 2 * n)
                ctx.apply_gradient(0, 2 * n)
                batch, _ = ctx.estimate_pose_batch(ref, ref + 1, raise_on_pair_failure=True)
                single, _ = ctx.estimate_pose_batch([2], [3], raise_on_pair_failure=True)
                gx = ctx.get_plane(0, 1, capi.PLANE_GRADX)
                out.append((batch.copy(), single[0].copy(), gx.copy()))
        except Exception as e:      # a failure in a thread must fail the test, not vanish
            out.append(e)

    serial = [[] for _ in work]
    for (c, ctx, want), out in zip(work, serial):
        drive(c, ctx, want, out, 1)
    together = [[] for _ in work]
    threads = [threading.Thread(target=drive, args=(c, ctx, want, out, 6)) for (c, ctx, want), out in zip(work, together)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for (c, ctx, want), s_out, t_out in zip(work, serial, together):
        assert len(t_out) == 6 and not any(isinstance(r, Exception) for r in s_out + t_out), [r for r in s_out + t_out if isinstance(r, Exception)]
        b0, s0, g0 = s_out[0]
        for i in range(c["n"]):
            assert np.array_equal(b0[i].view(np.uint32), want[i % 3].view(np.uint32)), i
        for b, s1, g in t_out:
            assert np.array_equal(b.view(np.uint32), b0.view(np.uint32)) and np.array_equal(s1.view(np.uint32), s0.view(np.uint32))
            assert np.array_equal(g, g0)
        ctx.close()


@pytest.mark.gpu
def test_every_tuning_switch_changes_no_bit(capi, O, synth):
    """uwt_tuning chooses HOW launches are laid out, never WHAT is computed: a batch and a lone pair under the defaults and with
    every switch flipped one at a time — parts, tail update, slicing target, coarse forms, gradient overlap, first look,
    chained flow, speculation, fused stages, batch pyramids, typed loads, streamed planes — through the whole per-frame path
    (pyramids, gradients, alignment): the same poses bit for bit, the oracle's.  Fixed and early-exit schedules."""
    import torch
    w, h, n = 320, 240, 20
    intr = (262.5, 262.5, 159.5, 119.5)
    settings = [dict(), dict(split=1), dict(split=3, split_min=2), dict(split_min_px=1), dict(tail_update=0), dict(tail_update=2),
                dict(target_blocks=64), dict(target_blocks=5000), dict(coarse=0), dict(coarse_batch_px=0), dict(coarse_batch_px=30000),
                dict(overlap_gradients=0), dict(first_poll=1), dict(first_poll=7), dict(chained=0), dict(chained=1), dict(speculation=0),
                dict(fused_stages=0), dict(pyramid_batch=0), dict(typed_loads=0), dict(stream_bytes=0), dict(stream_bytes=0, typed_loads=0),
                dict(split_min_px=1, typed_loads=0, tail_update=2)]
    for over in (dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0, has_depth=1), dict(has_depth=1)):
        pairs = [synth.render_pair(w, h, *intr, seed=9700 + s, with_depth=True, max_t=0.012, max_deg=0.6)[:3] for s in range(4)]
        po = O.default_params(w, h, *intr, **over)
        want = [O.align_pair(po, *p)[1] for p in pairs]
        frames = np.stack([f for i in range(n) for f in pairs[i % 4][:2]])
        depth = np.stack([pairs[i % 4][2] for i in range(n) for _ in (0, 1)])
        ref = np.arange(n, dtype=np.int32) * 2
        first = None
        for t in settings:
            ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over), tuning=t or None)
            ctx.upload_frames(0, frames, depth)
            buf = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
            ctx.track_batch_async(0, 2 * n, ref, ref + 1, buf.data_ptr())
            ctx.sync()
            batch = buf.cpu().numpy()
            lone, _ = ctx.estimate_pose_batch([2], [3], raise_on_pair_failure=True)
            few, _ = ctx.estimate_pose_batch(ref[:5], ref[:5] + 1, raise_on_pair_failure=True)
            ctx.close()
            if first is None:
                first = batch
                for i in range(n):
                    assert np.array_equal(batch[i].view(np.uint32), want[i % 4].view(np.uint32)), (over, i)
            assert np.array_equal(batch.view(np.uint32), first.view(np.uint32)), (over, t)
            assert np.array_equal(lone[0].view(np.uint32), first[1].view(np.uint32)), (over, t)
            assert np.array_equal(few.view(np.uint32), first[:5].view(np.uint32)), (over, t)


@pytest.mark.gpu
def test_bad_arguments_return_a_status_and_touch_nothing(capi, synth):
    """Slot ranges, levels, counts and pointers out of range come back as UWT_ERR_INVALID_ARG / UWT_ERR_CAPACITY (a range whose
    end overflows 32 bits included) and the context keeps working."""
    import ctypes as C
    w, h, intr = 64, 48, (64.0, 64.0, 31.5, 23.5)
    ctx = capi.Context(capi.default_params(w, h, *intr, n_levels=3, first_level=2, last_level=0, max_frames=4, max_pairs=2))
    L, H = capi.lib(), ctx._h
    ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=3)
    ctx.upload_frames(0, np.stack([ref, tgt])); ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
    good, _ = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
    big = 2 ** 31 - 1
    gray = np.zeros((h, w), np.uint8)
    gp = gray.ctypes.data_as(C.POINTER(C.c_uint8))
    one = (C.c_int32 * 1)(0); far = (C.c_int32 * 1)(7); neg = (C.c_int32 * 1)(-1)
    pose = np.zeros(7, np.float32); pp = pose.ctypes.data_as(C.POINTER(C.c_float))
    st3 = (capi.Stats * 3)()
    for rc in (L.uwt_build_pyramids(H, big, 1), L.uwt_build_pyramids(H, big, big), L.uwt_build_pyramids(H, 3, 2), L.uwt_build_pyramids(H, -1, 1),
               L.uwt_apply_gradient(H, 0, -1), L.uwt_apply_gradient(H, big - 1, 2),
               L.uwt_upload_frames(H, 4, 1, gp, None), L.uwt_upload_frames(H, big, 1, gp, None), L.uwt_upload_frames(H, 0, 1, None, None),
               L.uwt_set_frame(H, 4, gp, C.c_size_t(w), None, C.c_size_t(0)), L.uwt_set_frame(H, 0, gp, C.c_size_t(w - 1), None, C.c_size_t(0)),
               L.uwt_estimate_pose_batch(H, 1, one, far, pp, st3), L.uwt_estimate_pose_batch(H, 1, neg, one, pp, st3),
               L.uwt_estimate_pose_batch(H, 0, one, one, pp, st3), L.uwt_estimate_pose_batch(H, 1, None, one, pp, st3)):
        assert rc == capi.ERR_INVALID_ARG
    three = (C.c_int32 * 3)(0, 0, 0)
    assert L.uwt_estimate_pose_batch(H, 3, three, three, pp, st3) == capi.ERR_CAPACITY
    lv = capi.Level()
    assert L.uwt_level_info(H, 3, C.byref(lv)) == capi.ERR_INVALID_ARG and L.uwt_level_info(H, -1, C.byref(lv)) == capi.ERR_INVALID_ARG
    out = C.c_void_p()
    assert L.uwt_plane_device_ptr(H, 0, 0, 9, C.byref(out)) == capi.ERR_INVALID_ARG
    assert L.uwt_plane_device_ptr(H, 0, 0, capi.PLANE_DEPTH, C.byref(out)) == capi.ERR_INVALID_ARG     # no depth plane in this context
    again, _ = ctx.estimate_pose_batch([0], [1], raise_on_pair_failure=True)
    assert np.array_equal(good, again)
    ctx.close()
