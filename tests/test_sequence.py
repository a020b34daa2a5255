"""Sequential tracking of a (synthetic) recorded sequence through the batched path: directory order, PNG reading,
chunking across the context capacity, trajectory hand-off."""
import importlib
import os

import numpy as np
import pytest


def _write_sequence(tmp, synth, n, w, h):
    from PIL import Image
    img_dir, dep_dir = tmp / "rgb", tmp / "depth"
    img_dir.mkdir(); dep_dir.mkdir()
    base = synth.texture(w + 64, h + 32, seed=3)
    frames, depths = [], []
    for i in range(n):
        f = np.ascontiguousarray(base[8 + i // 3: 8 + i // 3 + h, 2 * i: 2 * i + w])       # a slow pan
        d = np.full((h, w), 5000 + 10 * i, np.uint16)
        Image.fromarray(f).save(img_dir / ("%06d.png" % (n - i)))                          # names sort in reverse creation order
        Image.fromarray(d).save(dep_dir / ("%06d.png" % (n - i)))
        frames.append(f); depths.append(d)
    return str(img_dir), str(dep_dir), frames[::-1], depths[::-1]


def test_listing_reading_cropping(tmp_path, synth):
    S = importlib.import_module("uw-slam_amd.sequence")
    img_dir, dep_dir, frames, depths = _write_sequence(tmp_path, synth, 5, 96, 64)
    names = S.list_sorted(img_dir)
    assert [os.path.basename(n) for n in names] == ["%06d.png" % i for i in range(1, 6)]
    assert np.array_equal(S.load_gray(names[0]), frames[0])
    assert np.array_equal(S.load_depth(S.list_sorted(dep_dir)[2]), depths[2]) and S.load_depth(S.list_sorted(dep_dir)[2]).dtype == np.uint16
    c, x0, y0 = S.centre_crop(frames[0], 64, 48)
    assert c.shape == (48, 64) and (x0, y0) == (16, 8) and np.array_equal(c, frames[0][8:56, 16:80])
    with pytest.raises(ValueError):
        S.centre_crop(frames[0], 200, 10)
    # evaluation helpers: a rigidly moved copy has zero ATE
    rng = np.random.default_rng(0)
    P = rng.normal(size=(30, 3))
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    R *= np.sign(np.linalg.det(R))
    assert S.ate_rmse(P, (R @ P.T).T + [1, 2, 3]) < 1e-9
    assert S.rpe_translation(P, P + [0.0, 3.0, 4.0]) == pytest.approx(5.0)


@pytest.mark.gpu
def test_sequence_tracking_equals_pairwise_and_oracle(tmp_path, synth, O):
    S = importlib.import_module("uw-slam_amd.sequence")
    w, h, n = 160, 96, 11
    img_dir, dep_dir, _, _ = _write_sequence(tmp_path, synth, n, w, h)
    frames = [S.load_gray(p) for p in S.list_sorted(img_dir)]
    depths = [S.load_depth(p) for p in S.list_sorted(dep_dir)]
    intr = (131.25, 131.25, 79.5, 47.5)
    trk = S.SequenceTracker(w, h, *intr, depth=True, chunk=4)        # 10 pairs in chunks of 4, 4, 2
    poses, stats = trk.track(frames, depths)
    assert poses.shape == (n - 1, 7) and all(s["status"] == 0 for s in stats)
    p = O.default_params(w, h, *intr, has_depth=1)
    for i in (0, 3, 4, 9):                                            # chunk interiors and boundaries
        st, pose_cpu, _ = O.align_pair(p, frames[i], frames[i + 1], depths[i])
        assert st == 0 and np.array_equal(poses[i], pose_cpu)
    traj = trk.trajectory(poses)
    assert np.array_equal(traj, O.accumulate_trajectory(poses))
    ref = trk.trajectory(poses, reference_visualiser=True)
    assert np.array_equal(ref, O.accumulate_trajectory(poses, t_scale=40.0, reference_axes=True))
    T = importlib.import_module("uw-slam_amd.trajectory")
    T.write_reference_csv(tmp_path / "out.csv", ref)
    est, _ = T.read_reference_csv(tmp_path / "out.csv")
    assert np.allclose(est, ref)
    trk.close()


@pytest.mark.gpu
def test_streaming_many_chunks_every_pair_against_the_oracle(synth, O):
    """More chunks than the context's dependency rings hold (8): slot ranges and staging blocks are reused many times while
    copies and alignments overlap; every pose must still be the oracle's."""
    S = importlib.import_module("uw-slam_amd.sequence")
    w, h, n = 160, 96, 90
    intr = (131.25, 131.25, 79.5, 47.5)
    base = synth.texture(w + 2 * n + 8, h + n // 2 + 8, seed=11)
    frames = [np.ascontiguousarray(base[4 + i // 2: 4 + i // 2 + h, 2 * i: 2 * i + w]) for i in range(n)]
    depths = [np.full((h, w), 4000 + 25 * i, np.uint16) for i in range(n)]
    for d in depths:
        d[::7, ::5] = 0                                             # invalid-depth holes
    trk = S.SequenceTracker(w, h, *intr, depth=True, chunk=4)        # 89 pairs: 22 chunks of 4 and one of 1
    poses, stats = trk.track(frames, depths)
    again, _ = trk.track(frames, depths)                             # the same context, ranges dirty from the first pass
    assert np.array_equal(poses.view(np.uint32), again.view(np.uint32))
    p = O.default_params(w, h, *intr, has_depth=1)
    for i in range(n - 1):
        st, pose_cpu, _ = O.align_pair(p, frames[i], frames[i + 1], depths[i])
        assert st == stats[i]["status"]
        if st == 0:
            assert np.array_equal(poses[i].view(np.uint32), pose_cpu.view(np.uint32)), i
    trk.close()
