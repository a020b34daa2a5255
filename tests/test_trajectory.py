"""trajectory accumulation (Visualizer::UpdateMessages) and its file formats."""
import importlib

ARITH_INDEPENDENT = True   # nothing here depends on the arithmetic set (tests/conftest.py): run once

import numpy as np
import pytest


def _random_poses(O, n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([O.se3_exp((rng.normal(0, 1, 6) * [0.01, 0.01, 0.01, 0.02, 0.02, 0.02]).astype(np.float32))
                     for _ in range(n)])


def test_oracle_trajectory_is_the_matrix_prefix_product(O):
    poses = _random_poses(O, 40, 1)
    traj = O.accumulate_trajectory(poses)
    M = np.eye(4)
    for i in range(40):
        M = M @ O.se3_matrix(poses[i]).astype(np.float64)
        assert np.allclose(O.se3_matrix(traj[i]), M, atol=2e-5)
    # reference visualiser mode: translation x40 before composing, published position (-z, -x, -y) (Visualizer.cpp:307-320)
    ref = O.accumulate_trajectory(poses, t_scale=40.0, reference_axes=True)
    M = np.eye(4)
    for i in range(40):
        T = O.se3_matrix(poses[i]).astype(np.float64)
        T[:3, 3] *= 40
        M = M @ T
        assert np.allclose(ref[i, 4:], [-M[2, 3], -M[0, 3], -M[1, 3]], atol=1e-3)
    start = O.se3_exp(np.array([1, 2, 3, 0.1, 0.2, 0.3], np.float32))
    one = O.accumulate_trajectory(poses[:1], start=start)
    assert np.array_equal(one[0], O.se3_mul(start, poses[0]))


def test_csv_and_groundtruth_formats(tmp_path):
    T = importlib.import_module("uw-slam_amd.trajectory")
    rng = np.random.default_rng(0)
    est, gt = rng.normal(size=(5, 7)), rng.normal(size=(5, 7))
    p = tmp_path / "out.csv"
    T.write_reference_csv(p, est, gt)
    first = open(p).readline().strip().split(",")
    assert len(first) == 14 and float(first[0]) == est[0, 4] and float(first[6]) == est[0, 3] and float(first[7]) == gt[0, 4]
    e2, g2 = T.read_reference_csv(p)
    assert np.array_equal(e2, est) and np.array_equal(g2, gt)
    tum = tmp_path / "groundtruth.txt"
    tum.write_text("# ground truth trajectory\n# file: x\n# timestamp tx ty tz qx qy qz qw\n"
                   "1.5 1 2 3 0 0 0 1\n2.5 4 5 6 0.5 0.5 0.5 0.5\n")
    ts, poses = T.read_groundtruth_tum(tum)
    assert ts.tolist() == [1.5, 2.5] and poses[1].tolist() == [0.5, 0.5, 0.5, 0.5, 4, 5, 6]
    eu = tmp_path / "data.csv"
    eu.write_text("#timestamp,p_x,p_y,p_z,q_w,q_x,q_y,q_z,v_x\n100,1,2,3,1,0,0,0,9\n")
    ts, poses = T.read_groundtruth_euroc(eu)
    assert ts.tolist() == [100.0] and poses[0].tolist() == [0, 0, 0, 1, 1, 2, 3]
    out = tmp_path / "traj.txt"
    T.write_tum(out, [1.5, 2.5], poses.repeat(2, 0))
    assert open(out).readline().split()[1:4] == ["1", "2", "3"]
    assert T.ground_truth_indices(1000, 100, 2).tolist()[:2] == [20, 30]
    assert T.ground_truth_indices(1000, 100, 2, euroc=True)[0] == 620


@pytest.mark.gpu
def test_gpu_trajectory_bit_exact(O):
    capi = importlib.import_module("uw-slam_amd.capi")
    ctx = capi.Context(capi.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=3, first_level=2, last_level=0))
    poses = _random_poses(O, 300, 2)
    start = O.se3_exp(np.array([0.5, -0.2, 0.1, 0.3, 0.1, -0.2], np.float32))
    for kw in (dict(), dict(t_scale=40.0, reference_axes=True), dict(start=start, t_scale=2.0)):
        a = ctx.accumulate_trajectory(poses, **kw)
        b = O.accumulate_trajectory(poses, **kw)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert ctx.accumulate_trajectory(np.zeros((0, 7), np.float32)).shape == (0, 7)


@pytest.mark.gpu
def test_gpu_trajectory_scan_agrees_with_the_sequential_product():
    """uwt_accumulate_trajectory_scan: a prefix product regrouped — equal to the sequential accumulation to float rounding;
    one and two poses (no regrouping possible) bit for bit."""
    import importlib
    capi = importlib.import_module("uw-slam_amd.capi")
    ctx = capi.Context(capi.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=3, first_level=2, last_level=0))
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 1024, 1025, 5000):
        q = rng.normal(0, 0.02, (n, 4)).astype(np.float32); q[:, 3] = 1.0
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        poses = np.concatenate([q, rng.normal(0, 0.01, (n, 3)).astype(np.float32)], axis=1)
        start = np.array([0.1, -0.2, 0.05, 0.97, 1.0, 2.0, 3.0], np.float32)
        start[:4] /= np.linalg.norm(start[:4])
        seq = ctx.accumulate_trajectory(poses, start)
        par = ctx.accumulate_trajectory(poses, start, scan=True)
        if n <= 2:
            assert np.array_equal(seq, par)
        dq = np.minimum(np.abs(seq[:, :4] - par[:, :4]).max(), np.abs(seq[:, :4] + par[:, :4]).max())
        assert dq < 2e-5 and np.abs(seq[:, 4:] - par[:, 4:]).max() < 1e-4 * max(1.0, np.abs(seq[:, 4:]).max()), (n, dq)
        ref = ctx.accumulate_trajectory(poses, start, 40.0, True, scan=True)       # scale and axis permutation in scan form too
        ref_seq = ctx.accumulate_trajectory(poses, start, 40.0, True)
        assert np.abs(ref - ref_seq).max() < 1e-3 * max(1.0, np.abs(ref_seq).max())
    ctx.close()


def test_ground_truth_conventions_camera_to_world_against_tracker_pairs(synth):
    """A ground-truth FILE holds camera-to-world poses G_k; the tracker's pose of pair k maps the previous camera's coordinates
    to the current one's (X_{k+1} = T_k X_k, src/Tracker.cpp:595), i.e. T_k = G_{k+1}^-1 G_k — the INVERSE of G_k^-1 G_{k+1}.
    With the synthetic camera's physical poses written as ground truth: pair_ground_truth recovers the scene's true per-pair
    motion, camera_trajectory of the true pairs recovers G_0^-1 G_k, and the wrong (un-inverted) pairing is visibly wrong."""
    T = importlib.import_module("uw-slam_amd.trajectory")
    S = importlib.import_module("uw-slam_amd.sequence")
    _, _, rel, absp = synth.render_sequence(64, 48, 64.0, 64.0, 31.5, 23.5, n=12, seed=3, margin=(32, 32))
    c2w = synth.camera_to_world_poses(rel)
    assert np.allclose(c2w[0], [0, 0, 0, 1, 0, 0, 0])
    gt_pairs = T.pair_ground_truth(c2w)
    assert np.allclose(gt_pairs, rel, atol=1e-9)                       # the tracker's convention
    assert np.allclose(T.camera_trajectory(rel), T.from_first(c2w), atol=1e-9)
    assert np.allclose(T.from_first(c2w), c2w[1:], atol=1e-9)          # (G_0 = identity here)
    wrong = T.relative_poses(c2w)                                      # G_k^-1 G_{k+1}: the inverse motion
    step = np.linalg.norm(rel[:, 4:], axis=1).mean()
    assert S.rpe_translation(rel[:, 4:], wrong[:, 4:]) > 1.5 * step    # about twice the motion, as the advisor measured
    assert np.allclose(T.invert(T.invert(rel)), rel, atol=1e-12)
    # the Visualizer-style accumulation (previous * SE3(q, t)) is a different trajectory: compose_from(rel) = absp, not c2w
    assert np.allclose(T.compose_from(rel), absp[1:], atol=1e-9)
    assert not np.allclose(T.compose_from(rel)[:, 4:], c2w[1:, 4:], atol=1e-4)
