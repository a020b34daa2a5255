"""Trajectory I/O next to the tracking path: the reference visualiser's 14-column CSV
(src/Visualizer.cpp:386-400), TUM-format trajectory files, and the two ground-truth readers
(ReadGroundTruthTUM / ReadGroundTruthEUROC, src/Visualizer.cpp:449-505).  The pose accumulation itself
(Visualizer::UpdateMessages, :304-325) runs in the library: Context.accumulate_trajectory / uwt_accumulate_trajectory.
"""
import numpy as np


def write_reference_csv(path, est, gt=None):
    """One row per frame: est x,y,z,qx,qy,qz,qw then gt x,y,z,qx,qy,qz,qw (src/Visualizer.cpp:386-400).
    `est`, `gt`: n x 7 arrays laid out qx qy qz qw tx ty tz (the library's pose layout)."""
    est = np.asarray(est, np.float64).reshape(-1, 7)
    gt = np.zeros_like(est) if gt is None else np.asarray(gt, np.float64).reshape(-1, 7)
    with open(path, "w") as f:
        for e, g in zip(est, gt):
            vals = [e[4], e[5], e[6], e[0], e[1], e[2], e[3], g[4], g[5], g[6], g[0], g[1], g[2], g[3]]
            f.write(",".join(repr(float(v)) for v in vals) + "\n")


def read_reference_csv(path):
    rows = np.loadtxt(path, delimiter=",", ndmin=2)
    def to_pose(b):
        return np.concatenate([b[:, 3:7], b[:, 0:3]], axis=1)
    return to_pose(rows[:, :7]), to_pose(rows[:, 7:14])


def write_tum(path, timestamps, traj):
    """TUM RGB-D trajectory format: 'timestamp tx ty tz qx qy qz qw' per line."""
    traj = np.asarray(traj, np.float64).reshape(-1, 7)
    with open(path, "w") as f:
        for t, p in zip(timestamps, traj):
            f.write("%.6f %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n" % (t, p[4], p[5], p[6], p[0], p[1], p[2], p[3]))


def read_groundtruth_tum(path):
    """src/Visualizer.cpp:449-477: skip 3 header lines; 'timestamp tx ty tz qx qy qz qw' separated by spaces.
    Returns (timestamps, n x 7 poses in the library layout)."""
    ts, rows = [], []
    with open(path) as f:
        lines = f.read().splitlines()[3:]
    for line in lines:
        if not line.strip():
            continue
        v = [float(x) for x in line.split(" ")[:8]]
        ts.append(v[0])
        rows.append([v[4], v[5], v[6], v[7], v[1], v[2], v[3]])
    return np.array(ts), np.array(rows, np.float64).reshape(-1, 7)


def read_groundtruth_euroc(path):
    """src/Visualizer.cpp:479-505: skip 1 header line; 'timestamp,px,py,pz,qw,qx,qy,qz,...' (EUROC state csv).
    The reference stores the first 7 values after the timestamp in file order; returned here in the library layout."""
    ts, rows = [], []
    with open(path) as f:
        lines = f.read().splitlines()[1:]
    for line in lines:
        if not line.strip():
            continue
        v = line.split(",")
        ts.append(float(v[0]))
        px, py, pz, qw, qx, qy, qz = [float(x) for x in v[1:8]]
        rows.append([qx, qy, qz, qw, px, py, pz])
    return np.array(ts), np.array(rows, np.float64).reshape(-1, 7)


def ground_truth_indices(n_gt, n_images, start_index, euroc=False):
    """Index schedule of the reference: step = n_gt // n_images, index = start * step (+600 for EUROC,
    src/Visualizer.cpp:476, 504)."""
    step = n_gt // max(n_images, 1)
    first = start_index * step + (600 if euroc else 0)
    return first + step * np.arange(n_images)


def _qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ v


def relative_poses(abs_poses):
    """rel[k] = abs[k]^-1 * abs[k+1] (float64) for n absolute poses in the library layout — the inverse of compose_from: the
    per-step factors of a trajectory that was ACCUMULATED the way Visualizer::UpdateMessages does it, final = previous *
    SE3(q, t) (src/Visualizer.cpp:313).  NOT the tracker's per-pair pose of a camera-to-world ground-truth file — that is
    pair_ground_truth (the inverse)."""
    P = np.asarray(abs_poses, np.float64).reshape(-1, 7)
    out = np.zeros((max(len(P) - 1, 0), 7))
    for k in range(len(P) - 1):
        qa = P[k, :4] / np.linalg.norm(P[k, :4])
        qb = P[k + 1, :4] / np.linalg.norm(P[k + 1, :4])
        qi = qa * [-1, -1, -1, 1]
        out[k, :4] = _qmul(qi, qb)
        out[k, 4:] = _qrot(qi, P[k + 1, 4:] - P[k, 4:])
    return out


def invert(poses):
    """Inverse of each SE(3) pose (n x 7, library layout), float64."""
    P = np.asarray(poses, np.float64).reshape(-1, 7)
    out = np.zeros_like(P)
    for k, p in enumerate(P):
        qi = p[:4] / np.linalg.norm(p[:4]) * [-1, -1, -1, 1]
        out[k, :4] = qi
        out[k, 4:] = -_qrot(qi, p[4:])
    return out


def pair_ground_truth(cam_to_world):
    """Per-pair ground truth in the TRACKER's convention from a camera-to-world trajectory G_0 .. G_{n-1} (what TUM's
    groundtruth.txt and EUROC's state estimate hold): previous_frame->rigid_transformation_ maps the previous camera's
    coordinates to the current camera's, X_{k+1} = T_k X_k (src/Tracker.cpp:595, the warp of :1450), so
    T_k = G_{k+1}^-1 * G_k — the inverse of relative_poses(G)[k]."""
    return invert(relative_poses(cam_to_world))


def camera_trajectory(pair_poses, start=None):
    """Camera-to-world poses C_1 .. C_n from the tracker's per-pair poses T_0 .. T_{n-1} (X_{k+1} = T_k X_k):
    C_{k+1} = C_k * T_k^-1, C_0 = start (identity: the first camera's frame is the world).  Comparable with a camera-to-world
    ground truth expressed relative to its first pose, from_first(G)."""
    return compose_from(invert(pair_poses), start)


def from_first(cam_to_world):
    """G_0^-1 * G_k for k = 1 .. n-1: a camera-to-world trajectory expressed in its first camera's frame."""
    return compose_from(relative_poses(cam_to_world))


def compose_from(rel, start=None):
    """abs[k] = start * rel[0] * ... * rel[k] (float64), one row per relative pose — the same product the library's
    uwt_accumulate_trajectory forms in f32."""
    rel = np.asarray(rel, np.float64).reshape(-1, 7)
    q = np.array([0, 0, 0, 1.0]) if start is None else np.asarray(start[:4], np.float64)
    t = np.zeros(3) if start is None else np.asarray(start[4:], np.float64)
    out = np.zeros_like(rel)
    for k, r in enumerate(rel):
        t = t + _qrot(q, r[4:])
        q = _qmul(q, r[:4] / np.linalg.norm(r[:4]))
        out[k, :4], out[k, 4:] = q, t
    return out


def rpe_rotation(est_rel, gt_rel):
    """RMSE of the angle of est^-1 * gt per pair (radians)."""
    e = np.asarray(est_rel, np.float64).reshape(-1, 7)
    g = np.asarray(gt_rel, np.float64).reshape(-1, 7)
    ang = []
    for a, b in zip(e, g):
        d = _qmul(a[:4] * [-1, -1, -1, 1] / np.linalg.norm(a[:4]), b[:4] / np.linalg.norm(b[:4]))
        ang.append(2.0 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3])))
    return float(np.sqrt(np.mean(np.square(ang)))) if ang else 0.0
