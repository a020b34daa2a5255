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
