"""Deterministic synthetic frame pairs: band-limited noise textures re-rendered under a small
random SE(3) for a fronto-parallel plane at depth z.  Host-side test/bench input only.
"""
import numpy as np
from scipy import ndimage


def texture(w, h, seed, sigma=3.0):
    """Standard-normal field -> Gaussian blur (sigma px) -> min-max to [0,255] u8."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((h, w)).astype(np.float32)
    f = ndimage.gaussian_filter(f, sigma, mode="wrap")
    lo, hi = float(f.min()), float(f.max())
    return np.clip(np.rint((f - lo) * (255.0 / (hi - lo))), 0, 255).astype(np.uint8)


def rodrigues(rvec):
    rvec = np.asarray(rvec, np.float64)
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def random_motion(rng, max_t=0.01, max_deg=0.5):
    """‖t‖ ~ U(0, max_t) m, rotation angle ~ U(0, max_deg)."""
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, max_deg))
    tdir = rng.standard_normal(3)
    tdir /= np.linalg.norm(tdir)
    return rodrigues(axis * ang), tdir * rng.uniform(0, max_t)


def render_pair(w, h, fx, fy, cx, cy, seed, z=1.0, max_t=0.01, max_deg=0.5, with_depth=False):
    """Returns (ref u8, tgt u8, depth u16 or None, R, t).  tgt(u') = ref(H^-1 u'), H = K (R + t n^T / z) K^-1."""
    rng = np.random.default_rng(seed + 7919)
    ref = texture(w, h, seed)
    R, t = random_motion(rng, max_t, max_deg)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    H = K @ (R + np.outer(t, [0, 0, 1.0]) / z) @ np.linalg.inv(K)
    Hi = np.linalg.inv(H)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    u = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den
    v = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
    tgt = ndimage.map_coordinates(ref.astype(np.float32), [v, u], order=1, mode="reflect")
    tgt = np.clip(np.rint(tgt), 0, 255).astype(np.uint8)
    depth = None
    if with_depth:
        # TUM-style u16 depth at the reference's 0.0002 m/unit scale (Tracker.cpp:1261), with a few invalid zeros
        depth = np.full((h, w), int(round(z / 0.0002)), np.uint16)
        holes = rng.random((h, w)) < 0.01
        depth[holes] = 0
    return ref, tgt, depth, R, t


def shifted_pair(w, h, seed, dx=2, dy=0):
    """Reference + the same texture rolled by integer pixels (known-answer cases)."""
    ref = texture(w, h, seed)
    return ref, np.roll(ref, (dy, dx), axis=(0, 1))


def _quat_from_R(R):
    """Unit quaternion (qx, qy, qz, qw) of a rotation matrix (Shepperd's method)."""
    R = np.asarray(R, np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = 2.0 * np.sqrt(tr + 1.0)
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
        q = [0.0, 0.0, 0.0, (R[k, j] - R[j, k]) / s]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q) * (1.0 if q[3] >= 0 else -1.0)


def render_sequence(w, h, fx, fy, cx, cy, n, seed, z=1.0, step_t=0.006, step_deg=0.25, with_depth=False, margin=(192, 128)):
    """A camera moving smoothly in front of a textured plane (fronto-parallel at depth z for frame 0): the stand-in for a
    recorded sequence (BASELINE configs 1, 2, 5).  Frame i sees the plane under X_i = R_i X_0 + t_i, i.e. through the
    homography H_i = K (R_i + t_i n^T / z) K^-1 of frame 0's pixels; the inter-frame motion stays below step_t metres and
    step_deg degrees.  Returns (frames [n] u8, depths [n] u16 or None, rel [n-1, 7], abs [n, 7]):
      rel[i]  = the true rigid transformation of pair (i, i+1) in the tracker's convention, X_{i+1} = T X_i
                (previous_frame->rigid_transformation_, src/Tracker.cpp:595), laid out qx qy qz qw tx ty tz;
      abs[k]  = rel[0] * rel[1] * ... * rel[k-1] (abs[0] = identity): the trajectory Visualizer::UpdateMessages would
                accumulate from the true relative poses (src/Visualizer.cpp:304-325) — the ground truth of the run.
    depth is the plane's true per-pixel depth in each frame at 0.0002 m per unit (src/Tracker.cpp:1261) with 1 % holes."""
    rng = np.random.default_rng(seed + 104729)
    mx, my = margin
    tex = texture(w + 2 * mx, h + 2 * my, seed).astype(np.float32)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Ki = np.linalg.inv(K)
    # a smooth closed-ish path: sums of two sinusoids per axis with random phases, scaled to the step bounds
    ph = rng.uniform(0, 2 * np.pi, size=(2, 6))
    i = np.arange(n)[:, None]
    path = np.sin(2 * np.pi * i / max(n, 2) * 1.0 + ph[0]) + 0.5 * np.sin(2 * np.pi * i / max(n, 2) * 2.3 + ph[1])
    path -= path[0]
    dmax = np.abs(np.diff(path, axis=0)).max(axis=0) if n > 1 else np.ones(6)
    tvec = path[:, :3] / np.maximum(dmax[:3], 1e-12) * (step_t / np.sqrt(3.0))
    rvec = path[:, 3:] / np.maximum(dmax[3:], 1e-12) * (np.deg2rad(step_deg) / np.sqrt(3.0))
    Rs = [rodrigues(r) for r in rvec]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    frames, depths = [], ([] if with_depth else None)
    nrm = np.array([0.0, 0.0, 1.0])
    for k in range(n):
        R, t = Rs[k], tvec[k]
        Hi = np.linalg.inv(K @ (R + np.outer(t, nrm) / z) @ Ki)
        den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
        u = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den
        v = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
        f = ndimage.map_coordinates(tex, [v + my, u + mx], order=1, mode="reflect")
        frames.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
        if with_depth:
            m = R @ nrm                                   # plane in frame k: m . X = z + m . t
            ray = Ki[2, 0] * xs + Ki[2, 1] * ys + Ki[2, 2]
            mdot = m[0] * (Ki[0, 0] * xs + Ki[0, 1] * ys + Ki[0, 2]) + m[1] * (Ki[1, 0] * xs + Ki[1, 1] * ys + Ki[1, 2]) + m[2] * ray
            Z = (z + m @ t) / mdot
            d = np.clip(np.rint(Z / 0.0002), 0, 65535).astype(np.uint16)
            d[rng.random((h, w)) < 0.01] = 0
            depths.append(d)
    rel = np.zeros((max(n - 1, 0), 7))
    absp = np.zeros((n, 7)); absp[:, 3] = 1.0
    A = np.eye(4)
    for k in range(n - 1):
        T0 = np.eye(4); T0[:3, :3] = Rs[k]; T0[:3, 3] = tvec[k]
        T1 = np.eye(4); T1[:3, :3] = Rs[k + 1]; T1[:3, 3] = tvec[k + 1]
        T = T1 @ np.linalg.inv(T0)                        # X_{k+1} = T X_k
        rel[k] = np.concatenate([_quat_from_R(T[:3, :3]), T[:3, 3]])
        A = A @ T
        absp[k + 1] = np.concatenate([_quat_from_R(A[:3, :3]), A[:3, 3]])
    return frames, depths, rel, absp


def camera_to_world_poses(rel):
    """The physical camera-to-world poses G_0 .. G_n of render_sequence's camera (world = the first camera's frame) from its
    true per-pair transformations rel[k] (X_{k+1} = rel[k] X_k): G_0 = I, G_{k+1} = G_k * rel[k]^-1.  What a motion-capture
    ground-truth file of the sequence would hold (TUM groundtruth.txt, EUROC state estimate)."""
    rel = np.asarray(rel, np.float64).reshape(-1, 7)
    out = np.zeros((len(rel) + 1, 7)); out[:, 3] = 1.0
    G = np.eye(4)
    for k, r in enumerate(rel):
        x, y, z, w = r[:4] / np.linalg.norm(r[:4])
        T = np.eye(4)
        T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
        T[:3, 3] = r[4:]
        G = G @ np.linalg.inv(T)
        out[k + 1] = np.concatenate([_quat_from_R(G[:3, :3]), G[:3, 3]])
    return out


def write_tum_layout(root, frames, depths, abs_poses, gt_per_frame=4):
    """A directory in the layout launch/uw_slamTUM.launch:5-8 points the reference at: rgb/<stamp>.png, depth/<stamp>.png and
    groundtruth.txt — 3 comment lines, then 'timestamp tx ty tz qx qy qz qw' (src/Visualizer.cpp:449-477), `gt_per_frame`
    ground-truth rows per image (the reference strides through the file with step = rows // images, :476), every row the
    pose of the image it falls on.  Returns (rgb dir, depth dir or None, groundtruth path)."""
    import os
    from PIL import Image
    rgb = os.path.join(root, "rgb"); os.makedirs(rgb, exist_ok=True)
    dep = None
    if depths is not None:
        dep = os.path.join(root, "depth"); os.makedirs(dep, exist_ok=True)
    for i, f in enumerate(frames):
        stamp = "%.6f" % (1305031100.0 + i / 30.0)
        Image.fromarray(f).save(os.path.join(rgb, stamp + ".png"))
        if depths is not None:
            Image.fromarray(depths[i]).save(os.path.join(dep, stamp + ".png"))
    gt = os.path.join(root, "groundtruth.txt")
    with open(gt, "w") as fh:
        fh.write("# ground truth trajectory\n# file: synthetic plane sequence\n# timestamp tx ty tz qx qy qz qw\n")
        for i, p in enumerate(abs_poses):
            for r in range(gt_per_frame):
                fh.write("%.4f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n"
                         % (1305031100.0 + (i + r / gt_per_frame) / 30.0, p[4], p[5], p[6], p[0], p[1], p[2], p[3]))
    return rgb, dep, gt


def write_euroc_layout(root, frames, abs_poses, gt_per_frame=10, lead_rows=600):
    """mav0/cam0/data/<ns>.png + mav0/state_groundtruth_estimate0/data.csv (launch/uw_slamEUROC.launch:5-8): one header line,
    then 'timestamp,px,py,pz,qw,qx,qy,qz,...' rows (src/Visualizer.cpp:479-505).  The reference starts `lead_rows` = 600 rows
    into the file (:504); those rows repeat the first pose.  Returns (image dir, csv path)."""
    import os
    from PIL import Image
    img = os.path.join(root, "mav0", "cam0", "data"); os.makedirs(img, exist_ok=True)
    gtd = os.path.join(root, "mav0", "state_groundtruth_estimate0"); os.makedirs(gtd, exist_ok=True)
    for i, f in enumerate(frames):
        Image.fromarray(f).save(os.path.join(img, "%d.png" % (1403636579763555584 + i * 50000000)))
    path = os.path.join(gtd, "data.csv")
    n = len(frames)
    # the reference reads row 600 + step * i for image i with step = rows // images: both must hold at once, which for a
    # short sequence needs step >= 602 - n (a real MH_01 has 36 382 rows for 3 682 images: step 9)
    step = max(gt_per_frame, lead_rows + 2 - n)
    total = step * n + n - 1
    with open(path, "w") as fh:
        fh.write("#timestamp,p_RS_R_x [m],p_RS_R_y [m],p_RS_R_z [m],q_RS_w [],q_RS_x [],q_RS_y [],q_RS_z [],v,v,v,bw,bw,bw,ba,ba,ba\n")
        for r in range(total):
            k = r - lead_rows
            i = 0 if k < 0 else min(k // step, n - 1)
            p = abs_poses[i]
            fh.write("%d,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,0,0,0,0,0,0,0,0,0\n"
                     % (1403636579763555584 + r * 5000000, p[4], p[5], p[6], p[3], p[0], p[1], p[2]))
    return img, path
