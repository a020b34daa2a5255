#!/usr/bin/env python3
"""Track a recorded sequence (EUROC cam0 / TUM rgb[+depth] directory layout, launch/uw_slam*.launch:5-8) through the
GPU path and write the trajectory in the reference visualiser's CSV format and in TUM format.

  python tools/track_sequence.py --images <dir> [--depth <dir>] --fx .. --fy .. --cx .. --cy .. \
         [--width 640 --height 480] [--groundtruth <file> --tum|--euroc] [--weights huber] [--bilinear] --out traj

No dataset ships with this repository (none is available offline); the synthetic test in tests/test_sequence.py
exercises the same code path.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", required=True)
    ap.add_argument("--depth")
    ap.add_argument("--fx", type=float, required=True); ap.add_argument("--fy", type=float, required=True)
    ap.add_argument("--cx", type=float, required=True); ap.add_argument("--cy", type=float, required=True)
    ap.add_argument("--width", type=int, default=640); ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--start", type=int, default=0); ap.add_argument("--count", type=int, default=0)
    ap.add_argument("--weights", choices=["identity", "tukey", "huber"], default="identity")
    ap.add_argument("--bilinear", action="store_true")
    ap.add_argument("--arith", choices=["opencv", "legacy"], default="opencv", help="arithmetic set (include/uwt.h uwt_arith)")
    ap.add_argument("--fixed-iters", type=int, default=0, help="0: the reference schedule (levels 4..1, early exit)")
    ap.add_argument("--groundtruth"); ap.add_argument("--euroc", action="store_true"); ap.add_argument("--tum", action="store_true")
    ap.add_argument("--out", default="trajectory")
    a = ap.parse_args()
    S = importlib.import_module("uw-slam_amd.sequence")
    T = importlib.import_module("uw-slam_amd.trajectory")
    names = S.list_sorted(a.images)[a.start:]
    dnames = S.list_sorted(a.depth)[a.start:] if a.depth else None
    if a.count:
        names = names[:a.count]
        dnames = dnames[:a.count] if dnames else None
    first = S.load_gray(names[0])
    _, x0, y0 = S.centre_crop(first, a.width, a.height)
    frames = [S.centre_crop(S.load_gray(n), a.width, a.height)[0] for n in names]
    depths = [S.centre_crop(S.load_depth(n), a.width, a.height)[0] for n in dnames] if dnames else None
    over = dict(weights={"identity": 0, "tukey": 1, "huber": 2}[a.weights], sampler=int(a.bilinear), arith={"opencv": 0, "legacy": 1}[a.arith])
    if a.fixed_iters:
        over.update(n_levels=4, first_level=3, last_level=0, max_iters=a.fixed_iters, early_exit=0)
    trk = S.SequenceTracker(a.width, a.height, a.fx, a.fy, a.cx - x0, a.cy - y0, depth=bool(depths), **over)
    t0 = time.perf_counter()
    poses, stats = trk.track(frames, depths)
    dt = time.perf_counter() - t0
    traj = trk.trajectory(poses)
    ref = trk.trajectory(poses, reference_visualiser=True)
    gt = None
    bad = sum(s["status"] != 0 for s in stats)
    metrics = dict(pairs=int(len(poses)), failed=int(bad), seconds=dt, crop_offset=[int(x0), int(y0)],
                   iterations=[int(s["iterations"]) for s in stats])
    if a.groundtruth:
        ts, gtp = (T.read_groundtruth_euroc if a.euroc else T.read_groundtruth_tum)(a.groundtruth)
        idx_all = np.clip(T.ground_truth_indices(len(gtp), len(names), a.start, euroc=a.euroc), 0, len(gtp) - 1)
        gt = gtp[idx_all[1:]]
        # The files hold camera-to-world poses G_k; the tracker's pose of pair k maps the previous camera's coordinates to
        # the current one's, T_k = G_{k+1}^-1 G_k.  RPE: per-pair estimate against that; ATE: the camera trajectory the
        # estimates imply, C_{k+1} = C_k T_k^-1, against G_0^-1 G_k after a rigid alignment.  (The Visualizer-style
        # accumulation previous * SE3(q, t) stays a separate output: <out>_tum.txt / <out>_reference.csv.)
        g_pair = T.pair_ground_truth(gtp[idx_all])
        cam = T.camera_trajectory(poses)
        metrics["ate_rmse_m"] = S.ate_rmse(cam[:, 4:], T.from_first(gtp[idx_all])[:, 4:])
        metrics["rpe_trans_rmse_m"] = S.rpe_translation(poses[:, 4:], g_pair[:, 4:])
        metrics["rpe_rot_rmse_rad"] = T.rpe_rotation(poses, g_pair)
        np.savetxt(a.out + "_camera_tum.txt", np.concatenate([np.arange(1, len(cam) + 1)[:, None], cam[:, 4:], cam[:, :4]], axis=1), fmt="%.9g")
        print("ATE RMSE %.4f m, RPE %.5f m / %.5f rad over %d poses"
              % (metrics["ate_rmse_m"], metrics["rpe_trans_rmse_m"], metrics["rpe_rot_rmse_rad"], len(traj)))
    T.write_reference_csv(a.out + "_reference.csv", ref, gt)
    T.write_tum(a.out + "_tum.txt", np.arange(len(traj), dtype=np.float64), traj)
    np.save(a.out + "_poses.npy", poses)
    import json
    with open(a.out + "_metrics.json", "w") as f:
        json.dump(metrics, f)
    print("%d pairs in %.3f s (%.1f pairs/s incl. upload), %d failed; wrote %s_reference.csv, %s_tum.txt"
          % (len(poses), dt, len(poses) / dt, bad, a.out, a.out))


if __name__ == "__main__":
    main()
