#!/usr/bin/env python3
"""Track a recorded sequence (EUROC cam0 / TUM rgb[+depth] directory layout, launch/uw_slam*.launch:5-8) through the
GPU path and write the trajectory in the reference visualiser's CSV format and in TUM format.

  python tools/track_sequence.py --images <dir> [--depth <dir>] --fx .. --fy .. --cx .. --cy .. \
         [--width 640 --height 480] [--groundtruth <file> --tum|--euroc] [--weights huber] [--bilinear] --out traj
  python tools/track_sequence.py --images <EUROC cam0/data> --fx 458.654 --fy 457.296 --cx 367.215 --cy 248.375 \
         --distortion=-0.28340811,0.07395907,0.00019359,1.76187114e-05 --rectified-size 736,480 --out traj
      the reference's own EUROC path (calibration/calibrationEUROC.xml): every frame rectified with the maps of CameraModel
      (src/CameraModel.cpp:84-90), cropped to the window System::CalculateROI finds on the first one (src/System.cpp:148-191: a
      data-dependent, odd size) and tracked AT THAT SIZE with the unshifted new camera matrix, as the reference does

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
    ap.add_argument("--distortion", default="", help="k1,k2,p1,p2: rectify + ROI-crop like the reference (System.cpp:148-191, 231-236); "
                    "--width / --height are then ignored: the frame size is the ROI's")
    ap.add_argument("--rectified-size", default="", help="W,H of the rectified frame (out_width / out_height of the calibration file); default: the input size")
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
    fx, fy, cx, cy = a.fx, a.fy, a.cx, a.cy
    if a.distortion:
        capi = importlib.import_module("uw-slam_amd.capi")
        dist = [float(v) for v in a.distortion.split(",")]
        ih, iw = first.shape
        ow, oh = [int(v) for v in a.rectified_size.split(",")] if a.rectified_size else (iw, ih)
        ing = capi.Ingest([a.fx, a.fy, a.cx, a.cy], dist, iw, ih, ow, oh)
        x0, y0, a.width, a.height = [int(v) for v in ing.calculate_roi(first)]          # w_, h_ = the ROI's (src/System.cpp:186-190)
        fx, fy, cx, cy = [float(v) for v in ing.newK]                                    # K_ = GetK(): not shifted by the crop (:105-112)
        frames = [np.ascontiguousarray(ing.undistort(S.load_gray(n))[y0:y0 + a.height, x0:x0 + a.width]) for n in names]
        depths = None
        ing.close()
        cx, cy = cx + x0, cy + y0                                                        # (undone below: the reference keeps cx, cy)
    else:
        _, x0, y0 = S.centre_crop(first, a.width, a.height)
        frames = [S.centre_crop(S.load_gray(n), a.width, a.height)[0] for n in names]
        depths = [S.centre_crop(S.load_depth(n), a.width, a.height)[0] for n in dnames] if dnames else None
    over = dict(weights={"identity": 0, "tukey": 1, "huber": 2}[a.weights], sampler=int(a.bilinear), arith={"opencv": 0, "legacy": 1}[a.arith])
    if a.fixed_iters:
        over.update(n_levels=4, first_level=3, last_level=0, max_iters=a.fixed_iters, early_exit=0)
    trk = S.SequenceTracker(a.width, a.height, fx, fy, cx - x0, cy - y0, depth=bool(depths), **over)
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
