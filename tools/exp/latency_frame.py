#!/usr/bin/env python3
"""What one NEW FRAME costs the drop-in, wall clock, synchronous: the frame crosses PCIe from page-locked memory
(uwt_upload_frames_async), ONE call prepares it and aligns it to the previous frame (uwt_track_batch_host_async: image pyramid of the
new slot, depth pyramid and gradients of the reference slot, the alignment, the pose into host memory), the host waits for the ticket —
System::AddFrame + System::Tracking of the reference (src/System.cpp:193-251) on a sequence, frame after frame through a ring of slots.
Sizes: 640 x 480 with depth (the bench's 4 x 10 schedule and the reference's), the EUROC sizes and ROI-like sizes no power of two divides
(z = 1, reference schedule: 5 levels, 4 -> 1, early exit).  python tools/exp/latency_frame.py [frames]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
E = (458.654, 457.296, 359.215, 248.375)     # EUROC cam0 (the calibration of the reference's own sample configuration)
FIXED = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)
CASES = [("640x480 depth, 4 x 10", 640, 480, (525.0, 525.0, 319.5, 239.5), True, FIXED),
         ("640x480 depth, reference", 640, 480, (525.0, 525.0, 319.5, 239.5), True, {}),
         ("752x480 z=1, reference", 752, 480, E, False, {}),
         ("736x480 z=1, reference", 736, 480, E, False, {}),
         ("725x465 z=1, reference", 725, 465, E, False, {}),
         ("733x471 z=1, reference", 733, 471, E, False, {}),
         ("735x479 z=1, reference", 735, 479, E, False, {}),
         # the frame as the reference holds it: a pageable cv::Mat VIEW into the rectified 752 x 480 image (images_[0] = distortion(ROI),
         # src/System.cpp:235), handed to uwt_set_frame with the parent's step
         ("725x465 z=1, reference, set_frame(view)", 725, 465, E, False, {}),
         ("640x480 depth, reference, set_frame(view)", 640, 480, (525.0, 525.0, 319.5, 239.5), True, {})]
N_DISTINCT = 12
for name, w, h, intr, with_depth, over in CASES:
    out = synth.render_sequence(w, h, *intr, N_DISTINCT, 5, with_depth=with_depth, margin=(96, 64))
    frames, depths = out[0], out[1]
    ring = 3
    view = "view" in name
    if view:
        parents = [np.pad(f, ((8, 7), (13, 14))) for f in frames]
        parents_d = [np.pad(d, ((8, 7), (13, 14))) for d in depths] if with_depth else None
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=ring, max_pairs=1, has_depth=int(with_depth), **over))
    pf = capi.pinned_empty((1, h, w), np.uint8)
    pd = capi.pinned_empty((1, h, w), np.uint16) if with_depth else None
    hp, hs = capi.pinned_empty((1, 7), np.float32), capi.pinned_empty((1, 4), np.int32)
    ctx.upload_frames(0, frames[0][None], depths[0][None] if with_depth else None)
    ctx.build_pyramids(0, 1)
    prev, ts, evals = 0, [], []
    # the sequence walks forth and back over its distinct frames: every step is a real inter-frame motion
    order = [i for _ in range(n_frames // (2 * N_DISTINCT - 2) + 2) for i in list(range(N_DISTINCT)) + list(range(N_DISTINCT - 2, 0, -1))][1:n_frames + 11]
    for k, fi in enumerate(order):
        slot = (prev + 1) % ring
        t0 = time.perf_counter()
        if view:
            ctx.set_frame(slot, parents[fi][8:8 + h, 13:13 + w], parents_d[fi][8:8 + h, 13:13 + w] if with_depth else None)
        else:
            pf[0] = frames[fi]                      # (the host's own copy into page-locked memory is part of a frame's cost)
            if with_depth:
                pd[0] = depths[fi]
            ctx.upload_frames_async(slot, pf, pd)
        tk = ctx.track_batch_host_async(slot, 1, [prev], [slot], hp, hs, grad_refs_only=True)
        ctx.wait_ticket(tk)
        dt = time.perf_counter() - t0
        if k >= 10:
            ts.append(dt)
            evals.append(int(hs[0, 1]))
        prev = slot
    ts = np.array(ts) * 1e3
    print("%-42s %.3f ms per frame (median %.3f, p95 %.3f), %.1f evaluations on average, %d frames" %
          (name, ts.mean(), np.median(ts), np.percentile(ts, 95), np.mean(evals), len(ts)), flush=True)
    ctx.close()
