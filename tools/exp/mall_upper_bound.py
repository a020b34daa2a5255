#!/usr/bin/env python3
"""Upper bound of what keeping a chunk's planes in the 256 MB memory-side cache could return: the default batch (1024 pairs of
640x480, 4 x 10) with every pair naming the slots of one of only D distinct pairs — same launches, same arithmetic, D x 2.46 MB
of planes at level 0 instead of 2.5 GB.  Prints alignments/s and the per-level launch durations for several D."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
capi = importlib.import_module("uw-slam_amd.capi")
w, h, P, U = 640, 480, 1024, 128
intr = (525.0, 525.0, 319.5, 239.5)
gen = bench._cpp_generator()
over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0, has_depth=1)
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * P, max_pairs=P, **over))
refs, tgts, deps = zip(*[gen(w, h, intr, g, True) for g in range(U)])
for i0 in range(0, P, 128):
    fr = np.empty((256, h, w), np.uint8); fr[0::2] = np.stack(refs); fr[1::2] = np.stack(tgts)
    dp = np.empty((256, h, w), np.uint16); dp[0::2] = np.stack(deps); dp[1::2] = np.stack(deps)
    ctx.upload_frames(2 * i0, fr, dp)
buf = torch.empty((P, 7), dtype=torch.float32, device="cuda")
all_slots = np.arange(P, dtype=np.int32) * 2
ctx.track_batch_async(0, 2 * P, all_slots, all_slots + 1, buf.data_ptr()); ctx.sync()      # every slot prepared
for D in (1024, 256, 64, 32, 8):
    ref = (np.arange(P, dtype=np.int32) % D) * 2
    for rep in range(2):
        for _ in range(5):
            ctx.track_batch_async(0, 2 * D, ref, ref + 1, buf.data_ptr(), grad_refs_only=False)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.track_batch_async(0, 2 * D, ref, ref + 1, buf.data_ptr(), grad_refs_only=False)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 20
    ctx.profile_enable(1)
    ctx.track_batch_async(0, 2 * D, ref, ref + 1, buf.data_ptr(), grad_refs_only=False); ctx.sync()
    lv = ctx.profile_read_levels(); clk = ctx.profile_clock()
    ctx.profile_enable(0)
    print("D=%4d  %.3f ms/step  %.0f alignments/s (pyramids of %d frames)  level ms/eval: %s  clock %.3f GHz"
          % (D, dt * 1e3, P / dt, 2 * D, ["%.4f" % (ms / n) for ms, n in lv if n], clk), flush=True)
