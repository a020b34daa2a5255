#!/usr/bin/env python3
"""Experiment (library built with -DUWT_EXP_STAMPS): wall-clock stamps (100 MHz) of the phases of the last two k_iterate
launches of a single-pair alignment: entry, after the update, after the pixel loop, after the record store."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h = 640, 480
intr = (525.0, 525.0, 319.5, 239.5)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=3, z=1.0, with_depth=True)
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2, max_pairs=1, n_levels=levels, first_level=levels - 1, last_level=0,
                                       max_iters=10, early_exit=0, has_depth=1))
ctx.upload_frames(0, np.stack([ref, tgt]), np.stack([dep, dep]))
ctx.build_pyramids(0, 2)
ctx.apply_gradient(0, 2)
for _ in range(20):
    ctx.estimate_pose_batch([0], [1])
L = capi.lib()
n_rec = 150
out = [np.zeros((n_rec, 64), np.uint32) for _ in range(2)]
for par in range(2):
    st = L.uwt_exp_read_records(ctx._h if hasattr(ctx, "_h") else ctx.handle, par, out[par].ctypes.data_as(C.POINTER(C.c_uint32)), n_rec * 64)
    assert st == 0
recs = []
for par in range(2):
    r = out[par][:, 60:64].astype(np.int64)
    r = r[r[:, 0] != 0]
    recs.append(r)
# order the two launches by entry time
recs.sort(key=lambda r: r[:, 0].min())
t0 = recs[0][:, 0].min()
for name, r in zip(("launch N-1", "launch N"), recs):
    r = (r - t0) * 10  # ns
    print("%s: %d blocks" % (name, len(r)))
    for k, lab in enumerate(("entry", "update done", "pixel loop done", "record stored")):
        print("   %-16s min %7d  median %7d  max %7d ns" % (lab, r[:, k].min(), np.median(r[:, k]), r[:, k].max()))
    print("   per-block: update %.0f  loop %.0f  reduce+store %.0f ns (medians)" % (
        np.median(r[:, 1] - r[:, 0]), np.median(r[:, 2] - r[:, 1]), np.median(r[:, 3] - r[:, 2])))
g = np.zeros(16, np.uint32)
assert L.uwt_exp_read_records(ctx._h, 2, g.ctypes.data_as(C.POINTER(C.c_uint32)), 16) == 0
g = (g.astype(np.int64) - int(g[0])) * 10
print("update of block 0, last launch (ns from its start):")
for i, lab in enumerate(("start", "parts folded", "-", "-", "-", "sums visible", "solve begins", "solve done", "exp done", "compose done", "state broadcast")):
    print("   %-16s %7d" % (lab, g[i]))
print("block reduction of block 0, last k_iterate (ns from its start):")
for i, lab in ((11, "start"), (12, "pass 0 folded"), (13, "pass 1 folded"), (14, "segments visible"), (15, "record store issued")):
    print("   %-20s %7d" % (lab, g[i] - g[11]))
ctx.close()
