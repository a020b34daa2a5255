#!/usr/bin/env python3
"""Every way a frame reaches its slot — uwt_upload_frames, uwt_upload_frames_async (page-locked and pageable memory), uwt_set_frame
(tight rows, a view into a parent, a column of a much wider parent) — at random sizes (most of them with pitched device rows), random
slot windows and batch sizes, contexts created and destroyed all along (the staging areas are allocated, grown and freed): level 0 of
the image and depth planes of EVERY slot read back and compared with what was uploaded last.  python tools/exp/upload_fuzz.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
checked = bad = calls = 0
t0 = time.time()
for case in range(cases):
    w, h = (int(rng.integers(5, 200)), int(rng.integers(5, 130))) if rng.random() < 0.7 else (int(rng.integers(200, 800)), int(rng.integers(130, 500)))
    if rng.random() < 0.25:
        w = (w + 3) // 4 * 4                      # tight device rows
    dep = bool(rng.random() < 0.6)
    slots = int(rng.integers(1, 14))
    ctx = capi.Context(capi.default_params(w, h, 0.8 * w, 0.8 * w, w / 2, h / 2, max_frames=slots, max_pairs=1, n_levels=1, first_level=0, last_level=0, has_depth=int(dep)))
    want_g = [None] * slots
    want_d = [None] * slots
    keep = []                                     # host arrays of asynchronous uploads stay alive until the sync
    for step in range(int(rng.integers(1, 9))):
        mode = int(rng.integers(0, 6))
        first = int(rng.integers(0, slots))
        n = int(rng.integers(1, slots - first + 1)) if mode < 3 else 1
        g = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
        d = rng.integers(0, 65536, (n, h, w)).astype(np.uint16) if dep else None
        calls += 1
        if mode == 0:
            ctx.upload_frames(first, g, d)
        elif mode == 1:                           # page-locked, asynchronous
            pg = capi.pinned_empty((n, h, w), np.uint8); pg[:] = g
            pd = None
            if dep:
                pd = capi.pinned_empty((n, h, w), np.uint16); pd[:] = d
            ctx.upload_frames_async(first, pg, pd); keep.append((pg, pd))
        elif mode == 2:                           # pageable, asynchronous (allowed; the runtime stages it)
            ctx.upload_frames_async(first, g, d); keep.append((g, d))
        elif mode == 3:
            ctx.set_frame(first, g[0], d[0] if dep else None)
        elif mode == 4:                           # a view into the corner of a parent
            py, px = int(rng.integers(0, 9)), int(rng.integers(1, 40))
            pg = np.pad(g[0], ((py, 0), (px, 0))); pd = np.pad(d[0], ((py, 0), (px, 0))) if dep else None
            ctx.set_frame(first, pg[py:, px:], pd[py:, px:] if dep else None)
        else:                                     # a column of a parent more than four times as wide
            pg = np.pad(g[0], ((0, 0), (w, 3 * w + 5))); pd = np.pad(d[0], ((0, 0), (w, 3 * w + 5))) if dep else None
            ctx.set_frame(first, pg[:, w:2 * w], pd[:, w:2 * w] if dep else None)
        for i in range(n):
            want_g[first + i] = g[i]
            if dep:
                want_d[first + i] = d[i]
        if rng.random() < 0.4:
            ctx.sync(); keep.clear()
    ctx.sync()
    for s in range(slots):
        if want_g[s] is None:
            continue
        checked += 1
        ok = np.array_equal(ctx.get_plane(s, 0, capi.PLANE_IMAGE), want_g[s]) and (not dep or np.array_equal(ctx.get_plane(s, 0, capi.PLANE_DEPTH), want_d[s]))
        if not ok:
            bad += 1
            if bad <= 10:
                gi = ctx.get_plane(s, 0, capi.PLANE_IMAGE)
                print("DIFFERS case %d %dx%d depth %d slot %d/%d: image %d px differ%s" % (case, w, h, dep, s, slots, int((gi != want_g[s]).sum()),
                      ", depth %d px differ" % int((ctx.get_plane(s, 0, capi.PLANE_DEPTH) != want_d[s]).sum()) if dep else ""), flush=True)
    ctx.close()
print("upload fuzz seed %d: %d contexts, %d upload calls, %d slots read back, %d differ, %.0f s" % (seed, cases, calls, checked, bad, time.time() - t0))
sys.exit(1 if bad else 0)
