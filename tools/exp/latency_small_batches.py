import importlib, sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
w, h = [int(x) for x in os.environ.get("WH", "640,480").split(",")]
f = 525.0 * w / 640.0
intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=3, z=1.0, with_depth=True)
LV = int(os.environ.get("LEVELS", "4"))
for name, over in (("fixed", dict(n_levels=LV, first_level=LV - 1, last_level=0, max_iters=10, early_exit=0, has_depth=1)), ("reference", dict(has_depth=1))):
    for n in [int(x) for x in os.environ.get("NS", "2,3,4,6,8,12").split(",")]:
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * n, max_pairs=n, **over))
        for i in range(n):
            ctx.upload_frames(2 * i, np.stack([ref, tgt]), np.stack([dep, dep]))
        ctx.build_pyramids(0, 2 * n); ctx.apply_gradient(0, 2 * n)
        rs = np.arange(n) * 2
        for _ in range(10): ctx.estimate_pose_batch(rs, rs + 1)
        t0 = time.perf_counter()
        for _ in range(100): ctx.estimate_pose_batch(rs, rs + 1)
        print("%dx%d %s n=%d chained=%s: %.3f ms" % (w, h, name, n, os.environ.get("UWT_CHAINED", "auto"), (time.perf_counter() - t0) / 100 * 1e3), flush=True) if True else print("%s n=%d chained=%s: %.3f ms" % (name, n, os.environ.get("UWT_CHAINED", "auto"), (time.perf_counter() - t0) / 100 * 1e3), flush=True)
        ctx.close()
