import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
g = np.load(os.path.join(ROOT, "tests", "golden", "pair_64x48_fixed.npz"))
h, w = g["ref"].shape
ctx = capi.Context(capi.default_params(w, h, *[float(v) for v in g["intr"]], max_frames=2, max_pairs=1, n_levels=3, first_level=2, last_level=0, max_iters=6, early_exit=0))
ctx.upload_frames(0, np.stack([g["ref"], g["tgt"]]))
ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
np.set_printoptions(precision=9, linewidth=200)
for lvl, pose in ((2, np.array([0,0,0,1,0,0,0],np.float32)), (2, g["trace_pose"][0]), (0, g["trace_pose"][11])):
    out = ctx.residual_jacobian(0, 1, lvl, pose, dump=True)
    v = out["valid"].astype(bool)
    Jd = out["J"][v].astype(np.float64); r = out["r"][v].astype(np.float64)
    Aex = Jd.T @ Jd
    print("lvl", lvl, "nv", out["n_valid"], "rel diff GPU vs exact (per entry):")
    print(np.abs(out["A"] - Aex) / np.abs(Aex).clip(1e-300))
    print("jtr rel:", np.abs(out["jtr"] - Jd.T @ r) / (np.abs(Jd).T @ np.abs(r)))
row = 1
print("golden A row1 vs exact-from-GPU-J at golden pose[0]:")
out = ctx.residual_jacobian(0, 1, 2, g["trace_pose"][0], dump=True)
v = out["valid"].astype(bool); Jd = out["J"][v].astype(np.float64)
print(np.abs(g["trace_A"][1] - (Jd.T@Jd)) / np.abs(Jd.T@Jd))
