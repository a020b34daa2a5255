"""Diagnostic: step the GPU pipeline iteration by iteration through the per-stage C-ABI entry points and print its
divergence from a golden oracle trace."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
name = sys.argv[1] if len(sys.argv) > 1 else "pair_64x48_fixed"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
h, w = g["ref"].shape
over = {}
for k, v in zip(g["over_keys"], g["over_vals"]):
    over[str(k)] = float(v) if str(k) in ("gain", "z_factor", "angle_factor", "epsilon") else int(v)
ctx = capi.Context(capi.default_params(w, h, *[float(v) for v in g["intr"]], max_frames=2, max_pairs=1, **over))
ctx.upload_frames(0, np.stack([g["ref"], g["tgt"]]))
ctx.build_pyramids(0, 2); ctx.apply_gradient(0, 2)
p = ctx.params
pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
row = 0
for lvl in range(p.first_level, p.last_level - 1, -1):
    for k in range(p.max_iters):
        if row >= len(g["trace_level"]) or g["trace_level"][row] != lvl:
            break
        out = ctx.residual_jacobian(0, 1, lvl, pose, dump=False)
        A = out["A"].astype(np.float32); b = (-(p.gain * out["jtr"])).astype(np.float32)
        gA, gb = g["trace_A"][row], g["trace_b"][row]
        sc = np.sqrt(np.outer(np.diag(gA), np.diag(gA))) + 1e-30
        d, Ai, ok = ctx.solve_delta(A, b)
        d_g, _, _ = ctx.solve_delta(gA, gb)   # GPU solve on the golden A,b
        print("lvl %d k %d nv %d/%d sr2 %d/%d  relA %.2e relb %.2e  |delta-gd| %.2e |solve(goldenAb)-gd| %.2e |delta| %.2e" % (
            lvl, k, out["n_valid"], g["trace_n_valid"][row], out["sum_r2"], g["trace_sum_r2"][row],
            np.abs((A - gA) / sc).max(), np.abs(b - gb).max() / (np.abs(gb).max() + 1e-30),
            np.abs(d - g["trace_delta"][row]).max(), np.abs(d_g - g["trace_delta"][row]).max(), np.abs(d).max()))
        if g["trace_exited"][row]:
            row += 1
            break
        pose = ctx.se3_mul(pose, ctx.se3_exp(d))
        print("      pose diff q %.2e t %.2e" % (np.abs(pose[:4] - g["trace_pose"][row][:4]).max(), np.abs(pose[4:] - g["trace_pose"][row][4:]).max()))
        row += 1
    if lvl != 0:
        pose = ctx.se3_handoff(pose, p.handoff_scale_t)
