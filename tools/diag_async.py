import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
import torch
MID = (131.25, 131.25, 79.5, 47.5)
w, h, n = 160, 96, 80
over = dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0)
# dirty the device memory first
junk = torch.full((256 * 1024 * 1024,), 0x7f, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
frames = []
for s in range(n):
    ref, tgt, _, _, _ = synth.render_pair(w, h, *MID, seed=3000 + s)
    frames += [ref, tgt]
frames = np.stack(frames)
ref_s, tgt_s = np.arange(n) * 2, np.arange(n) * 2 + 1
p = O.default_params(w, h, *MID, **over)
cpu = np.stack([O.align_pair(p, frames[2 * i], frames[2 * i + 1])[1] for i in range(n)])
for trial in range(3):
    ctx = capi.Context(capi.default_params(w, h, *MID, max_frames=2 * n, max_pairs=n, **over))
    ctx.upload_frames(0, frames)
    d_poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    ctx.track_batch_async(0, 2 * n, ref_s, tgt_s, d_poses.data_ptr(), None, grad_refs_only=True)
    ctx.sync()
    a = d_poses.cpu().numpy()
    ctx2 = capi.Context(capi.default_params(w, h, *MID, max_frames=2 * n, max_pairs=n, **over))
    ctx2.upload_frames(0, frames); ctx2.build_pyramids(0, 2 * n); ctx2.apply_gradient(0, 2 * n)
    b, _ = ctx2.estimate_pose_batch(ref_s, tgt_s)
    bad_a = [i for i in range(n) if not np.array_equal(a[i], cpu[i])]
    bad_b = [i for i in range(n) if not np.array_equal(b[i], cpu[i])]
    print("trial", trial, "async!=cpu:", bad_a, " sync!=cpu:", bad_b)
    ctx.close(); ctx2.close()
