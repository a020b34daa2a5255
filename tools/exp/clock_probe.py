"""Diagnostic (lib_clock.so build only): in-kernel shader clock under the real level-0 load."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
import torch
w, h, P = 640, 480, 256
intr = (525.0, 525.0, 319.5, 239.5)
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * P, max_pairs=P, n_levels=1, first_level=0, last_level=0,
                                       max_iters=10, early_exit=0, has_depth=1))
ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=1, with_depth=True)
fr = np.empty((2 * P, h, w), np.uint8); fr[0::2] = ref; fr[1::2] = tgt
dp = np.empty((2 * P, h, w), np.uint16); dp[:] = dep
ctx.upload_frames(0, fr, dp)
poses = torch.empty((P, 7), dtype=torch.float32, device="cuda")
for _ in range(3):
    ctx.track_batch_async(0, 2 * P, np.arange(P) * 2, np.arange(P) * 2 + 1, poses.data_ptr())
ctx.sync()
out = (C.c_uint32 * 64)()
fs = []
for pair in (0, 50, 100, 200, 255):
    for sl in (0, 5, 18):
        capi.lib().uwt_debug_read_record(ctx._h, pair, sl, 0, out)
        d = np.frombuffer(out, np.uint64)
        clk, rt = int(d[30]), int(d[31])
        fs.append(clk / rt * 100e6 / 1e9)
        print("pair %3d slice %2d: %d shader clocks over %.2f us -> %.3f GHz" % (pair, sl, clk, rt / 100.0, fs[-1]))
print("median clock %.3f GHz" % np.median(fs))
