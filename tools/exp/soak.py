#!/usr/bin/env python3
"""Determinism soak: one resident batch aligned S times through the whole per-frame path; every step's poses must equal the
first step's bit for bit (a race in the hand-written waits of the typed kernels, in the ticketed tail update or between the two
streams of a split batch would show as a difference), and the first pairs the oracle's.
usage: soak.py [--steps 300] [--pairs 1024] [--arith opencv|legacy] [--intrinsics fx,fy,cx,cy] [--weights huber] [--no-depth]"""
import argparse, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
capi = importlib.import_module("uw-slam_amd.capi")
from oracle import oracle as O
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300); ap.add_argument("--pairs", type=int, default=1024)
ap.add_argument("--arith", default="opencv"); ap.add_argument("--intrinsics", default=""); ap.add_argument("--weights", default="identity")
ap.add_argument("--no-depth", action="store_true")
a = ap.parse_args()
w, h, P, U = 640, 480, a.pairs, 32
intr = tuple(float(v) for v in a.intrinsics.split(",")) if a.intrinsics else (525.0, 525.0, 319.5, 239.5)
over = dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0, has_depth=0 if a.no_depth else 1,
            arith={"opencv": 0, "legacy": 1}[a.arith], weights={"identity": 0, "tukey": 1, "huber": 2}[a.weights])
gen = bench._cpp_generator()
trip = [gen(w, h, intr, g, not a.no_depth) for g in range(U)]
ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=2 * P, max_pairs=P, **over))
for i0 in range(0, P, 128):
    ix = np.arange(i0, min(P, i0 + 128)) % U
    fr = np.empty((2 * len(ix), h, w), np.uint8); fr[0::2] = np.stack([trip[i][0] for i in ix]); fr[1::2] = np.stack([trip[i][1] for i in ix])
    dp = None
    if not a.no_depth:
        dp = np.empty((2 * len(ix), h, w), np.uint16); dp[0::2] = np.stack([trip[i][2] for i in ix]); dp[1::2] = dp[0::2]
    ctx.upload_frames(2 * i0, fr, dp)
buf = [torch.zeros((P, 7), dtype=torch.float32, device="cuda") for _ in range(2)]
ref = np.arange(P, dtype=np.int32) * 2
first, bad = None, 0
for s in range(a.steps):
    ctx.track_batch_async(0, 2 * P, ref, ref + 1, buf[s & 1].data_ptr())
    if s:                                   # compare the previous step while this one runs
        ctx_prev = buf[(s - 1) & 1]
    ctx.sync()
    cur = buf[s & 1].cpu().numpy()
    if first is None:
        first = cur.copy()
    elif not np.array_equal(cur.view(np.uint32), first.view(np.uint32)):
        bad += 1
po = O.default_params(w, h, *intr, **{k: v for k, v in over.items()})
cpu_ok = sum(np.array_equal(first[u].view(np.uint32), O.align_pair(po, trip[u][0], trip[u][1], trip[u][2] if not a.no_depth else None)[1].view(np.uint32)) for u in range(8))
tiled = int(np.array_equal(first.view(np.uint32), first[np.arange(P) % U].view(np.uint32)))
print("soak %s %s %s%s: %d steps of %d pairs, %d steps differ from the first; tiled copies equal: %d; first 8 pairs equal to the oracle: %d"
      % (a.arith, a.weights, "fx!=fy " if intr[0] != intr[1] else "", "no depth" if a.no_depth else "depth", a.steps, P, bad, tiled, cpu_ok))
sys.exit(1 if bad or not tiled or cpu_ok != 8 else 0)
