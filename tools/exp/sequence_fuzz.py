"""The drop-in's own usage, asynchronous: a recorded-like sequence tracked frame by frame through a RING of slots.  Per frame:
uwt_upload_frames_async(slot, frame k+1), then uwt_track_batch_host_async preparing ONLY that slot with the one pair
(previous slot -> reference, new slot -> target) and grad_refs_only = 1 — the reference's gradients and depth pyramid are
built in this call for a slot OUTSIDE the prepared range, its image pyramid comes from the call before — and nothing is
waited for until several frames later.  Ring lengths from 2 (every upload lands on the slot the call in flight reads as its
reference's predecessor) up; queue depths 1..4; both schedules; robust weights.  Every pose against the oracle's for that
pair of frames.  python tools/exp/sequence_fuzz.py [frames] [seed]"""
import importlib, os, sys, time, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
total = bad = 0
t0 = time.time()
pool = ThreadPoolExecutor(min(32, os.cpu_count() or 4))
CASES = [(160, 96, dict(n_levels=4, first_level=3, last_level=0, max_iters=6, early_exit=0)), (320, 240, dict()),
         (208, 112, dict(n_levels=5, first_level=4, last_level=1, max_iters=5, early_exit=1, weights=2)),
         (640, 480, dict(n_levels=4, first_level=3, last_level=0, max_iters=4, early_exit=0, weights=1)), (736, 480, dict()),
         # sizes no power of two divides (round 6): the uploads go through the staging area and the row-spreading kernel
         (365, 233, dict()), (183, 119, dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0, weights=2))]
for ci, (w, h, over) in enumerate(CASES):
    f = float(np.float32(0.8 * w))
    intr = (f, f if ci % 2 == 0 else float(np.float32(f * 0.997)), float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
    over = dict(over, has_depth=1)
    frames, depths, _, _ = synth.render_sequence(w, h, *intr, n_frames, seed * 10 + ci, with_depth=True, margin=(96, 64))
    po = O.default_params(w, h, *intr, **over)
    want = list(pool.map(lambda k: O.align_pair(po, frames[k], frames[k + 1], depths[k]), range(n_frames - 1)))
    for ring in (2, 3, 5, 8):
        ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=ring, max_pairs=1, **over))
        depth_q = int(rng.integers(1, 5))
        sets = [(capi.pinned_empty((1, h, w), np.uint8), capi.pinned_empty((1, h, w), np.uint16), capi.pinned_empty((1, 7), np.float32),
                 capi.pinned_empty((1, 4), np.int32)) for _ in range(6)]
        ctx.upload_frames(0, frames[0][None], depths[0][None]); ctx.build_pyramids(0, 1)   # the first frame (System::AddFrame)
        inflight = collections.deque()
        prev = 0
        for k in range(n_frames - 1):
            slot = (prev + 1) % ring
            pf, pd, hp, hs = sets[k % len(sets)]
            pf[0], pd[0] = frames[k + 1], depths[k + 1]
            hp[...] = np.nan
            ctx.upload_frames_async(slot, pf, pd)
            tk = ctx.track_batch_host_async(slot, 1, [prev], [slot], hp, hs, grad_refs_only=True)
            inflight.append((tk, k, hp, hs))
            prev = slot
            while len(inflight) > depth_q or (k == n_frames - 2 and inflight):
                tk, kk, hp_, hs_ = inflight.popleft()
                ctx.wait_ticket(tk)
                cs, cp, _ = want[kk]
                total += 1
                same = np.array_equal(hp_[0].view(np.uint32), cp.view(np.uint32)) or (np.isnan(hp_[0]).all() and np.isnan(cp).all())
                if int(hs_[0, 0]) != cs or not same:
                    bad += 1
                    if bad <= 10:
                        print("DIFFERS %dx%d ring %d queue %d frame %d: status %d/%d\n  got  %s\n  want %s" % (w, h, ring, depth_q, kk, int(hs_[0, 0]), cs, hp_[0], cp), flush=True)
            if rng.random() < 0.1:
                depth_q = int(rng.integers(1, 5))
        ctx.sync(); ctx.close()
    print("%dx%d %s: %d frame pairs so far, %d differ, %.0f s" % (w, h, {k: v for k, v in over.items() if k != "has_depth"} or "reference constants", total, bad, time.time() - t0), flush=True)
print("sequence fuzz seed %d: %d frame pairs through rings of 2..8 slots, %d differ from the oracle" % (seed, total, bad))
sys.exit(1 if bad else 0)
