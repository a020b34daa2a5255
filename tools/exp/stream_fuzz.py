"""Ordering fuzz of the streaming form: uwt_upload_frames_async into slot ranges whose previous contents may still be read by
tracker calls in flight, uwt_track_batch_host_async on them, results collected several calls later (uwt_wait_ticket) — the
pipeline of System::AddFrame + System::Tracking with nothing waited for in between.  Every call's poses must be the oracle's for
the frames that were uploaded to its slots immediately before it: an upload that overtakes the alignment still reading the
slots, or an alignment that starts before its upload has landed, shows up as a wrong pose.  Random slot ranges (aligned halves,
overlapping windows), batch sizes, queue depths, both schedules.  python tools/exp/stream_fuzz.py [steps] [seed]"""
import importlib, os, sys, time, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("uw-slam_amd.capi")
synth = importlib.import_module("uw-slam_amd.synth")
from oracle import oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
total = bad = 0
t0 = time.time()
for (w, h, nl, early, B) in [(160, 96, 4, 0, 48), (320, 240, 4, 1, 24), (208, 112, 5, 0, 32), (640, 480, 4, 0, 12),
                          (183, 119, 4, 0, 32), (365, 233, 5, 1, 16)]:   # pitched rows: uploads through the staging area (round 6)
    f = float(np.float32(0.8 * w))
    intr = (f, f, float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5)))
    over = dict(n_levels=nl, first_level=nl - 1, last_level=0 if not early else 1, max_iters=6, early_exit=early, has_depth=1)
    S = 4 * B                                   # frame slots: room for two batches of B pairs
    ctx = capi.Context(capi.default_params(w, h, *intr, max_frames=S, max_pairs=B, **over))
    po = O.default_params(w, h, *intr, **over)
    U = 10
    pairs = [synth.render_pair(w, h, *intr, seed=seed * 777 + 31 * nl + s, with_depth=True, max_t=0.004 + 0.002 * (s % 4), max_deg=0.2 + 0.15 * (s % 3)) for s in range(U)]
    want = [O.align_pair(po, p[0], p[1], p[2]) for p in pairs]
    inflight = collections.deque()
    pool = [(capi.pinned_empty((2 * B, h, w), np.uint8), capi.pinned_empty((2 * B, h, w), np.uint16), capi.pinned_empty((B, 7), np.float32),
             capi.pinned_empty((B, 4), np.int32)) for _ in range(6)]
    depth_q = int(rng.integers(1, 5))
    for step in range(steps):
        n = int(rng.choice([1, 2, B // 4, B // 2, B]))
        first = int(rng.integers(0, S - 2 * n + 1)) if rng.random() < 0.5 else int(rng.choice([0, 2 * B]))   # any window / the two halves
        first = min(first, S - 2 * n)
        ids = rng.integers(0, U, n)
        PF, PD, HP, HS = pool[step % len(pool)]          # (a set is reused only after its call has been waited for: the queue is shorter than the pool)
        pf, pd = PF[:2 * n], PD[:2 * n]
        for i, u in enumerate(ids):
            pf[2 * i], pf[2 * i + 1] = pairs[u][0], pairs[u][1]
            pd[2 * i] = pairs[u][2]; pd[2 * i + 1] = 0           # the tracker never reads the target frame's depth
        hp, hs = HP[:n], HS[:n]
        hp[...] = np.nan
        ctx.upload_frames_async(first, pf, pd)
        ref = first + 2 * np.arange(n)
        tk = ctx.track_batch_host_async(first, 2 * n, ref, ref + 1, hp, hs, grad_refs_only=bool(rng.random() < 0.7))
        inflight.append((tk, ids, hp, hs, pf, pd, first, step))
        while len(inflight) > depth_q or (step == steps - 1 and inflight):
            tk, ids_, hp_, hs_, _pf, _pd, first_, step_ = inflight.popleft()
            ctx.wait_ticket(tk)
            for i, u in enumerate(ids_):
                total += 1
                cs, cp, _ = want[u]
                if int(hs_[i, 0]) != cs or not np.array_equal(hp_[i].view(np.uint32), cp.view(np.uint32)):
                    bad += 1
                    if bad <= 10:
                        print("DIFFERS %dx%d step %d (first slot %d, %d pairs, queue depth %d) pair %d: status %d/%d\n  got  %s\n  want %s" % (
                            w, h, step_, first_, len(ids_), depth_q, i, int(hs_[i, 0]), cs, hp_[i], cp), flush=True)
        if rng.random() < 0.05:
            depth_q = int(rng.integers(1, 5))
    ctx.sync(); ctx.close()
    print("%dx%d x %d levels, %s schedule, up to %d pairs per call: %d alignments so far, %d differ, %.0f s" % (w, h, nl, "early-exit" if early else "fixed", B, total, bad, time.time() - t0), flush=True)
print("stream fuzz seed %d: %d alignments through the asynchronous pipeline, %d differ from the oracle" % (seed, total, bad))
sys.exit(1 if bad else 0)
