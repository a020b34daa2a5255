#!/usr/bin/env python3
"""bench.py — frame-pair alignments/s of the MI355X direct SE(3) tracking path (BASELINE.json metric).

One "step" = one pass of the whole hot path over one resident batch: image pyramids of every frame of the batch, depth
pyramids and gradients of the reference frames + the full coarse-to-fine Gauss-Newton alignment of every pair (levels 3..0
of a 4-level pyramid, 10 iterations per level, no early exit), poses written to HBM.

N > 1: one process per GPU, pairs sharded round-robin (pair i -> rank i mod N), one RCCL all_gather of the solved poses
per step, un-shuffled into global pair order on every rank.  Weak scaling by default (--pairs per GPU); --total-pairs T
fixes the total work instead (strong scaling, SURVEY.md §8e: 8192 pairs over 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --total-pairs 8192            # starts the 8 ranks itself; fails if fewer than 8 GPUs are visible
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 10 --warmup 3

Prints ONE JSON line on rank 0.
"""
import argparse
import faulthandler
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime deals a process's streams onto a few hardware queues in turn (4 by default).  A batch runs as two halves
# on two streams whose launches fill each other's gaps; in a process that holds more streams — an RCCL communicator's, in
# every multi-GPU rank — the two can be dealt the same queue, where they simply run one after the other: 12.57 -> 12.93 ms per
# step with a communicator present, 13.04 with 2 queues, 12.62 with a communicator and 8 queues.  Set before anything
# initialises the runtime (read once, at its start-up).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ALG_BYTES_PER_PIXEL_ITER = 10          # I_ref 1 + gx 2 + gy 2 + z 4 + I_tgt 1
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0                  # MI355X_MICROARCH.md §Chip-level parameters: measured copy ceiling
# Facts about the dominant kernel that cannot be measured from inside this process (counter passes, compiler resource usage),
# per arithmetic set, each naming its source file, stamped with the source id (uwt_source_id(): sha256 of sources + flags) of the
# libuwt_hip.so they were collected on (tools/make_profile_facts.py).  They are quoted only while the library this process
# loaded reports that id — hipcc's output is not byte-reproducible, so the binary's own hash would not survive a rebuild.
PROFILE_FACTS = os.path.join(ROOT, "profiles", "r06", "k_residual_facts.json")


def _quoted_facts(capi, arith):
    """(facts of this arithmetic set or {}, note).  Empty when the loaded library is not built from the sources the facts were collected on."""
    if not os.path.exists(PROFILE_FACTS):
        return {}, "no %s" % os.path.relpath(PROFILE_FACTS, ROOT)
    allf = json.load(open(PROFILE_FACTS))
    sid = capi.source_id()
    if allf.get("library_source_id") != sid:
        return {}, ("quoted counter / resource facts withheld: %s was collected on a libuwt_hip.so built from sources %s..., this run "
                    "loaded one built from %s... (re-run tools/collect_profiles.sh + tools/publish_profiles.sh)"
                    % (os.path.relpath(PROFILE_FACTS, ROOT), str(allf.get("library_source_id"))[:12], sid[:12]))
    return dict(allf.get("sets", {}).get(arith, {})), "facts collected on a library built from these sources (uwt_source_id %s...)" % sid[:12]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1024,
                    help="resident frame pairs per GPU (BASELINE config 4: 8192 pairs over 8 GPUs = 1024 per GPU); weak scaling")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="total pairs of the job, split round-robin over the GPUs (strong scaling); overrides --pairs")
    ap.add_argument("--unique", type=int, default=128, help="distinct synthetic pairs generated per rank (tiled to its shard); "
                    "every one of them is checked against the CPU oracle after the timed region")
    ap.add_argument("--generator", choices=["cpp", "numpy"], default="cpp",
                    help="cpp: the deterministic C++ generator shared with tools/uwt_bench (tools/uwt_gen.h through "
                         "tools/libuwt_gen.so; SURVEY.md §8d); numpy: uw-slam_amd/synth.py")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--acc", choices=["f64", "f32"], default="f64")
    ap.add_argument("--arith", choices=["opencv", "legacy"], default="opencv",
                    help="arithmetic set of the OpenCV steps (include/uwt.h uwt_arith): opencv = what OpenCV 3.x's generic gemm / "
                         "MatExpr / solve paths compute (default; which set a given reference build computes is unpinned: both are measured, arith_sets); legacy = rounds 1-3 (f32 FMA chains, inverse then multiply)")
    ap.add_argument("--no-depth", action="store_true")
    ap.add_argument("--intrinsics", default="",
                    help="fx,fy,cx,cy of level 0 (default: TUM-like square pixels, 525 * width / 640, principal point at the centre); "
                         "EUROC (calibration/calibrationEUROC.xml:16-21): 458.654,457.296,367.215,248.375")
    ap.add_argument("--weights", choices=["identity", "tukey", "huber"], default="identity",
                    help="robust weights (general path; identity is the reference's live setting)")
    ap.add_argument("--bilinear", action="store_true", help="bilinear sampler extension (general path)")
    ap.add_argument("--reference-schedule", action="store_true",
                    help="the reference's EstimatePose constants: 5 levels, iterate 4..1, <= 50 iterations, early exit "
                         "(src/Tracker.cpp:364-372); not the headline workload")
    ap.add_argument("--tuning", default="", help="launch-shape switches for A/B runs: comma-separated uwt_tuning fields, e.g. "
                    "typed_loads=0,split=1 (never changes results; include/uwt.h)")
    ap.add_argument("--dump-poses", default="", help="rank 0 writes the last step's poses of the whole job, global pair order, [total, 7] float32 (.npy)")
    ap.add_argument("--n1-value", type=float, default=0.0,
                    help="alignments/s of the same workload on ONE GPU (a line of this program at --gpus 1): a multi-GPU line then "
                         "carries efficiency_vs_n1 = value / (n_gpus x this) beside per_rank_ms")
    ap.add_argument("--cpu-pairs", type=int, default=96,
                    help="alignments timed on the CPU oracle, 1 thread (rank 0, N=1 only; 0 = skip); ~10 s at 640x480")
    ap.add_argument("--no-profile", action="store_true",
                    help="skip the event-bracketed step, the compute-only step and the secondary figures after the timed region")
    return ap.parse_args(argv)


def streaming_figure(capi, params, frames, depth, P, rounds, resident_poses):
    """Secondary figure (never `value`): the same alignments with every frame crossing PCIe first, overlapped — a context
    of its own with two slot ranges that alternate; the batch is uploaded from page-locked memory on the copy stream
    (uwt_upload_frames_async: reference frames with depth, target frames without — the tracker reads depth of reference
    frames only) into one range while the batch in the other range is aligned, results come back through
    uwt_track_batch_host_async.  Same pairs per launch as the timed run.  Returns a dict, poses checked bit for bit against
    the resident run."""
    h, w = frames.shape[1:]
    over = {k: getattr(params, k) for k in ("n_levels", "first_level", "last_level", "max_iters", "early_exit", "has_depth",
                                            "accumulate_f64", "weights", "sampler", "device", "arith")}
    ctx = capi.Context(capi.default_params(w, h, params.fx, params.fy, params.cx, params.cy, max_frames=4 * P, max_pairs=P, **over))
    g_ref = capi.pinned_empty((P, h, w), np.uint8); g_ref[:] = frames[0::2]
    g_tgt = capi.pinned_empty((P, h, w), np.uint8); g_tgt[:] = frames[1::2]
    d_ref = None
    if depth is not None:
        d_ref = capi.pinned_empty((P, h, w), np.uint16); d_ref[:] = depth[0::2]
    h_poses = [capi.pinned_empty((P, 7), np.float32) for _ in range(2)]
    nbytes = g_ref.nbytes + g_tgt.nbytes + (d_ref.nbytes if d_ref is not None else 0)

    def enqueue(c):
        base = c * 2 * P                                       # slot range of this batch: refs first, then targets
        ctx.upload_frames_async(base, g_ref, d_ref)
        ctx.upload_frames_async(base + P, g_tgt, None)
        ref = base + np.arange(P, dtype=np.int32)
        return ctx.track_batch_host_async(base, 2 * P, ref, ref + P, h_poses[c])

    # PCIe alone: the uploads of one batch, nothing else running
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.upload_frames_async(0, g_ref, d_ref)
        ctx.upload_frames_async(P, g_tgt, None)
    ctx.sync()
    pcie_gbs = 3 * nbytes / (time.perf_counter() - t0) / 1e9
    for c in (0, 1):                                           # warm-up round
        enqueue(c)
    ctx.sync()
    tickets = []
    t0 = time.perf_counter()
    for r in range(rounds):
        for c in (0, 1):
            tickets.append(enqueue(c))
            if len(tickets) > 2:
                ctx.wait_ticket(tickets[-3])                   # at most two batches queued behind the running one
    ctx.sync()
    dt = time.perf_counter() - t0
    same = min(int(sum(np.array_equal(hp[i].view(np.uint32), resident_poses[i].view(np.uint32)) for i in range(P))) for hp in h_poses)
    rate = rounds * 2 * P / dt
    bound = pcie_gbs * 1e9 / (nbytes / P)
    ctx.close()
    return {"value": round(rate, 2), "unit": "alignments/s per GPU", "pcie_GBs": round(pcie_gbs, 2),
            "bytes_per_pair": int(nbytes / P), "pcie_bound_alignments_per_s": round(bound, 1),
            "frac_of_pcie_bound": round(rate / bound, 4), "pairs_per_launch": P,
            "poses_bit_identical_to_resident": same, "pairs_checked": P,
            "batches_timed": 2 * rounds,
            "note": "page-locked host memory, H2D on the copy stream overlapped with the alignment of the previous batch; "
                    "reference frames cross with their depth plane, target frames without; the timed span includes the "
                    "pipeline's fill (first upload) and drain (last alignment)"}


def latency_figure(capi, params, frames, depth, resident_pose, reps=200):
    """Secondary figure (never `value`): the drop-in use, one pair per synchronous call (Tracker::EstimatePose on a frame
    pair whose pyramids and gradients are in place, src/System.cpp:193-223) — the bench's own schedule, and the reference's
    (5 levels, iterate 4..1, early exit).  A context of its own; pose of the bench schedule checked against the batch's."""
    h, w = frames.shape[1:]
    out = {"unit": "ms per alignment, one pair per call", "calls_timed": reps}
    keep = {k: getattr(params, k) for k in ("n_levels", "first_level", "last_level", "max_iters", "early_exit", "has_depth",
                                            "accumulate_f64", "device", "arith")}
    for name in ("bench_schedule", "reference_schedule"):
        over = dict(keep) if name == "bench_schedule" else dict(has_depth=keep["has_depth"], accumulate_f64=keep["accumulate_f64"], device=keep["device"], arith=keep["arith"])
        if name == "reference_schedule" and min(w, h) < 16:    # five levels: the coarsest grid is size >> 4
            continue
        ctx = capi.Context(capi.default_params(w, h, params.fx, params.fy, params.cx, params.cy, max_frames=2, max_pairs=1, **over))
        ctx.upload_frames(0, frames[0:2], depth[0:2] if depth is not None else None)
        ctx.build_pyramids(0, 2)
        ctx.apply_gradient(0, 2)
        for _ in range(10):
            poses, stats = ctx.estimate_pose_batch([0], [1])
        t0 = time.perf_counter()
        for _ in range(reps):
            poses, stats = ctx.estimate_pose_batch([0], [1])
        out[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        out[name + "_evaluations"] = int(stats[0]["iterations"])
        if name == "bench_schedule" and resident_pose is not None:
            out["pose_bit_identical_to_batch"] = bool(np.array_equal(poses[0].view(np.uint32), resident_pose.view(np.uint32)))
        ctx.close()
    return out


def other_arith_figure(capi, params, pair_block, Pa, steps, warmup, dev):
    """alignments/s of the same step (pyramids, gradients, alignment of Pa resident pairs) under the arithmetic set the timed
    run did NOT use; a context of its own, inputs resident before the clock starts."""
    import torch
    h, w = params.height, params.width
    over = {k: getattr(params, k) for k in ("n_levels", "first_level", "last_level", "max_iters", "early_exit", "has_depth",
                                            "accumulate_f64", "weights", "sampler", "device")}
    over["arith"] = 1 - params.arith
    ctx = capi.Context(capi.default_params(w, h, params.fx, params.fy, params.cx, params.cy, max_frames=2 * Pa, max_pairs=Pa, **over))
    for i0 in range(0, Pa, 128):
        fr, dp = pair_block(i0, min(Pa, i0 + 128))
        ctx.upload_frames(2 * i0, fr, dp)
    buf = torch.empty((Pa, 7), dtype=torch.float32, device=dev)
    ref = np.arange(Pa, dtype=np.int32) * 2
    for _ in range(warmup):
        ctx.track_batch_async(0, 2 * Pa, ref, ref + 1, buf.data_ptr())
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.track_batch_async(0, 2 * Pa, ref, ref + 1, buf.data_ptr())
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.close()
    return {"arith": "legacy" if over["arith"] == 1 else "opencv", "value": round(Pa * steps / dt, 2), "unit": "alignments/s",
            "ms_per_step": round(1e3 * dt / steps, 4), "pairs": Pa, "steps": steps}


def main(args, standin=None):
    """One rank of the job.  `standin` (tests only: tests/test_bench_ranks_cpu.py) replaces the HIP library's binding by an object
    with the same Context surface that fills the pose buffers from the uploaded frames on the CPU: the rank body — sharding,
    uploads in blocks, the step loop through GatherPipeline, the barrier, the MAX-reduce of the step time, the JSON line — then
    runs over gloo on a machine without a GPU.  Never set by this program itself: without it the HIP library is required."""
    # A stalled run ends with every thread's Python traceback instead of sitting there until the caller's clock runs out.
    faulthandler.dump_traceback_later(int(os.environ.get("UWT_BENCH_WATCHDOG_S", "1500")), exit=True)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if standin is None:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d needs device %d, %d visible" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    use_dist = world > 1 or "RANK" in os.environ          # under a launcher the RCCL path runs even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if standin is None:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    capi = standin if standin is not None else importlib.import_module("uw-slam_amd.capi")
    synth = importlib.import_module("uw-slam_amd.synth")
    distm = importlib.import_module("uw-slam_amd.dist")

    w, h = args.width, args.height
    strong = args.total_pairs > 0
    total = args.total_pairs if strong else args.pairs * world
    if total < world:   # (decided from the arguments alone: every rank leaves here, none is left waiting in a collective)
        raise SystemExit("bench.py: %d pairs over %d GPUs: a rank would own no pair" % (total, world))
    my_pairs = distm.shard_round_robin(total, world, rank)     # global pair ids owned by this rank: i mod N == rank
    P = len(my_pairs)
    if P < 1:
        raise SystemExit("bench.py: rank %d owns no pair (%d pairs over %d GPUs)" % (rank, total, world))
    f = 525.0 * w / 640.0                                  # TUM-like intrinsics (calibrationTUM.xml:18-22), scaled
    intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
    if args.intrinsics:
        intr = tuple(float(v) for v in args.intrinsics.split(","))
        if len(intr) != 4:
            raise SystemExit("bench.py: --intrinsics takes fx,fy,cx,cy")
    has_depth = 0 if args.no_depth else 1
    over = dict(n_levels=args.levels, first_level=args.levels - 1, last_level=0, max_iters=args.iters, early_exit=0,
                has_depth=has_depth, accumulate_f64=1 if args.acc == "f64" else 0,
                weights={"identity": 0, "tukey": 1, "huber": 2}[args.weights], sampler=int(args.bilinear),
                arith={"opencv": 0, "legacy": 1}[args.arith])
    if args.reference_schedule:
        over.update(n_levels=5, first_level=4, last_level=1, max_iters=50, early_exit=1)
    params = capi.default_params(w, h, *intr, max_frames=2 * P, max_pairs=P, device=local_rank, **over)
    # (before the context: nothing may start a child process once the GPU is initialised — under rocprofv3 a child would
    # inherit the profiler's preloaded library; _cpp_generator builds the generator, if it has to, with that environment stripped)
    gen = _cpp_generator() if args.generator == "cpp" else None
    # under a profiler, per-kernel statistics are to describe whole-batch launches, one at a time: the two-halves-on-two-streams
    # form of a batch overlaps launches of half the size, whose durations a trace cannot tell apart from waiting
    tuning = dict(split=1) if _under_profiler() else {}
    for kv in filter(None, args.tuning.split(",")):
        if "=" not in kv:
            raise SystemExit("bench.py: --tuning takes field=value items separated by commas; got %r" % kv)
        k, v = kv.split("=", 1)
        if k.strip() not in [f[0] for f in capi.Tuning._fields_ if f[0] != "reserved"]:
            raise SystemExit("bench.py: --tuning: no uwt_tuning field %r (include/uwt.h: %s)"
                             % (k.strip(), ", ".join(f[0] for f in capi.Tuning._fields_ if f[0] != "reserved")))
        tuning[k.strip()] = int(v)
    ctx = capi.Context(params, tuning=tuning or None)

    U = min(args.unique, P)
    refs, tgts, deps = [], [], []
    for u in range(U):
        gid = int(my_pairs[u])
        if gen is not None:                                 # same pair ids -> same inputs as tools/uwt_bench
            ref, tgt, dep = gen(w, h, intr, gid, bool(has_depth))
        else:
            ref, tgt, dep, _, _ = synth.render_pair(w, h, *intr, seed=gid, z=0.8 + 0.4 * ((gid * 7) % 11) / 10.0,
                                                    with_depth=bool(has_depth))
        refs.append(ref); tgts.append(tgt); deps.append(dep)
    ref_stack, tgt_stack = np.stack(refs), np.stack(tgts)
    dep_stack = np.stack(deps) if has_depth else None

    def pair_block(i0, i1):
        """frames (and depth) of resident pairs [i0, i1): pair i repeats distinct pair i mod U; slots 2i (reference), 2i + 1 (target)"""
        ix = np.arange(i0, i1) % U
        fr = np.empty((2 * (i1 - i0), h, w), np.uint8)
        fr[0::2] = ref_stack[ix]
        fr[1::2] = tgt_stack[ix]
        dp = None
        if has_depth:
            dp = np.empty((2 * (i1 - i0), h, w), np.uint16)
            dp[0::2] = dep_stack[ix]
            dp[1::2] = dp[0::2]
        return fr, dp

    # inputs resident in HBM before the timed region.  Uploaded in blocks of CHUNK pairs built from the distinct ones: the host
    # never holds the whole shard (8192 pairs of 640x480 with depth are 15 GB), only the block in flight
    CHUNK = 128
    upload_s, upload_bytes = 0.0, 0
    for i0 in range(0, P, CHUNK):
        fr, dp = pair_block(i0, min(P, i0 + CHUNK))
        t_up = time.perf_counter()
        ctx.upload_frames(2 * i0, fr, dp)
        upload_s += time.perf_counter() - t_up             # blocking copies from pageable numpy memory (secondary figure)
        upload_bytes += fr.nbytes + (dp.nbytes if dp is not None else 0)
    del fr, dp
    ref_slots = np.arange(P, dtype=np.int32) * 2
    tgt_slots = ref_slots + 1

    # torch sees the context's HIP stream as an external stream.  The RCCL gather and the un-shuffle into global pair order
    # run on a stream of their own behind an event recorded after the alignment (no host synchronisation per step), so the
    # next step's kernels do not queue behind the collective: pose and gather buffers alternate between two sets, and a
    # step waits (on the device) for the gather that last read the set it is about to overwrite.  (Measured with a world of one:
    # the gather itself costs a step 0.07-0.1 ms; what cost 0.3 ms more was the communicator's streams crowding the hardware
    # queues, see GPU_MAX_HW_QUEUES at the top.)
    pipe = distm.GatherPipeline(total, P, dev, ctx.stream()) if use_dist else None
    single_buf = None if use_dist else torch.empty((P, 7), dtype=torch.float32, device=dev)

    def step():
        if use_dist:
            pipe.step(lambda buf: ctx.track_batch_async(0, 2 * P, ref_slots, tgt_slots, buf.data_ptr()))
        else:
            ctx.track_batch_async(0, 2 * P, ref_slots, tgt_slots, single_buf.data_ptr())

    def fence():
        ctx.sync()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            if dev.type == "cuda":
                torch.cuda.synchronize()

    t_w = time.perf_counter()
    for _ in range(args.warmup):
        step()
    fence()
    warm_s = time.perf_counter() - t_w
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    per_rank_ms = [1e3 * dt / args.steps]
    if use_dist:
        # every rank's own time for the K steps (barrier to barrier), then the job's: the slowest rank's
        mine_t = torch.tensor([dt], dtype=torch.float64, device=dev)
        all_t = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(all_t, mine_t)
        per_rank_ms = [round(1e3 * float(v) / args.steps, 4) for v in all_t.cpu().tolist()]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gpu_poses = (pipe.last_local() if use_dist else single_buf).cpu().numpy()
    all_poses = pipe.gathered.cpu().numpy() if use_dist else gpu_poses

    # ---- after the timed region: one step with HIP events around every residual launch (each event pair drains the
    # stream, so it is kept out of `value`), then one step of the kernel's compute-only twin (its instruction-issue
    # floor).  Each is preceded by untimed steps of the same kind: the chip holds a lower clock for the first
    # milliseconds after the pause of the copy above, and the durations should be those of the sustained state.
    res_ms = res_launches = res_pixels = 0
    res_levels = []
    co_ms = 0.0
    clock_ghz = co_clock_ghz = 0.0
    t_p = time.perf_counter()
    if not args.no_profile:
        for _ in range(max(2, args.warmup)):
            step()
        ctx.profile_enable(1)
        step()
        fence()
        res_ms, res_launches, res_pixels = ctx.profile_read()
        res_levels = ctx.profile_read_levels()
        clock_ghz = ctx.profile_clock()
        if not (args.bilinear or args.weights != "identity" or args.acc != "f64"):
            ctx.profile_enable(2)                           # compute-only, no events: re-heat in this mode
            for _ in range(max(2, args.warmup)):
                step()
            ctx.profile_enable(3)                           # compute-only + events
            step()                                          # poses of these steps are meaningless by construction
            fence()
            co_ms = ctx.profile_read()[0]
            co_clock_ghz = ctx.profile_clock()              # the clock the chip holds under the twin (no memory traffic)
        ctx.profile_enable(0)
        step()                                              # leave the real poses behind
        fence()
    prof_s = time.perf_counter() - t_p
    streaming = None
    # (not under a profiler: its launches carry the residual kernel's name and, running beside the copies, would blur the
    # per-kernel statistics that are compared with roofline.avg_launch_ms)
    Ps = min(P, 1024)                                       # pairs of the secondary legs (their own contexts hold 4 Ps and 2 Ps slots)
    if not args.no_profile and world == 1 and not args.reference_schedule and not _under_profiler():
        fr, dp = pair_block(0, Ps)
        streaming = streaming_figure(capi, params, fr, dp, Ps, max(3, args.steps // 2), gpu_poses[:Ps])   # 2 batches per round
        del fr, dp
    latency = None
    if not args.no_profile and world == 1 and not args.bilinear and args.weights == "identity" and not _under_profiler():
        fr, dp = pair_block(0, 1)
        latency = latency_figure(capi, params, fr, dp, None if args.reference_schedule else gpu_poses[0])
    # The same workload under the OTHER arithmetic set, measured in this run on a context of its own (never `value`): the default
    # set (opencv: what OpenCV's generic code paths compute) costs ~19 % of the throughput the legacy set (f32 FMA chains) has,
    # and which of the two a given reference build computes is unpinned — so both are reported, side by side.
    other = None
    if not args.no_profile and world == 1 and not _under_profiler():
        other = other_arith_figure(capi, params, pair_block, Ps, args.steps, args.warmup, dev)

    value = total * args.steps / dt
    px_per_align = sum((w >> l) * (h >> l) for l in range(args.levels))

    out = {
        "metric": "frame-pair alignments/sec (%dx%d, %d pyr lvls)" % (w, h, args.levels),
        "value": round(value, 2),
        "unit": "alignments/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        # where this process kept the GPU busy (seconds of wall time): a utilisation sampler with a period of seconds sees
        # little of a run whose GPU work is a few tenths of a second between CPU legs (generation, the CPU baseline)
        "gpu_active_s": {"warmup_steps": round(warm_s, 3), "timed_steps": round(dt, 3),
                         "event_bracketed_and_compute_only_steps": round(prof_s, 3)},
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (%s)" % ("deterministic C++ generator tools/uwt_gen.h, shared with tools/uwt_bench" if args.generator == "cpp" else "uw-slam_amd/synth.py"),
        "config": {
            "workload": ("synthetic %dx%d pairs, %s, "
                         "dense points%s, %d pairs in total, %d resident on rank 0 (%d distinct), image pyramids of both frames + depth pyramid and gradients of the reference frame + alignment per step"
                         % (w, h, "reference schedule: 5 pyramid levels, iterate 4..1, <= 50 iterations, early exit"
                            if args.reference_schedule else
                            "%d pyramid levels (0..%d), %d GN iterations/level, no early exit" % (args.levels, args.levels - 1, args.iters),
                            ", u16 depth plane" if has_depth else ", z=1", total, P, U)),
            "arithmetic": ("opencv: cv::gemm products (4-term rigid, 2-term Jacobian row, N-long normal equations) accumulated in f64 and "
                           "rounded once, MatExpr-folded unprojection, A.inv()*b as cv::solve (LU on the right-hand side)"
                           if args.arith == "opencv" else
                           "legacy (rounds 1-3): 4-/2-term products as f32 FMA chains, (x-cx)*invfx as written, inverse then f64-accumulated product"),
            "normal_equation_accumulation": args.acc, "weights": args.weights, "tuning": tuning or "defaults",
            "sampler": "bilinear" if args.bilinear else "nearest", "total_pairs": total, "pairs_on_rank0": P,
            "sharding": "round-robin pairs, RCCL all_gather of poses, global order on every rank" if use_dist else "single GPU",
        },
    }
    if rank == 0:
        if args.dump_poses:
            np.save(args.dump_poses, all_poses)
        if use_dist:
            # every rank's block must have arrived in global pair order: rank 0's own pairs sit at i = rank + k * world,
            # and pairs that repeat a distinct input (same seed modulo the tiling) must carry identical poses
            assert all_poses.shape == (total, 7)
            assert np.array_equal(all_poses[rank::world], gpu_poses), "gathered poses of rank 0 differ from its own"
            assert np.isfinite(all_poses).all() and (np.abs(np.linalg.norm(all_poses[:, :4], axis=1) - 1.0) < 1e-3).all()
            out["config"]["gather"] = {"collective_ran": bool(pipe.collective), "world": world,
                                       "rank0_block_bitwise_equal_to_its_own_poses": True}
            # what the driver's command measures: `--gpus N` with the default --pairs keeps 1024 pairs PER RANK (weak scaling:
            # 8 ranks = BASELINE config 4's 8192 pairs); --total-pairs T fixes the job instead (strong)
            out["per_rank_ms"] = per_rank_ms
            if args.n1_value > 0:
                out["efficiency_vs_n1"] = round(value / (world * args.n1_value), 4)
                out["n1_value"] = args.n1_value
        # SURVEY §8(d) secondary figure: the same step with the batch's frames crossing PCIe first (never `value`)
        out["h2d_inclusive"] = {"value": round(P / (upload_s + dt / args.steps), 2), "unit": "alignments/s per GPU",
                                "upload_ms": round(upload_s * 1e3, 2), "upload_GBs": round(upload_bytes / upload_s / 1e9, 2),
                                "note": "one blocking upload of the rank's resident batch from pageable host memory + one step"}
        if streaming:
            out["streaming"] = streaming
        if latency:
            out["single_pair_latency"] = latency
        if other:
            mine = {"arith": args.arith, "value": round(value, 2), "unit": "alignments/s", "ms_per_step": round(1e3 * dt / args.steps, 4),
                    "pairs": P, "steps": args.steps}
            out["arith_sets"] = {args.arith: mine, other["arith"]: other,
                                 "note": "the same step under both arithmetic sets of the OpenCV expressions on the path (include/uwt.h "
                                         "uwt_arith), measured in this run; `value` is the set named in config.arithmetic"}
        if res_launches:
            facts, facts_note = _quoted_facts(capi, args.arith)
            alg_bytes = ALG_BYTES_PER_PIXEL_ITER * res_pixels
            achieved = alg_bytes / (res_ms * 1e-3) / 1e9
            default_shape = (w, h, args.levels, has_depth) == (640, 480, 4, 1) and not args.reference_schedule
            traffic_px = facts.get("hbm_bytes_per_pixel_iteration") if default_shape else None
            roof = {
                # `bound` names the roofline `achieved` / `peak` / `frac` are priced against (HBM bandwidth, the
                # algorithmic bytes); what actually limits the kernel is vector-instruction issue: `limited_by`,
                # and `valu` says how close the kernel runs to that unit's peak.
                "bound": "hbm", "limited_by": "valu", "kernel": "k_residual (fused warp+residual+Jacobian+reduction)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 4),
                # the same launches on the bytes the PMC counters see (u16 depth instead of the f32 z the 10 B assume)
                "frac_counter_bytes": round(achieved * traffic_px / ALG_BYTES_PER_PIXEL_ITER / HBM_PEAK_GBS, 4) if traffic_px else None,
                "counter_bytes_per_pixel_iteration": traffic_px,
                "per_level": [{"level": l, "evaluations": int(n), "avg_ms_per_evaluation": round(ms / n, 5),
                               "algorithmic_GBs": round(ALG_BYTES_PER_PIXEL_ITER * P * (w >> l) * (h >> l) / (ms / n * 1e-3) / 1e9, 1)}
                              for l, (ms, n) in enumerate(res_levels) if n],
                "traffic": int(traffic_px * res_pixels / res_launches) if traffic_px else None,
                "traffic_source": facts.get("hbm_bytes_source") if traffic_px else None,
                "launches": int(res_launches), "avg_launch_ms": round(res_ms / res_launches, 5),
                "algorithmic_bytes_per_launch_avg": int(alg_bytes / res_launches),
                "launch_form": "whole-batch launches, one at a time, a k_gn_update launch behind each (the profiled step after the "
                               "timed region; the timed steps run the batch as two halves on two streams, whose overlapping launches "
                               "have no duration of their own, with the update in the tail of the evaluation's own launch)",
                "whole_job_effective_GBs": round(value / world * (ALG_BYTES_PER_PIXEL_ITER * px_per_align * args.iters
                                                                  + 5 * px_per_align + 2.5 * px_per_align) / 1e9, 1),
            }
            valu = {"shader_clock_GHz": round(clock_ghz, 3),
                    "clock_source": "s_memtime / s_memrealtime deltas written by the blocks of the profiled launches"}
            if co_ms:
                # same launches, same instruction stream, no memory operation: what the VALU alone takes
                valu["compute_only_avg_launch_ms"] = round(co_ms / res_launches, 5)
                valu["valu_issue_frac"] = round(co_ms / res_ms, 4)
                if co_clock_ghz:
                    # the twin moves no data: if the chip clocks higher under it, part of valu_issue_frac is clock, not stalls
                    valu["compute_only_shader_clock_GHz"] = round(co_clock_ghz, 3)
                    if clock_ghz:
                        valu["valu_issue_frac_in_cycles"] = round(co_ms * co_clock_ghz / (res_ms * clock_ghz), 4)
                valu["note"] = ("valu_issue_frac = duration of the kernel's compute-only twin (every load of the loop replaced "
                                "by register arithmetic) / duration of the kernel, measured back to back in this run")
            if clock_ghz:
                # SIMD cycles the chip spent per pixel-iteration: 1024 SIMDs x 64 lanes
                valu["simd_cycles_per_pixel"] = round(res_ms * 1e-3 * clock_ghz * 1e9 * 1024 * 64 / res_pixels, 1)
                vipp = facts.get("valu_instructions_per_pixel") if default_shape and args.weights == "identity" and not args.bilinear \
                    and args.acc == "f64" else None
                if vipp:
                    # the bound the line names, as a fraction of its peak: vector instructions the kernel retires per second over
                    # what the chip can issue — 1024 SIMDs x 16 lanes per cycle (a 64-lane instruction occupies its SIMD for four
                    # cycles) at the clock the chip held under this kernel
                    valu["frac_of_valu_peak"] = round(vipp * (res_pixels / (res_ms * 1e-3)) / (1024 * 16 * clock_ghz * 1e9), 4)
                    valu["valu_peak_lane_instructions_per_s"] = round(1024 * 16 * clock_ghz * 1e9, 0)
            # counter and compiler facts of the production instantiation, only for the workload they were collected on
            for k in ("valu_instructions_per_pixel", "f64_fma_per_pixel", "instruction_mix_source", "vgprs", "waves_per_simd",
                      "lds_bytes_per_block", "resource_source"):
                valu[k] = facts.get(k) if default_shape and args.weights == "identity" and not args.bilinear and args.acc == "f64" else None
            roof["facts_note"] = facts_note
            roof["valu"] = valu
            if args.reference_schedule:
                # early exit: a launch is counted (and its level's pixels with it) whether or not its pairs have already left
                # the level, so bytes / time overstates what the kernel moved; the durations stay, the fractions go
                for k in ("achieved", "frac", "frac_of_measured_copy_peak", "frac_counter_bytes"):
                    roof[k] = None
                for e in roof["per_level"]:
                    e["algorithmic_GBs"] = None
                roof["note"] = "early-exit schedule: launches whose pairs have left the level fall through; no meaningful bytes per launch"
            out["roofline"] = roof
        if world == 1 and args.cpu_pairs > 0:
            from oracle import oracle as O          # test infrastructure, used here only as the timed CPU baseline/checker
            po = O.default_params(w, h, *intr, **{k: v for k, v in over.items() if k != "accumulate_f64"})
            n_cpu = args.cpu_pairs                  # cycles over the U distinct pairs
            t0 = time.perf_counter()
            for k in range(n_cpu):
                u = k % U
                O.align_pair(po, refs[u], tgts[u], deps[u] if has_depth else None)
            t_cpu = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": round(n_cpu / t_cpu, 4), "unit": "alignments/s", "cores": 1, "kind": "port",
                "sample": "%d alignments over the batch's %d distinct pairs through oracle/uwt_oracle.c "
                          "(pyramid+gradients+EstimatePose), 1 thread, %.1f s" % (n_cpu, U, t_cpu),
            }
            # the same work spread over every host core (ctypes releases the GIL): pair-parallel thread pool
            from concurrent.futures import ThreadPoolExecutor
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = max(1, min(cores, 64))                  # a thread pool of ctypes calls stops scaling well before that
            def _one(k):
                u = k % U
                return O.align_pair(po, refs[u], tgts[u], deps[u] if has_depth else None)[1]
            n_mt = max(cores * 8, U)                        # covers every distinct pair: the parity leg below checks them all
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                mt_poses = list(ex.map(_one, range(n_mt)))
            t_mt = time.perf_counter() - t0
            cpu_poses = [mt_poses[u] for u in range(U)]
            dr, dtr, bit = [], [], 0
            for u in range(U):
                a, b = gpu_poses[u].astype(np.float64), cpu_poses[u].astype(np.float64)
                wv = abs(float(np.dot(a[:4], b[:4])))
                v = b[3] * a[:3] - a[3] * b[:3] - np.cross(a[:3], b[:3])
                dr.append(2.0 * np.arctan2(np.linalg.norm(v), wv))
                dtr.append(float(np.linalg.norm(a[4:] - b[4:])))
                bit += int(np.array_equal(gpu_poses[u].view(np.uint32), cpu_poses[u].view(np.uint32)))
            out["cpu_baseline_all_cores"] = {"value": round(n_mt / t_mt, 4), "unit": "alignments/s", "cores": cores,
                                             "kind": "port", "sample": "%d alignments, thread pool, %.1f s" % (n_mt, t_mt)}
            # every resident pair repeats one of the U distinct inputs: all P poses are covered by comparing each with its original
            tiled_same = int(sum(np.array_equal(gpu_poses[i].view(np.uint32), gpu_poses[i % U].view(np.uint32)) for i in range(P)))
            out["parity"] = {"pairs": len(cpu_poses), "max_rot_rad": float(np.max(dr)), "max_trans_m": float(np.max(dtr)),
                             "bit_identical_poses": bit, "tolerance": "1e-4 rad / 1e-4 m",
                             "resident_pairs_equal_to_their_distinct_original": tiled_same, "resident_pairs": P}
            assert tiled_same == P, "tiled copies of a pair differ within the batch"

        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def _cpp_generator():
    """tools/libuwt_gen.so (make -C tools; __graft_entry__.build() does it): the benchmarks' deterministic C++ input generator."""
    import ctypes as C
    path = os.path.join(ROOT, "tools", "libuwt_gen.so")
    if not os.path.exists(path):
        # normally built by __graft_entry__.build() / make -C tools.  Built here only from a process that has not touched the
        # GPU yet (run_rank calls this before it creates its context), and without the profiler's preload in the child.
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER"))}
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "libuwt_gen.so"], env=env)
    lib = C.CDLL(path)
    lib.uwt_gen_pair.restype = C.c_int
    lib.uwt_gen_pair.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(C.c_double)]

    def gen(w, h, intr, gid, with_depth):
        ref = np.empty((h, w), np.uint8); tgt = np.empty((h, w), np.uint8)
        dep = np.empty((h, w), np.uint16) if with_depth else None
        z = C.c_double()
        rc = lib.uwt_gen_pair(w, h, *[float(v) for v in intr], gid, ref.ctypes.data, tgt.ctypes.data,
                              dep.ctypes.data if dep is not None else None, C.byref(z))
        if rc:
            raise RuntimeError("uwt_gen_pair failed")
        return ref, tgt, dep
    return gen


def _under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(args):
    """The process the user starts never touches the GPU: it starts one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's JSON line and fails if any rank fails.  A single-GPU run gets
    a time limit and ONE retry (a box-level stall before any GPU work, seen once in ~300 runs, then costs a retry instead
    of the measurement); a multi-GPU job is not retried.  Not used under torchrun (RANK is set: the launcher owns the
    ranks) or under a profiler (the profiled process has to be the one doing the work)."""
    n = args.gpus
    if n < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if n > 1:
        import torch                                   # counting devices does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to run a smaller job under that name\n" % (n, have))
            sys.exit(2)
    limit = int(os.environ.get("UWT_BENCH_CHILD_TIMEOUT_S", "900"))
    for attempt in range(2 if n == 1 else 1):
        port = _free_port()
        procs = []
        for r in range(n):
            env = dict(os.environ, UWT_BENCH_CHILD="1")
            if n > 1:
                env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        deadline = time.time() + limit
        timed_out = False
        out0 = b""
        try:
            out0, _ = procs[0].communicate(timeout=limit)
            for p in procs[1:]:
                p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            timed_out = True
        if timed_out:
            for p in procs:
                if p.poll() is None:
                    p.kill()                           # the exact processes started above
            for p in procs:
                try:
                    p.communicate(timeout=10)
                except Exception:
                    pass
            sys.stderr.write("bench.py: attempt %d exceeded %d s, killed%s\n" % (attempt + 1, limit, "; retrying" if attempt == 0 and n == 1 else ""))
            continue
        sys.stdout.write(out0.decode())
        sys.stdout.flush()
        rc = max(abs(p.returncode or 0) for p in procs)
        sys.exit(0 if rc == 0 else 1)
    sys.exit(1)


if __name__ == "__main__":
    _args = parse_args()
    if os.environ.get("UWT_BENCH_CHILD") == "1" or "RANK" in os.environ or _under_profiler():
        main(_args)
    else:
        _launch(_args)
