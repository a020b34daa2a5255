"""bench.py's rank body at world 2 over gloo, on a machine without a GPU: the HIP library's binding is replaced by a stand-in with
the same Context surface whose "alignment" stamps every pose with a signature of the two frames it was given — so the sharding
(pair i -> rank i mod N), the uploads in blocks, the step loop through GatherPipeline (alternating buffers, one all_gather per
step, the un-shuffle into global pair order), the barrier and the MAX-reduce of the step time, and the JSON line's multi-GPU
fields (n_gpus, total_pairs, scaling, per_rank_ms, efficiency_vs_n1) are executed as the driver's
`torch.distributed.run ... bench.py --gpus N` executes them.  (The GPU path itself is covered by the -m gpu tests; what never ran
anywhere before this test is the `world > 1` branch of bench.py.)"""
import ctypes
import importlib.util
import json
import os
import socket
import sys
import types

ARITH_INDEPENDENT = True

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def signature(frame):
    return float(frame.astype(np.float64).mean()) / 255.0


def make_standin():
    """An object with the part of uw-slam_amd/capi.py bench.py's rank body touches."""
    mod = types.SimpleNamespace()

    class Tuning(ctypes.Structure):
        _fields_ = [("split", ctypes.c_int32), ("typed_loads", ctypes.c_int32), ("reserved", ctypes.c_int32 * 4)]

    class Context:
        def __init__(self, params, tuning=None):
            self.params, self.tuning, self.sig, self.steps = params, tuning, {}, 0

        def upload_frames(self, first_slot, gray, depth=None):
            assert gray.shape[1:] == (self.params.height, self.params.width) and first_slot + gray.shape[0] <= self.params.max_frames
            assert (depth is None) == (not self.params.has_depth)
            for i in range(gray.shape[0]):
                self.sig[first_slot + i] = signature(gray[i])

        def track_batch_async(self, first_slot, n_frames, ref_slots, tgt_slots, d_poses_ptr, d_stats_ptr=None, grad_refs_only=True):
            assert len(ref_slots) <= self.params.max_pairs and n_frames == 2 * len(ref_slots)
            poses = np.zeros((len(ref_slots), 7), np.float32)
            poses[:, 3] = 1.0
            for k, (a, b) in enumerate(zip(ref_slots, tgt_slots)):
                poses[k, 4:] = (self.sig[int(a)], self.sig[int(b)], float(k))
            ctypes.memmove(d_poses_ptr, poses.ctypes.data, poses.nbytes)
            self.steps += 1

        def stream(self):
            return 0

        def sync(self):
            pass

        def close(self):
            pass

    def default_params(width, height, fx, fy, cx, cy, **over):
        return types.SimpleNamespace(width=width, height=height, fx=fx, fy=fy, cx=cx, cy=cy, **over)

    mod.Tuning, mod.Context, mod.default_params = Tuning, Context, default_params
    return mod


def _load_bench():
    spec = importlib.util.spec_from_file_location("uwt_bench_py", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rank(rank, world, port, argv, out_path, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    bench = _load_bench()
    args = bench.parse_args(argv)
    if rank == 0:
        sys.stdout = open(out_path, "w")
    try:
        bench.main(args, standin=make_standin())
        q.put((rank, "ok"))
    except SystemExit as e:
        q.put((rank, "exit: %s" % (e,)))
    finally:
        if rank == 0:
            sys.stdout.close()


def _run(world, argv, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out_path = str(tmp_path / "rank0.json")
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, argv, out_path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res, out_path


@pytest.mark.parametrize("mode", ["weak", "strong_uneven"])
def test_bench_rank_body_world2_over_gloo(tmp_path, mode):
    world, w, h = 2, 64, 48
    poses_path = str(tmp_path / "poses.npy")
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "2", "--width", str(w), "--height", str(h), "--levels", "3", "--unique", "5",
            "--no-profile", "--cpu-pairs", "0", "--dump-poses", poses_path, "--n1-value", "100.0", "--generator", "numpy", "--tuning", "split=1"]
    argv += ["--pairs", "6"] if mode == "weak" else ["--total-pairs", "11"]
    total = 12 if mode == "weak" else 11
    res, out_path = _run(world, argv, tmp_path)
    assert res == {0: "ok", 1: "ok"}
    line = open(out_path).read().strip().splitlines()
    assert len(line) == 1                                              # ONE JSON line, from rank 0
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["unit"] == "alignments/s"
    assert d["scaling"] == ("weak" if mode == "weak" else "strong")
    assert d["config"]["total_pairs"] == total and d["config"]["pairs_on_rank0"] == (6 if mode == "weak" else 6)
    assert d["config"]["gather"] == {"collective_ran": True, "world": 2, "rank0_block_bitwise_equal_to_its_own_poses": True}
    assert len(d["per_rank_ms"]) == 2 and all(v > 0 for v in d["per_rank_ms"])
    # the job's time is the slowest rank's (MAX over ranks), the value the whole job's pairs over it
    assert abs(d["ms_per_step"] - max(d["per_rank_ms"])) <= 1e-3 * max(d["per_rank_ms"]) + 1e-3
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["value"]
    assert abs(d["efficiency_vs_n1"] - d["value"] / 200.0) < 1e-3 and d["n1_value"] == 100.0
    # the gathered poses, in GLOBAL pair order: pair i lives on rank i mod 2 as its local pair i // 2, and carries the signatures of
    # the frames generated for global id (rank's u-th distinct pair repeats: local pair k shows distinct pair k mod U of that rank)
    synth = importlib.import_module("uw-slam_amd.synth")
    f = 525.0 * w / 640.0
    intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
    poses = np.load(poses_path)
    assert poses.shape == (total, 7)
    for i in range(total):
        r, k = i % world, i // world
        n_local = len(range(r, total, world))
        U = min(5, n_local)
        gid = r + (k % U) * world                                      # the distinct pair local pair k repeats
        ref, tgt, _, _, _ = synth.render_pair(w, h, *intr, seed=gid, z=0.8 + 0.4 * ((gid * 7) % 11) / 10.0, with_depth=True)
        assert poses[i, 3] == 1.0 and poses[i, 6] == float(k), i
        assert poses[i, 4] == np.float32(signature(ref)) and poses[i, 5] == np.float32(signature(tgt)), i


def test_bench_rank_body_failure_paths(tmp_path):
    base = ["--steps", "1", "--warmup", "0", "--width", "64", "--height", "48", "--levels", "3", "--no-profile", "--cpu-pairs", "0", "--generator", "numpy"]
    # WORLD_SIZE does not match --gpus: every rank refuses before anything starts
    res, _ = _run(2, ["--gpus", "3"] + base, tmp_path)
    assert all(v.startswith("exit: bench.py: --gpus 3 but WORLD_SIZE 2") for v in res.values()), res
    # fewer pairs than ranks: every rank leaves (none waits in a collective for one that left)
    res, _ = _run(2, ["--gpus", "2", "--total-pairs", "1"] + base, tmp_path)
    assert all("a rank would own no pair" in v for v in res.values()), res
    # a bad --tuning item names the fields instead of crashing
    res, _ = _run(1, ["--gpus", "1", "--tuning", "residual_plane=0"] + base, tmp_path)
    assert "no uwt_tuning field 'residual_plane'" in res[0], res
    res, _ = _run(1, ["--gpus", "1", "--tuning", "split"] + base, tmp_path)
    assert "field=value" in res[0], res
