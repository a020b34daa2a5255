"""world_size-2 gloo test of the pair sharding + pose gather used by the multi-GPU batched mode (no GPU needed:
the per-rank 'solver' here is a stand-in that stamps each pose with its global pair id)."""
import importlib
import os
import socket

ARITH_INDEPENDENT = True   # nothing here depends on the arithmetic set (tests/conftest.py): run once

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("uw-slam_amd.dist")
    ids = d.shard_round_robin(n_pairs, world, rank)
    local = torch.zeros((len(ids), 7), dtype=torch.float32)
    for k, gid in enumerate(ids):
        local[k] = torch.arange(7, dtype=torch.float32) + 10.0 * gid
    glob = d.gather_poses(local, n_pairs)
    if rank == 0:
        q.put(glob.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 7, 1])
def test_round_robin_shard_and_gather_world2(n_pairs):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    glob = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = np.arange(7, dtype=np.float32)[None, :] + 10.0 * np.arange(n_pairs, dtype=np.float32)[:, None]
    assert np.array_equal(glob, expect)


def test_shard_partition_properties():
    d = importlib.import_module("uw-slam_amd.dist")
    for n, g in [(8192, 8), (10, 3), (5, 8)]:
        ids = [d.shard_round_robin(n, g, r) for r in range(g)]
        allid = np.sort(np.concatenate(ids))
        assert np.array_equal(allid, np.arange(n))
        assert d.shard_sizes(n, g) == [len(x) for x in ids]
        assert all((x % g == r).all() for r, x in enumerate(ids))


def _pipeline_worker(rank, world, port, n_pairs, steps, q):
    """bench.py's rank logic with a CPU stand-in for the context: shard -> `align` stamps every pose of the shard with its
    global pair id and the step number -> GatherPipeline (the double-buffered exchange bench.py runs per step)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("uw-slam_amd.dist")
    mine = d.shard_round_robin(n_pairs, world, rank)                 # bench.py: my_pairs
    pipe = d.GatherPipeline(n_pairs, len(mine), torch.device("cpu"))
    assert pipe.collective and pipe.gatherers[0].world == world
    seen = []
    bufs = set()
    for k in range(steps):
        def align(buf, k=k):
            bufs.add(buf.data_ptr())
            buf.copy_(torch.arange(7, dtype=torch.float32)[None, :] + 10.0 * torch.from_numpy(mine).to(torch.float32)[:, None] + 1000.0 * k)
        glob = pipe.step(align)
        assert pipe.wait() is glob                                   # (CPU: nothing to wait for; the handle is the same buffer)
        pipe.release()                                               # (CPU: nothing to order either)
        seen.append(glob.clone().numpy())
        assert np.array_equal(pipe.last_local().numpy()[:, 0], 10.0 * mine + 1000.0 * k)
    assert len(bufs) == min(2, steps)                                # the two pose buffers take turns
    dist.barrier()
    if rank == 0:
        q.put(np.stack(seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 7])
def test_bench_step_pipeline_world2_double_buffered_gather(n_pairs):
    """The step loop bench.py runs per rank (uw-slam_amd/dist.py GatherPipeline: two pose buffers, a gatherer each, the
    exchange behind the alignment) executed at world size 2 over gloo for five steps: every step's gathered block is that
    step's poses of ALL pairs in global pair order — even and uneven shards."""
    world, steps = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, n_pairs, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    for k in range(steps):
        expect = np.arange(7, dtype=np.float32)[None, :] + 10.0 * np.arange(n_pairs, dtype=np.float32)[:, None] + 1000.0 * k
        assert np.array_equal(seen[k], expect), k


@pytest.mark.parametrize("n_pairs", [8192, 8191])
def test_bench_step_pipeline_world8_config4_shards(n_pairs):
    """BASELINE config 4's exchange at its own size: 8 192 pairs (and 8 191: uneven shards, the padded send block) sharded
    round-robin over 8 ranks, the bench's double-buffered step loop for three steps over gloo — every step's gathered block is
    all pairs' poses of that step in global pair order."""
    world, steps = 8, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, n_pairs, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for k in range(steps):
        expect = np.arange(7, dtype=np.float32)[None, :] + 10.0 * np.arange(n_pairs, dtype=np.float32)[:, None] + 1000.0 * k
        assert np.array_equal(seen[k], expect), k
