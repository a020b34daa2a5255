"""Multi-GPU sharding of independent frame pairs: pair i -> rank i mod G, every rank runs the
single-GPU batched pipeline on its shard, then ONE all-gather of the float[7] poses (RCCL over xGMI on GPUs, gloo in
the CPU tests).  There is no other collective on the path: alignments never exchange data (every pair starts from
identity, src/Tracker.cpp:385)."""
import numpy as np


def shard_round_robin(n_pairs, world, rank):
    """Global ids of the pairs rank `rank` owns."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


def shard_sizes(n_pairs, world):
    return [len(range(r, n_pairs, world)) for r in range(world)]


def gather_poses(local_poses, n_pairs, group=None):
    """all_gather of per-rank pose blocks ([n_local, 7] float32 torch tensors) and un-shuffle into global pair order.
    Ranks may own different counts (n_pairs not divisible by world): blocks are padded to the largest shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_poses
    sizes = shard_sizes(n_pairs, world)
    m = max(sizes)
    pad = torch.zeros((m, 7), dtype=local_poses.dtype, device=local_poses.device)
    pad[: local_poses.shape[0]] = local_poses
    out = torch.empty((world * m, 7), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, m, 7)
    glob = torch.empty((n_pairs, 7), dtype=local_poses.dtype, device=local_poses.device)
    for r in range(world):
        glob[r::world] = out[r, : sizes[r]]
    return glob
