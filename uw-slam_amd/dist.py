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


class PoseGatherer:
    """The per-step exchange of a sharded batch, with everything it needs allocated once: a padded send block (ranks may
    own different counts when n_pairs is not divisible by the world size), the gathered rank-major buffer, and the
    permutation that turns rank-major into global pair order (global pair i = entry i // G of rank i mod G) — one
    all_gather_into_tensor and one index_select per call, both enqueued on the caller's current stream; no host loop,
    no allocation, no synchronisation."""

    def __init__(self, n_pairs, device, dtype=None, group=None, stage_send=False):
        """stage_send: always go through the padded send block (what ranks with uneven shards do), so that a world of one
        exercises that copy too."""
        import torch
        import torch.distributed as dist
        self.group = group
        self.collective = dist.is_initialized()   # a process group exists: the collective runs, whatever the world size
        self.n_pairs = int(n_pairs)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        dtype = dtype or torch.float32
        sizes = shard_sizes(self.n_pairs, self.world)
        self.n_local = sizes[self.rank]
        self.m = max(sizes) if sizes else 0
        self.even = all(s == self.m for s in sizes) and not stage_send
        self.send = None if self.even else torch.zeros((self.m, 7), dtype=dtype, device=device)
        self.gathered = torch.empty((self.world * self.m, 7), dtype=dtype, device=device)
        i = torch.arange(self.n_pairs, dtype=torch.int64)
        self.perm = ((i % self.world) * self.m + i // self.world).to(device)
        self.out = torch.empty((self.n_pairs, 7), dtype=dtype, device=device)

    def gather(self, local_poses):
        """local_poses: [n_local, 7] tensor of this rank's shard, in shard order.  Returns the [n_pairs, 7] tensor in
        global pair order (a buffer owned by the gatherer, overwritten by the next call)."""
        import torch
        import torch.distributed as dist
        if not self.collective:   # no process group (plain single-GPU run): nothing to exchange
            return local_poses
        # With a process group the exchange runs even for a world of one, so that a one-GPU box executes the very calls
        # (all_gather_into_tensor on the caller's stream, the un-shuffle) that N ranks do.
        block = local_poses
        if not self.even:
            self.send[: self.n_local].copy_(local_poses)
            block = self.send
        dist.all_gather_into_tensor(self.gathered, block, group=self.group)
        torch.index_select(self.gathered, 0, self.perm, out=self.out)
        return self.out


def gather_poses(local_poses, n_pairs, group=None):
    """One-off form of PoseGatherer.gather (allocates; a loop should keep a PoseGatherer)."""
    return PoseGatherer(n_pairs, local_poses.device, local_poses.dtype, group).gather(local_poses)
