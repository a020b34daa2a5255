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


class GatherPipeline:
    """The step loop of the sharded batched mode (bench.py): every step enqueues this rank's alignment into one of two pose
    buffers and, behind it, the exchange (PoseGatherer.gather) on a stream of its own — no host synchronisation per step:

      step k:  [wait: the gather that last read poses[k % 2] has finished]  ->  align(poses[k % 2]) on the context's stream
               -> event align_done  ->  on the gather stream: wait align_done, all_gather + un-shuffle into the gatherer's
               own output buffer, event gather_done

    so the next step's kernels never queue behind the collective, and a pose buffer is not overwritten while its gather may
    still read it.  On a CPU device (the gloo tests) there are no streams: the same calls run synchronously in this order.
    `align(buf)` must enqueue (CUDA) or perform (CPU) the alignment that fills `buf` ([n_local, 7]) in shard order."""

    N_BUF = 2

    def __init__(self, n_pairs, n_local, device, ctx_stream_handle=None, group=None, dtype=None):
        import torch
        self.torch = torch
        self.cuda = torch.device(device).type == "cuda"
        dtype = dtype or torch.float32
        self.poses = [torch.empty((n_local, 7), dtype=dtype, device=device) for _ in range(self.N_BUF)]
        self.gatherers = [PoseGatherer(n_pairs, device, dtype, group) for _ in range(self.N_BUF)]
        self.pending = [False] * self.N_BUF
        self.step_no = 0
        self.last = 0
        self.gathered = None
        if self.cuda:
            self.ctx_stream = torch.cuda.ExternalStream(ctx_stream_handle, device=device)
            self.gather_stream = torch.cuda.Stream(device=device)
            self.align_done = [torch.cuda.Event() for _ in range(self.N_BUF)]
            self.gather_done = [torch.cuda.Event() for _ in range(self.N_BUF)]
            self.consumed = [None] * self.N_BUF    # release(): a consumer's reads of gatherers[b]'s output have been enqueued up to here

    @property
    def collective(self):
        return self.gatherers[0].collective

    def step(self, align):
        b = self.step_no % self.N_BUF
        self.step_no += 1
        self.last = b
        if self.cuda:
            if self.pending[b]:
                self.ctx_stream.wait_event(self.gather_done[b])      # the gather of two steps ago has read poses[b]
            align(self.poses[b])
            self.align_done[b].record(self.ctx_stream)
            with self.torch.cuda.stream(self.gather_stream):
                self.gather_stream.wait_event(self.align_done[b])
                if self.consumed[b] is not None:   # a consumer on another stream may still be reading this gatherer's output
                    self.gather_stream.wait_event(self.consumed[b])
                    self.consumed[b] = None
                self.gathered = self.gatherers[b].gather(self.poses[b])   # RCCL all_gather over xGMI + permutation to global order
                self.gather_done[b].record(self.gather_stream)
        else:
            align(self.poses[b])
            self.gathered = self.gatherers[b].gather(self.poses[b])
        self.pending[b] = True
        return self.gathered      # valid on another stream only behind wait(); see there

    def last_local(self):
        return self.poses[self.last]

    def wait(self, stream=None):
        """Orders `stream` (default: torch's current stream) behind the exchange of the last step(): the tensor step() returned
        — a buffer of the gatherer, overwritten by the next step but one — may be read on that stream after this call, and only
        until the next-but-one step() (a consumer whose reads may still be queued by then says so with release()).  step() itself returns with the all_gather and the un-shuffle still in flight on the
        pipeline's private stream; without this (or a device synchronisation, as bench.py's fence) a consumer races them.
        No-op on a CPU device (the gloo form runs synchronously)."""
        if self.cuda and self.pending[self.last]:
            (stream or self.torch.cuda.current_stream()).wait_event(self.gather_done[self.last])
        return self.gathered

    def release(self, stream=None):
        """The other half of wait(): call it once the kernels that read the tensor wait() returned have been ENQUEUED on `stream`
        (default: torch's current stream).  It records how far that stream has come; the step that next writes the same gatherer
        buffer (the next but one) makes the pipeline's private stream wait for that point before its all_gather and un-shuffle
        overwrite the buffer.  Without it the "valid until the next-but-one step()" rule of wait() is host order only: a consumer
        kernel still queued on its stream when that step is enqueued could be overtaken.  bench.py needs neither (it fences the
        device before it reads); a consumer that pipelines steps against its own stream needs both.  No-op on a CPU device."""
        if self.cuda and self.pending[self.last]:
            ev = self.torch.cuda.Event()
            ev.record(stream or self.torch.cuda.current_stream())
            self.consumed[self.last] = ev
