"""Trajectory sharding of the rollout over the GPUs of one node.

The path has no cross-trajectory arithmetic (GroupNorm / InstanceNorm / LayerNorm / attention
are per sample), so rank r simply owns a contiguous slice of the batch and runs
the whole rollout on it: no collective in the data path.  When every rank wants the full
result, finished step-blocks are all-gathered (RCCL over xGMI on GPUs; gloo in CPU tests) on
a side stream while the remaining steps are still being computed.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int):
    """Contiguous split; the first (global_batch % world) ranks get one more trajectory."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_lengths(T: int, chunk: int):
    """Step-block schedule of the overlapped gather: long blocks first, `chunk`-step blocks last.  Every block
    boundary drains the engine's stream pipeline (~0.8 ms on NS2d-128, tools/chunk_cost.py) and the gather of the LAST
    block is the only exposed one, so the blocks halve (T/2, T/4, ...) down to `chunk`: 64 steps, chunk 8 ->
    [32, 16, 8, 8]."""
    chunk = max(1, min(chunk, T))
    lens, rem = [], T
    while rem >= 4 * chunk:
        take = (rem // 2) // chunk * chunk
        lens.append(take)
        rem -= take
    lens += [min(chunk, rem - i) for i in range(0, rem, chunk)]
    return lens


class ChunkedGatherRollout:
    """encode once, then rollout in step-blocks; block i is gathered while block i+1 runs.

    encode(x) -> z ; rollout_latent(z, steps, out) -> z_next  (out: [B, steps, C, H, W], written in place).
    All ranks must hold the same local batch size (all_gather_into_tensor)."""

    def __init__(self, encode, rollout_latent, frame_shape, B, T, chunk, device, group=None, gather=True):
        self.encode, self.rollout_latent = encode, rollout_latent
        self.group, self.gather = group, gather
        self.world = dist.get_world_size(group) if (gather and dist.is_initialized()) else 1
        self.lens = chunk_lengths(T, chunk)
        self.bufs = [torch.empty((B, n) + tuple(frame_shape), dtype=torch.float32, device=device) for n in self.lens]
        self.gathered = None
        self.comm_stream = None
        if self.gather and self.world > 1:
            # concatenated along dim 0 in rank order (the form every backend accepts)
            self.gathered = [torch.empty((self.world * b.shape[0],) + tuple(b.shape[1:]), dtype=torch.float32,
                                         device=device) for b in self.bufs]
            if torch.device(device).type == "cuda":
                self.comm_stream = torch.cuda.Stream(device=device)

    def run(self, x):
        z = self.encode(x)
        works = []
        for i, buf in enumerate(self.bufs):
            z = self.rollout_latent(z, buf.shape[1], buf)
            if self.gathered is None:
                continue
            if self.comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    works.append(dist.all_gather_into_tensor(self.gathered[i], buf, group=self.group, async_op=True))
            else:
                works.append(dist.all_gather_into_tensor(self.gathered[i], buf, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        return self.bufs

    def assemble(self):
        """[world*B, T, C, H, W] on every rank (concatenation in rank order), for checks."""
        if self.gathered is None:
            return torch.cat(self.bufs, dim=1)
        return torch.cat(self.gathered, dim=1)


class EndGatherRollout:
    """The north star's literal form: every rank runs its whole shard's rollout, then ONE all-gather of the decoded
    shards [B,T,C,H,W] into a preallocated [world*B,T,C,H,W] buffer (rank order along dim 0).  No collective and no
    host synchronisation inside the rollout; the gather is fully exposed (it is what `exposed_gather_ms` measures).

    rollout(x, out) writes out [B,T,C,H,W] in place."""

    def __init__(self, rollout, frame_shape, B, T, device, group=None, gather=True, collective_at_world1=False):
        """collective_at_world1: issue the all-gather even when the process group has ONE rank (`bench.py --force-dist`:
        the RCCL path -- communicator creation, the collective on the real shard size, the receive buffer -- executed
        on a one-GPU box; the result must equal the shard bit for bit)."""
        self.rollout, self.group = rollout, group
        self.world = dist.get_world_size(group) if (gather and dist.is_initialized()) else 1
        self.out = torch.empty((B, T) + tuple(frame_shape), dtype=torch.float32, device=device)
        self.full = None
        if self.world > 1 or (collective_at_world1 and gather and dist.is_initialized()):
            self.full = torch.empty((self.world * B, T) + tuple(frame_shape), dtype=torch.float32, device=device)
        self.on_gpu = torch.device(device).type == "cuda"
        self.bufs = [self.out]
        self.exposed_ms = []            # per run: time between the end of the rollout and the end of the gather

    def run(self, x):
        import time
        self.rollout(x, self.out)
        if self.full is None:
            return self.bufs
        if self.on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(self.full, self.out, group=self.group)
            e1.record()
            self._pending = (e0, e1)
        else:
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(self.full, self.out, group=self.group)
            self.exposed_ms.append((time.perf_counter() - t0) * 1e3)
        return self.bufs

    def finish_timing(self):
        """(GPU) resolve the events of the last run; call after a synchronise."""
        p = getattr(self, "_pending", None)
        if p is not None:
            self.exposed_ms.append(p[0].elapsed_time(p[1]))
            self._pending = None

    def assemble(self):
        return self.out if self.full is None else self.full


def gather_metrics(frame, seq, group=None):
    """Validation over sharded trajectories without moving the fields: each rank reduces its own decoded rollout to
    the frame-wise [b,T,C] and sequence-wise [b,C] errors (lns_amd.metrics.relative_l2, one pass over its shard) and
    only those are all-gathered -- KBs per rank instead of the [b,T,C,H,W] fields.  Returns the
    concatenation over ranks in rank order, i.e. the rows of train_stage2_ns2d.py:254-257 for the global batch.
    All ranks must hold the same local batch size."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return frame, seq
    world = dist.get_world_size(group)
    f_all = torch.empty((world * frame.shape[0],) + tuple(frame.shape[1:]), dtype=frame.dtype, device=frame.device)
    s_all = torch.empty((world * seq.shape[0],) + tuple(seq.shape[1:]), dtype=seq.dtype, device=seq.device)
    dist.all_gather_into_tensor(f_all, frame.contiguous(), group=group)
    dist.all_gather_into_tensor(s_all, seq.contiguous(), group=group)
    return f_all, s_all
