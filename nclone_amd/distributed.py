"""Multi-GPU plumbing: environments are independent units, so the path shards with NO data-path collective.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).
Rank r owns the contiguous env block [r * n / world, (r + 1) * n / world) and a full copy of the level tables
(SURVEY.md 8(e)).  The only collective is the optional observation gather of BASELINE.json config 4:
one all_gather of the packed game_state block per step.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_envs(total_envs, rank, world):
    """Contiguous block partition; remainders go to the first ranks. Returns (start, count)."""
    base, rem = divmod(int(total_envs), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def shard_level_ids(global_level_ids, rank, world):
    """Slice of the global env -> level assignment owned by `rank`."""
    start, count = shard_envs(len(global_level_ids), rank, world)
    return np.asarray(global_level_ids)[start:start + count]


def gather_observations(local, out=None):
    """all_gather of a [n_local, ...] tensor into [world * n_local, ...] (equal shard sizes).

    With 8192 envs x 41 f32 per rank this is 1.3 MB per rank per step: far below one xGMI link, so a single
    un-bucketed ring all-gather is the right size."""
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class OverlappedObsGather:
    """The config-4 observation gather taken off the step's critical path: `submit()` snapshots the rank's packed observation
    block (one device-to-device copy of 201 B/env, ordered after the step on the caller's stream -- the next step may then
    overwrite the block) and hands the collective to a SIDE stream; the caller enqueues step t + 1; `wait()` returns the gathered
    bytes of step t ([world * packed_bytes] uint8, two buffers alternate) without the step's stream ever having waited for the
    collective.  Backend "nccl" (= RCCL over xGMI): all_gather_into_tensor(async_op=True) on the side stream, the consumer stream
    waits on an event, the host never blocks.  Any other backend (the gloo rehearsal): the snapshot is copied to pinned host
    memory on the side stream and the (blocking) host collective runs inside `wait()`, i.e. while step t + 1 executes.  The bytes
    are those of the serial gather (tests/test_dist_gloo.py, bench.py --gather-obs reports `overlapped.bytes_equal_serial`)."""

    def __init__(self, packed, world=None):
        self.packed = packed
        self.world = dist.get_world_size() if world is None else int(world)
        self.cuda = bool(packed.is_cuda)
        self.nccl = self.cuda and dist.get_backend() == "nccl"
        n = packed.numel()
        self.snap = [torch.empty_like(packed) for _ in range(2)]
        if self.nccl:
            self.out = [torch.empty(self.world * n, dtype=torch.uint8, device=packed.device) for _ in range(2)]
        else:
            self.out = [torch.empty(self.world * n, dtype=torch.uint8) for _ in range(2)]
            self.host = [torch.empty(n, dtype=torch.uint8, pin_memory=self.cuda) for _ in range(2)] if self.cuda else self.snap
        self.side = torch.cuda.Stream(device=packed.device) if self.cuda else None
        self.k = 0
        self._inflight = [None, None]   # per buffer: ("nccl", work) | ("host", event) | ("cpu", None)

    def submit(self):
        """Call right after the step has been enqueued on the current stream."""
        i = self.k & 1
        assert self._inflight[i] is None, "wait() for the previous submission of this buffer first (depth 2)"
        if not self.cuda:
            self.snap[i].copy_(self.packed)
            self._inflight[i] = ("cpu", None)
        else:
            main = torch.cuda.current_stream(self.packed.device)
            self.snap[i].copy_(self.packed, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                if self.nccl:
                    self._inflight[i] = ("nccl", dist.all_gather_into_tensor(self.out[i], self.snap[i], async_op=True))
                else:
                    self.host[i].copy_(self.snap[i], non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self.side)
                    self._inflight[i] = ("host", done)
        self.k += 1
        return i

    def wait(self, consumer_stream=None):
        """Gathered bytes of the OLDEST outstanding submission."""
        i = (self.k - 1) & 1 if self._inflight[self.k & 1] is None else self.k & 1
        kind, h = self._inflight[i]
        self._inflight[i] = None
        if kind == "nccl":
            with torch.cuda.stream(self.side):
                h.wait()
                done = torch.cuda.Event()
                done.record(self.side)
            (consumer_stream or torch.cuda.current_stream(self.packed.device)).wait_event(done)
        else:
            if kind == "host":
                h.synchronize()
            dist.all_gather_into_tensor(self.out[i], self.host[i])
        return self.out[i]
