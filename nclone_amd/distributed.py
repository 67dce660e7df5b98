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
