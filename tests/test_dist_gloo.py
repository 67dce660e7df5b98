"""Multi-process path on CPU (gloo, world_size 2): env sharding + the config-4 observation gather + max-over-ranks
timing reduction.  Each rank steps ITS shard with the CPU oracle standing in for the device (the collective plumbing
is what is under test here, not the kernels)."""
import os
import socket

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


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nclone_amd.distributed import gather_observations, max_over_ranks, shard_envs, shard_level_ids
    from nclone_amd.levels import curriculum0_levels
    from oracle import oracle as om

    levels, _ = curriculum0_levels()
    global_ids = (np.arange(n_total) // 4) % len(levels)
    start, count = shard_envs(n_total, rank, world)
    ids = shard_level_ids(global_ids, rank, world)
    assert len(ids) == count
    sims = []
    for e in range(count):
        o = om.Oracle("pow")
        o.load(levels[ids[e]])
        sims.append(o)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 6, size=(5, n_total))
    local = torch.zeros((count, 41), dtype=torch.float32)
    for s in range(5):
        for e in range(count):
            sims[e].env_step(int(acts[s, start + e]), 4)
            local[e, :40] = torch.from_numpy(sims[e].ninja_state().astype(np.float32))
            local[e, 40] = (10000 - sims[e].frame) / 10000
        full = gather_observations(local)
    t = max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((full.numpy(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gather_matches_single_process():
    from nclone_amd.levels import curriculum0_levels
    from oracle import oracle as om

    om.build()
    n_total = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0 and full.shape == (n_total, 41)
    # single-process reference
    levels, _ = curriculum0_levels()
    ids = (np.arange(n_total) // 4) % len(levels)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 6, size=(5, n_total))
    for e in range(n_total):
        o = om.Oracle("pow")
        o.load(levels[ids[e]])
        for s in range(5):
            o.env_step(int(acts[s, e]), 4)
        assert np.array_equal(full[e, :40], o.ninja_state().astype(np.float32)), e
