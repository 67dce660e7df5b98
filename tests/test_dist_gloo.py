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


def _packed_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nclone_amd.engine import OutputBlock, _ALWAYS

    blk = OutputBlock(n, "cpu", set(_ALWAYS))
    g = torch.Generator().manual_seed(100 + rank)
    blk.t["game_state"].copy_(torch.rand((n, 41), generator=g))
    blk.t["entity_pos"].copy_(torch.rand((n, 6), generator=g))
    blk.t["reward"].copy_(torch.rand((n,), generator=g))
    blk.t["frames"].copy_(torch.randint(0, 5, (n,), generator=g).to(torch.int16))
    blk.t["action_mask"].copy_(torch.randint(0, 2, (n, 6), generator=g).to(torch.int8))
    blk.t["flags"].copy_(torch.randint(0, 64, (n,), generator=g).to(torch.uint8))
    packed = blk.packed()
    gathered = torch.empty(world * packed.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, packed)          # ONE collective for the whole observation of a rank
    parts = blk.split_packed(gathered, world)
    if rank == 0:
        q.put({k: v.numpy() for k, v in parts.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_packed_observation_block_gather_world_size_2():
    """Config 4's gather: the packed observation block (game_state, entity_positions, reward, frames, action_mask, flags)
    of every rank moves with ONE all_gather and splits back into per-rank field views."""
    n = 96
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        assert np.array_equal(parts["game_state"][rank], torch.rand((n, 41), generator=g).numpy())
        assert np.array_equal(parts["entity_pos"][rank], torch.rand((n, 6), generator=g).numpy())
        assert np.array_equal(parts["reward"][rank], torch.rand((n,), generator=g).numpy())
        assert np.array_equal(parts["frames"][rank], torch.randint(0, 5, (n,), generator=g).to(torch.int16).numpy())
        assert np.array_equal(parts["action_mask"][rank], torch.randint(0, 2, (n, 6), generator=g).to(torch.int8).numpy())
        assert np.array_equal(parts["flags"][rank], torch.randint(0, 64, (n,), generator=g).to(torch.uint8).numpy())


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nclone_amd.distributed import OverlappedObsGather

    n = 5000
    g = torch.Generator().manual_seed(100 + rank)
    packed = torch.zeros(n, dtype=torch.uint8)
    og = OverlappedObsGather(packed)
    serial = torch.empty(world * n, dtype=torch.uint8)
    ok = True
    prev = None
    for step in range(7):
        packed.copy_(torch.randint(0, 256, (n,), generator=g).to(torch.uint8))       # "step t" writes the block
        dist.all_gather_into_tensor(serial, packed)                                    # the serial gather of the same bytes
        og.submit()
        want = serial.clone()
        packed.copy_(torch.randint(0, 256, (n,), generator=g).to(torch.uint8))       # "step t + 1" overwrites it before wait()
        got = og.wait()
        ok = ok and bool(torch.equal(got, want))
        if prev is not None:
            ok = ok and not torch.equal(got, prev)
        prev = got.clone()
    # depth 2: two submissions outstanding, collected in order
    packed.fill_(7 + rank); og.submit(); a = torch.full((n,), 7, dtype=torch.uint8)
    packed.fill_(9 + rank); og.submit()
    first, second = og.wait().clone(), og.wait().clone()
    ok = ok and bool((first[:n] == 7).all() and (first[n:] == 8).all() and (second[:n] == 9).all() and (second[n:] == 10).all())
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_returns_the_serial_bytes():
    """OverlappedObsGather (the double-buffered config-4 gather) on world_size 2 / gloo / CPU tensors: every
    step's gathered bytes equal a serial all_gather of the same block even though the block is overwritten before wait()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_bench_bare_invocation_starts_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` without a launcher starts 2 rank processes itself; here (no GPU) the ranks fail, which must
    surface as a non-zero exit code and no JSON line -- not as a silent 1-rank run."""
    import subprocess
    import sys

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_two_ranks_on_one_gpu")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--preroll", "1", "--envs-per-gpu", "64", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The N > 1 launch path rehearsed on one GPU: `bench.py --gpus 2 --backend gloo --device 0` (bare invocation) runs two
    rank processes on the HIP path and rank 0 prints n_gpus 2 with the gathered-observation figure."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0", "--steps", "20",
                        "--warmup", "5", "--preroll", "10", "--envs-per-gpu", "1024", "--workload", "c3mixed", "--gather-obs",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["value"] > 0
    assert line["with_obs_gather"]["own_shard_roundtrip_ok"] and line["with_obs_gather"]["value"] > 0
    assert line["with_obs_gather"]["overlapped"]["bytes_equal_serial"] and line["with_obs_gather"]["overlapped"]["value"] > 0
    assert line["config"]["preroll_steps"] == 10 and line["launch_us"]["p95"] >= line["launch_us"]["p50"]
