#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched N++ stepper on MI355X (BASELINE.json metric, config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one npp_step launch: every environment executes one Gymnasium step = frame_skip (4) physics ticks
(early stop on win/death), truncation check, observation (game_state f32[41], action_mask i8[6],
entity_positions f32[6], flags, reward, frames), in-kernel auto-reset.  Inputs (actions for all K+W steps)
are resident in HBM before the timed region.  Workload: 8192 envs per GPU on 128 "curriculum 0" levels
(exit+switch only) x 64 envs, actions iid uniform {0..5} from numpy default_rng(rank).
Weak scaling: each rank (one process per GPU) steps its own 8192 envs; there is no data-path collective
(--gather-obs adds the RCCL all_gather of config 4 for inspection; it is off for the headline metric).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 8192
FRAME_SKIP = 4
# algorithmic HBM bytes per env-step for the game_state-only observation
# (state read 160 + state write 160 + action 1 + outputs 201)
ALGO_BYTES_PER_ENV_STEP = 522
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def committed_traffic():
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs,
    gfx950 x2 correction on FETCH_SIZE) of this same command; tools/profile_round.sh collects them and
    tools/summarize_profiles.py writes profiles/<round>_summary.json.  bench.py cannot run PMC passes on itself, so
    it reports the newest committed figure (or null)."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            t = json.load(open(f)).get("traffic", {}).get("hbm_bytes_per_launch")
            if t:
                best = (float(t), os.path.basename(f))
        except Exception:
            pass
    return best


def cpu_baseline(levels, seconds_target=12.0):
    """Time the CPU oracle ("port": C restatement of the reference tick, bit-checked against reference
    fixtures) on this box's host cores with OpenMP, on a bounded sample of the same workload."""
    from oracle import oracle as om

    om.build()
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    n_envs = 16 * threads
    sims = []
    for e in range(n_envs):
        o = om.Oracle("pow")
        o.load(levels[(e * 7) % len(levels)])
        sims.append(o)
    rng = np.random.default_rng(12345)
    # calibrate
    a = rng.integers(0, 6, size=(50, n_envs)).astype(np.uint8)
    t0 = time.perf_counter()
    om.run_batch(sims, a, FRAME_SKIP, 10000, threads)
    dt = time.perf_counter() - t0
    rate = 50 * n_envs / max(dt, 1e-6)
    steps = int(max(100, min(200000, seconds_target * rate / n_envs)))
    a = rng.integers(0, 6, size=(steps, n_envs)).astype(np.uint8)
    t0 = time.perf_counter()
    ticks = om.run_batch(sims, a, FRAME_SKIP, 10000, threads)
    dt = time.perf_counter() - t0
    return {
        "value": steps * n_envs / dt,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d envs x %d steps (frame_skip 4, %d ticks) of the same level set and action distribution, "
                  "%d OpenMP threads, %.1f s" % (n_envs, steps, ticks, threads, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--gather-obs", action="store_true", help="RCCL all_gather of game_state each step (config 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-streams", type=int, default=4,
                    help="also time the same K steps with the envs split into this many independent sub-batches on "
                         "separate HIP streams (reported as async_subbatches, never as value); 0 = skip")
    ap.add_argument("--open-loop-chunk", type=int, default=50,
                    help="also time the K steps as npp_step_many launches of this many steps (open_loop_rollout; 0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing the "
                                                      "multi-rank path on one GPU together with --device)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (rehearsal on a one-GPU box)")
    ap.add_argument("--workload", default="c0", choices=["c0", "mines", "doors", "zoo"],
                    help="c0 = config 2 (the headline metric); the others are secondary level sets")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is not None:
        local_rank = args.device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        dist = None
        torch.cuda.set_device(local_rank)

    from nclone_amd.engine import NppBatch
    from nclone_amd import levels as level_sets

    levels, tags = {"c0": level_sets.curriculum0_levels, "mines": level_sets.mine_levels, "doors": level_sets.door_levels,
                    "zoo": level_sets.zoo_levels}[args.workload]()
    n = args.envs_per_gpu
    K, W = args.steps, args.warmup
    b = NppBatch(n, device=local_rank, autoreset=True)
    b.load_levels(levels)
    # 64 consecutive envs (one wavefront / workgroup) per level, levels repeated round-robin
    b.assign_levels((np.arange(n) // 64) % len(levels))
    rng = np.random.default_rng(rank)
    acts = torch.from_numpy(rng.integers(0, 6, size=(K + W, n)).astype(np.uint8)).cuda()
    gathered = None
    if args.gather_obs and dist is not None:
        gathered = torch.empty((world * n, 41), dtype=torch.float32, device="cuda")

    def one(k):
        b.step(acts[k], FRAME_SKIP, want_terminal=False)
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, b.game_state)

    for k in range(W):
        one(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(W, W + K):
        one(k)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    done_frac = float((b.flags & 3).ne(0).float().mean().item())
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # secondary figure: the same envs as S independent sub-batches, each stepped K times on its own HIP stream with no
    # cross-stream synchronisation until the end (an asynchronous / double-buffered vector env).  A synchronous step lasts
    # as long as its slowest env's serial fp64 chain; independent sub-batches let other envs' work fill that tail.
    async_rep = None
    S = args.async_streams
    if S > 1 and world == 1 and n % (S * 64) == 0 and not args.gather_obs:   # single-GPU runs only: a secondary figure
        sub = n // S
        streams = [torch.cuda.Stream() for _ in range(S)]
        subs = []
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                sb = NppBatch(sub, device=local_rank, autoreset=True, stream=streams[k])
                sb.load_levels(levels)
                sb.assign_levels(((np.arange(sub) + k * sub) // 64) % len(levels))
                subs.append(sb)
        torch.cuda.synchronize()
        views = [acts[:, k * sub:(k + 1) * sub].contiguous() for k in range(S)]

        def run_async(k0, k1):
            for t in range(k0, k1):
                for k in range(S):
                    subs[k].step(views[k][t], FRAME_SKIP, want_terminal=False)

        run_async(0, W)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        run_async(W, W + K)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        adt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([adt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            adt = float(tt.item())
        async_rep = {"streams": S, "envs_per_stream": sub, "value": world * n * K / adt, "unit": "env-steps/s",
                     "ms_per_step_all_streams": adt * 1e3 / K,
                     "note": "same envs, levels and actions as `value`, stepped as independent sub-batches on separate HIP "
                             "streams (no barrier between sub-batches); not the headline metric"}
        for sb in subs:
            sb.close()

    # secondary figure: the same K steps as launches of 50 steps each (npp_step_many): open-loop action sequences, as in
    # batched checkpoint replay; wavefronts run through their steps without waiting for the slowest env of every step
    many_rep = None
    if world == 1 and not args.gather_obs and args.open_loop_chunk > 0 and K >= args.open_loop_chunk:
        mb = NppBatch(n, device=local_rank, autoreset=True)
        mb.load_levels(levels)
        mb.assign_levels((np.arange(n) // 64) % len(levels))
        chunk = args.open_loop_chunk
        mb.step_many(acts[:W])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done_steps = 0
        for k0 in range(W, W + K - chunk + 1, chunk):
            mb.step_many(acts[k0:k0 + chunk])
            done_steps += chunk
        torch.cuda.synchronize()
        mdt = time.perf_counter() - t0
        many_rep = {"steps_per_launch": chunk, "steps": done_steps, "value": n * done_steps / mdt, "unit": "env-steps/s",
                    "note": "npp_step_many: open-loop action sequences (no observation between steps), per-step flags / "
                            "rewards still written; not the headline metric"}
        mb.close()

    if rank == 0:
        value = world * n * K / dt
        launch_us = dev_ms * 1e3 / K   # HIP events on the launch stream: average duration per npp_step launch
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (launch_us * 1e-6) / 1e9
        line = {
            "metric": "env-steps/sec (whole node) at N parallel envs",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("config 2: %d envs/GPU, curriculum_level=0 (exit+switch only: 78 bc_replays maps + "
                             "maze:tiny/hills:simple seeds 100001-100025 = %d levels x 64 envs), game_state+"
                             "action_mask+entity_positions obs, frame_skip 4, uniform random actions, auto-reset"
                             % (n, len(levels))) if args.workload == "c0" else
                            ("secondary level set '%s': %d envs/GPU on %d levels x 64 envs, same observation and action "
                             "distribution as config 2" % (args.workload, n, len(levels))),
                "envs_per_gpu": n,
                "frame_skip": FRAME_SKIP,
                "ticks_per_s": value * FRAME_SKIP,
                "gather_obs": bool(gathered is not None),
                "terminated_frac_last_step": done_frac,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (committed_traffic() or (None, None))[0],
                "traffic_source": (committed_traffic() or (None, None))[1],
                "kernel": "npp_step_kernel",
                "avg_launch_us": launch_us,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                "note": "latency/divergence-bound fp64 scalar chains (about 7k dependent flops per env-step); "
                        "the HBM fraction is tiny by construction (SURVEY.md 8(d))",
            },
        }
        if async_rep is not None:
            line["async_subbatches"] = async_rep
        if many_rep is not None:
            line["open_loop_rollout"] = many_rep
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(levels)
            except Exception as e:  # the baseline is a reported figure, never a dependency of the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
