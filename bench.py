#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched N++ stepper on MI355X (BASELINE.json metric; default = config 2).

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one npp_step launch: every environment executes one Gymnasium step = frame_skip (4) physics ticks
(early stop on win/death), truncation check, observation (game_state f32[41], action_mask i8[6],
entity_positions f32[6], flags, reward, frames), in-kernel auto-reset.  Inputs (actions for every step) are resident
in HBM before the timed region.

Workloads:
  c0       config 2 (headline): 8192 envs/GPU on 128 exit+switch levels x 64 envs, uniform random actions
  mines    config 3 level set; with --player-frame the 84x84 raster of every env is rendered EVERY step inside the
           timed region (npp_render_player_frame) and a second roofline is reported for the render kernel
  c3mixed  config 4: the 512-level mixed set, 8192 envs/GPU (65 536 on 8 GPUs); --gather-obs adds the RCCL
           all_gather of the packed observation block (game_state + entity_positions + reward + frames +
           action_mask + flags) and the line reports the rate with and without it
  doors    config 5 level set (physics + game_state; the reachability observation has its own benchmark line)
  zoo      the 26 entity-zoo maps

Before --warmup is honoured a fixed pre-roll of PREROLL_STEPS launches runs (reported as preroll_steps): episodes
desynchronise, clocks ramp and npp_step's build-variant autotuner (256 + 9 x 48 launches) reaches its decision, so the figure
does not depend on the caller's warm-up.  Every launch of the timed region
is bracketed by HIP events on the launch stream: mean / p50 / p95 / max are reported, together with the levels that own
the slowest 5 % of the launches (the env with the most depenetration iterations in that launch, npp_step_out.d_work).

Without --workload the line carries every BASELINE config that fits the run (round 3):
  N = 1:  value = config 2 (c0) and `other_configs` = {config3 (mines + player_frame every step), config4_shard (c3mixed, one GPU's
          8192 of the 65 536 envs), config5_full_obs (doors, every Dict observation every step)}, --other-steps (200) timed steps
          each after their own pre-roll, each with its own roofline block(s); config5_full_obs is timed twice -- every kernel on one
          stream (`serial`, the pass the per-kernel times and rooflines come from) and with the observation overlap of
          include/npp_amd.h npp_set_obs_overlap (`obs_overlap`, --obs-overlap: same work, same bits) -- `value` is the better one;
  N > 1:  value = config 2 on N GPUs (so that N = 1 agrees with the line above) and `config4` = c3mixed on the N GPUs with and
          without the RCCL all_gather of the packed observation block.
With --workload X the line is that workload alone, as before.

Weak scaling: each rank (one process per GPU) steps its own envs; there is no data-path collective unless --gather-obs.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 8192
FRAME_SKIP = 4
PREROLL_STEPS = 1000
# algorithmic HBM bytes per env-step for the game_state-only observation
# (state read 160 + state write 160 + action 1 + outputs 201)
ALGO_BYTES_PER_ENV_STEP = 522
# player_frame: 7056 B written + ~0.2 KB of state / level tables read per env
ALGO_BYTES_PER_FRAME = 7056 + 200
# global_view: 176 x 100 bytes written per env + the env's state / draw records read (the per-level view comes from L2)
ALGO_BYTES_PER_GLOBAL_VIEW = 17600 + 300
# reachability: 38 + 3 floats out, position / key / level in, the env's cache row read or written (npp_reach_kernel.hip)
ALGO_BYTES_PER_REACH = 340
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def committed_traffic(kernel="step", workload="c0", variant=None):
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs) of this same
    command; tools/profile_round.sh collects them and tools/summarize_profiles.py writes profiles/<round>_summary.json (the
    headline workload), <round>_render_summary.json (config 3: mines + player_frame) and <round>_c5_summary.json (config 5: doors,
    full observation), each with a `by_kernel` table.  bench.py cannot run PMC passes on itself, so it reports the newest
    committed figure for the kernel of the workload's own profile (or null) and names the file."""
    import glob

    if workload == "zoo":
        return None   # no committed PMC profile of the zoo kernels
    suffix = {"mines": "_render_summary.json", "doors": "_c5_summary.json"}.get(workload, "_summary.json")
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*" + suffix))):
        name = os.path.basename(f)
        if suffix == "_summary.json" and (name.endswith("_render_summary.json") or name.endswith("_c5_summary.json")):
            continue
        try:
            j = json.load(open(f))
            t = j.get("by_kernel", {}).get(kernel, {}).get("hbm_bytes_per_launch")
            if kernel == "step" and variant is not None:   # the row of the build variant this run launched (the profile holds all three)
                tv = j.get("by_kernel", {}).get("step", {}).get("variants", {}).get(str(int(variant)), {}).get("hbm_bytes_per_launch")
                if tv:
                    t, name = tv, name + " (step build variant %d)" % int(variant)
            if t is None and kernel == "step":
                t = j.get("traffic", {}).get("hbm_bytes_per_launch")          # round 1-2 files
            if t is None and kernel == "player_frame":
                t = j.get("render_traffic", {}).get("hbm_bytes_per_launch")
            if t:
                best = (float(t), name)
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--preroll", type=int, default=PREROLL_STEPS, help="fixed untimed launches before --warmup")
    ap.add_argument("--gather-obs", action="store_true",
                    help="also time the steps with the RCCL all_gather of the packed observation block (config 4)")
    ap.add_argument("--player-frame", action="store_true",
                    help="render the 84x84 player_frame of every env each step inside the timed region (config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-streams", type=int, default=4,
                    help="also time the same K steps with the envs split into this many independent sub-batches on "
                         "separate HIP streams (reported as async_subbatches, never as value); 0 = skip")
    ap.add_argument("--open-loop-chunk", type=int, default=50,
                    help="also time the K steps as npp_step_many launches of this many steps (open_loop_rollout; 0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing the "
                                                      "multi-rank path on one GPU together with --device)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (rehearsal on a one-GPU box)")
    ap.add_argument("--full-obs", action="store_true",
                    help="config 5: every Dict observation inside the timed region -- spatial_context (in the step kernel), "
                         "switch_states, player_frame, global_view, reachability_features + mine_sdf_features; per-kernel times "
                         "are reported under obs_kernels")
    ap.add_argument("--workload", default=None, choices=["c0", "mines", "doors", "zoo", "c3mixed"],
                    help="time this workload alone (c0 = config 2, the headline metric); default: c0 as `value` plus the other "
                         "BASELINE configs in their own blocks, see the module docstring")
    ap.add_argument("--other-steps", type=int, default=200,
                    help="timed steps of each block of `other_configs` / `config4` (0 = leave them out)")
    ap.add_argument("--step-variant", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="pin a build variant of the step kernel (A/B runs); -1 = the autotuner decides (default)")
    ap.add_argument("--obs-overlap", type=str, default="50",
                    help="full-obs workloads: comma-separated percentages for npp_set_obs_overlap (the most expensive workgroups of the step "
                         "go to a second stream and the observation kernels of the other envs run beside them; same bits).  Each is timed "
                         "after the serial pass and reported under `obs_overlap`; the block's `value` is the best of them, the serial pass "
                         "stays as `serial`.  \"\" = serial only")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="seconds after which `bench.py --gpus N` gives up on its rank processes")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when this process starts the ranks itself")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's line.
    The parent never touches torch.cuda / HIP (and never execs): each child initialises its own GPU.  All children are polled:
    the first one that exits non-zero ends the run (the others are terminated, then killed after a grace period) with its exit
    code, and so does --rank-timeout; the other ranks' stdout is relayed to stderr with a rank prefix, stderr is inherited."""
    import socket
    import threading

    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE))

    def relay(r, pipe):
        for raw in iter(pipe.readline, b""):
            if r == 0:
                sys.stdout.write(raw.decode(errors="replace"))
                sys.stdout.flush()
            else:
                sys.stderr.write("[rank %d] %s" % (r, raw.decode(errors="replace")))
                sys.stderr.flush()

    threads = [threading.Thread(target=relay, args=(r, p.stdout), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    deadline = time.monotonic() + args.rank_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            sys.stderr.write("bench.py: rank processes still running after %.0f s, giving up\n" % args.rank_timeout)
            rc = 124
            break
        time.sleep(0.2)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for t in threads:
        t.join(timeout=5.0)
    return rc


def percentiles(us):
    a = np.sort(np.asarray(us, dtype=np.float64))
    return {"mean": float(a.mean()), "p50": float(a[len(a) // 2]), "p95": float(a[min(len(a) - 1, int(0.95 * len(a)))]),
            "max": float(a[-1]), "min": float(a[0])}


DESCS = {
    "c0": "config 2: %d envs/GPU, curriculum_level=0 (exit+switch only: 78 bc_replays maps + maze:tiny/hills:simple "
          "seeds 100001-100025 = %d levels x 64 envs), game_state+action_mask+entity_positions obs",
    "mines": "config 3: %d envs/GPU, curriculum_level=2 (mines: 20 bc_replays maps + 44 generated corridor levels = %d levels x 64 envs)",
    "c3mixed": "config 4: %d envs/GPU (x n_gpus), curriculum_level=3 mixed map set (%d levels: c0 + mines + 320 generated "
               "simpler/simple levels) x 64 envs",
    "doors": "config 5: %d envs/GPU, curriculum_level=4 (locked doors / switches, %d levels x 64 envs)",
    "zoo": "secondary level set 'zoo': %d envs/GPU on the %d entity-zoo maps x 64 envs",
}


def hbm_roofline(kernel, algo_bytes, n, us, traffic=(None, None), note=None):
    ach = algo_bytes * n / (us["mean"] * 1e-6) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic[0],
         "traffic_source": traffic[1], "kernel": kernel, "avg_launch_us": us["mean"], "algorithmic_bytes_per_launch": algo_bytes * n}
    if note:
        r["note"] = note
    return r


STEP_NOTE = ("issue-cadence-bound fp64 scalar chains (about 7k dependent flops per env-step; DESIGN.md 4.1): the HBM fraction is tiny by "
             "construction (SURVEY.md 8(d)); traffic = rocprofv3 FETCH_SIZE + WRITE_SIZE per launch of the committed profile named in "
             "traffic_source (a process cannot run PMC passes on itself)")


def run_workload(ctx, workload, K, W, P, player_frame=False, full_obs=False, gather_obs=False, secondaries=False):
    """Time K npp_step launches (+ the observation kernels asked for) of one workload after P + W untimed ones.  Returns the
    block of figures for that workload (rank 0) -- `value` is the whole-job rate over all ranks."""
    import torch

    from nclone_amd import levels as level_sets
    from nclone_amd.engine import NppBatch

    args, rank, world, dist, n = ctx["args"], ctx["rank"], ctx["world"], ctx["dist"], ctx["n"]
    barrier, max_over_ranks = ctx["barrier"], ctx["max_over_ranks"]
    levels, tags = {"c0": level_sets.curriculum0_levels, "mines": level_sets.mine_levels, "doors": level_sets.door_levels,
                    "zoo": level_sets.zoo_levels, "c3mixed": level_sets.c3_mixed_levels}[workload]()
    outputs = ["work"] + (["player_frame"] if player_frame else [])
    if full_obs:
        outputs += ["spatial_context", "switch_states", "player_frame", "global_view", "reachability_features", "mine_sdf_features"]
    b = NppBatch(n, device=ctx["local_rank"], autoreset=True, outputs=outputs)
    # (switch_states comes out of the reachability launch: npp_reachability_ex)
    STAGES = (("player_frame", lambda: b.render_player_frame()), ("global_view", lambda: b.render_global_view()),
              ("reachability", lambda: b.reachability(with_switch_states=True))) if full_obs else ()
    b.load_levels(levels)
    if args.step_variant >= 0:
        b.set_step_variant(args.step_variant)
    # 64 consecutive envs (one wavefront / workgroup) per level, levels repeated round-robin; with more ranks than one the
    # global env index decides, so that 8 x 8192 envs cover all 512 levels of the mixed set twice
    env_level = ((np.arange(n) + rank * n) // 64) % len(levels)
    b.assign_levels(env_level)
    rng = np.random.default_rng(rank + (2 if workload == "c3mixed" else 0))
    total = P + W + K
    acts = torch.from_numpy(rng.integers(0, 6, size=(total, n)).astype(np.uint8)).cuda()
    work = torch.zeros((K, n), dtype=torch.int16, device="cuda")
    stream = b.stream
    stage_us = {}

    def one(k):
        b.step(acts[k], FRAME_SKIP, want_terminal=False)
        if player_frame:
            b.render_player_frame()
        for _name, fn in STAGES:
            fn()

    def timed(k0, gather=None):
        """K launches from step index k0 on, bracketed by barrier + synchronize; per-launch HIP events on the launch stream."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        mids = [torch.cuda.Event(enable_timing=True) for _ in range(K)] if player_frame else None
        sevs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(STAGES))] for _ in range(K)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record(stream)
        for k in range(K):
            b.step(acts[k0 + k], FRAME_SKIP, want_terminal=False, work_out=work[k])
            if mids is not None:
                mids[k].record(stream)
                b.render_player_frame()
            for j, (_name, fn) in enumerate(STAGES):
                sevs[k][j].record(stream)
                fn()
            if gather is not None:
                gather()
            evs[k + 1].record(stream)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        if STAGES:
            step_us = [evs[k].elapsed_time(sevs[k][0]) * 1e3 for k in range(K)]
            render_us = None
            for j, (name, _fn) in enumerate(STAGES):
                stage_us[name] = [sevs[k][j].elapsed_time(sevs[k][j + 1] if j + 1 < len(STAGES) else evs[k + 1]) * 1e3 for k in range(K)]
        elif mids is None:
            step_us = [evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(K)]
            render_us = None
        else:
            step_us = [evs[k].elapsed_time(mids[k]) * 1e3 for k in range(K)]
            render_us = [mids[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(K)]
        return dt, step_us, render_us

    for k in range(P + W):
        one(k)
    dt, step_us, render_us = timed(P + W)
    done_frac = float((b.flags & 3).ne(0).float().mean().item())
    # build variant of the G = 16 step kernels picked by npp_step's autotuner during the pre-roll (include/npp_amd.h: npp_set_step_variant)
    sv = b.step_variant()
    step_variant = {"variant": int(sv[0]), "tuned": bool(sv[1]),
                    "meaning": "0: 2 wavefronts/SIMD, 2 candidate slots; 1: 2 wavefronts/SIMD, 1 slot; 2: 1 wavefront/SIMD, 2 slots"}

    # which levels own the slow launches: the env with the most depenetration iterations in each of the slowest 5 %
    lt = np.asarray(step_us)
    slow = np.argsort(lt)[-max(1, K // 20):]
    wk = work[torch.from_numpy(np.sort(slow)).cuda()].cpu().numpy().astype(np.int64) & 0xffff
    owners = env_level[np.argmax(wk, axis=1)]
    ids, cnt = np.unique(owners, return_counts=True)
    top = np.argsort(-cnt)[:6]
    stragglers = {"launches": int(len(slow)), "iterations_max_env_mean": float(wk.max(axis=1).mean()),
                  "iterations_all_envs_mean": float(wk.mean()),
                  "levels": [{"level_id": int(ids[i]), "tag": tags[int(ids[i])], "launches": int(cnt[i])} for i in top]}

    # observation overlap (include/npp_amd.h npp_set_obs_overlap): the same K steps with the expensive workgroups of the step on a
    # second stream and one observation kernel per part; the outputs are the serial ones (tests/test_gpu_round3.py)
    overlap_rep = None
    if full_obs and args.obs_overlap:   # (config 3 -- short step, one observation kernel -- loses with it: DESIGN.md 4.9)
        overlap_rep = []
        for spec in [x.strip() for x in args.obs_overlap.split(",") if x.strip()]:
            pct = [int(c) for c in spec.split("+")]   # "40" = one cut, "6+25+50" = three
            b.set_obs_overlap(pct)
            for k in range(P, P + W):
                one(k)
                b.join()
            barrier()
            t0 = time.perf_counter()
            for k in range(K):
                b.step(acts[P + W + k], FRAME_SKIP, want_terminal=False)
                for _name, fn in STAGES:
                    fn()
                b.join()
            barrier()
            odt = max_over_ranks(time.perf_counter() - t0)
            overlap_rep.append({"cuts_percent": pct, "value": world * n * K / odt, "unit": "env-steps/s", "ms_per_step": odt * 1e3 / K})
        b.set_obs_overlap(0)

    gather_rep = None
    if gather_obs and dist is not None:
        packed = b.out.packed()
        if args.backend == "nccl":
            gathered = torch.empty(world * packed.numel(), dtype=torch.uint8, device="cuda")

            def gather():
                dist.all_gather_into_tensor(gathered, packed)
        else:   # gloo rehearsal: host tensors
            gathered = torch.empty(world * packed.numel(), dtype=torch.uint8)
            host = torch.empty(packed.numel(), dtype=torch.uint8)

            def gather():
                host.copy_(packed)
                dist.all_gather_into_tensor(gathered, host)
        for k in range(P, P + W):
            one(k)
            gather()
        gdt, _, _ = timed(P + W, gather)
        parts = b.out.split_packed(gathered, world)
        ok = bool(torch.equal(parts["game_state"][rank].to(b.game_state.device), b.game_state))
        # the same with the gather taken off the critical path (nclone_amd.distributed.OverlappedObsGather): the collective of step t
        # runs on a side stream while step t + 1 executes.  Guarded: RCCL has never run in the build environment, and a failure of
        # this secondary figure must not take the line (and the serial figure) with it.
        overlapped = None
        try:
            from nclone_amd.distributed import OverlappedObsGather

            og = OverlappedObsGather(packed, world)
            same = True
            for k in range(P, P + 3):   # the bytes are those of the serial gather
                b.step(acts[k], FRAME_SKIP, want_terminal=False)
                gather()
                torch.cuda.synchronize()
                want = gathered.clone().cpu()
                og.submit()
                b.step(acts[k + 1], FRAME_SKIP, want_terminal=False)   # overwrites the block before the gather is collected
                got = og.wait()
                torch.cuda.synchronize()
                same = same and bool(torch.equal(got.cpu(), want))
            barrier()
            t0 = time.perf_counter()
            for k in range(K):
                b.step(acts[P + W + k], FRAME_SKIP, want_terminal=False)
                og.submit()
                if k:
                    og.wait()
            og.wait()
            barrier()
            odt = max_over_ranks(time.perf_counter() - t0)
            overlapped = {"value": world * n * K / odt, "unit": "env-steps/s", "ms_per_step": odt * 1e3 / K, "bytes_equal_serial": same,
                          "how": "snapshot of the packed block + all_gather on a side stream while the next step runs (double-buffered); "
                                 "with gloo the host collective blocks inside wait()"}
        except Exception as e:   # noqa: BLE001
            overlapped = {"value": None, "error": repr(e)[:300]}
        gather_rep = {"value": world * n * K / gdt, "unit": "env-steps/s", "ms_per_step": gdt * 1e3 / K,
                      "overlapped": overlapped,
                      "bytes_per_rank_per_step": int(packed.numel()), "collective": "all_gather_into_tensor (1 per step)",
                      "own_shard_roundtrip_ok": ok,
                      "fields": "game_state f32[41], entity_positions f32[6], reward f32, frames i16, action_mask i8[6], flags u8"}

    # secondary figure: the asynchronous vector env (nclone_amd.async_env.NppAsyncVecEnvironment's engine): the same envs as S
    # independent sub-batches, each stepped on its own HIP stream with no cross-stream synchronisation until the end
    async_rep = None
    S = args.async_streams
    if secondaries and S > 1 and world == 1 and n % (S * 64) == 0 and not gather_obs and not player_frame and not full_obs:
        from nclone_amd.async_env import AsyncBatches

        ab = AsyncBatches(n, S, device=ctx["local_rank"], autoreset=True)
        ab.load_levels(levels)
        ab.assign_levels(env_level)
        torch.cuda.synchronize()
        views = [acts[:, k * (n // S):(k + 1) * (n // S)].contiguous() for k in range(S)]
        for t in range(P + W):
            ab.step_async([v[t] for v in views], FRAME_SKIP)
        ab.wait()
        t0 = time.perf_counter()
        for t in range(P + W, total):
            ab.step_async([v[t] for v in views], FRAME_SKIP)
        ab.wait()
        adt = time.perf_counter() - t0
        async_rep = {"streams": S, "envs_per_stream": n // S, "value": n * K / adt, "unit": "env-steps/s",
                     "ms_per_step_all_streams": adt * 1e3 / K,
                     "note": "same envs, levels and actions as `value`, stepped as independent sub-batches on separate HIP "
                             "streams (NppAsyncVecEnvironment); not the headline metric"}
        ab.close()

    # secondary figure: the same K steps as launches of 50 steps each (npp_step_many): open-loop action sequences, as in
    # batched checkpoint replay; wavefronts run through their steps without waiting for the slowest env of every step
    many_rep = None
    if secondaries and world == 1 and not gather_obs and not player_frame and not full_obs and args.open_loop_chunk > 0 and K >= args.open_loop_chunk:
        mb = NppBatch(n, device=ctx["local_rank"], autoreset=True)
        mb.load_levels(levels)
        mb.assign_levels(env_level)
        chunk = args.open_loop_chunk
        mb.step_many(acts[:P + W])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done_steps = 0
        for k0 in range(P + W, total - chunk + 1, chunk):
            mb.step_many(acts[k0:k0 + chunk])
            done_steps += chunk
        torch.cuda.synchronize()
        mdt = time.perf_counter() - t0
        many_rep = {"steps_per_launch": chunk, "steps": done_steps, "value": n * done_steps / mdt, "unit": "env-steps/s",
                    "note": "npp_step_many: open-loop action sequences (no observation between steps), per-step flags / "
                            "rewards still written; not the headline metric"}
        mb.close()
    b.close()
    del b
    torch.cuda.empty_cache()

    value = world * n * K / dt
    pl = percentiles(step_us)
    desc = DESCS[workload] % (n, len(levels))
    if player_frame:
        desc += ", player_frame 84x84 rendered every step"
    if full_obs:
        desc += (", full Dict obs every step: game_state, action_mask, entity_positions, spatial_context, switch_states, player_frame, "
                 "global_view, reachability_features, mine_sdf_features (0 level(s) dropped)")
    if gather_rep is not None:
        desc += ", RCCL gather of the packed obs reported beside"
    blk = {
        "value": value, "unit": "env-steps/s", "steps": K, "ms_per_step": dt * 1e3 / K,
        "config": {"workload": desc + ", frame_skip 4, uniform random actions, auto-reset", "envs_per_gpu": n, "frame_skip": FRAME_SKIP,
                   "ticks_per_s": value * FRAME_SKIP, "preroll_steps": P, "player_frame": bool(player_frame or full_obs),
                   "gather_obs": bool(gather_rep is not None), "terminated_frac_last_step": done_frac},
        "launch_us": pl, "step_variant": step_variant, "stragglers": stragglers,
        "roofline": hbm_roofline("npp_step_kernel", ALGO_BYTES_PER_ENV_STEP, n, pl,
                                 committed_traffic("step", workload, step_variant["variant"]) or (None, None), STEP_NOTE),
    }
    if full_obs:
        blk["config"]["full_obs"] = True
    if render_us is not None:
        pr = percentiles(render_us)
        blk["roofline_render"] = hbm_roofline("npp_render_kernel", ALGO_BYTES_PER_FRAME, n, pr, committed_traffic("player_frame", workload) or (None, None),
                                              "7056 B written + ~0.2 KB read per env; parity of the raster is unpinned (no cairo/cv2 reference frame)")
        blk["roofline_render"]["launch_us"] = pr
    if stage_us:
        ok = {k: percentiles(v) for k, v in stage_us.items()}
        blk["obs_kernels"] = dict(ok)
        blk["obs_kernels"]["note"] = ("HIP-event time of each observation kernel per step on the launch stream; npp_step includes "
                                      "spatial_context; reachability = table look-ups for the envs whose (cell, switch) key changed, "
                                      "and switch_states from the same launch (npp_reachability_ex)")
        blk["roofline_render"] = hbm_roofline("npp_render_kernel", ALGO_BYTES_PER_FRAME, n, ok["player_frame"], committed_traffic("player_frame", workload) or (None, None),
                                              "player_frame; raster parity unpinned (no cairo/cv2 reference frame)")
        blk["roofline_global_view"] = hbm_roofline("npp_global_view_kernel", ALGO_BYTES_PER_GLOBAL_VIEW, n, ok["global_view"], committed_traffic("global_view", workload) or (None, None),
                                                   "17 600 B written per env (the per-level view it patches is read through L2)")
        blk["roofline_reach"] = hbm_roofline("npp_reach_kernel", ALGO_BYTES_PER_REACH, n, ok["reachability"], committed_traffic("reachability", workload) or (None, None),
                                             "latency-bound table look-ups for the envs whose cache key changed")
    if overlap_rep:
        blk["obs_overlap"] = overlap_rep
        best = max(overlap_rep, key=lambda o: o["value"])
        blk["serial"] = {"value": value, "unit": "env-steps/s", "ms_per_step": dt * 1e3 / K,
                         "note": "every kernel on one stream; the per-kernel HIP-event times and rooflines of this block are from this pass"}
        if best["value"] > value:
            blk["value"], blk["ms_per_step"] = best["value"], best["ms_per_step"]
            blk["config"]["ticks_per_s"] = best["value"] * FRAME_SKIP
            blk["config"]["obs_overlap_cuts_percent"] = best["cuts_percent"]
    if gather_rep is not None:
        blk["with_obs_gather"] = gather_rep
    if async_rep is not None:
        blk["async_subbatches"] = async_rep
    if many_rep is not None:
        blk["open_loop_rollout"] = many_rep
    blk["_levels"] = levels
    return blk


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

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
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if dist is None:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    ctx = {"args": args, "rank": rank, "local_rank": local_rank, "world": world, "dist": dist, "n": args.envs_per_gpu,
           "barrier": barrier, "max_over_ranks": max_over_ranks}
    K, W, P = args.steps, args.warmup, args.preroll
    single = args.workload is not None
    head = run_workload(ctx, args.workload or "c0", K, W, P, player_frame=args.player_frame, full_obs=args.full_obs,
                        gather_obs=args.gather_obs, secondaries=True)
    others = {}
    Ko = args.other_steps
    if not single and Ko > 0 and not (args.player_frame or args.full_obs or args.gather_obs):
        Wo = min(W, 50)
        if world == 1:
            # BASELINE configs 3, 4 (one GPU's shard) and 5 in the driver's default run
            others["config3"] = run_workload(ctx, "mines", Ko, Wo, P, player_frame=True)
            others["config4_shard"] = run_workload(ctx, "c3mixed", Ko, Wo, P)
            others["config5_full_obs"] = run_workload(ctx, "doors", Ko, Wo, P, full_obs=True)
        else:
            # the north-star multi-GPU config: the mixed set on N GPUs, with and without the RCCL observation gather.  Guarded (on every
            # rank alike): the headline must survive a failure of this block
            try:
                others["config4"] = run_workload(ctx, "c3mixed", Ko, Wo, P, gather_obs=True)
            except Exception as e:   # noqa: BLE001
                others["config4"] = {"value": None, "error": repr(e)[:400]}

    if rank == 0:
        levels = head.pop("_levels")
        line = {
            "metric": "env-steps/sec (whole node) at N parallel envs",
            "value": head["value"],
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
        }
        for k, v in head.items():
            if k not in line:
                line[k] = v
        if others:
            for blk in others.values():
                blk.pop("_levels", None)
                blk["warmup"] = min(W, 50)
            if world == 1:
                line["other_configs"] = others
            else:
                line["config4"] = others["config4"]
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
