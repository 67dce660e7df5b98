"""Round-3 GPU tests: BASELINE config 5 as a whole (the full Dict step at 8192 envs on ALL 21 door levels, including the one
whose reachability goes through the reference's physics A*), the bench line's other_configs / config4 blocks, the dynamic
truncation limit, and the Gymnasium surface's checkpoint option."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 8192
FULL = ("positions", "spatial_context", "switch_states", "player_frame", "global_view", "reachability_features", "mine_sdf_features",
        "reach_status")


def test_config5_full_dict_step_on_the_whole_door_set(oracle_mod):
    """BASELINE config 5: 8192 envs, curriculum 4 (locked doors / switches), every Dict observation every step, on the whole
    21-level door set (round 2 dropped `hcorr:door:100053`, whose exit door sits 18 px from its switch).  Checked:
      * every output of the block is identical between the two replicas of each 32-env group (no env leaks into another);
      * physics + game_state of a 96-env sample against the CPU oracle in the env's reset sequence (load, reset(), fast_reset());
      * reachability_features of a sample that covers every level -- incl. the A* level -- against the HOST build of the same
        function driven by the recorded positions / episode boundaries (cache rule, per-episode dictionary, live mine counts);
      * player_frame = the reference's axis-swapped crop of the whole rendered frame, global_view = OpenCV-order area reduction of
        the whole frame, for sampled envs; switch_states against the entity dump."""
    from nclone_amd import _native as nat
    from nclone_amd.engine import NppBatch, compile_level_entities, reach_level_info
    from nclone_amd.levels import door_levels
    from tests.test_gpu_render import _area_tabs, _reduce_frame

    levels, tags = door_levels()
    assert len(levels) == 21 and all(reach_level_info(m)["supported"] for m in levels)
    lib = nat.lib()
    level_ids = (np.arange(N) // 64) % len(levels)
    steps = 48
    acts_np = np.random.default_rng(33).integers(0, 6, size=(steps, N)).astype(np.uint8).reshape(steps, N // 64, 64)
    acts_np[:, :, 32:] = acts_np[:, :, :32]
    acts_np = acts_np.reshape(steps, N)
    acts = torch.from_numpy(acts_np).cuda()
    b = NppBatch(N, autoreset=True, fast_reset=True, outputs=FULL)
    b.load_levels(levels)
    b.assign_levels(level_ids)
    b.reset()                      # the env's first reset(): Simulator.reset
    b.set_truncation_limit(120)    # episodes end inside the run on every level
    # sample: 3 envs of every level (first block of the level) + extra envs of the A* level
    first_block = {li: int(np.flatnonzero(level_ids == li)[0]) for li in range(len(levels))}
    astar = tags.index("hcorr:door:100053")
    sample = sorted({first_block[li] + k for li in range(len(levels)) for k in (0, 7, 19)} |
                    {int(e) for e in np.flatnonzero(level_ids == astar)[:24]})
    sample = np.array(sample)
    hist_pos, hist_flags, hist_feat = [], [], []

    def observe_all():
        b.switch_states()
        b.render_player_frame()
        b.render_global_view()
        b.reachability()

    b.observe()
    observe_all()
    h = b.to_host(("positions", "reachability_features", "reach_status", "flags"))
    assert not h["reach_status"].any()
    hist_pos.append(h["positions"][sample, :2].copy())
    hist_flags.append(np.zeros(len(sample), np.uint8))
    hist_feat.append(h["reachability_features"][sample].copy())
    for t in range(steps):
        b.step(acts[t])
        observe_all()
        h = b.to_host(("positions", "reachability_features", "reach_status", "flags"))
        assert not h["reach_status"].any(), t
        hist_pos.append(h["positions"][sample, :2].copy())
        hist_flags.append(h["flags"][sample].copy())
        hist_feat.append(h["reachability_features"][sample].copy())
    out = {k: v.copy() for k, v in b.to_host().items()}
    f, di = b.dump_state()
    # ---- replicas
    for k, v in out.items():
        if k in ("work",):
            continue
        blk = v.reshape((N // 64, 64) + v.shape[1:])
        assert np.array_equal(blk[:, :32], blk[:, 32:]), k
    assert out["player_frame"].std() > 10 and out["global_view"].std() > 10
    # ---- physics of a sample vs the oracle
    for e in np.random.default_rng(4).choice(N, size=96, replace=False):
        o = oracle_mod.Oracle("mul")
        assert o.load(levels[level_ids[e]]) == 0
        o.reset()
        for t in range(steps):
            k, fl = o.env_step(int(acts_np[t, e]), 4)
            if fl or o.frame >= 120:
                o.fast_reset()
        of, od = o.core()
        assert np.array_equal(f[e], of), (tags[level_ids[e]], e)
        assert np.array_equal(di[e, :22], od[:22]), (tags[level_ids[e]], e)
        assert np.abs(out["game_state"][e, :40] - o.ninja_state().astype(np.float32)).max() <= 2e-6
    # ---- reachability of the sample vs the host build of the same function along the recorded rollout
    P = np.stack(hist_pos, axis=1)          # [sample, steps + 1, 2]
    FL = np.stack(hist_flags, axis=1)
    RF = np.stack(hist_feat, axis=1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    recomputed = astar_rows = 0
    for si, e in enumerate(sample):
        m = np.ascontiguousarray(levels[level_ids[e]])
        ents = compile_level_entities(m)
        total = int((ents[:, 0] == 1).sum())
        key, rows, pending = None, [], 1
        for t in range(steps + 1):
            fl = int(FL[si, t])
            ended = (fl & 11) != 0                                  # won / dead / truncated: the env was reset inside the step
            if ended:
                pending = 1
            sw = ((fl & 4) != 0) and not ended
            k = (int(P[si, t, 0] // 24), int(P[si, t, 1] // 24), sw)
            if k != key:
                key = k
                rows.append((t, pending))
                pending = 0
        ts = np.array([r[0] for r in rows])
        pos = np.ascontiguousarray(P[si, ts])
        ne = np.array([r[1] for r in rows], dtype=np.uint8)
        deadly = np.rint(RF[si, ts, 11].astype(np.float64) * total).astype(np.int32)
        mines = np.ascontiguousarray(np.stack([np.full(len(ts), total, np.int32), deadly], axis=1))
        exp = np.zeros((len(ts), 38), np.float32)
        st = np.zeros(len(ts), np.int32)
        assert lib.npp_reach_rollout_host(m.ctypes.data_as(C.POINTER(C.c_double)), m.size, p(pos), p(mines), p(ne), len(ts), p(exp), p(st),
                                          None) == 0
        assert not st.any()
        cur = 0
        for t in range(steps + 1):
            while cur + 1 < len(ts) and ts[cur + 1] <= t:
                cur += 1
            assert np.array_equal(RF[si, t], exp[cur]), (tags[level_ids[e]], int(e), t)
        recomputed += len(ts)
        astar_rows += len(ts) if level_ids[e] == astar else 0
    assert recomputed > 4 * len(sample) and astar_rows > 50
    # ---- frames of sampled envs: crop / reduction of the whole rendered frame
    xtab, ytab = _area_tabs(100, 1056.0 / 100, 1056), _area_tabs(176, 600.0 / 176, 600)
    gv = b.out.t["global_view"]
    for e in sample[::9]:
        full_t = b.render_frame(int(e), 1)[0, :, :, 0]
        assert torch.equal(gv[int(e), :, :, 0], _reduce_frame(full_t.float(), xtab, ytab)), int(e)
        full = full_t.cpu().numpy()
        px, py = out["positions"][e, 0], out["positions"][e, 1]
        r0, r1, c0, c1 = max(0, int(px - 42)), min(600, int(px + 42)), max(0, int(py - 42)), min(1056, int(py + 42))   # the axis swap
        hh, ww = max(0, r1 - r0), max(0, c1 - c0)
        ref = np.zeros((84, 84), np.uint8)
        if hh and ww:
            ref[(84 - hh) // 2:(84 - hh) // 2 + hh, (84 - ww) // 2:(84 - ww) // 2 + ww] = full[r0:r1, c0:c1]
        assert np.array_equal(out["player_frame"][e, :, :, 0], ref), int(e)
    # ---- switch_states: collected flags follow the entity states
    for e in sample[::5]:
        ents = compile_level_entities(levels[level_ids[e]])
        st_e = b.dump_entities(int(e))
        locked = np.flatnonzero(ents[:, 0] == 6)[:5]
        ss = out["switch_states"][e].reshape(5, 5)
        for j, slot in enumerate(locked):
            x, y = np.float32(np.clip(ents[slot, 1] / 1056.0, 0.0, 1.0)), np.float32(np.clip(ents[slot, 2] / 600.0, 0.0, 1.0))
            assert np.array_equal(ss[j], np.array([x, y, x, y, 0.0 if (st_e[slot] & 1) else 1.0], np.float32)), (int(e), j)
        assert not ss[len(locked):].any()
    b.close()


def test_bench_default_line_carries_the_other_configs():
    """`python bench.py` (N = 1, no --workload): value = config 2 and other_configs = configs 3, 4-shard and 5 with their own
    rooflines; config 5 runs on its whole level set."""
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--preroll", "40", "--envs-per-gpu",
                        "1024", "--other-steps", "12", "--no-cpu-baseline", "--async-streams", "0", "--open-loop-chunk", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 30 and line["value"] > 0 and "config 2" in line["config"]["workload"]
    oc = line["other_configs"]
    assert set(oc) == {"config3", "config4_shard", "config5_full_obs"}
    for k, blk in oc.items():
        assert blk["value"] > 0 and blk["steps"] == 12 and blk["roofline"]["frac"] > 0, k
    assert oc["config3"]["roofline_render"]["kernel"] == "npp_render_kernel" and oc["config3"]["roofline_render"]["frac"] > 0
    c5 = oc["config5_full_obs"]
    assert "0 level(s) dropped" in c5["config"]["workload"] and "21 levels" in c5["config"]["workload"]
    assert c5["roofline_global_view"]["frac"] > 0 and c5["roofline_reach"]["frac"] > 0
    assert set(c5["obs_kernels"]) >= {"player_frame", "global_view", "reachability"}   # switch_states comes out of the reachability launch
    assert c5["serial"]["value"] > 0 and c5["obs_overlap"][0]["cuts_percent"] == [50] and c5["value"] >= c5["serial"]["value"]


def test_bench_two_ranks_default_line_reports_config4():
    """`python bench.py --gpus 2` without --workload (what a SCALE run launches): value = config 2 on both ranks and a `config4`
    block = the mixed set with and without the observation gather.  Rehearsed on one GPU over gloo."""
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0", "--steps", "20",
                        "--warmup", "5", "--preroll", "10", "--envs-per-gpu", "1024", "--other-steps", "10", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and "config 2" in line["config"]["workload"] and line["value"] > 0
    c4 = line["config4"]
    assert "config 4" in c4["config"]["workload"] and c4["value"] > 0
    assert c4["with_obs_gather"]["own_shard_roundtrip_ok"] and c4["with_obs_gather"]["value"] > 0


def test_dynamic_truncation_limit_on_the_device():
    """npp_set_dynamic_truncation: every env truncates at its LEVEL's limit (int(clip(sqrt(surface area) * 500, 1200, 10000)),
    pinned against the reference function in test_host_cpu.py) and game_state[40] = max(0, (limit - frame) / limit); a reassigned
    env follows its new level; a manual limit overrides until the next assignment."""
    from nclone_amd.engine import NppBatch, level_truncation_limit
    from nclone_amd.levels import curriculum0_levels, door_levels

    cand = curriculum0_levels()[0][:40] + door_levels()[0]
    lims = [level_truncation_limit(m)[0] for m in cand]
    pick = [int(np.argmin(lims)), int(np.argsort(lims)[len(lims) // 2]), int(np.argmax(lims))]
    levels = [cand[i] for i in pick]
    L = [lims[i] for i in pick]
    assert L[0] == 1200 and L[0] < L[1] <= L[2]
    n = 192
    lvl = np.arange(n) // 64
    b = NppBatch(n, autoreset=True)
    b.load_levels(levels)
    b.assign_levels(lvl)
    b.set_dynamic_truncation(True)
    b.reset()
    noop = torch.zeros(n, dtype=torch.uint8, device="cuda")
    first_trunc = np.full(n, -1)
    frames_alive = np.zeros(n, dtype=np.int64)
    lim_e = np.array(L)[lvl]
    for s in range(L[1] // 4 + 8):
        b.step(noop)
        h = b.to_host(("flags", "frames", "game_state"))
        fl = h["flags"]
        frames_alive += h["frames"].astype(np.int64)
        trunc = (fl & 8) != 0
        done = (fl & 11) != 0
        # an env still in its episode: time_remaining of the observation = (limit - frame) / limit
        alive = ~done
        exp = np.maximum(0.0, (lim_e - frames_alive) / lim_e).astype(np.float32)
        assert np.array_equal(h["game_state"][alive, 40], exp[alive]), s
        assert not (trunc & (frames_alive < lim_e)).any(), s          # never before the level's limit
        first_trunc[(first_trunc < 0) & trunc] = s
        frames_alive[done] = 0
    # standing still nothing else ends the episode on these levels for at least some envs: they truncate exactly at the limit
    for k in (0, 1):
        sel = (lvl == k) & (first_trunc >= 0)
        assert sel.any() and (first_trunc[sel] == (L[k] + 3) // 4 - 1).all(), (k, L[k], np.unique(first_trunc[sel]))
    # reassignment: the first 64 envs move to the level with the middle limit; a manual limit holds until then
    b.set_truncation_limit(40)
    b.assign_levels(np.full(64, 1, dtype=np.int32), env_ids=np.arange(64, dtype=np.int32))
    b.reset()
    b.step(noop)
    gs = b.to_host(("game_state",))["game_state"]
    assert np.array_equal(gs[:64, 40], np.full(64, np.float32((L[1] - 4) / L[1])))
    assert np.array_equal(gs[64:128, 40], np.full(64, np.float32((40 - 4) / 40)))
    b.close()


def test_reset_checkpoint_option_and_seed():
    """NppVecEnvironment.reset(options={"checkpoint": ...}) (base_environment.py:1769-1789): an action sequence is replayed from the
    spawn (frame_skip ticks per action, as ActionReplayer does), "snapshot" restores the device-side copy -- both give the state
    the original episode had, bit for bit; seed= makes action_space_sample() reproducible."""
    from nclone_amd.levels import door_levels, mine_levels
    from nclone_amd.vec_env import NppVecEnvironment

    levels = mine_levels()[0][:3] + door_levels()[0][:3]
    n = 6 * 64
    v = NppVecEnvironment(levels, n, autoreset=False, truncation_limit=10000)
    v.reset(seed=7)
    acts = np.stack([v.action_space_sample() for _ in range(25)])
    v.reset(seed=7)
    assert np.array_equal(acts, np.stack([v.action_space_sample() for _ in range(25)]))
    for t in range(25):
        v.step(acts[t])
    f0, i0 = v.batch.dump_state()
    obs0 = {k: (x.clone() if isinstance(x, torch.Tensor) else x) for k, x in v.batch.out.t.items()}
    v.snapshot()
    for t in range(10):
        v.step(acts[t])
    f1, _ = v.batch.dump_state()
    assert not np.array_equal(f0, f1)
    obs, info = v.reset(options={"checkpoint": "snapshot"})
    f2, i2 = v.batch.dump_state()
    assert np.array_equal(f0, f2) and np.array_equal(i0[:, :26], i2[:, :26]) and info["restored_snapshot"]
    assert torch.equal(obs["game_state"], obs0["game_state"])
    # replay: per-env sequences [N, K] and one shared sequence
    obs, info = v.reset(options={"checkpoint": {"action_sequence": acts.T}})
    f3, i3 = v.batch.dump_state()
    assert info["checkpoint_replay"] and np.array_equal(f0, f3) and np.array_equal(i0[:, :26], i3[:, :26])
    assert torch.equal(obs["game_state"], obs0["game_state"])

    class Ckpt:                      # the attribute names of the reference's StateCheckpoint (state_checkpoint.py)
        action_sequence = [2, 2, 5, 5, 2, 0, 1, 4]
        source_frame_skip = 4

    obs, info = v.reset(options={"checkpoint": Ckpt()})
    f4, _ = v.batch.dump_state()
    v.reset()
    for a in Ckpt.action_sequence:
        v.step(np.full(n, a, dtype=np.uint8))
    f5, _ = v.batch.dump_state()
    assert np.array_equal(f4, f5)
    with pytest.raises(ValueError):
        v.reset(options={"checkpoint": {"action_sequence": np.zeros((3, 4), np.uint8)}})
    v.close()


def test_switch_states_match_the_reference_methods():
    """switch_states against the reference's OWN code: tests/golden/obs.npz holds what NppEnvironment._build_switch_states_array /
    _extract_locked_door_positions (npp_environment.py:1782-1847; run by tests/golden/make_golden_obs.py) return on the live
    locked-door entities along 400-step rollouts of the 21 door levels -- collected flags flipping as switches are touched and
    coming back after resets, the 'door' position falling back to the switch's, six doors cut to five.  Bit for bit, every step."""
    from nclone_amd.engine import NppBatch

    z = np.load(os.path.join(ROOT, "tests", "golden", "obs.npz"))
    names = bytes(z["names"]).decode().split("\n")
    n = len(names)
    assert n == 21
    b = NppBatch(n, autoreset=True, outputs=("positions", "switch_states"), fast_reset=False)   # the fixture resets with Simulator.reset
    b.load_levels([z["m%d" % k] for k in range(n)])
    b.assign_levels(np.arange(n))
    b.set_truncation_limit(100000)
    A = np.stack([z["a%d" % k] for k in range(n)])
    P = np.stack([z["p%d" % k] for k in range(n)])
    S = np.stack([z["s%d" % k] for k in range(n)])
    assert (S[:, :, 4::5] == 1).any() and (S[:, :, 4::5] == 0).any() and (S[19, 0, :25:5] > 0).all()
    b.observe()
    b.switch_states()
    h = b.to_host(("positions", "switch_states"))
    assert np.array_equal(h["switch_states"], S[:, 0])
    for t in range(A.shape[1]):
        b.step(torch.from_numpy(np.ascontiguousarray(A[:, t])).cuda())
        b.switch_states()
        h = b.to_host(("positions", "switch_states"))
        assert np.array_equal(h["positions"][:, :2], P[:, t + 1]), t
        bad = np.flatnonzero((h["switch_states"] != S[:, t + 1]).any(axis=1))
        assert len(bad) == 0, (t, [names[i] for i in bad])
    b.close()


@pytest.mark.parametrize("cuts", [12, 50, (6, 25, 50)])
def test_observation_overlap_produces_the_serial_bits(cuts):
    """npp_set_obs_overlap(_parts): the step's heavy-first order cut into two or four launches on streams of their own, one
    observation kernel per part.  Every
    output of the block -- and the state behind it -- equals the unsplit run's after every step, across auto-resets, a masked
    reset and a snapshot / restore (entry points that join the two streams themselves)."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import door_levels

    levels, _tags = door_levels()
    n = 2048
    level_ids = (np.arange(n) // 32) % len(levels)
    steps = 40
    acts = torch.from_numpy(np.random.default_rng(5).integers(0, 6, size=(steps, n)).astype(np.uint8)).cuda()
    runs = []
    for pct in (0, cuts):
        b = NppBatch(n, autoreset=True, fast_reset=True, outputs=FULL)
        b.load_levels(levels)
        b.assign_levels(level_ids)
        b.set_step_variant(1)          # a pinned build: the split starts with the first step (the autotuner never splits a timed launch)
        b.set_obs_overlap(pct)
        b.reset()
        b.set_truncation_limit(25)
        hist = []
        for t in range(steps):
            b.step(acts[t])
            b.switch_states()
            b.render_player_frame()
            b.render_global_view()
            b.reachability()
            if t == 10:
                b.snapshot()
            if t == 20:
                mask = np.zeros(n, np.uint8)
                mask[::3] = 1
                b.reset(mask)
            if t == 30:
                mask = np.zeros(n, np.uint8)
                mask[1::2] = 1
                b.restore(mask)
            if t % 3 == 0 or t in (10, 20, 30):
                hist.append({k: v.copy() for k, v in b.to_host().items()})
        f, di = b.dump_state()
        runs.append((hist, f, di))
        del b
    (h0, f0, d0), (h1, f1, d1) = runs
    assert np.array_equal(f0, f1, equal_nan=True) and np.array_equal(d0, d1)
    for a, c in zip(h0, h1):
        assert a.keys() == c.keys()
        for k in a:
            assert np.array_equal(a[k], c[k], equal_nan=True), k


def test_reachability_ex_writes_the_switch_states_of_the_separate_kernel():
    """npp_reachability_ex: switch_states from the reachability launch = npp_switch_states's output, and the other outputs are
    those of npp_reachability (door levels, locked doors collected along the way)."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import door_levels

    levels, _tags = door_levels()
    n = 1024
    outs = ("switch_states", "reachability_features", "mine_sdf_features", "reach_status")
    acts = torch.from_numpy(np.random.default_rng(9).integers(0, 6, size=(120, n)).astype(np.uint8)).cuda()
    a, b = (NppBatch(n, autoreset=True, outputs=outs) for _ in range(2))
    for x in (a, b):
        x.load_levels(levels)
        x.assign_levels((np.arange(n) // 16) % len(levels))
        x.reset()
    seen_collected = False
    for t in range(120):
        a.step(acts[t]); a.switch_states(); a.reachability()
        b.step(acts[t]); b.reachability(with_switch_states=True)
        if t % 8 == 7:
            ha, hb = a.to_host(), b.to_host()
            for k in outs:
                assert np.array_equal(ha[k], hb[k]), (t, k)
            seen_collected = seen_collected or bool((ha["switch_states"].reshape(n, 5, 5)[:, :, 4] == 0).any() and (ha["switch_states"] != 0).any())
    assert (ha["switch_states"] != 0).any()


def test_vec_env_with_observation_overlap_returns_the_same_dict():
    """NppVecEnvironment(obs_overlap=...) -- split step, one observation kernel per part, switch_states out of the reachability
    launch, join before the Dict is handed out -- returns the observations, rewards and flags of the plain env, step after step."""
    from nclone_amd.levels import door_levels
    from nclone_amd.vec_env import NppVecEnvironment

    levels, _tags = door_levels()
    n = 1024
    kw = dict(enable_visual_observations=True, enable_spatial_context=True, enable_switch_states=True, enable_reachability=True,
              output="numpy", truncation_limit=60)
    a = NppVecEnvironment(levels, n, **kw)
    b = NppVecEnvironment(levels, n, obs_overlap=45, **kw)
    b.batch.set_step_variant(0)   # pinned: the split starts with the first step
    a.batch.set_step_variant(0)
    oa, _ = a.reset(seed=1)
    ob, _ = b.reset(seed=1)
    acts = np.random.default_rng(2).integers(0, 6, size=(50, n))
    for t in range(50):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        for k in ra[0]:
            assert np.array_equal(ra[0][k], rb[0][k], equal_nan=True), (t, k)
        assert np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]) and np.array_equal(ra[3], rb[3]), t
    assert set(ra[0]) >= {"player_frame", "global_view", "switch_states", "reachability_features", "spatial_context"}
    a.close(); b.close()


def test_observation_overlap_argument_checks():
    """npp_set_obs_overlap_parts refuses cuts that are not ascending percentages in (0, 100) or more than three of them; switching
    the overlap off and on again keeps working (streams are chosen once per caller stream)."""
    from nclone_amd import _native as nat
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import door_levels

    levels, _ = door_levels()
    b = NppBatch(256, autoreset=True, outputs=("player_frame",))
    b.load_levels(levels[:4])
    b.assign_levels((np.arange(256) // 64) % 4)
    lib = nat.lib()
    for bad in ([0], [100], [50, 40], [10, 10], [10, 20, 30, 40]):
        arr = (C.c_int * len(bad))(*bad)
        assert lib.npp_set_obs_overlap_parts(b.h, arr, len(bad)) == 1, bad          # NPP_ERR_INVALID
    assert lib.npp_set_obs_overlap(b.h, 100) == 1 and lib.npp_set_obs_overlap(b.h, -1) == 1
    acts = torch.zeros(256, dtype=torch.uint8, device="cuda")
    b.set_step_variant(1)
    for cuts in (30, 0, (20, 60), 0, 50):
        b.set_obs_overlap(cuts)
        for _ in range(3):
            b.step(acts)
            b.render_player_frame()
        b.join()
    b.sync()
    b.close()
