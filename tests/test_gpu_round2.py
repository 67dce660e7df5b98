"""Round-2 GPU tests: config 4's mixed level set at one GPU's shard size, the boundary additions (pass-through position
scalars, single-block host staging, device guard, npp_reset_ex), the zoo-block regression of the round-1 fault, snapshot /
restore together with repositioned entities, and the asynchronous vector env."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N = 8192


def test_config4_mixed_set_shard_vs_oracle(oracle_mod):
    """One GPU's shard of config 4 (8192 of the 65 536 envs) on the 512-level mixed set: replicas hold identical bits, a
    192-env sample re-simulated by the CPU oracle matches bit for bit, G = 1 equals G = 16."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import c3_mixed_levels

    levels, tags = c3_mixed_levels()
    assert len(levels) == 512
    # shard 3 of 8: global env index decides the level, as in bench.py
    level_ids = ((np.arange(N) + 3 * N) // 64) % len(levels)
    steps = 60
    acts_np = np.random.default_rng(2).integers(0, 6, size=(steps, N)).astype(np.uint8).reshape(steps, N // 64, 64)
    acts_np[:, :, 32:] = acts_np[:, :, :32]
    acts_np = acts_np.reshape(steps, N)
    acts = torch.from_numpy(acts_np).cuda()

    def run(g, nsteps):
        b = NppBatch(N, autoreset=True, outputs=("work",))
        b.load_levels(levels)
        if g:
            b.set_launch_geometry(g, 0)
        b.assign_levels(level_ids)
        hist, work = [], []
        for s in range(nsteps):
            b.step(acts[s])
            hist.append(b.flags.clone())
            work.append(b.work.clone())
        f, i = b.dump_state()
        return b, f, i, torch.stack(hist).cpu().numpy(), torch.stack(work).cpu().numpy()

    b, f, i, flags, work = run(0, steps)
    fb, ib = f.reshape(N // 64, 64, -1), i.reshape(N // 64, 64, -1)
    assert np.array_equal(fb[:, :32], fb[:, 32:]) and np.array_equal(ib[:, :32, :27], ib[:, 32:, :27])
    assert work.max() > 0 and work.max() <= 4 * 4 * 32       # 4 ticks x 4 substeps x 32 iterations at most
    sample = np.random.default_rng(5).choice(N, size=192, replace=False)
    for e in sample:
        o = oracle_mod.Oracle("mul")
        assert o.load(levels[level_ids[e]]) == 0
        for s in range(steps):
            k, fl = o.env_step(int(acts_np[s, e]), 4)
            got = int(flags[s, e])
            assert (1 if got & 1 else (2 if got & 2 else 0)) == fl, (tags[level_ids[e]], e, s)
            if fl or o.frame >= 10000:
                o.reset()
        of, od = o.core()
        assert np.array_equal(f[e], of), (tags[level_ids[e]], e)
        assert np.array_equal(i[e, :22], od[:22]), (tags[level_ids[e]], e)
        assert np.array_equal(b.dump_entities(int(e)), o.entity_states()), (tags[level_ids[e]], e)
    assert (flags & 3).any()
    _, f1, i1, fl1, w1 = run(1, 20)
    _, f16, i16, fl16, w16 = run(16, 20)
    assert np.array_equal(f1, f16) and np.array_equal(i1, i16) and np.array_equal(fl1, fl16) and np.array_equal(w1, w16)


def test_passthrough_positions_and_single_block_staging(golden):
    """player_x/y, switch_x/y, exit_door_x/y, switch_activated (observation_processor.py:374-399) are the unrounded fp64
    values; output="numpy" stages every enabled output with one copy into alternating pinned blocks."""
    from nclone_amd.levels import curriculum0_levels
    from nclone_amd.vec_env import NppEnvironment, NppVecEnvironment

    levels, _ = curriculum0_levels()
    n = 256
    v = NppVecEnvironment(levels[:4], n, output="numpy", enable_spatial_context=True)
    assert "spatial_context" in v.observation_space.spaces
    obs0, _ = v.reset()
    keys = {"game_state", "action_mask", "entity_positions", "spatial_context", "player_x", "player_y", "switch_x", "switch_y",
            "exit_door_x", "exit_door_y", "switch_activated"}
    assert set(obs0) == keys
    f, i = v.batch.dump_state()
    assert obs0["player_x"].dtype == np.float64 and np.array_equal(obs0["player_x"], f[:, 0]) and np.array_equal(obs0["player_y"], f[:, 1])
    assert np.array_equal(obs0["entity_positions"][:, 2], (obs0["switch_x"] / 1056.0).astype(np.float32))
    assert np.array_equal(obs0["entity_positions"][:, 5], (obs0["exit_door_y"] / 600.0).astype(np.float32))
    assert not obs0["switch_activated"].any()
    x0 = obs0["player_x"].copy()
    obs1, rew, term, trunc, info = v.step(np.full(n, 2, dtype=np.uint8))
    f, _ = v.batch.dump_state()
    assert np.array_equal(obs1["player_x"], f[:, 0]) and (obs1["player_x"] != x0).any()
    assert np.array_equal(obs0["player_x"], x0)      # the previous observation's staging block is still intact
    assert rew.shape == (n,) and info["frames_executed"].shape == (n,) and info["terminal_observation"].shape == (n, 41)
    v.close()
    e = NppEnvironment(map_data=levels[0])
    o, _ = e.reset()
    assert isinstance(o["player_x"], float) and isinstance(o["switch_activated"], bool) and o["game_state"].shape == (41,)
    e.close()


def test_zoo_block_covers_locked_door_levels():
    """Regression for the round-1 GPU fault: the per-env zoo block was sized over the levels WITH moving
    entities only, while the reset kernel initialises the doors of every level -- a locked-door-only level with more doors
    than any zoo level overran its block.  Plain set -> set_entity_pos -> a set whose locked-door levels have more doors
    than its zoo level; every env's block must stay inside its own words."""
    from nclone_amd import _native as nat
    from nclone_amd.engine import NppBatch, _as_f64_blob
    from nclone_amd.levels import curriculum0_levels, door_levels, zoo_levels

    c0, _ = curriculum0_levels()
    doors, _ = door_levels()
    zoo, _ = zoo_levels()
    lib = nat.lib()

    def plan(levels):
        blob, off = _as_f64_blob(levels)
        d, m, w = C.c_int(), C.c_int(), C.c_int()
        nat.check(None, lib.npp_plan_zoo_block(blob.ctypes.data_as(C.POINTER(C.c_double)), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                               len(levels), C.byref(d), C.byref(m), C.byref(w)))
        return d.value, m.value, w.value

    per_door = [plan([lv])[0] for lv in doors]
    zoo_pick = min(range(len(zoo)), key=lambda k: plan([zoo[k]])[0])
    many = int(np.argmax(per_door))
    assert per_door[many] > plan([zoo[zoo_pick]])[0]          # the shape of the round-1 failure
    n = 256
    b = NppBatch(n, autoreset=True)
    b.load_levels(c0[:2])
    b.assign_levels(np.arange(n) % 2)
    b.set_entity_pos(5, 0, 300.0, 300.0)                      # zoo kernels now run on a set without any zoo level
    acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(30, n)).astype(np.uint8)).cuda()
    for s in range(10):
        b.step(acts[s])
    mixed = [zoo[zoo_pick], doors[many], doors[(many + 1) % len(doors)]]
    b.load_levels(mixed)
    lv = np.arange(n) % 3
    b.assign_levels(lv)
    for s in range(30):
        b.step(acts[s])
    b.reset()
    b.sync()
    # the same envs on their own (no neighbours to be overrun by) must agree bit for bit
    for k in range(3):
        s1 = NppBatch(64, autoreset=True)
        s1.load_levels(mixed)
        s1.assign_levels(np.full(64, k))
        sel = np.nonzero(lv == k)[0][:64]
        b2 = NppBatch(n, autoreset=True)
        b2.load_levels(mixed)
        b2.assign_levels(lv)
        for s in range(30):
            s1.step(acts[s][torch.from_numpy(sel).cuda()].contiguous())
            b2.step(acts[s])
        f1, i1 = s1.dump_state()
        f2, i2 = b2.dump_state()
        assert np.array_equal(f1, f2[sel]) and np.array_equal(i1[:, :27], i2[sel, :27]), k
        assert np.array_equal(s1.entity_checksum(), b2.entity_checksum()[sel]), k


def test_snapshot_restore_keeps_repositioned_entities():
    """npp_snapshot / npp_restore with npp_set_entity_pos in between: the restored zoo block carries the
    override again, and the host must dispatch the zoo kernels again -- the moved switch keeps being honoured."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import curriculum0_levels

    levels, _ = curriculum0_levels()
    n = 128

    def fresh():
        b = NppBatch(n, autoreset=False)
        b.load_levels(levels[:2])
        b.assign_levels(np.arange(n) // 64)
        return b

    acts = torch.from_numpy(np.random.default_rng(3).integers(0, 6, size=(80, n)).astype(np.uint8)).cuda()
    a = fresh()
    f0, _ = a.dump_state()
    sx, sy = float(f0[7, 0]) + 30.0, float(f0[7, 1])          # put env 7's switch right next to its spawn
    a.set_entity_pos(7, 0, sx, sy)
    a.snapshot()
    a.set_entity_pos(7, 0, float("nan"), float("nan"))         # clear: host n_ovr -> 0, plain kernels would run
    a.restore()
    for s in range(80):
        a.step(acts[s])
    ref = fresh()
    ref.set_entity_pos(7, 0, sx, sy)
    for s in range(80):
        ref.step(acts[s])
    fa, ia = a.dump_state()
    fr, ir = ref.dump_state()
    assert np.array_equal(fa, fr) and np.array_equal(ia[:, :27], ir[:, :27])
    assert np.array_equal(a.entity_checksum(), ref.entity_checksum())
    # error paths leave the assignment untouched
    from nclone_amd._native import NppError

    with pytest.raises(NppError):
        a.assign_levels([0, 1, 99], env_ids=[0, 1, 2])
    _, i2 = a.dump_state()
    assert (i2[:, 27] == np.arange(n) // 64).all()


def test_async_vec_env_matches_sync():
    """NppAsyncVecEnvironment (S sub-batches on S HIP streams, step_async / step_wait) returns bit-identical results to the
    synchronous NppVecEnvironment on the same envs, levels and actions, including auto-resets and partial waits."""
    from nclone_amd.async_env import NppAsyncVecEnvironment
    from nclone_amd.levels import mine_levels
    from nclone_amd.vec_env import NppVecEnvironment

    levels, _ = mine_levels()
    n = 1024
    lv = (np.arange(n) // 64) % 12
    sync = NppVecEnvironment(levels[:12], n, level_ids=lv, output="numpy")
    asy = NppAsyncVecEnvironment(levels[:12], n, n_streams=4, level_ids=lv, output="numpy")
    o1, _ = sync.reset()
    o2, _ = asy.reset()
    for k in ("game_state", "action_mask", "entity_positions", "player_x"):
        assert np.array_equal(o1[k], o2[k]), k
    rng = np.random.default_rng(11)
    done = 0
    for s in range(150):
        a = rng.integers(0, 6, size=n).astype(np.uint8)
        r1 = sync.step(a)
        if s % 3 == 0:      # whole-batch protocol
            r2 = asy.step(a)
        else:               # partial protocol: enqueue all, collect sub-batches out of order
            asy.step_async(a)
            parts = {k: asy.step_wait_partial(k) for k in (2, 0, 3, 1)}
            r2 = tuple(asy._cat([parts[k][i] for k in range(4)]) for i in range(5))
        for k in ("game_state", "action_mask", "entity_positions", "player_x", "player_y", "switch_activated"):
            assert np.array_equal(r1[0][k], r2[0][k]), (s, k)
        assert np.array_equal(r1[1], r2[1]) and np.array_equal(r1[2], r2[2]) and np.array_equal(r1[3], r2[3]), s
        assert np.array_equal(r1[4]["frames_executed"], r2[4]["frames_executed"])
        done += int(r1[2].sum())
    assert done > 0
    sync.close()
    asy.close()


def test_device_guard_and_reset_modes():
    """Entry points run with the handle's device current and leave the caller's current device as it was;
    npp_reset_ex validates its mode."""
    from nclone_amd import _native as nat
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import curriculum0_levels

    levels, _ = curriculum0_levels()
    before = torch.cuda.current_device()
    b = NppBatch(64, device=0)
    b.load_levels(levels[:1])
    acts = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    b.step(acts)
    b.sync()
    assert torch.cuda.current_device() == before
    assert nat.lib().npp_reset_ex(b.h, None, 7) == 1          # NPP_ERR_INVALID
    b.reset(mode="full")
    b.reset(mode="fast")
    b.reset()
    f, i = b.dump_state()
    assert (i[:, 22] == 0).all()
    b.close()


@pytest.mark.parametrize("fixture", ["fast", "official"])
def test_fast_reset_vs_reference_fixtures(golden, fixture):
    """Simulator.fast_reset (nsim.py:78-140) on the device against the reference's own runs (tests/golden/fast.npz: 58 rollouts
    on mine / locked-door / mine-soup levels and on every zoo map, resets forced mid-flight, fast and full resets mixed;
    official.npz: 15 rollouts on the reference's five official tutorial levels, `nclone/maps/test-maps/`):
    every step's final ninja state within the north-star bars (f32 positions within 1e-5, discrete state identical), the
    entity checksum (positions, speeds and state codes of every entity: movers that keep going across a fast reset),
    game_state and action mask."""
    from nclone_amd.engine import NppBatch

    g = golden.z(fixture)
    names = golden.names(fixture)
    n = len(names)
    levels = [g["m%d" % r] for r in range(n)]
    b = NppBatch(n, autoreset=False)
    b.load_levels(levels)
    b.assign_levels(np.arange(n))
    steps = [len(g["a%d" % r]) for r in range(n)]
    acts = np.zeros((max(steps), n), dtype=np.uint8)
    for r in range(n):
        acts[:steps[r], r] = g["a%d" % r]
    d_acts = torch.from_numpy(acts).cuda()
    row = np.zeros(n, dtype=np.int64)
    checked = 0
    for s in range(max(steps)):
        b.step(d_acts[s])
        f, i = b.dump_state()
        cs = b.entity_checksum()
        gs = b.game_state.cpu().numpy()
        mk = b.action_mask.cpu().numpy()
        fr = b.frames.cpu().numpy()
        full = np.zeros(n, dtype=np.uint8)
        fast = np.zeros(n, dtype=np.uint8)
        for r in range(n):
            if s >= steps[r]:
                continue
            ex, term, frame, mode = (int(v) for v in g["s%d" % r][s])
            assert int(fr[r]) == ex, (names[r], s)
            row[r] += ex
            T, D, E = g["t%d" % r][row[r] - 1], g["d%d" % r][row[r] - 1], g["e%d" % r][row[r] - 1]
            assert np.abs(f[r, :2].astype(np.float32) - T[:2].astype(np.float32)).max() <= 1e-5, (names[r], s, f[r, :4], T)
            assert np.array_equal(f[r, :4], T), (names[r], s, f[r, :4], T)      # in fact the same bits on this corpus
            assert np.array_equal(i[r, :20].clip(0, 255), D), (names[r], s)
            assert np.allclose(cs[r], E, rtol=0, atol=1e-9) and np.array_equal(cs[r, 4:], E[4:]), (names[r], s, cs[r], E)
            assert np.abs(gs[r, :40] - g["g%d" % r][s]).max() <= 2e-6, (names[r], s)
            assert int(sum(int(v) << k for k, v in enumerate(mk[r]))) == int(g["k%d" % r][s]), (names[r], s)
            full[r] = mode == 1
            fast[r] = mode == 2
            checked += 1
        if full.any():
            b.reset(full, mode="full")
        if fast.any():
            b.reset(fast, mode="fast")
    assert checked == sum(steps)


@pytest.mark.parametrize("explicit_first_reset", [True, False])
def test_fast_reset_autoreset_vs_oracle(oracle_mod, explicit_first_reset):
    """In-kernel auto-reset under NPP_FLAG_FAST_RESET (what NppVecEnvironment uses by default) on zoo, mine and door levels, in
    the reference env's own sequence: load_map at construction (base_environment.py:318), the FIRST reset() reloads
    the map = Simulator.reset (npp_environment.py:518-557, _last_reset_map_name is None), every later same-level reset is
    Simulator.fast_reset.  explicit_first_reset=True: npp_reset (mode 0) plays that first reset() and every auto-reset is fast;
    False: nobody calls reset, so the first AUTO-reset of each env is the full one.  Bit-identical to the oracle doing
    load -> reset() -> ... -> fast_reset(); and different from the full-reset handle on the zoo levels (movers keep going)."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import door_levels, mine_levels, zoo_levels

    levels = zoo_levels()[0][:10] + mine_levels()[0][:6] + door_levels()[0][:4]
    n = 4 * len(levels)
    lvl = np.arange(n) % len(levels)
    steps = 160
    acts = np.random.default_rng(17).integers(0, 6, size=(steps, n)).astype(np.uint8)
    d_acts = torch.from_numpy(acts).cuda()

    def run(fast):
        b = NppBatch(n, autoreset=True, fast_reset=fast)
        b.load_levels(levels)
        b.assign_levels(lvl)
        if explicit_first_reset:
            b.reset()
        b.set_truncation_limit(150)          # forces resets mid-flight on top of deaths
        for s in range(steps):
            b.step(d_acts[s])
        return b

    bf = run(True)
    f, i = bf.dump_state()
    cs = bf.entity_checksum()
    resets = 0
    for e in range(n):
        o = oracle_mod.Oracle("mul")
        o.load(levels[lvl[e]])
        had_full = False
        if explicit_first_reset:
            o.reset()
            had_full = True
        for s in range(steps):
            k, fl = o.env_step(int(acts[s, e]), 4)
            if fl or o.frame >= 150:
                if had_full:
                    o.fast_reset()
                else:
                    o.reset()
                    had_full = True
                resets += 1
        of, od = o.core()
        assert np.array_equal(f[e], of), (e, lvl[e], f[e], of)
        assert np.array_equal(i[e, :22], od[:22]), (e, lvl[e])
        assert np.array_equal(cs[e], o.entity_checksum()), (e, lvl[e], cs[e], o.entity_checksum())
    assert resets > n
    bs = run(False)
    cs2 = bs.entity_checksum()
    assert not np.array_equal(cs[lvl < 10], cs2[lvl < 10])


def test_step_variant_autotuner_reexamines_its_decision():
    """The tuner measures again after NPP_TUNE_AGAIN launches (16 384 by default; shortened here through the environment, in a child
    process because the library reads it once): two full tuning cycles inside 2000 launches, same final bits as a pinned run."""
    import subprocess
    import sys

    code = r'''
import numpy as np, torch
from nclone_amd.engine import NppBatch
from nclone_amd.levels import curriculum0_levels
levels, _ = curriculum0_levels()
n = 1024
lvl = (np.arange(n) // 64) % len(levels)
acts = torch.from_numpy(np.random.default_rng(6).integers(0, 6, size=(2000, n)).astype(np.uint8)).cuda()
outs = []
for pin in (-1, 0):
    b = NppBatch(n, autoreset=True)
    b.load_levels(levels); b.assign_levels(lvl); b.set_launch_geometry(16, 4); b.set_step_variant(pin)
    seen = set()
    for s in range(2000):
        b.step(acts[s])
        if s % 50 == 0:
            torch.cuda.synchronize()
            seen.add(b.step_variant())
    torch.cuda.synchronize()
    outs.append(b.dump_state())
    if pin < 0:
        assert any(t for _, t in seen) and any(not t for _, t in seen), seen   # both tuning and decided phases were observed
    b.close()
assert all(np.array_equal(x, y) for x, y in zip(*outs))
print("OK")
'''
    env = dict(os.environ, NPP_TUNE_AGAIN="300")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_step_variant_autotuner_decides_and_keeps_the_bits():
    """npp_step's autotuner (HIP-event windows over the three G = 16 build variants) reaches a decision without any synchronisation on
    the caller's side, and a run that goes through the tuning windows ends in the same state as a run pinned to variant 0."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import curriculum0_levels

    levels, _ = curriculum0_levels()
    n = 2048
    lvl = (np.arange(n) // 64) % len(levels)
    acts = torch.from_numpy(np.random.default_rng(5).integers(0, 6, size=(720, n)).astype(np.uint8)).cuda()
    outs = []
    for pin in (-1, 0):
        b = NppBatch(n, autoreset=True)
        b.load_levels(levels)
        b.assign_levels(lvl)
        b.set_launch_geometry(16, 4)
        b.set_step_variant(pin)
        for s in range(720):
            b.step(acts[s])
        torch.cuda.synchronize()
        b.step(acts[0])   # one more launch after the last window's events have completed: the tuner decides here
        v, tuned = b.step_variant()
        assert tuned and v in (0, 1, 2), (pin, v, tuned)
        outs.append(b.dump_state() + (b.game_state.cpu().numpy(),))
        b.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
