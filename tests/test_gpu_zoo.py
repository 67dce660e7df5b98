"""GPU parity for the entity zoo: regular / trap doors, launch pads, one-way platforms, zap and
mini drones, bounce blocks, thwumps, boost pads, death balls, shove thwumps.

Fixtures: tests/golden/zoo.npz (make_golden_zoo.py, produced by running the reference): the 26 bc_replays whose maps
hold those entities and 9 random-action rollouts on such maps, with a per-tick checksum over every entity.
The HIP path is compared BIT-FOR-BIT with the oracle's multiply-square twin (ninja state, discrete state, entity
checksum) and with the reference fixtures under the north-star bars (f32 positions within 1e-5, discrete state
identical); libm's pow is not correctly rounded, so death-ball speeds of the twin can differ from the reference in the
last bit (measured: <= 2e-15 on these fixtures, ninja trajectories identical).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

POS_TOL = 1e-5
GS_TOL = 2e-6
ENT_TOL = 1e-9


def _batch(n, **kw):
    from nclone_amd.engine import NppBatch

    return NppBatch(n, **kw)


def _disc(i):
    return i[:, :20].clip(0, 255).astype(np.uint8)


def _run_replays(golden, oracle_mod, idx, geometry=None, copies=1):
    c, z = golden.z("corpus"), golden.z("zoo")
    n = len(idx) * copies
    b = _batch(n, autoreset=False)
    b.load_levels([c["m%d" % i] for i in idx])
    if geometry:
        b.set_launch_geometry(*geometry)
    b.assign_levels(np.arange(n) % len(idx))          # first creation: replay semantics (no reset after the load)
    sims = []
    for i in idx:
        o = oracle_mod.Oracle("mul")
        o.load(c["m%d" % i].astype(np.float64))
        sims.append(o)
    T = [len(z["t%d" % i]) for i in idx]
    tmax = max(T)
    inputs = np.zeros((tmax, n), dtype=np.uint8)
    for e in range(n):
        k = e % len(idx)
        inputs[: T[k], e] = c["in%d" % idx[k]][: T[k]]
    d_in = torch.from_numpy(inputs).cuda()
    final = c["final"]
    ticks = 0
    for tick in range(tmax):
        b.tick(d_in[tick : tick + 1])
        f, di = b.dump_state()
        cs = b.entity_checksum()
        for k, i in enumerate(idx):
            if tick >= T[k]:
                continue
            h, j = oracle_mod.controls(int(inputs[tick, k]))
            sims[k].tick(h, j)
            of, od = sims[k].core()
            oc = sims[k].entity_checksum()
            for e in range(k, n, len(idx)):
                assert np.array_equal(f[e], of), (i, tick, e, f[e], of)
                assert np.array_equal(di[e, :22], od[:22]), (i, tick, e)
                assert np.array_equal(cs[e], oc), (i, tick, e, cs[e] - oc)
            # reference fixtures
            ref = z["t%d" % i][tick]
            d32 = np.abs(f[k, :4].astype(np.float32).astype(np.float64) - ref.astype(np.float32).astype(np.float64)).max()
            assert d32 <= POS_TOL, (i, tick, d32)
            assert np.array_equal(_disc(di[k : k + 1])[0], z["d%d" % i][tick]), (i, tick)
            assert np.abs(cs[k] - z["e%d" % i][tick]).max() <= ENT_TOL, (i, tick)
            if tick == T[k] - 1:
                assert int(final[i, 0]) == T[k] and di[k, 0] == int(final[i, 1])
                assert np.array_equal(f[k, :2], final[i, 2:4])
            ticks += 1
    return ticks, b.launch_geometry()


def test_zoo_replays_bit_exact(golden, oracle_mod):
    """All 26 zoo bc_replays, one env each, every tick: with the 104 others of test_gpu_parity.py that is 130 of 130."""
    z = golden.z("zoo")
    ticks, geo = _run_replays(golden, oracle_mod, [int(i) for i in z["idx"]])
    print("zoo replay ticks %d, geometry %s" % (ticks, geo))
    assert ticks == 5886


@pytest.mark.parametrize("geometry", [(1, 1), (4, 2), (64, 1)])
def test_zoo_lane_group_geometries(golden, oracle_mod, geometry):
    """Lane ownership of movers, ballots for list-order numbers and the merged neighbourhood walk must not depend on G.
    (The host may raise G when the zoo blocks of 64/G envs per wavefront do not fit the LDS budget.)"""
    z = golden.z("zoo")
    idx = [int(i) for i in z["idx"]][:6]
    ticks, geo = _run_replays(golden, oracle_mod, idx, geometry=geometry, copies=3)
    print("requested", geometry, "ran", geo)


def test_zoo_rollouts_step_api(golden):
    """npp_step with in-kernel auto-reset on 9 zoo maps against reference rollouts (hp.reset() on termination).  The
    first episode runs with the first creation of the entities (ball-ball repulsion on), later ones do not."""
    z = golden.z("zoo")
    n = int(z["n_rollouts"][0])
    b = _batch(n, autoreset=True)
    b.load_levels([z["rm%d" % r] for r in range(n)])
    b.assign_levels(np.arange(n))
    steps = len(z["ra0"])
    acts = np.stack([z["ra%d" % r] for r in range(n)], axis=1)
    d_acts = torch.from_numpy(acts).cuda()
    rows = np.zeros(n, dtype=np.int64)
    episodes = 0
    for s in range(steps):
        b.step(d_acts[s], frame_skip=4)
        b.sync()
        flags = b.flags.cpu().numpy()
        frames = b.frames.cpu().numpy().astype(np.int64)
        gs = b.game_state.cpu().numpy()
        term_gs = b.terminal_state.cpu().numpy()
        mask = b.action_mask.cpu().numpy()
        f, di = b.dump_state()
        cs = b.entity_checksum()
        for r in range(n):
            ex, term, frame = z["rs%d" % r][s]
            assert frames[r] == ex, (r, s, frames[r], ex)
            kind = 1 if flags[r] & 1 else (2 if flags[r] & 2 else 0)
            assert kind == term, (r, s, flags[r], term)
            rows[r] += ex
            got = term_gs[r] if term else gs[r]
            assert np.abs(got[:40] - z["rg%d" % r][s]).max() <= GS_TOL, (r, s)
            if not term:
                assert list(mask[r]) == [(int(z["rk%d" % r][s]) >> k) & 1 for k in range(6)], (r, s)
                ref = z["rt%d" % r][rows[r] - 1]
                assert np.abs(f[r, :4] - ref).max() <= POS_TOL, (r, s, f[r, :4], ref)
                assert np.array_equal(_disc(di[r : r + 1])[0], z["rd%d" % r][rows[r] - 1]), (r, s)
                assert np.abs(cs[r] - z["re%d" % r][rows[r] - 1]).max() <= ENT_TOL, (r, s, cs[r] - z["re%d" % r][rows[r] - 1])
            else:
                episodes += 1
                assert di[r, 22] == 0 and di[r, 0] == 0
    assert episodes >= 10
    print("zoo rollout steps %d x %d envs, episodes %d" % (steps, n, episodes))


def test_zoo_mixed_with_plain_levels_and_checkpoint(golden, oracle_mod):
    """Zoo and plain levels in one batch (interleaved inside wavefronts), snapshot / restore of the zoo blocks, and the
    oracle twin as referee at the end."""
    c, z, r = golden.z("corpus"), golden.z("zoo"), golden.z("rollouts")
    zi = [int(i) for i in z["idx"]][:5]
    levels = [c["m%d" % i].astype(np.float64) for i in zi] + [r["m%d" % k] for k in range(3)]
    n = 96
    b = _batch(n, autoreset=False)
    b.load_levels(levels)
    lvl = np.arange(n) % len(levels)
    b.assign_levels(lvl)
    rng = np.random.default_rng(17)
    acts = rng.integers(0, 6, size=(60, n)).astype(np.uint8)
    d = torch.from_numpy(acts).cuda()
    for s in range(30):
        b.step(d[s])
    b.snapshot()
    f0, i0 = b.dump_state()
    c0 = b.entity_checksum()
    for s in range(30, 60):
        b.step(d[s])
    f1, i1 = b.dump_state()
    c1 = b.entity_checksum()
    b.restore()
    fr, ir = b.dump_state()
    assert np.array_equal(fr, f0) and np.array_equal(ir, i0) and np.array_equal(b.entity_checksum(), c0)
    for s in range(30, 60):
        b.step(d[s])
    f2, i2 = b.dump_state()
    assert np.array_equal(f2, f1) and np.array_equal(i2, i1) and np.array_equal(b.entity_checksum(), c1)
    # referee: the oracle twin stepping the same actions (terminal envs are not stepped again without auto-reset)
    for e in range(0, n, 7):
        o = oracle_mod.Oracle("mul")
        o.load(levels[lvl[e]])
        done = False
        for s in range(60):
            if not done:
                _, fl = o.env_step(int(acts[s, e]), 4)
                done = fl != 0
        of, od = o.core()
        assert np.array_equal(f1[e], of), (e, lvl[e])
        assert np.array_equal(i1[e, :22], od[:22]), (e, lvl[e])
        assert np.array_equal(c1[e], o.entity_checksum()), (e, lvl[e])


def test_zoo_full_size_sample_vs_oracle(oracle_mod):
    """8192 envs on the 26 zoo maps (64 consecutive envs per map), 40 random-action steps with auto-reset; a sample of envs
    is replayed on the oracle twin and must match bit for bit (ninja state and entity checksum); determinism across a
    second run of the same inputs."""
    from nclone_amd.levels import zoo_levels

    levels, _ = zoo_levels()
    n = 8192
    lvl = (np.arange(n) // 64) % len(levels)
    rng = np.random.default_rng(99)
    acts = rng.integers(0, 6, size=(40, n)).astype(np.uint8)
    d = torch.from_numpy(acts).cuda()
    runs = []
    for rep in range(2):
        b = _batch(n, autoreset=True)
        b.load_levels(levels)
        b.assign_levels(lvl)
        resets = np.zeros(n, dtype=np.int64)
        for s in range(40):
            b.step(d[s])
            resets += (b.flags.cpu().numpy() & 11) != 0
        f, i = b.dump_state()
        runs.append((f, i, b.entity_checksum(), resets))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    f, i, cs, resets = runs[0]
    checked = 0
    for e in list(range(0, n, 97)) + list(np.nonzero(resets)[0][:20]):
        o = oracle_mod.Oracle("mul")
        o.load(levels[lvl[e]])
        for s in range(40):
            _, fl = o.env_step(int(acts[s, e]), 4)
            if fl or o.frame >= 10000:
                o.reset()
        of, od = o.core()
        assert np.array_equal(f[e], of), (e, lvl[e])
        assert np.array_equal(i[e, :22], od[:22]), (e, lvl[e])
        assert np.array_equal(cs[e], o.entity_checksum()), (e, lvl[e])
        checked += 1
    print("zoo full size: %d envs, %d checked against the oracle, %d episodes ended" % (n, checked, int(resets.sum())))


def test_all_130_replays_validate_in_one_batch(golden):
    """tools/test_replay_playback.py for the whole bc_replays corpus at once (one env per replay, plain and zoo levels
    mixed in the same wavefronts): tick count, win / death and final position of every replay as the reference recorded
    them (corpus.npz `final`); 128 of 130 win."""
    from nclone_amd.replay import CompactReplay, validate_replays

    c = golden.z("corpus")
    final = c["final"]
    reps = [CompactReplay(bytes(c["m%d" % i]), list(c["in%d" % i])) for i in range(len(final))]
    res = validate_replays(reps)
    wins = 0
    for i, r in enumerate(res):
        assert r["ticks"] == int(final[i, 0]), (i, r, final[i])
        assert r["won"] == (int(final[i, 1]) == 8) and r["died"] == (int(final[i, 1]) in (6, 7)), (i, r, final[i])
        assert (r["x"], r["y"]) == (final[i, 2], final[i, 3]), (i, r, final[i])
        wins += r["won"]
    assert len(res) == 130 and wins == int(np.sum(final[:, 1] == 8)) == 128


def test_zoo_edge_cases(golden, oracle_mod):
    """One env; spatial_context + terminal observation together with the zoo blocks in LDS; truncation; masked reset
    (a reset is a Simulator.reset(): the entities are no longer in their first creation)."""
    c, z = golden.z("corpus"), golden.z("zoo")
    i0 = int(z["idx"][0])
    m = c["m%d" % i0].astype(np.float64)
    # (a) a single env replaying its own inputs
    b = _batch(1, autoreset=False)
    b.load_levels([m])
    b.assign_levels(np.array([0]))
    T = len(z["t%d" % i0])
    b.tick(torch.from_numpy(c["in%d" % i0][:T].reshape(T, 1).copy()).cuda())
    f, di = b.dump_state()
    assert np.array_equal(f[0, :4], z["t%d" % i0][-1]) and di[0, 0] == 8
    # (b) spatial_context on zoo levels, with auto-reset and terminal observations, against the oracle twin
    n = 40
    lv = [c["m%d" % int(k)].astype(np.float64) for k in z["idx"][:4]]
    b = _batch(n, autoreset=True)
    b.load_levels(lv)
    lvl = np.arange(n) % 4
    b.assign_levels(lvl)
    b.enable_spatial_context()
    b.set_truncation_limit(90)
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 6, size=(50, n)).astype(np.uint8)
    d = torch.from_numpy(acts).cuda()
    sims = []
    for e in range(n):
        o = oracle_mod.Oracle("mul")
        o.load(lv[lvl[e]])
        sims.append(o)
    truncated = 0
    for s in range(50):
        b.step(d[s])
        fl = b.flags.cpu().numpy()
        sc = b.spatial_context.cpu().numpy()
        f, di = b.dump_state()
        for e in range(n):
            _, ofl = sims[e].env_step(int(acts[s, e]), 4)
            trunc = (not ofl) and sims[e].frame >= 90
            assert bool(fl[e] & 8) == trunc and bool(fl[e] & 3) == bool(ofl), (s, e, fl[e], ofl, sims[e].frame)
            truncated += trunc
            if ofl or trunc:
                sims[e].reset()
            assert np.array_equal(f[e], sims[e].core()[0]), (s, e)
            assert np.array_equal(sc[e], sims[e].spatial_context()), (s, e)
    assert truncated > 0
    # (c) masked reset: reset envs restart from spawn with re-created entities, the others are untouched
    f0, i0_ = b.dump_state()
    c0 = b.entity_checksum()
    mask = np.zeros(n, dtype=np.uint8)
    mask[::2] = 1
    b.reset(mask)
    f1, i1 = b.dump_state()
    c1 = b.entity_checksum()
    assert np.array_equal(f1[1::2], f0[1::2]) and np.array_equal(c1[1::2], c0[1::2])
    for e in range(0, n, 2):
        o = oracle_mod.Oracle("mul")
        o.load(lv[lvl[e]])
        assert np.array_equal(f1[e], o.core()[0]) and np.array_equal(c1[e], o.entity_checksum())


def test_set_entity_pos_matches_reference(golden, oracle_mod):
    """npp_set_entity_pos (curriculum repositioning, intermediate_goal_manager.py:698): exit switch and exit door moved onto
    the recorded trajectory of 8 replays (plain, mines and zoo levels); fixtures from the reference simulator
    (make_golden_moved.py).  Two episodes with a reset in between (positions persist); plus one untouched env per level
    in the same batch, which must still reproduce the original replay."""
    c, mv = golden.z("corpus"), golden.z("moved")
    idx = [int(i) for i in mv["idx"]]
    n = 2 * len(idx)
    b = _batch(n, autoreset=False)
    b.load_levels([c["m%d" % i] for i in idx])
    b.assign_levels(np.arange(n) % len(idx))
    for k, i in enumerate(idx):
        b.set_entity_pos(k, 0, *mv["sw%d" % i])
        b.set_entity_pos(k, 1, *mv["door%d" % i])
    final = c["final"]
    for ep in range(2):
        T = [int(mv["ep%d" % i][0]) if ep == 0 else len(mv["t%d" % i]) - int(mv["ep%d" % i][0]) for i in idx]
        base = [0 if ep == 0 else int(mv["ep%d" % i][0]) for i in idx]
        tmax = max(max(T), max(int(final[i, 0]) for i in idx) if ep == 0 else 0)
        inputs = np.zeros((tmax, n), dtype=np.uint8)
        for e in range(n):
            i = idx[e % len(idx)]
            seq = c["in%d" % i]
            m = min(tmax, len(seq))
            inputs[:m, e] = seq[:m]
        d_in = torch.from_numpy(inputs).cuda()
        done = np.zeros(n, dtype=bool)
        for tick in range(tmax):
            b.tick(d_in[tick : tick + 1])
            f, di = b.dump_state()
            for k, i in enumerate(idx):
                if tick < T[k]:
                    ref = mv["t%d" % i][base[k] + tick]
                    assert np.array_equal(f[k, :4], ref), (ep, i, tick, f[k, :4], ref)
                    assert np.array_equal(_disc(di[k : k + 1])[0], mv["d%d" % i][base[k] + tick]), (ep, i, tick)
                    if tick == T[k] - 1:
                        assert di[k, 0] == 8
                if ep == 0:
                    e = k + len(idx)      # the untouched twin
                    if tick == int(final[i, 0]) - 1:
                        assert di[e, 0] == int(final[i, 1]) and np.array_equal(f[e, :2], final[i, 2:4]), (i, tick)
        b.reset()
    # entity_positions observation reports the moved switch / door
    b.observe()
    ep_ = b.entity_pos.cpu().numpy()
    for k, i in enumerate(idx):
        sw, door = mv["sw%d" % i], mv["door%d" % i]
        want = np.array([sw[0] / 1056.0, sw[1] / 600.0, door[0] / 1056.0, door[1] / 600.0], dtype=np.float32)
        assert np.allclose(ep_[k, 2:6], want, atol=1e-7), (i, ep_[k], want)


from tests.fuzz_levels import fuzz_level as _fuzz_level  # noqa: E402


def test_zoo_fuzz_random_entities(oracle_mod):
    """Random entity soups on real tile sets: every kind incl. the ones the recorded corpus barely has (regular doors, trap
    doors, shove thwumps, diagonal launch pads / one-ways, all drone modes), GPU vs the oracle twin, bit for bit, through
    episodes with auto-reset.  The oracle is pinned by the reference's replays for the kinds they contain; for the rest this
    checks two independent implementations (C lists vs merged CSR walk) against each other."""
    from nclone_amd.levels import curriculum0_levels, mine_levels

    bases = curriculum0_levels()[0][::5] + mine_levels()[0][::6]
    rng = np.random.default_rng(2024)
    levels = [_fuzz_level(bases[k % len(bases)], rng, keep_away=(0.0 if k % 2 else 160.0)) for k in range(48)]
    n = len(levels)
    b = _batch(n, autoreset=True)
    b.load_levels(levels)
    b.assign_levels(np.arange(n))
    steps = 150
    acts = rng.integers(0, 6, size=(steps, n)).astype(np.uint8)
    acts[:, ::4] = 0      # a quarter of the envs just stand at the spawn: long episodes for the entities to wander
    d = torch.from_numpy(acts).cuda()
    sims = []
    for e in range(n):
        o = oracle_mod.Oracle("mul")
        o.load(levels[e])
        sims.append(o)
    episodes = 0
    for s in range(steps):
        b.step(d[s])
        f, di = b.dump_state()
        cs = b.entity_checksum()
        fl = b.flags.cpu().numpy()
        for e in range(n):
            _, ofl = sims[e].env_step(int(acts[s, e]), 4)
            assert bool(fl[e] & 3) == bool(ofl), (s, e, fl[e], ofl)
            if ofl:
                sims[e].reset()
                episodes += 1
            of, od = sims[e].core()
            assert np.array_equal(f[e], of), (s, e, f[e], of)
            assert np.array_equal(di[e, :22], od[:22]), (s, e)
            assert np.array_equal(cs[e], sims[e].entity_checksum()), (s, e, cs[e] - sims[e].entity_checksum())
    print("fuzz: %d levels x %d steps, %d episodes ended" % (n, steps, episodes))
    assert episodes > 5


def test_fuzz_levels_vs_reference_fixtures(golden):
    """The 28 entity-soup levels the REFERENCE was run on (fuzz.npz): npp_step with auto-reset against its per-tick ninja
    state, discrete state and entity checksum at every step boundary, and per-step frames / termination."""
    z = golden.z("fuzz")
    n = int(z["n"][0])
    b = _batch(n, autoreset=True)
    b.load_levels([z["m%d" % k] for k in range(n)])
    b.assign_levels(np.arange(n))
    steps = len(z["a0"])
    acts = np.stack([z["a%d" % k] for k in range(n)], axis=1)
    d = torch.from_numpy(acts).cuda()
    rows = np.zeros(n, dtype=np.int64)
    episodes = 0
    for s in range(steps):
        b.step(d[s], frame_skip=4)
        flags = b.flags.cpu().numpy()
        frames = b.frames.cpu().numpy().astype(np.int64)
        f, di = b.dump_state()
        cs = b.entity_checksum()
        for k in range(n):
            ex, term, frame = z["s%d" % k][s]
            assert frames[k] == ex, (k, s, frames[k], ex)
            assert (1 if flags[k] & 1 else (2 if flags[k] & 2 else 0)) == term, (k, s, flags[k], term)
            rows[k] += ex
            if not term:
                ref = z["t%d" % k][rows[k] - 1]
                assert np.abs(f[k, :4] - ref).max() <= POS_TOL, (k, s, f[k, :4], ref)
                assert np.array_equal(_disc(di[k : k + 1])[0], z["d%d" % k][rows[k] - 1]), (k, s)
                assert np.abs(cs[k] - z["e%d" % k][rows[k] - 1]).max() <= ENT_TOL, (k, s, cs[k] - z["e%d" % k][rows[k] - 1])
            else:
                episodes += 1
    assert episodes > 100
