"""The CPU oracle (oracle/nsim_oracle.c) against fixtures produced by RUNNING the reference
(tests/golden/make_golden.py).  Bit-exact: fp64 positions/velocities and every discrete field.
These tests pin the oracle; the GPU tests then compare the HIP path with the oracle and the same fixtures."""
import numpy as np
import pytest


def test_oracle_replays_bit_exact(golden, oracle_mod):
    c, t = golden.z("corpus"), golden.z("traj")
    idx = golden.in_scope_replays()
    assert len(idx) == 104
    ticks = 0
    for i in idx:
        o = oracle_mod.Oracle("pow")
        assert o.load(c["m%d" % i].astype(np.float64)) == 0
        T, D = t["t%d" % i], t["d%d" % i]
        for k in range(len(T)):
            h, j = oracle_mod.controls(int(c["in%d" % i][k]))
            o.tick(h, j)
            f, d = o.core()
            assert np.array_equal(f[:4], T[k]), (i, k)
            assert np.array_equal(d[:20].clip(0, 255), D[k]), (i, k)
            ticks += 1
        final = c["final"][i]
        assert int(final[0]) == len(T) and d[0] == int(final[1])
    assert ticks == 25575


def test_oracle_very_simple_100001_known_answers(golden, oracle_mod):
    """replay 20251022_142001_train_very_simple_100001: 37 inputs, win at tick 37,
    final position (812.0036311681246, 542.0), spawn (828, 540)."""
    c = golden.z("corpus")
    names = golden.names("corpus")
    i = names.index("20251022_142001_train_very_simple_100001")
    o = oracle_mod.Oracle("pow")
    o.load(c["m%d" % i].astype(np.float64))
    f, d = o.core()
    assert (f[0], f[1]) == (828.0, 540.0)
    for k, b in enumerate(c["in%d" % i]):
        h, j = oracle_mod.controls(int(b))
        o.tick(h, j)
        f, d = o.core()
        if d[0] == 8:
            break
    assert k + 1 == 37 and (f[0], f[1]) == (812.0036311681246, 542.0)


@pytest.mark.parametrize("variant", ["pow", "mul"])
def test_oracle_rollouts(golden, oracle_mod, variant):
    """Random-action frame-skip rollouts with reset on termination.  The libm-pow variant must match the reference's
    bits; the multiply-square variant (what the GPU computes) must too on this corpus."""
    r = golden.z("rollouts")
    names = golden.names("rollouts")
    for i in range(len(names)):
        o = oracle_mod.Oracle(variant)
        assert o.load(r["m%d" % i]) == 0
        T, D, S, G, K = r["t%d" % i], r["d%d" % i], r["s%d" % i], r["g%d" % i], r["k%d" % i]
        row = 0
        for s, a in enumerate(r["a%d" % i]):
            ex, fl = o.env_step(int(a), 4)
            f, d = o.core()
            row += ex
            assert (ex, fl, o.frame) == tuple(S[s]), (names[i], s)
            assert np.array_equal(f[:4], T[row - 1]), (names[i], s)
            assert np.array_equal(d[:20].clip(0, 255), D[row - 1]), (names[i], s)
            if variant == "pow":
                assert np.array_equal(o.ninja_state().astype(np.float32), G[s]), (names[i], s)
                assert o.action_mask() == K[s]
            if fl:
                o.reset()


def test_oracle_game_state_and_mask_on_replays(golden, oracle_mod):
    c, g = golden.z("corpus"), golden.z("gstate")
    for i in golden.in_scope_replays()[::4]:
        o = oracle_mod.Oracle("pow")
        o.load(c["m%d" % i].astype(np.float64))
        G, K = g["g%d" % i], g["k%d" % i]
        for k in range(len(G)):
            h, j = oracle_mod.controls(int(c["in%d" % i][k]))
            o.tick(h, j)
            assert np.array_equal(o.ninja_state().astype(np.float32), G[k]), (i, k)
            assert o.action_mask() == K[k]


def test_oracle_level_tables(golden, oracle_mod):
    """Ordered per-cell segment lists (first-hit-wins order matters) and entity tables of all 130 + 112 levels."""
    c, csr, lg = golden.z("corpus"), golden.z("csr"), golden.z("levels_gen")
    for i in range(len(golden.names("corpus"))):
        o = oracle_mod.Oracle("pow")
        o.load(c["m%d" % i].astype(np.float64))
        assert np.array_equal(o.dump_csr(), csr[bytes(c["csr%d" % i]).decode()]), i
        if "ent%d" % i in c.files:
            ref = c["ent%d" % i]
            got = o.dump_entities()
            assert np.array_equal(got[:, :6], ref[:, :6]) and np.array_equal(got[:, 7], ref[:, 7]), i
            mine = ref[:, 6] >= 0
            assert np.array_equal(got[mine, 6], ref[mine, 6])
    for k in range(len(golden.names("levels_gen"))):
        o = oracle_mod.Oracle("pow")
        assert o.load(lg["L%d" % k]) == 0
        assert np.array_equal(o.dump_csr(), csr[bytes(lg["csr%d" % k]).decode()]), k
        assert np.array_equal(o.dump_entities()[:, :6], lg["ent%d" % k][:, :6]), k


def test_oracle_all_130_replays_end_state(golden, oracle_mod):
    """Every bc_replay of the reference corpus (the entity zoo included) ends where the reference ends: tick count,
    final ninja state, final position (corpus.npz `final`, from make_golden.py)."""
    c = golden.z("corpus")
    final = c["final"]
    wins = 0
    for i in range(len(final)):
        o = oracle_mod.Oracle("pow")
        assert o.load(c["m%d" % i].astype(np.float64)) == 0      # nothing is "unsupported" any more
        n = 0
        for b in c["in%d" % i]:
            h, j = oracle_mod.controls(int(b))
            o.tick(h, j)
            n += 1
            if o.core()[1][0] in (6, 7, 8):
                break
        f, d = o.core()
        assert (n, int(d[0])) == (int(final[i, 0]), int(final[i, 1])), i
        assert f[0] == final[i, 2] and f[1] == final[i, 3], i
        wins += int(d[0]) == 8
    assert wins == int(np.sum(final[:, 1] == 8))


@pytest.mark.parametrize("variant", ["pow", "mul"])
def test_oracle_zoo_replays(golden, oracle_mod, variant):
    """The 26 replays with launch pads, one-ways, drones, bounce blocks, thwumps, boost pads, death balls, trap doors
    and shove thwumps (zoo.npz from make_golden_zoo.py): per-tick ninja state and an entity checksum.
    `pow` squares like CPython and must match every bit.  `mul` (the GPU twin) squares by multiplication: libm's pow
    is not correctly rounded, so death-ball speeds may differ in the last bit; the ninja must stay within the
    north-star tolerance (1e-5 px) with identical discrete state -- here it is in fact identical."""
    c, z = golden.z("corpus"), golden.z("zoo")
    ticks = 0
    for i in z["idx"]:
        o = oracle_mod.Oracle(variant)
        o.load(c["m%d" % i].astype(np.float64))
        T, D, E, G, K = z["t%d" % i], z["d%d" % i], z["e%d" % i], z["g%d" % i], z["k%d" % i]
        for k in range(len(T)):
            h, j = oracle_mod.controls(int(c["in%d" % i][k]))
            o.tick(h, j)
            f, d = o.core()
            e = o.entity_checksum()
            assert np.array_equal(d[:20].clip(0, 255), D[k]), (i, k)
            if variant == "pow":
                assert np.array_equal(f[:4], T[k]), (i, k)
                assert np.array_equal(e, E[k]), (i, k)
                assert np.array_equal(o.ninja_state().astype(np.float32), G[k]), (i, k)
                assert o.action_mask() == K[k]
            else:
                assert np.abs(f[:4] - T[k]).max() <= 1e-5, (i, k)
                assert np.abs(e - E[k]).max() <= 1e-9, (i, k)
        ticks += len(T)
    assert ticks == 5886


def test_oracle_zoo_rollouts(golden, oracle_mod):
    """Random-action frame-skip rollouts with reset on termination on 9 zoo maps (Entity.index keeps counting across
    Simulator.reset(), which switches the death-ball repulsion off after the first episode: reproduced)."""
    z = golden.z("zoo")
    for r in range(int(z["n_rollouts"][0])):
        o = oracle_mod.Oracle("pow")
        o.load(z["rm%d" % r])
        T, D, E, S, G, K = (z[k + str(r)] for k in ("rt", "rd", "re", "rs", "rg", "rk"))
        row = 0
        for s, a in enumerate(z["ra%d" % r]):
            h, j = oracle_mod.ACTIONS[a]
            ex = fl = 0
            for _ in range(4):
                o.tick(h, j)
                ex += 1
                f, d = o.core()
                assert np.array_equal(f[:4], T[row]) and np.array_equal(d[:20].clip(0, 255), D[row]), (r, s)
                assert np.array_equal(o.entity_checksum(), E[row]), (r, s)
                row += 1
                if d[0] in (6, 7, 8):
                    fl = 1 if d[0] == 8 else 2
                    break
            assert (ex, fl, o.frame) == tuple(S[s]), (r, s)
            assert np.array_equal(o.ninja_state().astype(np.float32), G[s]), (r, s)
            assert o.action_mask() == K[s]
            if fl:
                o.reset()


@pytest.mark.parametrize("variant", ["pow", "mul"])
def test_oracle_repositioned_switch_and_door(golden, oracle_mod, variant):
    """Curriculum repositioning (intermediate_goal_manager.py:698): exit switch and door moved onto the recorded trajectory
    (moved.npz from make_golden_moved.py, produced by the reference simulator with the same attribute updates), two
    episodes with a Simulator.reset() in between; every tick of both."""
    c, mv = golden.z("corpus"), golden.z("moved")
    for i in mv["idx"]:
        o = oracle_mod.Oracle(variant)
        o.load(c["m%d" % i].astype(np.float64))
        o.set_entity_pos(0, *mv["sw%d" % i])
        o.set_entity_pos(1, *mv["door%d" % i])
        T, D, ep1 = mv["t%d" % i], mv["d%d" % i], int(mv["ep%d" % i][0])
        k = 0
        for ep in range(2):
            for b in c["in%d" % i]:
                h, j = oracle_mod.controls(int(b))
                o.tick(h, j)
                f, d = o.core()
                assert np.array_equal(f[:4], T[k]), (i, ep, k, f[:4], T[k])
                assert np.array_equal(d[:20].clip(0, 255), D[k]), (i, ep, k)
                k += 1
                if d[0] in (6, 7, 8):
                    break
            assert k == (ep1 if ep == 0 else len(T)), (i, ep, k)
            o.reset()
        assert D[-1][0] == 8


def test_oracle_fuzz_levels_vs_reference(golden, oracle_mod):
    """28 random "entity soup" levels run through the reference (fuzz.npz, make_golden_fuzz.py): regular doors, trap doors,
    shove thwumps, every orientation and drone mode -- kinds the recorded replays barely contain.  Bit for bit, every tick,
    incl. the entity checksum and the resets."""
    z = golden.z("fuzz")
    ticks = 0
    for k in range(int(z["n"][0])):
        o = oracle_mod.Oracle("pow")
        o.load(z["m%d" % k])
        T, D, E, S = z["t%d" % k], z["d%d" % k], z["e%d" % k], z["s%d" % k]
        row = 0
        for s, a in enumerate(z["a%d" % k]):
            h, j = oracle_mod.ACTIONS[a]
            ex = fl = 0
            for _ in range(4):
                o.tick(h, j)
                ex += 1
                f, d = o.core()
                assert np.array_equal(f[:4], T[row]), (k, s, row, f[:4], T[row])
                assert np.array_equal(d[:20].clip(0, 255), D[row]), (k, s, row)
                assert np.array_equal(o.entity_checksum(), E[row]), (k, s, row, o.entity_checksum() - E[row])
                row += 1
                if d[0] in (6, 7, 8):
                    fl = 1 if d[0] == 8 else 2
                    break
            assert (ex, fl, o.frame) == tuple(S[s]), (k, s)
            if fl:
                o.reset()
        ticks += row
    assert ticks == 12537


def test_oracle_spatial_context(golden, oracle_mod):
    """spatial_context rows from the reference's spatial_context.py (loaded by file, see make_golden.py)."""
    r = golden.z("rollouts")
    names = golden.names("rollouts")
    for i in range(len(names)):
        o = oracle_mod.Oracle("pow")
        o.load(r["m%d" % i])
        SC, SCT = r["sc%d" % i], r["sct%d" % i]
        assert np.array_equal(o.spatial_context(), SC[0]), names[i]
        for s, a in enumerate(r["a%d" % i]):
            ex, fl = o.env_step(int(a), 4)
            assert np.array_equal(o.spatial_context(), SCT[s]), (names[i], s)
            if fl:
                o.reset()
                assert np.array_equal(o.spatial_context(), SC[s + 1]), (names[i], s)


def test_oracle_c3_mixed_rollouts_and_tables(golden, oracle_mod):
    """Config 4's extra 320 levels (categories `simpler` / `simple`, tests/golden/make_golden_c3.py): the reference's
    rollouts on every 20th level bit for bit, entity tables of all of them, ordered segment dumps of every 4th."""
    g = golden.z("levels_c3")
    names = golden.names("levels_c3")
    assert len(names) == 320
    for k in range(len(names)):
        o = oracle_mod.Oracle("pow")
        assert o.load(g["L%d" % k]) == 0, names[k]
        assert np.array_equal(o.dump_entities()[:, :6], g["ent%d" % k][:, :6]), names[k]
        if "csr%d" % k in g.files:
            assert np.array_equal(o.dump_csr(), g["c" + bytes(g["csr%d" % k]).decode()]), names[k]
    ticks = 0
    for r in range(int(g["n_rollouts"][0])):
        k = int(g["rl%d" % r][0])
        for variant in ("pow", "mul"):
            o = oracle_mod.Oracle(variant)
            assert o.load(g["L%d" % k]) == 0
            T, D, S, G, K = g["rt%d" % r], g["rd%d" % r], g["rs%d" % r], g["rg%d" % r], g["rk%d" % r]
            row = 0
            for s, a in enumerate(g["ra%d" % r]):
                ex, fl = o.env_step(int(a), 4)
                f, d = o.core()
                row += ex
                assert (ex, fl, o.frame) == tuple(S[s]), (names[k], s)
                assert np.array_equal(f[:4], T[row - 1]), (names[k], s, variant)
                assert np.array_equal(d[:20].clip(0, 255), D[row - 1]), (names[k], s)
                if variant == "pow":
                    assert np.array_equal(o.ninja_state().astype(np.float32), G[s]), (names[k], s)
                    assert o.action_mask() == K[s]
                if fl:
                    o.reset()
            ticks += row
    assert ticks == 2 * 19173


@pytest.mark.parametrize("fixture,rollouts,total_ticks", [("fast", 58, 54603), ("official", 15, 19199)])
def test_oracle_fast_reset_rollouts(golden, oracle_mod, fixture, rollouts, total_ticks):
    """Simulator.fast_reset (nsim.py:78-140) between episodes, mixed with full resets, on mine / locked-door levels and on
    every zoo map (tests/golden/make_golden_fast.py ran the reference), and on the reference's five official tutorial levels
    (`nclone/maps/test-maps/`, make_golden_official.py): ninja trajectory, discrete state, entity checksum at every tick and
    the per-entity states at every step, bit for bit."""
    g = golden.z(fixture)
    names = golden.names(fixture)
    assert len(names) == rollouts
    ticks = 0
    for r in range(len(names)):
        o = oracle_mod.Oracle("pow")
        assert o.load(g["m%d" % r]) == 0
        T, D, E, S, Q, G, K = (g["%s%d" % (k, r)] for k in "tdesqgk")
        row = 0
        for s, a in enumerate(g["a%d" % r]):
            h, j = oracle_mod.ACTIONS[int(a)]
            ex, term, frame, mode = (int(v) for v in S[s])
            for _ in range(ex):
                o.tick(h, j)
                f, d = o.core()
                assert np.array_equal(f[:4], T[row]), (names[r], s, row)
                assert np.array_equal(d[:20].clip(0, 255), D[row]), (names[r], s, row)
                assert np.array_equal(o.entity_checksum(), E[row]), (names[r], s, row, o.entity_checksum(), E[row])
                row += 1
            assert (1 if d[0] == 8 else (2 if d[0] in (6, 7) else 0)) == term
            assert np.array_equal(o.entity_states_dic(), Q[s]), (names[r], s)
            assert np.array_equal(o.ninja_state().astype(np.float32), G[s]), (names[r], s)
            assert o.action_mask() == K[s]
            if mode == 1:
                o.reset()
            elif mode == 2:
                o.fast_reset()
            else:
                assert o.frame == frame
        assert row == len(T)
        ticks += row
    assert ticks == total_ticks
