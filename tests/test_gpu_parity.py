"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(1) the golden fixtures produced by running the reference and (2) the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): discrete state and termination tick identical; float32 positions within
1e-5.  In addition the fp64 state is compared BIT-FOR-BIT with the oracle's multiply-square variant (the
kernel squares by multiplication; see oracle/nsim_oracle.c header).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

POS_TOL = 1e-5  # north-star tolerance on float32 positions
GS_TOL = 2e-6   # game_state f32 features (libm vs device atan2/sqrt last-bit differences after the f32 cast)


def _batch(n, **kw):
    from nclone_amd.engine import NppBatch

    return NppBatch(n, **kw)


def _disc_from_dump(i):
    """dump i32 row -> the 20 discrete fields of the golden `d` arrays."""
    return i[:, :20].clip(0, 255).astype(np.uint8)


def test_replays_match_reference_tick_for_tick(golden):
    """All in-scope bc_replays, one env per replay, every tick compared (tools/test_replay_playback.py semantics)."""
    c, t = golden.z("corpus"), golden.z("traj")
    idx = golden.in_scope_replays()
    n = len(idx)
    b = _batch(n, autoreset=False)
    b.load_levels([c["m%d" % i] for i in idx])
    b.assign_levels(np.arange(n))
    T = [len(t["t%d" % i]) for i in idx]
    tmax = max(T)
    inputs = np.zeros((tmax, n), dtype=np.uint8)
    for k, i in enumerate(idx):
        inputs[: T[k], k] = c["in%d" % i][: T[k]]
    d_inputs = torch.from_numpy(inputs).cuda()
    diverged = 0
    worst = 0.0
    exact = 0
    total = 0
    final = c["final"]
    for tick in range(tmax):
        b.tick(d_inputs[tick : tick + 1])
        f, di = b.dump_state()
        disc = _disc_from_dump(di)
        for k, i in enumerate(idx):
            if tick >= T[k]:
                continue
            if tick == T[k] - 1:
                # end-state table of the survey: ticks run, final ninja state, final position
                assert int(final[i, 0]) == T[k] and di[k, 0] == int(final[i, 1])
                assert np.array_equal(f[k, :2], final[i, 2:4])
            ref = t["t%d" % i][tick]
            total += 1
            d32 = np.abs(f[k, :4].astype(np.float32).astype(np.float64) - ref.astype(np.float32).astype(np.float64)).max()
            worst = max(worst, d32)
            if np.array_equal(f[k, :4], ref):
                exact += 1
            if d32 > POS_TOL or not np.array_equal(disc[k], t["d%d" % i][tick]):
                diverged += 1
    print("replay ticks %d, bit-exact fp64 ticks %d, worst f32 position diff %.3g, divergent ticks %d" % (total, exact, worst, diverged))
    assert diverged == 0


def test_replays_bit_exact_vs_oracle_mul(golden, oracle_mod):
    """fp64 state bit-for-bit against the oracle's multiply-square twin, incl. normals and old velocities."""
    c, t = golden.z("corpus"), golden.z("traj")
    idx = golden.in_scope_replays()[::3]
    n = len(idx)
    b = _batch(n, autoreset=False)
    b.load_levels([c["m%d" % i] for i in idx])
    b.assign_levels(np.arange(n))
    sims = []
    for i in idx:
        o = oracle_mod.Oracle("mul")
        o.load(c["m%d" % i].astype(np.float64))
        sims.append(o)
    T = [len(t["t%d" % i]) for i in idx]
    tmax = max(T)
    inputs = np.zeros((tmax, n), dtype=np.uint8)
    for k, i in enumerate(idx):
        inputs[: T[k], k] = c["in%d" % i][: T[k]]
    d_inputs = torch.from_numpy(inputs).cuda()
    for tick in range(tmax):
        b.tick(d_inputs[tick : tick + 1])
        f, di = b.dump_state()
        for k in range(n):
            h, j = oracle_mod.controls(int(inputs[tick, k]))
            sims[k].tick(h, j)
            of, od = sims[k].core()
            assert np.array_equal(f[k], of), (idx[k], tick, f[k], of)
            assert np.array_equal(di[k, :22], od[:22]), (idx[k], tick, di[k, :22], od[:22])
            if tick % 16 == 0:
                assert np.array_equal(b.dump_entities(k), sims[k].entity_states()), (idx[k], tick)


def test_rollouts_step_api_matches_reference(golden):
    """Random-action frame-skip rollouts through npp_step with in-kernel auto-reset, against reference rollouts
    (base_environment.py:535-609 loop + reset on termination): per-step frames executed, termination kind,
    game_state (f32[40] + time_remaining), action mask, and the fp64 state at every step boundary."""
    r = golden.z("rollouts")
    names = golden.names("rollouts")
    n = len(names)
    b = _batch(n, autoreset=True)
    b.load_levels([r["m%d" % i] for i in range(n)])
    b.assign_levels(np.arange(n))
    steps = len(r["a0"])
    acts = np.stack([r["a%d" % i] for i in range(n)], axis=1)  # [steps, n]
    d_acts = torch.from_numpy(acts).cuda()
    rows = np.zeros(n, dtype=np.int64)
    frames_before = np.zeros(n, dtype=np.int64)
    worst_gs = 0.0
    for s in range(steps):
        b.step(d_acts[s], frame_skip=4)
        b.sync()
        flags = b.flags.cpu().numpy()
        frames = b.frames.cpu().numpy().astype(np.int64)
        gs = b.game_state.cpu().numpy()
        term_gs = b.terminal_state.cpu().numpy()
        mask = b.action_mask.cpu().numpy()
        f, di = b.dump_state()
        for i in range(n):
            ex, term, frame = r["s%d" % i][s]
            assert frames[i] == ex, (names[i], s, frames[i], ex)
            kind = 1 if flags[i] & 1 else (2 if flags[i] & 2 else 0)
            assert kind == term, (names[i], s, flags[i], term)
            rows[i] += ex
            ref_gs = r["g%d" % i][s]
            got = term_gs[i] if term else gs[i]
            worst_gs = max(worst_gs, np.abs(got[:40] - ref_gs).max())
            assert np.abs(got[:40] - ref_gs).max() <= GS_TOL, (names[i], s, np.nonzero(np.abs(got[:40] - ref_gs) > GS_TOL))
            tr = max(0.0, (10000 - frame) / 10000)
            assert abs(got[40] - np.float32(tr)) <= 1e-7
            if not term:
                ref_mask = [(int(r["k%d" % i][s]) >> k) & 1 for k in range(6)]
                assert list(mask[i]) == ref_mask, (names[i], s)
                ref_row = r["t%d" % i][rows[i] - 1]
                assert np.array_equal(f[i, :4], ref_row), (names[i], s, f[i, :4], ref_row)
                assert np.array_equal(_disc_from_dump(di[i : i + 1])[0], r["d%d" % i][rows[i] - 1]), (names[i], s)
            else:
                # auto-reset: state is the spawn state again
                assert di[i, 22] == 0 and di[i, 0] == 0
    print("rollout steps %d x %d envs, worst game_state diff %.3g" % (steps, n, worst_gs))


def test_mixed_level_workgroups_match_uniform(golden):
    """The LDS-staged path (64 envs of one level per workgroup) and the global-memory path (mixed levels in a
    workgroup) must give identical results."""
    r = golden.z("rollouts")
    n_lv = 8
    levels = [r["m%d" % i] for i in range(n_lv)]
    steps = 120
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 6, size=(steps, 1)).astype(np.uint8)
    # uniform: 64 envs per level, all envs of a level get the same actions
    bu = _batch(64 * n_lv, autoreset=True)
    bu.load_levels(levels)
    bu.assign_levels(np.repeat(np.arange(n_lv), 64))
    # mixed: envs interleaved so every workgroup sees all levels
    bm = _batch(64 * n_lv, autoreset=True)
    bm.load_levels(levels)
    bm.assign_levels(np.tile(np.arange(n_lv), 64))
    a_all = torch.from_numpy(np.repeat(acts, 64 * n_lv, axis=1)).cuda()
    for s in range(steps):
        bu.step(a_all[s])
        bm.step(a_all[s])
    fu, iu = bu.dump_state()
    fm, im = bm.dump_state()
    for lvl in range(n_lv):
        u = fu[lvl * 64]
        assert np.array_equal(fu[lvl * 64 : (lvl + 1) * 64], np.tile(u, (64, 1)))
        assert np.array_equal(fm[lvl::n_lv], np.tile(u, (64, 1)))
        assert np.array_equal(im[lvl::n_lv, :27], np.tile(iu[lvl * 64, :27], (64, 1)))


def test_ragged_env_count_and_reset_mask(golden):
    """N not a multiple of 64, partial reset by mask, determinism (same inputs twice -> same bits)."""
    r = golden.z("rollouts")
    n = 70
    b = _batch(n, autoreset=False)
    b.load_levels([r["m5"], r["m24"]])
    b.assign_levels(np.arange(n) % 2)
    rng = np.random.default_rng(3)
    acts = torch.from_numpy(rng.integers(0, 6, size=(40, n)).astype(np.uint8)).cuda()
    for s in range(40):
        b.step(acts[s])
    f1, i1 = b.dump_state()
    mask = np.zeros(n, dtype=np.uint8)
    mask[::3] = 1
    b.reset(mask)
    f2, i2 = b.dump_state()
    assert np.array_equal(f2[1::3], f1[1::3]) and np.array_equal(f2[2::3], f1[2::3])
    assert np.all(i2[::3, 22] == 0) and np.all(i2[::3, 0] == 0)
    b.reset()
    for s in range(40):
        b.step(acts[s])
    f3, i3 = b.dump_state()
    assert np.array_equal(f3, f1) and np.array_equal(i3, i1)


def test_launch_geometries_bit_identical(golden):
    """Every lanes-per-env / waves-per-block geometry must give the same bits (DPP reductions keep the reference's
    first-wins tie-breaks and the wall-probe summation order)."""
    r = golden.z("rollouts")
    pick = [3, 8, 14, 21, 24, 25, 27, 28]
    levels = [r["m%d" % i] for i in pick]
    n = 200
    steps = 150
    rng = np.random.default_rng(11)
    acts = torch.from_numpy(rng.integers(0, 6, size=(steps, n)).astype(np.uint8)).cuda()
    ref = None
    # the three build variants of the G = 16 kernels (register cap / candidate slots; npp_set_step_variant) are geometries too
    for g, wpb, var in [(1, 1, 0), (1, 4, 0), (2, 2, 0), (4, 4, 0), (8, 1, 0), (8, 4, 0), (16, 4, 0), (16, 4, 1), (16, 4, 2), (16, 2, 1),
                        (32, 2, 0), (64, 4, 0)]:
        b = _batch(n, autoreset=True)
        b.load_levels(levels)
        b.set_launch_geometry(g, wpb)
        b.set_step_variant(var)
        assert b.launch_geometry()[0] == g
        assert b.step_variant() == (var if g == 16 else 0, True)
        b.assign_levels(np.arange(n) % len(levels))
        tot_flags = np.zeros(n, dtype=np.int64)
        for s in range(steps):
            b.step(acts[s])
            if s % 10 == 0:
                tot_flags += b.flags.cpu().numpy().astype(np.int64)
        f, di = b.dump_state()
        gs = b.game_state.cpu().numpy()
        ent = np.concatenate([b.dump_entities(e) for e in range(0, n, 17)])
        cur = (f, di, gs, tot_flags, ent)
        if ref is None:
            ref = cur
        else:
            for x, y in zip(ref, cur):
                assert np.array_equal(x, y), (g, wpb, var)


def test_spatial_context_matches_reference(golden, oracle_mod):
    """spatial_context (112 f32) against rows produced by the reference's own spatial_context module, including its
    position cache (stale between recomputes) and the reset behaviour of a vector env."""
    r = golden.z("rollouts")
    names = golden.names("rollouts")
    n = len(names)
    worst = 0.0
    # every lanes-per-env geometry: the nearest-eight selection is split over the lanes of a group (1, 2, 4 lanes own several of
    # the eight results, 8+ lanes one each)
    for g, wpb in [(16, 4), (1, 1), (2, 2), (4, 4), (8, 4), (64, 4)]:
        b = _batch(n, autoreset=True)
        b.load_levels([r["m%d" % i] for i in range(n)])
        b.set_launch_geometry(g, wpb)
        b.assign_levels(np.arange(n))
        b.enable_spatial_context()
        b.reset()
        b.observe()
        sc = b.spatial_context.cpu().numpy()
        for i in range(n):
            assert np.abs(sc[i] - r["sc%d" % i][0]).max() <= 1.2e-7, (g, names[i], "reset obs", np.abs(sc[i] - r["sc%d" % i][0]).max())
        acts = torch.from_numpy(np.stack([r["a%d" % i] for i in range(n)], axis=1)).cuda()
        for s in range(acts.shape[0]):
            b.step(acts[s])
            sc = b.spatial_context.cpu().numpy()
            for i in range(n):
                ref = r["sc%d" % i][s + 1]
                d = np.abs(sc[i] - ref).max()
                worst = max(worst, d)
                assert d <= 1.2e-7, (g, names[i], s, np.nonzero(np.abs(sc[i] - ref) > 1.2e-7)[0][:6])
        b.close()
    print("spatial_context worst abs diff %.3g" % worst)
