"""player_frame rasteriser (npp_render_player_frame) against the numpy restatement in tests/raster_ref.py.

No reference frames exist (cairo/pygame are not installed: parity unpinned beyond geometry), so the bar is the
tolerance proposed in SURVEY.md appendix C: pixels away from primitive edges exact, edge pixels within +-64,
mean absolute difference < 2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _check(golden, centered):
    from nclone_amd.engine import NppBatch
    from tests.raster_ref import player_frame

    r = golden.z("rollouts")
    pick = [3, 14, 24, 25, 27, 28, 21]   # mines, doors, slopes
    levels = [r["m%d" % i] for i in pick]
    n = len(levels) * 4
    b = NppBatch(n, autoreset=True, frame_centered=centered)
    b.load_levels(levels)
    lvl = np.arange(n) % len(levels)
    b.assign_levels(lvl)
    rng = np.random.default_rng(9)
    acts = torch.from_numpy(rng.integers(0, 6, size=(90, n)).astype(np.uint8)).cuda()
    out = torch.zeros((n, 84, 84), dtype=torch.uint8, device="cuda")
    checked = 0
    for s in range(90):
        b.step(acts[s])
        if s % 30 != 29:
            continue
        b.render_player_frame(out)
        frames = out.cpu().numpy()
        f, _ = b.dump_state()
        for e in range(n):
            ref, edge = player_frame(levels[lvl[e]], b.dump_entities(e), f[e, 0], f[e, 1], centered=centered)
            got = frames[e].astype(np.int64)
            d = np.abs(got - ref.astype(np.int64))
            bad = np.argwhere((d > 0) & ~edge)
            assert len(bad) == 0, (e, s, lvl[e], f[e, :2], [(tuple(p), int(got[tuple(p)]), int(ref[tuple(p)])) for p in bad[:8]])
            assert d.max(initial=0) <= 64
            assert d.mean() < 2.0
            checked += 1
    assert checked == 3 * n


def test_player_frame_reference_axis_swap(golden):
    _check(golden, centered=False)


def test_player_frame_centered(golden):
    _check(golden, centered=True)


def _zoo_batch_and_oracles(golden, oracle_mod, centered, n_steps=60):
    """A few zoo + plain levels stepped with random actions on the GPU and on the oracle twin (same bits), so that the
    oracle can tell the numpy rasteriser where every entity is."""
    from nclone_amd.engine import NppBatch

    c, z, r = golden.z("corpus"), golden.z("zoo"), golden.z("rollouts")
    idx = [int(i) for i in z["idx"]]
    levels = [c["m%d" % i].astype(np.float64) for i in (idx[0], idx[3], idx[12], idx[20])] + [r["m3"], r["m24"]]
    n = len(levels) * 2
    lvl = np.arange(n) % len(levels)
    b = NppBatch(n, autoreset=False, frame_centered=centered)
    b.load_levels(levels)
    b.assign_levels(lvl)
    rng = np.random.default_rng(31)
    acts = rng.integers(0, 6, size=(n_steps, n)).astype(np.uint8)
    d = torch.from_numpy(acts).cuda()
    for s in range(n_steps):
        b.step(d[s])
    sims = []
    for e in range(n):
        o = oracle_mod.Oracle("mul")
        o.load(levels[lvl[e]])
        done = False
        for s in range(n_steps):
            if not done:
                _, fl = o.env_step(int(acts[s, e]), 4)
                done = fl != 0
        sims.append(o)
    f, _ = b.dump_state()
    for e in range(n):
        assert np.array_equal(f[e], sims[e].core()[0])
    return b, sims, f


@pytest.mark.parametrize("centered", [False, True])
def test_player_frame_with_zoo_entities(golden, oracle_mod, centered):
    """Doors, launch pads / one-ways (oriented strokes), drones, bounce blocks and thwumps (squares), boost pads, death balls
    in the player frame; entity positions come from the oracle twin.  Same tolerance as the plain frames."""
    from tests.raster_ref import player_frame_rows

    b, sims, f = _zoo_batch_and_oracles(golden, oracle_mod, centered)
    out = torch.zeros((b.n, 84, 84), dtype=torch.uint8, device="cuda")
    b.render_player_frame(out)
    frames = out.cpu().numpy()
    drawn = 0
    for e in range(b.n):
        ref, edge = player_frame_rows(sims[e].tiles(), sims[e].draw_list(), f[e, 0], f[e, 1], centered=centered)
        got = frames[e].astype(np.int64)
        dlt = np.abs(got - ref.astype(np.int64))
        bad = np.argwhere((dlt > 0) & ~edge)
        assert len(bad) == 0, (e, f[e, :2], [(tuple(p), int(got[tuple(p)]), int(ref[tuple(p)])) for p in bad[:8]])
        assert dlt.max(initial=0) <= 64 and dlt.mean() < 2.0
        drawn += int(ref.any())
    assert drawn >= b.n // 2


def test_global_view(golden, oracle_mod):
    """global_view = the reference's cv2.resize(frame, (100, 176), INTER_AREA) of the whole canvas (swapped constants
    included), against the numpy restatement: destination pixels whose source rectangle touches no primitive edge are
    exact, the others within the edge tolerance scaled by the edge fraction."""
    from tests.raster_ref import global_view

    b, sims, f = _zoo_batch_and_oracles(golden, oracle_mod, False, n_steps=30)
    out = torch.zeros((b.n, 176, 100), dtype=torch.uint8, device="cuda")
    b.render_global_view(out)
    views = out.cpu().numpy()
    for e in range(0, b.n, 2)[:6]:
        ref, ef = global_view(sims[e].tiles(), sims[e].draw_list(), f[e, 0], f[e, 1])
        dlt = np.abs(views[e].astype(np.int64) - ref.astype(np.int64))
        assert np.all(dlt[ef == 0] <= 1), (e, np.argwhere((dlt > 1) & (ef == 0))[:5])    # float rounding of the mean only
        assert np.all(dlt <= 2 + 64 * ef), (e, dlt.max())
        assert dlt.mean() < 1.0
        assert ref.std() > 10       # a real picture, not a constant


def _area_tabs(n_dst, scale, ssize):
    """Padded (index, weight) tables of OpenCV's computeResizeAreaTab (tests/raster_ref._area_tab), float32."""
    from tests.raster_ref import _area_tab

    tabs = [_area_tab(d, np.float32(scale), ssize) for d in range(n_dst)]
    m = max(len(t) for t in tabs)
    idx = np.zeros((n_dst, m), dtype=np.int64)
    w = np.zeros((n_dst, m), dtype=np.float32)
    for d, t in enumerate(tabs):
        for j, (s_, ww) in enumerate(t):
            idx[d, j], w[d, j] = s_, ww
    return torch.from_numpy(idx).cuda(), torch.from_numpy(w).cuda()


def _reduce_frame(frame, xtab, ytab):
    """cv2.resize(frame, (100, 176), INTER_AREA) of one [600, 1056] float32 frame in OpenCV's accumulation order; padding
    entries carry weight 0 and add exactly 0."""
    (ix, wx), (iy, wy) = xtab, ytab
    hs = torch.zeros((600, 100), dtype=torch.float32, device="cuda")
    for j in range(ix.shape[1]):
        hs = hs + wx[:, j][None, :] * frame[:, ix[:, j]]
    acc = torch.zeros((176, 100), dtype=torch.float32, device="cuda")
    for j in range(iy.shape[1]):
        acc = acc + wy[:, j][:, None] * hs[iy[:, j]]
    return torch.clamp(torch.round(acc), 0, 255).to(torch.uint8)


def test_global_view_equals_reduction_of_the_whole_frame(golden, oracle_mod):
    """The global_view kernel patches a per-level table (the level's view right after a reset) in the destination cells that
    something dynamic can touch.  It must equal, bit for bit, the area reduction of the WHOLE rendered frame of the same env
    (npp_render_frame, every one of the 633 600 pixels through the same pixel function) -- on zoo levels after 40 and after
    400 steps (moved drones / thwumps, collected gold, opened doors, toggled mines, dead and respawned ninjas)."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import door_levels, mine_levels

    c, z = golden.z("corpus"), golden.z("zoo")
    levels = [c["m%d" % int(i)].astype(np.float64) for i in z["idx"]] + door_levels()[0][:8] + mine_levels()[0][:14]
    n = len(levels)
    b = NppBatch(n, autoreset=True)
    b.load_levels(levels)
    b.assign_levels(np.arange(n))
    warm = torch.from_numpy(np.random.default_rng(76).integers(0, 6, size=(40, n)).astype(np.uint8)).cuda()
    for t in range(40):
        b.step(warm[t])
    xtab, ytab = _area_tabs(100, 1056.0 / 100, 1056), _area_tabs(176, 600.0 / 176, 600)
    acts = torch.from_numpy(np.random.default_rng(77).integers(0, 6, size=(360, b.n)).astype(np.uint8)).cuda()
    changed_cells = 0
    for phase in range(2):
        out = torch.zeros((b.n, 176, 100), dtype=torch.uint8, device="cuda")
        b.render_global_view(out)
        for e in range(b.n):
            frame = b.render_frame(e, 1)[0, :, :, 0].float()
            ref = _reduce_frame(frame, xtab, ytab)
            assert torch.equal(out[e], ref), (phase, e, int((out[e] != ref).sum()))
        if phase == 0:
            first = out.clone()
            for t in range(acts.shape[0]):
                b.step(acts[t])
        else:
            changed_cells = int((out != first).sum())
    assert changed_cells > 100   # the picture did change between the two phases
    b.close()


def test_full_frame_consistent_with_player_frame_and_reference_raster(golden, oracle_mod):
    """render(): the whole 600 x 1056 frame.  Its crop around the player must BE the player_frame (same pixel function), and
    a band of it matches the numpy rasteriser under the usual tolerance."""
    from tests.raster_ref import render_canvas

    b, sims, f = _zoo_batch_and_oracles(golden, oracle_mod, True, n_steps=40)
    pf = torch.zeros((b.n, 84, 84), dtype=torch.uint8, device="cuda")
    b.render_player_frame(pf)
    pf = pf.cpu().numpy()
    for e in (0, 3, 5):
        full = b.render_frame(e, 1)[0, :, :, 0].cpu().numpy()
        px, py = f[e, 0], f[e, 1]
        r0, r1, c0, c1 = max(0, int(py - 42)), min(600, int(py + 42)), max(0, int(px - 42)), min(1056, int(px + 42))
        h, w = r1 - r0, c1 - c0
        top, left = (84 - h) // 2, (84 - w) // 2
        assert np.array_equal(pf[e][top:top + h, left:left + w], full[r0:r1, c0:c1]), e
        y0 = max(0, min(600 - 60, int(py) - 30))
        ref, edge = render_canvas(sims[e].tiles(), sims[e].draw_list(), px, py, 0, y0, 1056, 60)
        dlt = np.abs(full[y0:y0 + 60].astype(np.int64) - ref.astype(np.int64))
        assert len(np.argwhere((dlt > 0) & ~edge)) == 0 and dlt.max(initial=0) <= 64 and dlt.mean() < 2.0, e


def test_vec_env_surface(golden):
    """Gymnasium-shaped classes: keys, shapes, dtypes, unbatched adapter, facade replay to a win."""
    from nclone_amd.replay import CompactReplay, validate_replays
    from nclone_amd.vec_env import NppEnvironment, NppVecEnvironment, NPlayHeadless

    c = golden.z("corpus")
    levels = [c["m0"], c["m2"]]
    v = NppVecEnvironment(levels, 130, enable_visual_observations=True, output="numpy")
    obs, info = v.reset()
    assert obs["game_state"].shape == (130, 41) and obs["game_state"].dtype == np.float32
    assert obs["action_mask"].shape == (130, 6) and obs["action_mask"].dtype == np.int8
    assert obs["entity_positions"].shape == (130, 6) and obs["player_frame"].shape == (130, 84, 84, 1)
    assert obs["global_view"].shape == (130, 176, 100, 1) and obs["global_view"].dtype == np.uint8
    obs, rew, term, trunc, info = v.step(np.full(130, 2, dtype=np.uint8))
    assert rew.shape == (130,) and term.dtype == np.bool_ and trunc.shape == (130,)
    assert (info["frames_executed"] == 4).all()
    v.close()
    # replay 0 (very_simple_100001): 37 inputs, win at tick 37 at (812.0036311681246, 542.0) -- SURVEY.md section 4
    h = NPlayHeadless()
    h.load_map_from_map_data(c["m0"])
    won_at = None
    for k, byte in enumerate(c["in0"]):
        hor = 0 if ((byte >> 1) & 1 and (byte >> 2) & 1) else (-1 if (byte >> 2) & 1 else (1 if (byte >> 1) & 1 else 0))
        h.tick(hor, int(byte) & 1)
        if h.ninja_has_won():
            won_at = k + 1
            break
    assert won_at == 37 and h.ninja_position() == (812.0036311681246, 542.0) and h.sim.frame == 37
    h.exit()
    # single-env adapter: python scalars
    e = NppEnvironment(map_data=c["m0"])
    o, i = e.reset()
    assert o["game_state"].shape == (41,)
    o, rwd, term, trunc, inf = e.step(2)
    assert isinstance(rwd, float) and isinstance(term, bool) and inf["frame_skip_stats"]["frames_executed"] == 4
    e.close()
    # batched replay validation == the survey's table (128 of 130 win; here the in-scope subset)
    idx = golden.in_scope_replays()[:40]
    reps = [CompactReplay(bytes(c["m%d" % i]), list(c["in%d" % i])) for i in idx]
    res = validate_replays(reps)
    final = c["final"]
    for k, i in enumerate(idx):
        assert res[k]["ticks"] == int(final[i, 0]) and res[k]["won"] == (int(final[i, 1]) == 8)
        assert (res[k]["x"], res[k]["y"]) == (final[i, 2], final[i, 3])


def test_switch_states_and_facade_entity_views(golden):
    """switch_states (25 f32: 5 locked doors x [switch xy, "door" xy = switch xy as in the reference, collected]) on the
    locked-door levels, cross-checked with the entity dump; NPlayHeadless.locked_doors / get_mine_entities.
    The reference cannot run this code here (gymnasium), so the layout follows npp_environment.py:1782-1847 by reading."""
    from nclone_amd.engine import NppBatch, compile_level_entities
    from nclone_amd.levels import door_levels
    from nclone_amd.vec_env import NPlayHeadless, NppVecEnvironment

    levels, _ = door_levels()
    n = 4 * len(levels)
    b = NppBatch(n, autoreset=False)
    b.load_levels(levels)
    lvl = np.arange(n) % len(levels)
    b.assign_levels(lvl)
    rng = np.random.default_rng(4)
    acts = torch.from_numpy(rng.integers(0, 6, size=(200, n)).astype(np.uint8)).cuda()
    opened = 0
    for s in range(200):
        b.step(acts[s])
    ss = b.switch_states().cpu().numpy()
    for e in range(n):
        rows = compile_level_entities(levels[lvl[e]])
        st = b.dump_entities(e)
        doors = [(r, st[i]) for i, r in enumerate(rows) if int(r[0]) == 6][:5]
        want = np.zeros(25, dtype=np.float32)
        for k, (r, s_) in enumerate(doors):
            x = np.float32(np.clip(r[1] / 1056.0, 0.0, 1.0))
            y = np.float32(np.clip(r[2] / 600.0, 0.0, 1.0))
            want[5 * k : 5 * k + 5] = [x, y, x, y, 0.0 if (s_ & 1) else 1.0]
            opened += int(not (s_ & 1))
        assert np.array_equal(ss[e], want), (e, ss[e], want)
    assert len(levels) > 0
    v = NppVecEnvironment(levels[:2], 8, enable_switch_states=True)
    obs, _ = v.reset()
    assert obs["switch_states"].shape == (8, 25) and obs["switch_states"].dtype == torch.float32
    v.close()
    hp = NPlayHeadless()
    hp.load_map_from_map_data(levels[0])
    d = hp.locked_doors()
    assert len(d) >= 1 and all(x.active and x.closed for x in d)
    m1, m21 = hp.get_mine_entities()
    assert all(m.state == 0 for m in m1) and all(m.state == 1 for m in m21)
    hp.exit()


def test_vec_env_on_zoo_levels_with_all_observations(golden):
    """The Gymnasium-shaped vector env on levels full of moving entities with every observation switched on: shapes, dtypes,
    a few hundred steps with auto-reset, frames that actually show the entities (not a constant picture)."""
    from nclone_amd.levels import zoo_levels
    from nclone_amd.vec_env import NppVecEnvironment

    levels, _ = zoo_levels()
    v = NppVecEnvironment(levels[:6], 48, level_ids=np.arange(48) % 6, enable_visual_observations=True, enable_spatial_context=True,
                          enable_switch_states=True, output="numpy")
    obs, _ = v.reset()
    assert set(obs) == {"game_state", "action_mask", "entity_positions", "spatial_context", "switch_states", "player_frame", "global_view",
                        "player_x", "player_y", "switch_x", "switch_y", "exit_door_x", "exit_door_y", "switch_activated"}
    rng = np.random.default_rng(2)
    ended = 0
    for s in range(120):
        obs, rew, term, trunc, info = v.step(rng.integers(0, 6, size=48).astype(np.uint8))
        ended += int(term.sum() + trunc.sum())
    assert obs["global_view"].shape == (48, 176, 100, 1) and obs["global_view"].std() > 5
    assert obs["spatial_context"].shape == (48, 112) and obs["switch_states"].shape == (48, 25)
    assert np.isfinite(obs["game_state"]).all() and np.abs(obs["game_state"]).max() <= 1.0 + 1e-6
    assert ended > 0 and set(np.unique(info["death_cause_code"])) <= {0, 1, 2}
    v.close()


def test_player_frame_config3_full_size(golden):
    """Config 3 at size: 8192 envs on the mine level set, frames rendered after every step of a random rollout with
    auto-reset.  Replica envs (same level, same actions) must produce identical frames whatever workgroup renders them; a
    sample of frames is compared with the numpy restatement (same tolerance as the small tests; parity of the raster
    itself is unpinned -- no cairo / cv2 reference frame exists); frames are real pictures (not constant)."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import mine_levels
    from tests.raster_ref import player_frame

    levels, _ = mine_levels()
    N = 8192
    lvl = (np.arange(N) // 64) % len(levels)
    steps = 40
    acts_np = np.random.default_rng(1).integers(0, 6, size=(steps, N)).astype(np.uint8).reshape(steps, N // 64, 64)
    acts_np[:, :, 32:] = acts_np[:, :, :32]
    acts = torch.from_numpy(acts_np.reshape(steps, N)).cuda()
    b = NppBatch(N, autoreset=True, outputs=("player_frame",))
    b.load_levels(levels)
    b.assign_levels(lvl)
    for s in range(steps):
        b.step(acts[s])
        b.render_player_frame()
    frames = b.out.t["player_frame"].cpu().numpy()[..., 0]
    fb = frames.reshape(N // 64, 64, 84, 84)
    assert np.array_equal(fb[:, :32], fb[:, 32:])
    # with the reference's axis swap a player at x > 642 gets an all-padding frame: not every frame is a picture
    assert (frames.reshape(N, -1).std(axis=1) > 1).mean() > 0.3
    f, _ = b.dump_state()
    for e in np.random.default_rng(3).choice(N, size=48, replace=False):
        ref, edge = player_frame(levels[lvl[e]], b.dump_entities(int(e)), f[e, 0], f[e, 1], centered=False)
        d = np.abs(frames[e].astype(np.int64) - ref.astype(np.int64))
        assert len(np.argwhere((d > 0) & ~edge)) == 0 and d.max(initial=0) <= 64 and d.mean() < 2.0, e


@pytest.mark.gpu
@pytest.mark.parametrize("centered", [False, True])
def test_player_frame_windows_at_the_canvas_edges(golden, centered):
    """Windows clipped by the canvas on every side (fewer rows and / or fewer columns than 84: padding rows in the strip pass, the
    generic span path for narrow windows) on mine-dense levels: spawn positions moved to the map's edges and corners, frame of the
    reset state against the numpy restatement (same tolerance as the other raster tests; raster parity itself is unpinned)."""
    from nclone_amd.engine import NppBatch
    from nclone_amd.levels import mine_levels
    from tests.raster_ref import player_frame

    base, _ = mine_levels()
    dense = sorted(range(len(base)), key=lambda i: -len(base[i]))[:3]
    spots = [(30, 30), (300, 30), (30, 300), (620, 300), (300, 570), (1026, 30), (1026, 570), (30, 570), (48, 576), (558, 42), (642, 300),
             (528, 288), (41, 41), (43, 43)]
    levels = []
    for li in dense:
        for (x, y) in spots:
            m = np.array(base[li], dtype=np.float64).copy()
            m[1231], m[1232] = x // 6, y // 6
            levels.append(m)
    n = len(levels)
    b = NppBatch(n, autoreset=False, frame_centered=centered)
    b.load_levels(levels)
    b.assign_levels(np.arange(n))
    b.reset()
    out = torch.zeros((n, 84, 84), dtype=torch.uint8, device="cuda")
    b.render_player_frame(out)
    frames = out.cpu().numpy()
    f, _ = b.dump_state()
    shapes = set()
    for e in range(n):
        ref, edge = player_frame(levels[e], b.dump_entities(e), f[e, 0], f[e, 1], centered=centered)
        got = frames[e].astype(np.int64)
        d = np.abs(got - ref.astype(np.int64))
        bad = np.argwhere((d > 0) & ~edge)
        assert len(bad) == 0, (e, f[e, :2], [(tuple(p), int(got[tuple(p)]), int(ref[tuple(p)])) for p in bad[:8]])
        assert d.max(initial=0) <= 64 and d.mean() < 2.0
        shapes.add((int((ref.sum(axis=1) > 0).sum()) < 84, int((ref.sum(axis=0) > 0).sum()) < 84))
    assert len(shapes) >= 3   # full frames, frames with padding rows, frames with padding columns
