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
