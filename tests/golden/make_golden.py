#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

This script is the only place in the repository that imports the reference
(`/root/reference`, Python, importable in the build container only).  It is
committed so that the provenance of every fixture is reproducible; it is never
imported by the package, the tests, bench.py or smoke().  Nothing here copies
reference source text: the outputs are pure data (inputs + expected outputs).

Run (build container only):

    HOME=/tmp/orahome PYTHONPATH=/root/reference python3 tests/golden/make_golden.py

Reference entry points exercised (file:line in /root/reference):
  * nclone/nsim.py:51  Simulator.load, :62 reset, :221 tick
  * nclone/nplay_headless.py:735 get_ninja_state, :433 get_action_mask
  * nclone/replay/gameplay_recorder.py:67-129 (format re-parsed here, the
    package itself is not importable: it pulls gymnasium)
  * nclone/map_generation/generator_factory.py:75 create_from_preset
  * harness semantics of tools/test_replay_playback.py:14-80 and of
    nclone/gym_environment/base_environment.py:535-609 (frame-skip loop)

Outputs (all numpy .npz, no pickles):
  corpus.npz      all 130 bc_replays: map bytes, input bytes, entity signature,
                  oracle end state (ticks run, final ninja state, final x/y)
  traj.npz        per-tick fp64 (x, y, vx, vy) + discrete state for every replay
                  whose entity types are in IN_SCOPE_TYPES
  gstate.npz      per-tick get_ninja_state() f32[40] + action-mask bits, same set
  csr.npz         ordered per-cell collision-segment dump for every distinct map
  levels_gen.npz  generated level blobs (fp64) for the bench/parity level sets
  rollouts.npz    random-action frame-skip rollouts with reset-on-termination on
                  a spread of levels (replay maps + generated), per tick
"""
import glob
import hashlib
import os
import struct
import sys

import numpy as np

REF = "/root/reference"
if REF not in sys.path:
    sys.path.insert(0, REF)

from nclone.nsim import Simulator  # noqa: E402
from nclone.sim_config import SimConfig  # noqa: E402
from nclone.nplay_headless import NPlayHeadless  # noqa: E402
from nclone.map_generation.generator_factory import GeneratorFactory  # noqa: E402

def load_spatial_context():
    """nclone/gym_environment/spatial_context.py is numpy-only, but importing it normally runs the package
    __init__ of nclone.gym_environment, which needs gymnasium (not installed, an ordinary ModuleNotFoundError).
    The module file is therefore executed directly under its own dotted name, with a bare package object registered
    for nclone.gym_environment so that its relative imports (..constants) resolve to the real reference modules.
    Nothing is stubbed or emulated: every line that runs is the reference's."""
    import importlib.util
    import types

    pkg = types.ModuleType("nclone.gym_environment")
    pkg.__path__ = [os.path.join(REF, "nclone", "gym_environment")]
    sys.modules.setdefault("nclone.gym_environment", pkg)
    spec = importlib.util.spec_from_file_location(
        "nclone.gym_environment.spatial_context", os.path.join(REF, "nclone", "gym_environment", "spatial_context.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nclone.gym_environment.spatial_context"] = mod
    spec.loader.exec_module(mod)
    return mod


def inner_tiles(sim):
    """level_data.tiles as built by base_environment.py:3346-3354 (inner 23 x 42 area)."""
    t = np.zeros((23, 42), dtype=np.int32)
    for (x, y), tid in sim.tile_dic.items():
        if 0 <= x - 1 < 42 and 0 <= y - 1 < 23:
            t[y - 1, x - 1] = int(tid)
    return t


def spatial_context_row(sc, hp, tiles):
    """NppEnvironment._compute_spatial_context (npp_environment.py:2318-2360)."""
    pos = hp.ninja_position()
    vel = hp.ninja_velocity()
    grid = sc.compute_local_tile_grid(pos, tiles)
    m1, m21 = hp.get_mine_entities()
    ov = sc.compute_mine_overlay_from_entities(pos, vel, m1, m21, 1056.0, 600.0)
    return np.concatenate([grid, ov]).astype(np.float32)


OUT = os.path.dirname(os.path.abspath(__file__))
IN_SCOPE_TYPES = {1, 2, 3, 6, 21}
ACTIONS = [(0, 0), (-1, 0), (1, 0), (0, 1), (-1, 1), (1, 1)]
N_DISC = 20


def parse_replay(data):
    first = struct.unpack("<I", data[:4])[0]
    if first <= 100:
        _, ml, il = struct.unpack("<III", data[:12])
        m = data[13:13 + ml]
        i = data[13 + ml:13 + ml + il]
    else:
        ml, il = struct.unpack("<II", data[:8])
        m = data[8:8 + ml]
        i = data[8 + ml:8 + ml + il]
    return m, i


def controls(b):
    j = b & 1
    r = (b >> 1) & 1
    l = (b >> 2) & 1
    h = 0 if (l and r) else (-1 if l else (1 if r else 0))
    return h, j


def signature(sim):
    return tuple(sorted(k for k, v in sim.entity_dic.items() if v))


def exit_switch(sim):
    for e in reversed(sim.entity_dic.get(3, [])):
        if type(e).__name__ == "EntityExitSwitch":
            return e
    return None


def disc_row(sim):
    n = sim.ninja
    sw = exit_switch(sim)
    g_fall = n.applied_gravity > 0.05
    d_reg = n.applied_drag > 0.95
    wn = int(n.wall_normal)
    row = [
        n.state, int(n.airborn), int(n.walled), wn + 1,
        n.jump_buffer + 1, n.floor_buffer + 1, n.wall_buffer + 1, n.launch_pad_buffer + 1,
        min(n.floor_count, 255), min(n.ceiling_count, 255), min(n.jump_duration, 255),
        int(g_fall), int(d_reg), int(sw.active) if sw is not None else 2,
        min(n.gold_collected, 255), min(n.doors_opened, 255),
        min(n.frames_airborne, 255), min(n.state_change_frame, 255),
        int(n.airborn_old), int(n.jump_input_old),
    ]
    assert len(row) == N_DISC
    return row


def mask_bits(mask):
    b = 0
    for i, m in enumerate(mask):
        if m:
            b |= 1 << i
    return b


def dump_csr(sim):
    """Ordered dump of what the ninja's region queries can return (spatial index),
    x-major over cells, list order inside a cell."""
    rows = []
    idx = sim.spatial_segment_index
    for cx in range(44):
        for cy in range(25):
            info = idx.cell_data.get((cx, cy))
            if not info:
                continue
            for s in info["segments"]:
                if s.type == "linear":
                    rows.append([cx, cy, 0, s.x1, s.y1, s.x2, s.y2, int(s.oriented)])
                else:
                    rows.append([cx, cy, 1, s.xpos, s.ypos, s.hor, s.ver, int(s.convex)])
    a = np.array(rows, dtype=np.float64).reshape(-1, 8)
    assert np.all(a == np.round(a))
    return a.astype(np.int16)


def dump_entities(sim):
    """type, x, y, cell x, cell y, initial state/aux in map order per type key."""
    rows = []
    for k in sorted(sim.entity_dic):
        for e in sim.entity_dic[k]:
            st = getattr(e, "state", -1)
            rows.append([k, e.type, e.xpos, e.ypos, e.cell[0], e.cell[1], st, int(e.active)])
    return np.array(rows, dtype=np.float64).reshape(-1, 8)


def gen_level(gen_type, preset, seed):
    g = GeneratorFactory.create_from_preset(gen_type, preset, seed)
    g.generate(seed=seed)
    return np.array(g.map_data(), dtype=np.float64)


def level_types(m):
    types = set()
    i = 1230
    n = len(m)
    while i + 4 < n:
        t = int(m[i])
        types.add(t)
        if t in (6, 8):
            if i + 9 < n and m[i + 7] != 0 and m[i + 8] == 0 and m[i + 9] == 0:
                i += 10
            else:
                i += 9
        else:
            i += 5
    return types - {0, 4}


def to_list(m):
    """Reference wants a Python list; ints stay ints where integral (bit-identical maths)."""
    return [int(v) if float(v).is_integer() else float(v) for v in m]


def main():
    files = sorted(glob.glob(os.path.join(REF, "bc_replays", "*.replay")))
    assert len(files) == 130, len(files)
    names = []
    corpus = {}
    traj = {}
    gst = {}
    csr = {}
    final = np.zeros((len(files), 4), dtype=np.float64)
    sigs = []
    total_ticks = 0
    scope_ticks = 0
    for i, f in enumerate(files):
        name = os.path.basename(f)[:-len(".replay")]
        names.append(name)
        m, inp = parse_replay(open(f, "rb").read())
        corpus["m%d" % i] = np.frombuffer(m, dtype=np.uint8)
        corpus["in%d" % i] = np.frombuffer(inp, dtype=np.uint8)
        hp = NPlayHeadless(enable_rendering=False)
        hp.load_map_from_map_data(list(m))
        sim = hp.sim
        sig = signature(sim)
        sigs.append(",".join(map(str, sig)))
        in_scope = set(sig) <= IN_SCOPE_TYPES
        key = hashlib.sha256(bytes(m[184:1150])).hexdigest()[:16]
        if key not in csr:
            csr[key] = dump_csr(sim)
        corpus["csr%d" % i] = np.frombuffer(key.encode(), dtype=np.uint8)
        if in_scope:
            corpus["ent%d" % i] = dump_entities(sim)
        rows, drows, grows, krows = [], [], [], []
        n = 0
        for b in inp:
            h, j = controls(b)
            hp.tick(h, j)
            n += 1
            if in_scope:
                nj = sim.ninja
                rows.append([nj.xpos, nj.ypos, nj.xspeed, nj.yspeed])
                drows.append(disc_row(sim))
                grows.append(hp.get_ninja_state())
                krows.append(mask_bits(hp.get_action_mask()))
            if sim.ninja.state in (6, 7, 8):
                break
        total_ticks += n
        final[i] = [n, sim.ninja.state, sim.ninja.xpos, sim.ninja.ypos]
        if in_scope:
            scope_ticks += n
            traj["t%d" % i] = np.array(rows, dtype=np.float64).reshape(-1, 4)
            traj["d%d" % i] = np.array(drows, dtype=np.uint8).reshape(-1, N_DISC)
            gst["g%d" % i] = np.array(grows, dtype=np.float64).astype(np.float32).reshape(-1, 40)
            gst["k%d" % i] = np.array(krows, dtype=np.uint8)
    # two raw replay files (data the reference's corpus holds) for the file-format parser: one per header version
    for want, key in ((0, "raw_v0"), (1, "raw_v1")):
        for f in files:
            data = open(f, "rb").read()
            is_v1 = struct.unpack("<I", data[:4])[0] <= 100
            if int(is_v1) == want:
                corpus[key] = np.frombuffer(data, dtype=np.uint8)
                corpus[key + "_index"] = np.array([files.index(f)])
                break
    corpus["names"] = np.frombuffer("\n".join(names).encode(), dtype=np.uint8)
    corpus["sigs"] = np.frombuffer("\n".join(sigs).encode(), dtype=np.uint8)
    corpus["final"] = final
    print("replays", len(files), "ticks", total_ticks, "in-scope ticks", scope_ticks,
          "wins", int(np.sum(final[:, 1] == 8)))

    # ---- generated level sets ------------------------------------
    lv = {}
    lvnames = []

    def add_level(tag, m):
        lvnames.append(tag)
        lv["L%d" % (len(lvnames) - 1)] = m

    for s in range(100001, 100026):
        add_level("maze:tiny:%d" % s, gen_level("maze", "tiny", s))
    for s in range(100001, 100026):
        add_level("hills:simple:%d" % s, gen_level("hills", "simple", s))
    n_mines = 0
    for s in range(100001, 100045):
        try:
            m = gen_level("horizontal_corridor", "simplest_with_mines", s)
        except Exception:
            m = gen_level("horizontal_corridor", "simplest", s)
        if level_types(m) <= IN_SCOPE_TYPES:
            add_level("hcorr:mines:%d" % s, m)
            n_mines += 1
    n_doors = 0
    for s in range(100001, 100061):
        m = gen_level("horizontal_corridor", "simplest", s)
        t = level_types(m)
        if 6 in t and t <= IN_SCOPE_TYPES:
            add_level("hcorr:door:%d" % s, m)
            n_doors += 1
    for tm in ("switch-simple", "switch-puzzle-1", "switch-puzzle-2"):
        p = os.path.join(REF, "nclone", "test_maps", tm)
        if os.path.isfile(p):
            m = np.frombuffer(open(p, "rb").read(), dtype=np.uint8).astype(np.float64)
            if level_types(m) <= IN_SCOPE_TYPES:
                add_level("test_maps:%s" % tm, m)
    lv["names"] = np.frombuffer("\n".join(lvnames).encode(), dtype=np.uint8)
    print("generated levels", len(lvnames), "mines", n_mines, "doors", n_doors)
    for k, tag in enumerate(lvnames):
        sim = Simulator(SimConfig(enable_anim=False))
        sim.load(to_list(lv["L%d" % k]))
        key = hashlib.sha256(np.ascontiguousarray(lv["L%d" % k][184:1150]).tobytes()).hexdigest()[:16]
        lv["csr%d" % k] = np.frombuffer(key.encode(), dtype=np.uint8)
        lv["ent%d" % k] = dump_entities(sim)
        if key not in csr:
            csr[key] = dump_csr(sim)

    # ---- random-action rollouts with frame skip + reset on termination -------------
    ro = {}
    ro_levels = []
    rng_pick = np.random.default_rng(7)
    in_scope_idx = [i for i in range(len(files)) if "t%d" % i in traj]
    picks = list(rng_pick.choice(in_scope_idx, size=14, replace=False))
    # make sure mines and doors are represented
    for want in ("1,3", "3,6", "1,2,3", "1,2,3,21"):
        for i in in_scope_idx:
            if sigs[i] == want and i not in picks:
                picks.append(i)
                break
    for i in picks:
        ro_levels.append(("replay:%s" % names[i], corpus["m%d" % i].astype(np.float64)))
    for k, tag in enumerate(lvnames):
        if tag.endswith((":100001", ":100002", ":100003")) or tag.startswith("test_maps"):
            ro_levels.append((tag, lv["L%d" % k]))
    ro_names = []
    n_ro_ticks = 0
    sc = load_spatial_context()
    for r, (tag, m) in enumerate(ro_levels):
        ro_names.append(tag)
        hp = NPlayHeadless(enable_rendering=False)
        hp.load_map_from_map_data(to_list(m))
        sim = hp.sim
        tiles = inner_tiles(sim)
        sc.reset_mine_overlay_cache()          # npp_environment.py:569-571 (reset)
        sc_rows = [spatial_context_row(sc, hp, tiles)]   # observation returned by reset()
        sc_term_rows = []
        steps = 400
        acts = np.random.default_rng(1000 + r).integers(0, 6, size=steps).astype(np.uint8)
        rows, drows, srows, grows, krows = [], [], [], [], []
        for a in acts:
            h, j = ACTIONS[a]
            executed = 0
            term = 0
            for _ in range(4):
                hp.tick(h, j)
                executed += 1
                nj = sim.ninja
                rows.append([nj.xpos, nj.ypos, nj.xspeed, nj.yspeed])
                drows.append(disc_row(sim))
                if nj.state in (6, 7, 8):
                    term = 1 if nj.state == 8 else 2
                    break
            grows.append(hp.get_ninja_state())
            krows.append(mask_bits(hp.get_action_mask()))
            srows.append([executed, term, sim.frame])
            sc_term_rows.append(spatial_context_row(sc, hp, tiles))   # observation of this step (terminal if term)
            if term:
                hp.reset()
                sim = hp.sim
                sc.reset_mine_overlay_cache()
                sc_rows.append(spatial_context_row(sc, hp, tiles))    # what a vector env returns after auto-reset
            else:
                sc_rows.append(sc_term_rows[-1])
        n_ro_ticks += len(rows)
        ro["m%d" % r] = m
        ro["a%d" % r] = acts
        ro["t%d" % r] = np.array(rows, dtype=np.float64).reshape(-1, 4)
        ro["d%d" % r] = np.array(drows, dtype=np.uint8).reshape(-1, N_DISC)
        ro["s%d" % r] = np.array(srows, dtype=np.int32).reshape(-1, 3)
        ro["g%d" % r] = np.array(grows, dtype=np.float64).astype(np.float32).reshape(-1, 40)
        ro["k%d" % r] = np.array(krows, dtype=np.uint8)
        ro["sc%d" % r] = np.stack(sc_rows).astype(np.float32)        # [steps + 1, 112]: reset obs, then per step
        ro["sct%d" % r] = np.stack(sc_term_rows).astype(np.float32)  # [steps, 112]: pre-reset observation
    ro["names"] = np.frombuffer("\n".join(ro_names).encode(), dtype=np.uint8)
    print("rollouts", len(ro_names), "ticks", n_ro_ticks)

    np.savez_compressed(os.path.join(OUT, "corpus.npz"), **corpus)
    np.savez_compressed(os.path.join(OUT, "traj.npz"), **traj)
    np.savez_compressed(os.path.join(OUT, "gstate.npz"), **gst)
    np.savez_compressed(os.path.join(OUT, "csr.npz"), **csr)
    np.savez_compressed(os.path.join(OUT, "levels_gen.npz"), **lv)
    np.savez_compressed(os.path.join(OUT, "rollouts.npz"), **ro)
    for fn in ("corpus", "traj", "gstate", "csr", "levels_gen", "rollouts"):
        print(fn, os.path.getsize(os.path.join(OUT, fn + ".npz")))


if __name__ == "__main__":
    main()
