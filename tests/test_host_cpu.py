"""CPU-only checks of the product's host side: the native level compiler against the reference's ordered segment
and entity dumps, the C ABI surface (every symbol declared in include/npp_amd.h is exported), loud failure
without a GPU, replay file format, host helpers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from nclone_amd import build_native

    build_native.build()
    from nclone_amd import _native

    return _native


def test_header_symbols_exported(native):
    hdr = open(os.path.join(ROOT, "include", "npp_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(npp_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = C.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(native.EXPORTS) == declared


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/npp_amd.h must compile as C99 (no C++-isms, no torch / HIP types)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "npp_amd.h"\nint main(void) { npp_handle h = 0; (void)h; return NPP_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           "-c", str(src), "-o", str(tmp_path / "hdr.o")])


def test_level_compiler_matches_reference_tables(native, golden):
    from nclone_amd.engine import compile_level_entities, compile_level_segments

    c, csr, lg = golden.z("corpus"), golden.z("csr"), golden.z("levels_gen")
    kind = {1: 1, 21: 1, 2: 2, 3: 3, 4: 4, 6: 6}
    sigs = golden.names("corpus", "sigs")
    for i in range(len(sigs)):
        m = c["m%d" % i].astype(np.float64)
        rows, uns = compile_level_segments(m)
        assert np.array_equal(rows, csr[bytes(c["csr%d" % i]).decode()]), i
        assert uns == 0          # every entity type of the reference's factory is simulated
        if "ent%d" % i in c.files:
            ref = sorted((kind[int(t)], x, y, cx, cy) for _, t, x, y, cx, cy, _, _ in c["ent%d" % i])
            got = sorted((int(k), x, y, cx, cy) for k, x, y, cx, cy, _ in compile_level_entities(m))
            assert ref == got, i
    for k in range(len(golden.names("levels_gen"))):
        rows, uns = compile_level_segments(lg["L%d" % k])
        assert uns == 0 and np.array_equal(rows, csr[bytes(lg["csr%d" % k]).decode()]), k
        ref = sorted((kind[int(t)], x, y, cx, cy) for _, t, x, y, cx, cy, _, _ in lg["ent%d" % k])
        got = sorted((int(kk), x, y, cx, cy) for kk, x, y, cx, cy, _ in compile_level_entities(lg["L%d" % k]))
        assert ref == got, k


def test_level_compiler_c3_mixed_set(native, golden):
    """Config 4's 320 extra levels: entity tables of all, ordered segment dumps of every 4th (reference dumps)."""
    from nclone_amd.engine import compile_level_entities, compile_level_segments
    from nclone_amd.levels import c3_mixed_levels

    g = golden.z("levels_c3")
    kind = {1: 1, 21: 1, 2: 2, 3: 3, 4: 4, 6: 6}
    n = len(golden.names("levels_c3"))
    for k in range(n):
        m = g["L%d" % k]
        ref = sorted((kind[int(t)], x, y, cx, cy) for _, t, x, y, cx, cy, _, _ in g["ent%d" % k])
        got = sorted((int(kk), x, y, cx, cy) for kk, x, y, cx, cy, _ in compile_level_entities(m))
        assert ref == got, k
        if "csr%d" % k in g.files:
            rows, uns = compile_level_segments(m)
            assert uns == 0 and np.array_equal(rows, g["c" + bytes(g["csr%d" % k]).decode()]), k
    levels, tags = c3_mixed_levels()
    assert len(levels) == 512 and len(set(tags)) == 512


def test_level_compiler_zoo_tables_match_oracle(native, golden, oracle_mod):
    """Grid edges (what drones / thwumps test) and the mover table of the 26 zoo maps against the oracle's own build of
    them (the oracle is pinned by the reference's replays, where a wrong edge sends a drone the wrong way)."""
    from nclone_amd.engine import compile_level_zoo

    c, z = golden.z("corpus"), golden.z("zoo")
    for i in z["idx"]:
        m = c["m%d" % i].astype(np.float64)
        hor, ver, mov = compile_level_zoo(m)
        o = oracle_mod.Oracle("pow")
        o.load(m)
        ohor, over = o.edges()
        assert np.array_equal(hor, ohor) and np.array_equal(ver, over), i
        # movers: entity_dic order = type ascending, creation order inside a type
        key = mov[:, 0] * 1e6 + mov[:, 3]
        assert np.all(np.diff(key) > 0), i
        raw = m.astype(int)
        want = sum(int(t in (14, 17, 20, 25, 26, 28)) for t in raw[1230::5][: (len(raw) - 1230) // 5]) if 6 not in raw[1230::5] and 8 not in raw[1230::5] else len(mov)
        assert len(mov) == want, i


def test_level_compiler_edge_cases(native):
    from nclone_amd._native import NppError
    from nclone_amd.engine import compile_level_entities, compile_level_segments

    with pytest.raises(NppError):
        compile_level_segments(np.zeros(100))            # too short
    empty = np.zeros(1245)
    empty[1231], empty[1232] = 10, 10
    rows, uns = compile_level_segments(empty)             # empty interior: the border ring's inner and outer half-edges
    inner, outer = 2 * (2 * 42 + 2 * 23), 2 * (2 * 44 + 2 * 25)
    assert uns == 0 and len(rows) == inner + outer
    assert len(compile_level_entities(empty)) == 0
    full = empty.copy()
    full[184:1150] = 1                                    # solid interior: every edge cancels
    assert len(compile_level_segments(full)[0]) == outer
    glitch = empty.copy()
    glitch[184:1150] = 35                                 # glitched tiles 34..37 are empty for collision
    assert len(compile_level_segments(glitch)[0]) == len(rows)


def test_no_gpu_fails_loudly(native):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nclone_amd.engine import NppBatch

    with pytest.raises(RuntimeError):
        NppBatch(4)
    h = C.c_void_p()
    rc = native.lib().npp_create(4, 0, 0, C.byref(h))
    assert rc != 0 and b"no HIP device" in native.lib().npp_last_error(None)


def test_replay_format_v0_v1(golden):
    from nclone_amd.replay import CompactReplay, decode_input_to_controls

    c = golden.z("corpus")
    for key, ver in (("raw_v0", 0), ("raw_v1", 1)):
        raw = bytes(c[key])
        i = int(c[key + "_index"][0])
        r = CompactReplay.from_binary(raw)
        assert r.version == ver
        assert r.map_data == bytes(c["m%d" % i]) and r.input_sequence == list(c["in%d" % i])
        r2 = CompactReplay.from_binary(r.to_binary())
        assert r2.map_data == r.map_data and r2.input_sequence == r.input_sequence and r2.version == 1
    assert [decode_input_to_controls(b) for b in range(8)] == [(0, 0), (0, 1), (1, 0), (1, 1), (-1, 0), (-1, 1), (0, 0), (0, 1)]


def test_truncation_limit_formula():
    """gym_environment/truncation_calculator.py:19-57 evaluated by hand: (sqrt(A) * 20 + mines * 75) * 25 clipped to
    [1200, 10000]."""
    from nclone_amd.vec_env import calculate_truncation_limit

    assert calculate_truncation_limit(1.0, 0) == 1200            # 500 -> floor
    assert calculate_truncation_limit(36.0, 1) == 4875           # (120 + 75) * 25
    assert calculate_truncation_limit(100.0, 0) == 5000
    assert calculate_truncation_limit(400.0, 3) == 10000         # clipped to MAX_TIME_IN_FRAMES
    assert calculate_truncation_limit(50.0, 0) == int(np.sqrt(50.0) * 20.0 * 25)


def test_host_helpers():
    from nclone_amd import spaces
    from nclone_amd.distributed import shard_envs
    from nclone_amd.vec_env import ACTION_TABLE, controls_to_input_byte
    from nclone_amd.replay import decode_input_to_controls

    assert spaces.action_space().n == 6
    obs = spaces.observation_space(True)
    assert obs["game_state"].shape == (41,) and obs["player_frame"].shape == (84, 84, 1)
    for hor, jump in ACTION_TABLE:
        assert decode_input_to_controls(controls_to_input_byte(hor, jump)) == (hor, jump)
    parts = [shard_envs(65536 + 3, r, 8) for r in range(8)]
    assert sum(c for _, c in parts) == 65539 and parts[0][0] == 0
    assert all(parts[k][0] + parts[k][1] == parts[k + 1][0] for k in range(7))


def test_aabb_threshold_table_is_exact():
    """AABB_LO (npp_kernels.hip): entry i is the smallest double x with fl(x + 10) >= 12 i, and 12 i + 10 is the largest with
    fl(x - 10) <= 12 i -- the thresholds that replace the rounded box of get_single_closest_point's AABB test (physics.py:150-167).
    Checked on 64 doubles either side of every threshold."""
    import re

    src = open(os.path.join(ROOT, "nclone_amd", "csrc", "npp_kernels.hip")).read()
    body = src[src.index("__constant__ double AABB_LO[89] = {"):]
    body = body[:body.index("};")]
    vals = [float.fromhex(m) for m in re.findall(r"(-?0x1\.[0-9a-f]+p[+-]\d+)", body)]
    assert len(vals) == 89
    ten = np.float64(10.0)
    for i, t in enumerate(vals):
        b = np.float64(12.0 * i)
        x = np.float64(t)
        for _ in range(64):
            assert np.float64(x + ten) >= b
            x = np.nextafter(x, np.inf)
        x = np.nextafter(np.float64(t), -np.inf)
        for _ in range(64):
            assert not (np.float64(x + ten) >= b), i
            x = np.nextafter(x, -np.inf)
        x = np.float64(b + ten)
        for _ in range(64):
            assert np.float64(x - ten) <= b
            x = np.nextafter(x, -np.inf)
        x = np.nextafter(np.float64(b + ten), np.inf)
        for _ in range(64):
            assert not (np.float64(x - ten) <= b), i
            x = np.nextafter(x, np.inf)


def test_dynamic_truncation_limit_matches_reference():
    """npp_level_truncation_limit (the per-level limit npp_set_dynamic_truncation applies) and the Python helper against the
    reference's calculate_truncation_limit (tests/golden/trunc.npz: the function itself called on areas 0..4000 and on the
    surface areas of the 126 reachability fixture levels), and the surface area against the reference's `sa<k>`."""
    from nclone_amd.engine import level_truncation_limit
    from nclone_amd.vec_env import calculate_truncation_limit

    t = np.load(os.path.join(ROOT, "tests", "golden", "trunc.npz"))
    ref = {int(a): int(v) for a, v in zip(t["area"], t["limit"])}
    for a, v, vm in zip(t["area"], t["limit"], t["limit_mines"]):
        assert calculate_truncation_limit(int(a)) == int(v)
        assert [calculate_truncation_limit(int(a), m) for m in (1, 2, 5, 20)] == [int(x) for x in vm]
    assert 1200 in ref.values() and 10000 in ref.values() and len(set(ref.values())) > 300
    seen = set()
    for fx in ("reach.npz", "reach2.npz", "reach3.npz"):
        z = np.load(os.path.join(ROOT, "tests", "golden", fx))
        for k, name in enumerate(bytes(z["names"]).decode().split("\n")):
            lim, area = level_truncation_limit(z["m%d" % k])
            sa = int(z["sa%d" % k][0])
            assert area == sa and sa > 0, name
            assert lim == ref[sa], (name, sa)
            seen.add(lim)
    assert len(seen) > 20 and min(seen) == 1200
