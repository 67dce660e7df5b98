"""Level sets used by bench.py / smoke() / tests: raw map_data blobs stored as data fixtures.

The blobs under tests/golden/ are DATA (map bytes the reference's own replay corpus holds, and generated maps
dumped by tests/golden/make_golden.py); nothing here imports the reference or the oracle.
"""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_GOLDEN = os.path.join(_ROOT, "tests", "golden")


def _names(z, key="names"):
    return bytes(z[key]).decode().split("\n")


def curriculum0_levels():
    """Exit+switch-only levels ("curriculum 0"): the 78 bc_replays maps whose only
    entities are exit door + switch, plus generated maze:tiny and hills:simple seeds 100001..100025 = 128 levels."""
    c = np.load(os.path.join(_GOLDEN, "corpus.npz"))
    sigs = _names(c, "sigs")
    out, tags = [], []
    for i, s in enumerate(sigs):
        if s == "3":
            out.append(c["m%d" % i].astype(np.float64))
            tags.append("replay:%d" % i)
    g = np.load(os.path.join(_GOLDEN, "levels_gen.npz"))
    for k, n in enumerate(_names(g)):
        if n.startswith("maze:tiny") or n.startswith("hills:simple"):
            out.append(g["L%d" % k])
            tags.append(n)
    return out, tags


def mine_levels():
    """Levels with toggle mines ("curriculum 2", config 3): (1,3) replay maps + generated corridor levels."""
    c = np.load(os.path.join(_GOLDEN, "corpus.npz"))
    sigs = _names(c, "sigs")
    out, tags = [], []
    for i, s in enumerate(sigs):
        if s == "1,3":
            out.append(c["m%d" % i].astype(np.float64))
            tags.append("replay:%d" % i)
    g = np.load(os.path.join(_GOLDEN, "levels_gen.npz"))
    for k, n in enumerate(_names(g)):
        if n.startswith("hcorr:mines"):
            out.append(g["L%d" % k])
            tags.append(n)
    return out, tags


def door_levels():
    """Levels with locked doors ("curriculum 4", config 5)."""
    c = np.load(os.path.join(_GOLDEN, "corpus.npz"))
    sigs = _names(c, "sigs")
    out, tags = [], []
    for i, s in enumerate(sigs):
        if s == "3,6":
            out.append(c["m%d" % i].astype(np.float64))
            tags.append("replay:%d" % i)
    g = np.load(os.path.join(_GOLDEN, "levels_gen.npz"))
    for k, n in enumerate(_names(g)):
        if n.startswith("hcorr:door") or n.startswith("test_maps"):
            out.append(g["L%d" % k])
            tags.append(n)
    return out, tags


def c3_mixed_levels():
    """The 512-level "curriculum 3, mixed map set" of BASELINE.json config 4: the 128 curriculum-0
    levels + the 64 mine levels + 320 generated levels of the reference's `simpler` and `simple` categories restricted to
    entity types {1, 3, 4, 21} (tests/golden/make_golden_c3.py; map_generation/generator_configs.py:732-760)."""
    out, tags = curriculum0_levels()
    m, t = mine_levels()
    out, tags = out + m, tags + t
    g = np.load(os.path.join(_GOLDEN, "levels_c3.npz"))
    for k, n in enumerate(_names(g)):
        out.append(g["L%d" % k])
        tags.append(n)
    return out, tags


def zoo_levels():
    """The 26 bc_replays maps with the entity zoo (doors, launch / boost pads, one-ways, drones, bounce blocks, thwumps,
    death balls, shove thwumps): SURVEY.md 8(f) row 2."""
    c = np.load(os.path.join(_GOLDEN, "corpus.npz"))
    z = np.load(os.path.join(_GOLDEN, "zoo.npz"))
    out, tags = [], []
    for i in z["idx"]:
        out.append(c["m%d" % i].astype(np.float64))
        tags.append("replay:%d" % i)
    return out, tags
