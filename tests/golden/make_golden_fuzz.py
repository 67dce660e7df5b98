#!/usr/bin/env python3
"""Reference runs on random "entity soup" levels (tests/fuzz_levels.py): pins the entity kinds and orientations the recorded
replay corpus barely contains -- regular doors (type 5), trap doors, shove thwumps, diagonal launch pads / one-ways, every
drone mode, horizontal and vertical thwumps -- against the REAL simulator.  Build container only; outputs are pure data.

    HOME=/tmp/orahome PYTHONPATH=/root/reference python3 tests/golden/make_golden_fuzz.py

Output fuzz.npz: n, and per level k: m<k> map_data (f64), a<k> actions u8[steps], t<k> f64[T, 4], d<k> u8[T, 20],
e<k> f64[T, 6] entity checksum per tick (make_golden_zoo.ent_row), s<k> i32[steps, 3] (executed, term, frame).
Episodes end with hp.reset() like the rollouts of make_golden.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
from make_golden_zoo import ent_row  # noqa: E402
from fuzz_levels import fuzz_level  # noqa: E402

from nclone.nplay_headless import NPlayHeadless  # noqa: E402


def main():
    c = np.load(os.path.join(HERE, "corpus.npz"))
    g = np.load(os.path.join(HERE, "levels_gen.npz"))
    sigs = bytes(c["sigs"]).decode().split("\n")
    bases = [c["m%d" % i].astype(np.float64) for i, s in enumerate(sigs) if s == "3"][::6]
    names = bytes(g["names"]).decode().split("\n")
    bases += [g["L%d" % k] for k, nme in enumerate(names) if nme.startswith("hcorr:mines")][::9]
    rng = np.random.default_rng(777)
    out = {}
    n = 28
    steps = 120
    ticks = 0
    kinds = set()
    for k in range(n):
        m = fuzz_level(bases[k % len(bases)], rng, keep_away=(0.0 if k % 2 else 160.0))
        acts = rng.integers(0, 6, size=steps).astype(np.uint8)
        if k % 4 == 0:
            acts[:] = 0
        hp = NPlayHeadless(enable_rendering=False)
        hp.load_map_from_map_data(mg.to_list(m))
        sim = hp.sim
        kinds |= set(mg.signature(sim))
        rows, drows, erows, srows = [], [], [], []
        for a in acts:
            h, j = mg.ACTIONS[a]
            executed = term = 0
            for _ in range(4):
                hp.tick(h, j)
                executed += 1
                nj = sim.ninja
                rows.append([nj.xpos, nj.ypos, nj.xspeed, nj.yspeed])
                drows.append(mg.disc_row(sim))
                erows.append(ent_row(sim))
                if nj.state in (6, 7, 8):
                    term = 1 if nj.state == 8 else 2
                    break
            srows.append([executed, term, sim.frame])
            if term:
                hp.reset()
                sim = hp.sim
        ticks += len(rows)
        out["m%d" % k] = m
        out["a%d" % k] = acts
        out["t%d" % k] = np.array(rows, dtype=np.float64).reshape(-1, 4)
        out["d%d" % k] = np.array(drows, dtype=np.uint8).reshape(-1, mg.N_DISC)
        out["e%d" % k] = np.array(erows, dtype=np.float64).reshape(-1, 6)
        out["s%d" % k] = np.array(srows, dtype=np.int32).reshape(-1, 3)
        print(k, mg.signature(sim), "ticks", len(rows), "episodes", int(sum(s[1] != 0 for s in srows)), flush=True)
    out["n"] = np.array([n], dtype=np.int32)
    print("kinds seen", sorted(kinds), "ticks", ticks)
    p = os.path.join(HERE, "fuzz.npz")
    np.savez_compressed(p, **out)
    print("fuzz.npz", os.path.getsize(p))


if __name__ == "__main__":
    main()
