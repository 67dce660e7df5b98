"""Random "entity soup" levels for fuzzing (pure numpy; used by tests and by tests/golden/make_golden_fuzz.py)."""
import numpy as np


def fuzz_level(base, rng, keep_away=0.0):
    """A real map's tiles + exit, with random zoo entities appended at empty tiles (map_loader.py:84-141 record layout:
    5 values per entity, 9 for door types 6 / 8 with the switch at +6, +7)."""
    m = np.asarray(base, dtype=np.float64).copy()
    tiles = m[184:1150].reshape(23, 42)
    empty = np.argwhere(tiles == 0)
    if len(empty) < 8:
        return m
    extra = []
    n_balls = 0
    for _ in range(int(rng.integers(6, 28))):
        t = int(rng.choice([1, 21, 2, 5, 6, 8, 10, 11, 14, 17, 20, 24, 25, 26, 28], p=None))
        ry, rx = empty[rng.integers(len(empty))]
        # pixel position inside the tile (tile (rx, ry) of the inner grid sits at world cell (rx + 1, ry + 1)), in map units of 6 px
        x = (rx + 1) * 4 + int(rng.integers(0, 5))
        y = (ry + 1) * 4 + int(rng.integers(0, 5))
        if keep_away and abs(x * 6 - m[1231] * 6) + abs(y * 6 - m[1232] * 6) < keep_away:
            continue    # long episodes: nothing lethal next to the spawn
        orient = int(rng.integers(0, 8))
        mode = int(rng.integers(0, 4))
        if t in (14, 20, 26, 5, 6, 8):
            orient = int(rng.choice([0, 2, 4, 6]))
        if t in (6, 8):
            sy, sx = empty[rng.integers(len(empty))]
            extra += [t, x, y, orient, mode, 0, (sx + 1) * 4 + 2, (sy + 1) * 4 + 2, 0]
        else:
            extra += [t, x, y, orient, mode]
        n_balls += t == 25
    out = np.concatenate([m, np.array(extra, dtype=np.float64)])
    out[1200] = n_balls
    return out
