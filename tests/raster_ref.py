"""numpy restatement of the reference's grayscale frame + player_frame crop (TEST INFRASTRUCTURE).

Follows nclone/nsim_renderer.py:71-134,176-274, nclone/entity_renderer.py:57-215,
nclone/shared_tile_renderer.py:26-152 and nclone/gym_environment/observation_processor.py:207-282.
The reference renders with cairo + pygame, neither of which is installed, so no reference frames exist to pin
against: "parity unpinned beyond geometry" (SURVEY.md 8(c)).  This restatement uses 8x8 supersampling for edge
coverage (the kernel uses 4x4), so it is not the same arithmetic as the product.
"""
import numpy as np

SS = 8


def luma(r, g, b):
    return (77 * r + 150 * g + 29 * b) >> 8


def tile_polys():
    T = 24.0
    H = 12.0
    P = {}
    P[6] = [(0, 0), (T, 0), (0, T)]; P[7] = [(0, 0), (T, 0), (T, T)]; P[8] = [(0, T), (T, 0), (T, T)]; P[9] = [(0, 0), (0, T), (T, T)]
    P[18] = [(0, 0), (T, 0), (0, H)]; P[19] = [(0, 0), (T, 0), (T, H)]; P[20] = [(0, T), (T, T), (T, H)]; P[21] = [(0, T), (T, T), (0, H)]
    P[22] = [(0, 0), (T, 0), (T, H), (0, T)]; P[23] = [(0, H), (0, 0), (T, 0), (T, T)]
    P[24] = [(0, H), (T, 0), (T, T), (0, T)]; P[25] = [(0, 0), (T, H), (T, T), (0, T)]
    P[26] = [(H, 0), (0, 0), (0, T)]; P[27] = [(H, 0), (T, 0), (T, T)]; P[28] = [(H, T), (T, 0), (T, T)]; P[29] = [(H, T), (0, 0), (0, T)]
    P[30] = [(H, T), (0, T), (0, 0), (T, 0)]; P[31] = [(H, T), (T, T), (T, 0), (0, 0)]
    P[32] = [(H, 0), (0, T), (T, T), (T, 0)]; P[33] = [(H, 0), (T, T), (0, T), (0, 0)]
    return P


POLYS = tile_polys()


def tile_cover(t):
    """coverage [24,24] of tile id t on its own cell"""
    if t == 0:
        return np.zeros((24, 24))
    if t == 1 or t > 33:
        return np.ones((24, 24))
    n = 24 * SS
    c = (np.arange(n) + 0.5) / SS
    u, v = np.meshgrid(c, c)  # u = x, v = y; arrays indexed [y, x]
    if t == 2:
        m = v < 12
    elif t == 3:
        m = u >= 12
    elif t == 4:
        m = v >= 12
    elif t == 5:
        m = u < 12
    elif 10 <= t < 14:
        cx = 24.0 if t in (11, 12) else 0.0
        cy = 24.0 if t in (12, 13) else 0.0
        m = (u - cx) ** 2 + (v - cy) ** 2 <= 576.0
    elif 14 <= t < 18:
        cx = 24.0 if t in (14, 17) else 0.0
        cy = 24.0 if t in (14, 15) else 0.0
        m = (u - cx) ** 2 + (v - cy) ** 2 >= 576.0
    else:
        pts = POLYS[t]
        pos = np.ones_like(u, dtype=bool)
        neg = np.ones_like(u, dtype=bool)
        for i in range(len(pts)):
            ax, ay = pts[i]
            bx, by = pts[(i + 1) % len(pts)]
            cr = (bx - ax) * (v - ay) - (by - ay) * (u - ax)
            pos &= cr >= 0
            neg &= cr <= 0
        m = pos | neg
    return m.reshape(24, SS, 24, SS).mean(axis=(1, 3))


def parse_level(map_data):
    m = np.asarray(map_data, dtype=np.float64)
    tiles = np.ones((44, 25), dtype=np.int64)
    for x in range(42):
        for y in range(23):
            tiles[x + 1, y + 1] = int(m[184 + x + 42 * y])
    ents = []
    doors = []
    index = 1230
    exit_count = int(m[1156])
    n = len(m)
    while index + 4 < n:
        t = int(m[index])
        if t in (1, 21):
            ents.append({"type": t, "x": m[index + 1] * 6, "y": m[index + 2] * 6})
        elif t == 2:
            ents.append({"type": 2, "x": m[index + 1] * 6, "y": m[index + 2] * 6})
        elif t == 3:
            ci = index + 5 * exit_count
            ents.append({"type": 3, "x": m[index + 1] * 6, "y": m[index + 2] * 6})
            ents.append({"type": 4, "x": m[ci + 1] * 6, "y": m[ci + 2] * 6})
        elif t == 6:
            ents.append({"type": 6, "x": m[index + 6] * 6, "y": m[index + 7] * 6})
            dx, dy, o = m[index + 1] * 6, m[index + 2] * 6, m[index + 3]
            if o in (0, 4):
                doors.append((dx, dy - 12, dx, dy + 12, len(ents) - 1))
            else:
                doors.append((dx - 12, dy, dx + 12, dy, len(ents) - 1))
        if t in (6, 8):
            if index + 9 < n and m[index + 7] != 0 and m[index + 8] == 0 and m[index + 9] == 0:
                index += 10
            else:
                index += 9
        else:
            index += 5
    return tiles, ents, doors


COLORS = {1: luma(0x9E, 0x21, 0x26), 21: luma(0xCE, 0x41, 0x46), 2: luma(0xDB, 0xE1, 0x49), 3: luma(0x83, 0x83, 0x84),
          4: luma(0x6D, 0x97, 0xC3), 6: 0}
RANK = {1: 0, 2: 1, 3: 2, 4: 3, 6: 4, 21: 5}


def render_window(map_data, ent_states, px, py, x0, y0, w, h):
    """Gray canvas pixels [y0:y0+h, x0:x0+w] (uint8).  ent_states: per entity in map order (mines: state;
    exit door: switch_hit; others: active)."""
    tiles, ents, doors = parse_level(map_data)
    n = SS
    xs = (np.arange(w * n) + 0.5) / n + x0
    ys = (np.arange(h * n) + 0.5) / n + y0
    X, Y = np.meshgrid(xs, ys)
    eg = np.zeros((h, w))
    ea = np.zeros((h, w))
    partial = np.zeros((h, w), dtype=bool)

    def apply(mask, gray):
        nonlocal eg, ea, partial
        cov = mask.reshape(h, n, w, n).mean(axis=(1, 3))
        partial |= (cov > 0) & (cov < 1)
        eg = eg * (1 - cov) + gray * cov
        ea = ea * (1 - cov) + cov

    for (x1, y1, x2, y2, ei) in doors:
        if ent_states[ei] == 0:
            continue
        vx, vy = x2 - x1, y2 - y1
        ln = np.hypot(vx, vy)
        vx, vy = vx / ln, vy / ln
        al = (X - x1) * vx + (Y - y1) * vy
        pe = (X - x1) * vy - (Y - y1) * vx
        apply((al >= 0) & (al <= ln) & (np.abs(pe) <= 1.0), luma(0x79, 0x79, 0x88))
    order = sorted(range(len(ents)), key=lambda i: RANK[ents[i]["type"]])
    for i in order:
        e = ents[i]
        st = ent_states[i]
        t = e["type"]
        if t in (1, 21):
            r = {0: 4.0, 1: 3.5, 2: 4.5}[int(st)]
            g = COLORS[t]
        elif t == 3:
            r = 12.0
            g = luma(0, 0, 128) if st else COLORS[3]
        else:
            if st == 0:
                continue
            r = {2: 6.0, 4: 6.0, 6: 5.0}[t]
            g = COLORS[t]
        apply((X - e["x"]) ** 2 + (Y - e["y"]) ** 2 <= r * r, g)
    apply((X - px) ** 2 + (Y - py) ** 2 <= 100.0, 0)
    v = np.full((h, w), 202, dtype=np.int64)
    a8 = np.floor(ea * 255 + 0.5).astype(np.int64)
    g8 = np.floor(eg + 0.5).astype(np.int64)
    v = np.where(a8 > 0, (g8 * a8 + v * (255 - a8)) >> 8, v)
    # tiles
    cov = np.zeros((h, w))
    for yy in range(h):
        for xx in range(w):
            cx, cy = (x0 + xx) // 24, (y0 + yy) // 24
            if 0 <= cx < 44 and 0 <= cy < 25:
                t = int(tiles[cx, cy])
                if t:
                    cov[yy, xx] = _tile_cov_cache(t)[(y0 + yy) - cy * 24, (x0 + xx) - cx * 24]
    ta = np.floor(cov * 255 + 0.5).astype(np.int64)
    tg = np.floor(122 * cov + 0.5).astype(np.int64)
    v = np.where(ta > 0, (tg * ta + v * (255 - ta)) >> 8, v)
    return v.astype(np.uint8), ((cov > 0) & (cov < 1)) | partial


_CACHE = {}


def _tile_cov_cache(t):
    if t not in _CACHE:
        _CACHE[t] = tile_cover(t)
    return _CACHE[t]


def player_frame(map_data, ent_states, px, py, centered=False):
    """84x84 crop with the reference's axis swap (observation_processor.py:219-258). Returns (frame, edge_mask)."""
    rc, cc = (py, px) if centered else (px, py)
    row0, row1 = max(0, int(rc - 42)), min(600, int(rc + 42))
    col0, col1 = max(0, int(cc - 42)), min(1056, int(cc + 42))
    h, w = max(0, row1 - row0), max(0, col1 - col0)
    out = np.zeros((84, 84), dtype=np.uint8)
    edge = np.zeros((84, 84), dtype=bool)
    if h == 0 or w == 0:
        return out, edge
    win, em = render_window(map_data, ent_states, px, py, col0, row0, w, h)
    top, left = (84 - h) // 2, (84 - w) // 2
    out[top:top + h, left:left + w] = win
    edge[top:top + h, left:left + w] = em
    return out, edge
