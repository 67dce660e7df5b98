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


# ---- draw-list driven renderer (all entity kinds) -----------------------------------------------------------------
# rows: oracle.Oracle.draw_list() -- [type, x, y, active, state | switch_hit, closed, nx, ny, door x1, y1, x2, y2, shape*1000+size]
TYPE_GRAY = {1: luma(0x9E, 0x21, 0x26), 2: luma(0xDB, 0xE1, 0x49), 3: luma(0x83, 0x83, 0x84), 4: luma(0x6D, 0x97, 0xC3),
             6: 0, 8: 0, 10: luma(0x86, 0x87, 0x93), 11: luma(0x66, 0x66, 0x66), 14: luma(0x6E, 0xC9, 0xE0),
             17: luma(0xE3, 0xE3, 0xE5), 20: luma(0x83, 0x83, 0x84), 21: luma(0xCE, 0x41, 0x46), 24: luma(0x66, 0x66, 0x66),
             25: luma(0x15, 0xA7, 0xBD), 26: luma(0x6E, 0xC9, 0xE0), 28: luma(0x6E, 0xC9, 0xE0)}


def _stroke_mask(X, Y, x1, y1, x2, y2, hw):
    vx, vy = x2 - x1, y2 - y1
    ln = np.hypot(vx, vy)
    vx, vy = vx / ln, vy / ln
    al = (X - x1) * vx + (Y - y1) * vy
    pe = (X - x1) * vy - (Y - y1) * vx
    return (al >= 0) & (al <= ln) & (np.abs(pe) <= hw)


def render_canvas(tiles, rows, px, py, x0, y0, w, h, ss=SS):
    """Gray canvas pixels [y0:y0+h, x0:x0+w] for an arbitrary entity list (entity_renderer.py:57-215). Returns (u8, edge)."""
    eg = np.zeros((h, w))
    ea = np.zeros((h, w))
    partial = np.zeros((h, w), dtype=bool)

    def apply(bx0, by0, bx1, by1, maskfn, gray):
        # restrict the work to the primitive's bounding box (canvas pixel units, clipped to the window)
        ix0, ix1 = max(x0, int(np.floor(bx0)) - 1), min(x0 + w, int(np.ceil(bx1)) + 1)
        iy0, iy1 = max(y0, int(np.floor(by0)) - 1), min(y0 + h, int(np.ceil(by1)) + 1)
        if ix1 <= ix0 or iy1 <= iy0:
            return
        xs = (np.arange((ix1 - ix0) * ss) + 0.5) / ss + ix0
        ys = (np.arange((iy1 - iy0) * ss) + 0.5) / ss + iy0
        X, Y = np.meshgrid(xs, ys)
        cov = maskfn(X, Y).reshape(iy1 - iy0, ss, ix1 - ix0, ss).mean(axis=(1, 3))
        sl = (slice(iy0 - y0, iy1 - y0), slice(ix0 - x0, ix1 - x0))
        partial[sl] |= (cov > 0) & (cov < 1)
        eg[sl] = eg[sl] * (1 - cov) + gray * cov
        ea[sl] = ea[sl] * (1 - cov) + cov

    # closed door strokes first (DOORWIDTH 2, tile colour), entity_dic order
    for r in rows:
        if int(r[0]) in (5, 6, 8) and r[5] != 0 and (r[8] != r[10] or r[9] != r[11]):
            x1, y1, x2, y2 = r[8:12]
            apply(min(x1, x2) - 1, min(y1, y2) - 1, max(x1, x2) + 1, max(y1, y2) + 1,
                  lambda X, Y: _stroke_mask(X, Y, x1, y1, x2, y2, 1.0), luma(0x79, 0x79, 0x88))
    # groups by type in order of first appearance among the ACTIVE entities, creation order inside a group
    groups = {}
    for r in rows:
        t = int(r[0])
        if r[3] == 0 or t == 5:
            continue
        groups.setdefault(t, []).append(r)
    for t, grp in groups.items():
        for r in grp:
            x, y = r[1], r[2]
            shape, size = int(r[12]) // 1000, r[12] % 1000
            g = TYPE_GRAY.get(t, 0)
            if t == 3 and r[4] != 0:
                g = luma(0, 0, 128)
            if shape == 1:   # oriented: stroke of PLATFORMWIDTH 3 across the normal
                nx, ny = r[6], r[7]
                ax, ay, bx, by = x + ny * size, y - nx * size, x - ny * size, y + nx * size
                apply(min(ax, bx) - 2, min(ay, by) - 2, max(ax, bx) + 2, max(ay, by) + 2,
                      lambda X, Y: _stroke_mask(X, Y, ax, ay, bx, by, 1.5), g)
            elif shape == 2:
                apply(x - size, y - size, x + size, y + size,
                      lambda X, Y: (np.abs(X - x) <= size) & (np.abs(Y - y) <= size), g)
            else:
                apply(x - size, y - size, x + size, y + size, lambda X, Y: (X - x) ** 2 + (Y - y) ** 2 <= size * size, g)
    apply(px - 10, py - 10, px + 10, py + 10, lambda X, Y: (X - px) ** 2 + (Y - py) ** 2 <= 100.0, 0)
    v = np.full((h, w), 202, dtype=np.int64)
    a8 = np.floor(ea * 255 + 0.5).astype(np.int64)
    g8 = np.floor(eg + 0.5).astype(np.int64)
    v = np.where(a8 > 0, (g8 * a8 + v * (255 - a8)) >> 8, v)
    # tile layer
    xs = np.arange(x0, x0 + w)
    ys = np.arange(y0, y0 + h)
    cov = np.zeros((h, w))
    cxs, cys = xs // 24, ys // 24
    for cy in np.unique(cys):
        for cx in np.unique(cxs):
            if not (0 <= cx < 44 and 0 <= cy < 25):
                continue
            t = int(tiles[cx, cy])
            if not t:
                continue
            yy = ys[cys == cy]
            xx = xs[cxs == cx]
            cov[np.ix_(yy - y0, xx - x0)] = _tile_cov_cache(t)[np.ix_(yy - cy * 24, xx - cx * 24)]
    ta = np.floor(cov * 255 + 0.5).astype(np.int64)
    tg = np.floor(122 * cov + 0.5).astype(np.int64)
    v = np.where(ta > 0, (tg * ta + v * (255 - ta)) >> 8, v)
    return v.astype(np.uint8), ((cov > 0) & (cov < 1)) | partial


def player_frame_rows(tiles, rows, px, py, centered=False):
    rc, cc = (py, px) if centered else (px, py)
    row0, row1 = max(0, int(rc - 42)), min(600, int(rc + 42))
    col0, col1 = max(0, int(cc - 42)), min(1056, int(cc + 42))
    h, w = max(0, row1 - row0), max(0, col1 - col0)
    out = np.zeros((84, 84), dtype=np.uint8)
    edge = np.zeros((84, 84), dtype=bool)
    if h == 0 or w == 0:
        return out, edge
    win, em = render_canvas(tiles, rows, px, py, col0, row0, w, h)
    top, left = (84 - h) // 2, (84 - w) // 2
    out[top:top + h, left:left + w] = win
    edge[top:top + h, left:left + w] = em
    return out, edge


def _area_tab(d, scale, ssize):
    """OpenCV computeResizeAreaTab for one destination index: list of (source index, weight), float32 arithmetic."""
    f32 = np.float32
    f1 = f32(d) * f32(scale)
    f2 = f1 + f32(scale)
    cell = min(f32(scale), f32(ssize) - f1)
    s1 = int(np.ceil(f1))
    s2 = min(int(np.floor(f2)), ssize - 1)
    s1 = min(s1, s2)
    out = []
    if s1 - f1 > 1e-3:
        out.append((s1 - 1, f32(s1 - f1) / cell))
    for s in range(s1, s2):
        out.append((s, f32(1.0) / cell))
    if f2 - s2 > 1e-3:
        out.append((s2, min(min(f32(f2 - s2), f32(1.0)), cell) / cell))
    return out


def global_view(tiles, rows, px, py, ss=4):
    """cv2.resize(frame[600, 1056], (100, 176), INTER_AREA) as the reference calls it (observation_processor.py:304-328),
    i.e. 176 rows x 100 columns, anisotropic.  Returns (u8 [176, 100], edge fraction [176, 100])."""
    canvas, edge = render_canvas(tiles, rows, px, py, 0, 0, 1056, 600, ss=ss)
    c = canvas.astype(np.float32)
    e = edge.astype(np.float32)
    sy, sx = np.float32(600.0 / 176), np.float32(1056.0 / 100)
    wy = [_area_tab(r, sy, 600) for r in range(176)]
    wx = [_area_tab(q, sx, 1056) for q in range(100)]
    # OpenCV's resizeArea order: per source row the horizontal sums (x ascending), then the vertical accumulation (y ascending)
    hs = np.zeros((600, 100), dtype=np.float32)
    he = np.zeros((600, 100), dtype=np.float32)
    for q in range(100):
        for (s_, w) in wx[q]:
            hs[:, q] += np.float32(w) * c[:, s_]
            he[:, q] += np.float32(w) * e[:, s_]
    out = np.zeros((176, 100), dtype=np.float32)
    ef = np.zeros((176, 100), dtype=np.float32)
    for r in range(176):
        for (s_, w) in wy[r]:
            out[r] += np.float32(w) * hs[s_]
            ef[r] += np.float32(w) * he[s_]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8), ef
