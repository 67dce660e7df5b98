// npp_render.hip -- player_frame (84 x 84 u8) rasterised on device.
//
// Geometry and compositing follow the reference's grayscale render path:
//   nclone/nsim_renderer.py:71-134   background fill, entity layer, then tile layer on top
//   nclone/nsim_renderer.py:176-274  per-layer luma (77R+150G+29B)>>8 and alpha blend (src*a + dst*(255-a))>>8
//   nclone/entity_renderer.py:57-215 closed door strokes (width 2, tile colour), active entities as filled discs of
//                                    their RADIUS in per-type colours, exit door dark blue once its switch is hit,
//                                    ninja = black disc of radius 10 (animation data is absent)
//   nclone/shared_tile_renderer.py:26-152 tile shapes per tile id
//   nclone/gym_environment/observation_processor.py:207-282 the crop around the player, INCLUDING its axis swap
//                                    (rows are indexed by player_x, columns by player_y) unless `centered` is set
// Anti-aliasing: the reference draws with cairo (not reproducible bit-for-bit); edge coverage
// here is 4x4 supersampling, so pixels away from primitive edges are exact and edge pixels differ by a few levels.
#include <hip/hip_runtime.h>

#include "npp_internal.hpp"
#include "npp_level.hpp"

namespace npp {
namespace {

constexpr int FW = 84, FH = 84;
constexpr int MAX_DRAW = 160;   // drawables kept per window (discs + strokes)

// convex polygons of tile ids 6..9 and 18..33 in units of 12 px (shared_tile_renderer.py:50-152), n vertices then xy
__constant__ unsigned char POLY[34][9] = {
    {0}, {0}, {0}, {0}, {0}, {0},
    {3, 0, 0, 2, 0, 0, 2, 0, 0}, {3, 0, 0, 2, 0, 2, 2, 0, 0}, {3, 0, 2, 2, 0, 2, 2, 0, 0}, {3, 0, 0, 0, 2, 2, 2, 0, 0},
    {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0},
    {3, 0, 0, 2, 0, 0, 1, 0, 0}, {3, 0, 0, 2, 0, 2, 1, 0, 0}, {3, 0, 2, 2, 2, 2, 1, 0, 0}, {3, 0, 2, 2, 2, 0, 1, 0, 0},
    {4, 0, 0, 2, 0, 2, 1, 0, 2}, {4, 0, 1, 0, 0, 2, 0, 2, 2}, {4, 0, 1, 2, 0, 2, 2, 0, 2}, {4, 0, 0, 2, 1, 2, 2, 0, 2},
    {3, 1, 0, 0, 0, 0, 2, 0, 0}, {3, 1, 0, 2, 0, 2, 2, 0, 0}, {3, 1, 2, 2, 0, 2, 2, 0, 0}, {3, 1, 2, 0, 0, 0, 2, 0, 0},
    {4, 1, 2, 0, 2, 0, 0, 2, 0}, {4, 1, 2, 2, 2, 2, 0, 0, 0}, {4, 1, 0, 0, 2, 2, 2, 2, 0}, {4, 1, 0, 2, 2, 0, 2, 0, 0},
};

// One drawable of the entity layer.  shape 0: filled disc (x, y, r); 1: stroke p1 = (x, y), p2 = (x2, y2) of half width
// r with butt caps; 2: filled axis-aligned square of semi side r around (x, y).
struct Draw {
    float x, y, r;
    float x2, y2;
    float gray;   // luma of the fill colour
    int shape;
};

__device__ inline int luma(int r, int g, int b) { return (77 * r + 150 * g + 29 * b) >> 8; }

__device__ inline bool tile_inside(int t, float u, float v) {   // (u, v) in [0, 24)^2 of the cell
    if (t == 0) return false;
    if (t == 1 || t > 33) return true;
    if (t < 6) {
        if (t == 2) return v < 12.f;
        if (t == 3) return u >= 12.f;
        if (t == 4) return v >= 12.f;
        return u < 12.f;
    }
    if (t >= 10 && t < 14) {
        float cx = (t == 11 || t == 12) ? 24.f : 0.f, cy = (t == 12 || t == 13) ? 24.f : 0.f;
        float dx = u - cx, dy = v - cy;
        return dx * dx + dy * dy <= 576.f;
    }
    if (t >= 14 && t < 18) {
        float cx = (t == 14 || t == 17) ? 24.f : 0.f, cy = (t == 14 || t == 15) ? 24.f : 0.f;
        float dx = u - cx, dy = v - cy;
        return dx * dx + dy * dy >= 576.f;
    }
    const unsigned char *p = POLY[t];
    int n = p[0];
    bool pos = true, neg = true;
    for (int i = 0; i < n; i++) {
        float ax = 12.f * p[1 + 2 * i], ay = 12.f * p[2 + 2 * i];
        int j = (i + 1 == n) ? 0 : i + 1;
        float bx = 12.f * p[1 + 2 * j], by = 12.f * p[2 + 2 * j];
        float cr = (bx - ax) * (v - ay) - (by - ay) * (u - ax);
        pos = pos && cr >= 0.f;
        neg = neg && cr <= 0.f;
    }
    return pos || neg;
}

// 4x4 supersampled coverage count (0..16) of drawable d on canvas pixel (x, y)
__device__ inline int draw_cover(const Draw &d, int x, int y) {
    int cnt = 0;
    if (d.shape == 0) {
        float ddx = (x + 0.5f) - d.x, ddy = (y + 0.5f) - d.y, lim = d.r + 1.f;
        if (ddx * ddx + ddy * ddy > lim * lim) return 0;
        for (int sy = 0; sy < 4; sy++)
            for (int sx = 0; sx < 4; sx++) {
                float qx = x + (sx + 0.5f) * 0.25f - d.x, qy = y + (sy + 0.5f) * 0.25f - d.y;
                cnt += (qx * qx + qy * qy <= d.r * d.r) ? 1 : 0;
            }
    } else if (d.shape == 1) {
        float vx = d.x2 - d.x, vy = d.y2 - d.y, len = sqrtf(vx * vx + vy * vy);
        if (len <= 0.f) return 0;
        vx /= len; vy /= len;
        for (int sy = 0; sy < 4; sy++)
            for (int sx = 0; sx < 4; sx++) {
                float qx = x + (sx + 0.5f) * 0.25f - d.x, qy = y + (sy + 0.5f) * 0.25f - d.y;
                float al = qx * vx + qy * vy, pe = qx * vy - qy * vx;
                cnt += (al >= 0.f && al <= len && fabsf(pe) <= d.r) ? 1 : 0;
            }
    } else {
        if (fabsf((x + 0.5f) - d.x) > d.r + 1.f || fabsf((y + 0.5f) - d.y) > d.r + 1.f) return 0;
        for (int sy = 0; sy < 4; sy++)
            for (int sx = 0; sx < 4; sx++) {
                float qx = x + (sx + 0.5f) * 0.25f - d.x, qy = y + (sy + 0.5f) * 0.25f - d.y;
                cnt += (fabsf(qx) <= d.r && fabsf(qy) <= d.r) ? 1 : 0;
            }
    }
    return cnt;
}

// Gray value of canvas pixel (x, y): background, entity layer (premultiplied gray + alpha), tile layer on top
// (nsim_renderer.py:71-134, 176-274)
__device__ inline int canvas_pixel(const Draw *draw, int nd, const uint8_t *tiles, int x, int y) {
    float eg = 0.f, ea = 0.f;
    for (int k = 0; k < nd; k++) {
        const int cnt = draw_cover(draw[k], x, y);
        if (cnt) {
            float cov = cnt * (1.f / 16.f);
            eg = eg * (1.f - cov) + draw[k].gray * cov;
            ea = ea * (1.f - cov) + cov;
        }
    }
    int v = 202;   // int((0.299*203 + 0.587*202 + 0.114*208)) background (nsim_renderer.py:75-79)
    int a8 = (int)(ea * 255.f + 0.5f);
    if (a8 > 0) v = ((int)(eg + 0.5f) * a8 + v * (255 - a8)) >> 8;
    int cx = x / 24, cy = y / 24;
    int t = (cx >= 0 && cx < 44 && cy >= 0 && cy < 25) ? tiles[cx * 25 + cy] : 0;
    if (t) {
        int cnt = 0;
        if (t == 1 || t > 33) cnt = 16;
        else
            for (int sy = 0; sy < 4; sy++)
                for (int sx = 0; sx < 4; sx++)
                    cnt += tile_inside(t, (x - cx * 24) + (sx + 0.5f) * 0.25f, (y - cy * 24) + (sy + 0.5f) * 0.25f) ? 1 : 0;
        if (cnt) {
            float cov = cnt * (1.f / 16.f);
            int ta = (int)(cov * 255.f + 0.5f);
            int tg = (int)(122.f * cov + 0.5f);   // luma(0x79, 0x79, 0x88) = 122, premultiplied
            v = (tg * ta + v * (255 - ta)) >> 8;
        }
    }
    return v;
}

// Drawables that can touch the window [wx0, wx1] x [wy0, wy1], in the reference's draw order (later ones overwrite:
// cairo operator SOURCE): closed door strokes (entity_renderer.py:63-97), then the active entities grouped by type
// (:100-150: discs of RADIUS, squares of SEMI_SIDE, oriented entities as a stroke of PLATFORMWIDTH across their normal),
// then the ninja.  Called by one thread.
__device__ int build_draw_list(const KernelArgs &a, const LevelHdr &H, int env, double px, double py, float wx0, float wy0,
                               float wx1, float wy1, Draw *out, int cap) {
    int n = 0;
    const uint32_t *meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_ent_meta);
    const double *doors = reinterpret_cast<const double *>(a.blob + H.off_doors);
    auto state_of = [&](int slot) { return (a.ent_bits[(size_t)(slot >> 4) * a.n + env] >> ((slot & 15) * 2)) & 3u; };
    for (uint32_t d = 0; d < H.n_door && n < cap; d++) {
        const int slot = (int)doors[5 * d + 4];
        const uint32_t st = state_of(slot), kind = meta[slot] & 15u;
        // segment.active == door closed: locked = switch not collected, regular = bit 1, trap = switch collected
        const bool closed = kind == EK_LOCKED ? (st & 1u) != 0 : (kind == EK_DOOR_REG ? (st & 2u) != 0 : (st & 1u) == 0);
        if (!closed) continue;
        float x1 = (float)doors[5 * d], y1 = (float)doors[5 * d + 1], x2 = (float)doors[5 * d + 2], y2 = (float)doors[5 * d + 3];
        if (fmaxf(x1, x2) < wx0 || fminf(x1, x2) > wx1 || fmaxf(y1, y2) < wy0 || fminf(y1, y2) > wy1) continue;
        out[n++] = {x1, y1, 1.f, x2, y2, (float)luma(0x79, 0x79, 0x88), 1};   // DOORWIDTH 2
    }
    const uint16_t *order = reinterpret_cast<const uint16_t *>(a.blob + H.off_raster);
    const double *ex = reinterpret_cast<const double *>(a.blob + H.off_ent_x);
    const double *ey = reinterpret_cast<const double *>(a.blob + H.off_ent_y);
    const uint32_t *mov_meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_mov_meta);
    const double *zb = (a.zoo && H.has_zoo) ? a.zoo + (size_t)env * a.zoo_words + ZOO_HEAD + (a.zoo_doors + 1) / 2 : nullptr;
    const uint32_t n_draw = H.n_ent + H.n_mov;
    for (uint32_t k = 0; k < n_draw && n < cap; k++) {
        const uint32_t ref = order[k];
        if (ref & 0x8000u) {   // a mover: position from the env's zoo block
            const int m = (int)(ref & 0x7fffu);
            if (!zb) continue;
            const float x = (float)zb[ZOO_MOV_WORDS * m], y = (float)zb[ZOO_MOV_WORDS * m + 1];
            if (x < wx0 || x > wx1 || y < wy0 || y > wy1) continue;
            const uint32_t mk = mov_meta[m] & 7u;
            if (mk == MK_DRONE) out[n++] = {x, y, 7.5f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};
            else if (mk == MK_MINI) out[n++] = {x, y, 4.f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};
            else if (mk == MK_BOUNCE) out[n++] = {x, y, 9.f, 0.f, 0.f, (float)luma(0xE3, 0xE3, 0xE5), 2};
            else if (mk == MK_THWUMP) out[n++] = {x, y, 9.f, 0.f, 0.f, (float)luma(0x83, 0x83, 0x84), 2};
            else if (mk == MK_BALL) out[n++] = {x, y, 5.f, 0.f, 0.f, (float)luma(0x15, 0xA7, 0xBD), 0};
            else out[n++] = {x, y, 8.f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};   // shove thwump: RADIUS wins over SEMI_SIDE
            continue;
        }
        const int slot = (int)ref;
        float x = (float)ex[slot], y = (float)ey[slot];
        if (a.zoo) {   // npp_set_entity_pos
            const double *hd = a.zoo + (size_t)env * a.zoo_words;
            const uint32_t ovr = reinterpret_cast<const uint32_t *>(hd + 3)[0];
            if (slot == H.obs_switch && (ovr & ZOO_OVR_SWITCH)) { x = (float)hd[4]; y = (float)hd[5]; }
            if (slot == H.obs_door && (ovr & ZOO_OVR_DOOR)) { x = (float)hd[6]; y = (float)hd[7]; }
        }
        if (x < wx0 || x > wx1 || y < wy0 || y > wy1) continue;
        const uint32_t mm = meta[slot], kind = mm & 15u, type = (mm >> 24) & 63u;
        const uint32_t st = state_of(slot);
        if (kind == EK_MINE) {             // always active; radius follows the state
            out[n++] = {x, y, st == 0 ? 4.0f : (st == 1 ? 3.5f : 4.5f), 0.f, 0.f,
                        type == 1 ? (float)luma(0x9E, 0x21, 0x26) : (float)luma(0xCE, 0x41, 0x46), 0};
        } else if (kind == EK_EXIT) {      // always active; dark blue (0, 0, 0.5) once the switch was hit
            out[n++] = {x, y, 12.f, 0.f, 0.f, st ? (float)luma(0, 0, 128) : (float)luma(0x83, 0x83, 0x84), 0};
        } else if (kind == EK_DOOR_REG) {
            continue;                      // entity_renderer.py:104-105
        } else {
            if ((st & 1u) == 0) continue;  // collected gold / switches are inactive and not drawn
            if (kind == EK_GOLD) out[n++] = {x, y, 6.f, 0.f, 0.f, (float)luma(0xDB, 0xE1, 0x49), 0};
            else if (kind == EK_SWITCH) out[n++] = {x, y, 6.f, 0.f, 0.f, (float)luma(0x6D, 0x97, 0xC3), 0};
            else if (kind == EK_BOOST) out[n++] = {x, y, 6.f, 0.f, 0.f, (float)luma(0x66, 0x66, 0x66), 0};
            else if (kind == EK_LAUNCH || kind == EK_ONEWAY) {
                // _draw_oriented_entity: angle = atan2(nx, ny) + pi/2; end points (x +- sin(angle) R, y +- cos(angle) R)
                const uint32_t o = (mm >> 8) & 7u;
                const float dg = 0.70710678f;
                const int sx = (o == 0 || o == 1 || o == 7) ? 1 : ((o >= 3 && o <= 5) ? -1 : 0);
                const int sy = (o >= 1 && o <= 3) ? 1 : ((o >= 5) ? -1 : 0);
                const float nx = (o & 1) ? sx * dg : (float)sx, ny = (o & 1) ? sy * dg : (float)sy;
                const float R = kind == EK_LAUNCH ? 6.f : 12.f;
                out[n++] = {x + ny * R, y - nx * R, 1.5f, x - ny * R, y + nx * R,
                            kind == EK_LAUNCH ? (float)luma(0x86, 0x87, 0x93) : (float)luma(0x66, 0x66, 0x66), 1};   // PLATFORMWIDTH 3
            } else out[n++] = {x, y, 5.f, 0.f, 0.f, 0.f, 0};   // locked / trap door switch: black
        }
    }
    if (n < cap) out[n++] = {(float)px, (float)py, 10.f, 0.f, 0.f, 0.f, 0};   // the ninja, drawn last
    return n;
}

__global__ __launch_bounds__(256) void npp_render_kernel(KernelArgs a, uint8_t *out, int centered) {
    __shared__ Draw s_draw[MAX_DRAW];
    __shared__ int s_n;
    __shared__ int s_win[6];   // row0, col0, h, w, top_pad, left_pad
    const int env = blockIdx.x;
    if (env >= a.n) return;
    const LevelHdr &H = a.hdr[a.env_level[env]];
    const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
    if (threadIdx.x == 0) {
        // observation_processor.py:219-231: rows sliced with player_x, columns with player_y (axis swap)
        double rc = centered ? py : px, cc = centered ? px : py;
        int row0 = (int)(rc - 42), row1 = (int)(rc + 42), col0 = (int)(cc - 42), col1 = (int)(cc + 42);
        row0 = row0 < 0 ? 0 : row0; row1 = row1 > 600 ? 600 : row1;
        col0 = col0 < 0 ? 0 : col0; col1 = col1 > 1056 ? 1056 : col1;
        int h = row1 - row0, w = col1 - col0;
        h = h < 0 ? 0 : h; w = w < 0 ? 0 : w;
        if (h > FH) h = FH;
        if (w > FW) w = FW;
        s_win[0] = row0; s_win[1] = col0; s_win[2] = h; s_win[3] = w;
        s_win[4] = (FH - h) / 2; s_win[5] = (FW - w) / 2;
        s_n = build_draw_list(a, H, env, px, py, col0 - 16.f, row0 - 16.f, col0 + w + 16.f, row0 + h + 16.f, s_draw, MAX_DRAW);
    }
    __syncthreads();
    const int row0 = s_win[0], col0 = s_win[1], h = s_win[2], w = s_win[3], top = s_win[4], left = s_win[5];
    const int nd = s_n;
    const uint8_t *tiles = a.blob + H.off_tiles;
    uint8_t *dst = out + (size_t)env * FW * FH;
    for (int p = threadIdx.x; p < FW * FH; p += blockDim.x) {
        int r = p / FW, c = p - r * FW;
        int fr = r - top, fc = c - left;
        uint8_t val = 0;   // cv2.copyMakeBorder(..., value=0)
        if (fr >= 0 && fr < h && fc >= 0 && fc < w) val = (uint8_t)canvas_pixel(s_draw, nd, tiles, col0 + fc, row0 + fr);
        dst[p] = val;
    }
}

// global_view (observation_processor.py:304-328): cv2.resize(frame, (RENDERED_VIEW_WIDTH = 100, RENDERED_VIEW_HEIGHT = 176),
// INTER_AREA) of the (600 rows x 1056 columns) gray frame.  The reference's constants are swapped (constants.py:18-19:
// "100 / 6", "1056 / 6"), so the frame is squashed anisotropically: 600 rows -> 176 (x 3.409) and 1056 columns -> 100
// (x 10.56).  Reproduced as is.  INTER_AREA with a non-integer factor is the area-weighted mean of the source pixels under
// each destination pixel (OpenCV resizeArea: float weights from computeResizeAreaTab, rounded to nearest on store).
// One workgroup per (env, output row); thread = output column.
constexpr int GV_ROWS = 176, GV_COLS = 100, GV_DRAW = 224;

// OpenCV computeResizeAreaTab for one destination index: source range [s1 - (head > 0), s2 + (tail > 0)) with weights
__device__ inline void area_tab(int d, float scale, int ssize, int &s1, int &s2, float &whead, float &wmid, float &wtail) {
    const float f1 = d * scale, f2 = f1 + scale;
    const float cell = fminf(scale, ssize - f1);
    s1 = (int)ceilf(f1);
    s2 = (int)floorf(f2);
    s2 = s2 < ssize - 1 ? s2 : ssize - 1;
    s1 = s1 < s2 ? s1 : s2;
    whead = (s1 - f1 > 1e-3f) ? (s1 - f1) / cell : 0.f;
    wmid = 1.f / cell;
    wtail = (f2 - s2 > 1e-3f) ? fminf(fminf(f2 - s2, 1.f), cell) / cell : 0.f;
}

__global__ __launch_bounds__(128) void npp_global_view_kernel(KernelArgs a, uint8_t *out) {
    __shared__ Draw s_draw[GV_DRAW];
    __shared__ int s_n;
    const int env = blockIdx.x / GV_ROWS, r = blockIdx.x - env * GV_ROWS;
    if (env >= a.n) return;
    const LevelHdr &H = a.hdr[a.env_level[env]];
    const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
    const float sy = 600.f / GV_ROWS, sx = 1056.f / GV_COLS;
    int y1, y2;
    float wyh, wym, wyt;
    area_tab(r, sy, 600, y1, y2, wyh, wym, wyt);
    const int ya = wyh > 0.f ? y1 - 1 : y1, yb = wyt > 0.f ? y2 + 1 : y2;   // canvas rows [ya, yb)
    if (threadIdx.x == 0)
        s_n = build_draw_list(a, H, env, px, py, -16.f, ya - 16.f, 1056.f + 16.f, yb + 16.f, s_draw, GV_DRAW);
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= GV_COLS) return;
    int x1, x2;
    float wxh, wxm, wxt;
    area_tab(c, sx, 1056, x1, x2, wxh, wxm, wxt);
    const int xa = wxh > 0.f ? x1 - 1 : x1, xb = wxt > 0.f ? x2 + 1 : x2;
    const uint8_t *tiles = a.blob + H.off_tiles;
    // drawables near this destination pixel's source rectangle
    int near[24], nn = 0;
    for (int k = 0; k < s_n; k++) {
        const Draw &d = s_draw[k];
        float ext = d.shape == 1 ? 0.5f * sqrtf((d.x2 - d.x) * (d.x2 - d.x) + (d.y2 - d.y) * (d.y2 - d.y)) + d.r : d.r;
        ext = d.shape == 2 ? ext * 1.4143f : ext;
        float cx = d.shape == 1 ? 0.5f * (d.x + d.x2) : d.x, cy = d.shape == 1 ? 0.5f * (d.y + d.y2) : d.y;
        if (cx + ext + 1.f >= xa && cx - ext - 1.f <= xb && cy + ext + 1.f >= ya && cy - ext - 1.f <= yb && nn < 24) near[nn++] = k;
    }
    float acc = 0.f;
    for (int y = ya; y < yb; y++) {
        const float wy = (y < y1) ? wyh : (y < y2 ? wym : wyt);
        for (int x = xa; x < xb; x++) {
            const float wx = (x < x1) ? wxh : (x < x2 ? wxm : wxt);
            float eg = 0.f, ea = 0.f;
            for (int q = 0; q < nn; q++) {
                const Draw &d = s_draw[near[q]];
                const int cnt = draw_cover(d, x, y);
                if (cnt) {
                    float cov = cnt * (1.f / 16.f);
                    eg = eg * (1.f - cov) + d.gray * cov;
                    ea = ea * (1.f - cov) + cov;
                }
            }
            int v = 202;
            int a8 = (int)(ea * 255.f + 0.5f);
            if (a8 > 0) v = ((int)(eg + 0.5f) * a8 + v * (255 - a8)) >> 8;
            const int cx = x / 24, cy = y / 24;
            const int t = tiles[cx * 25 + cy];
            if (t) {
                int cnt = 0;
                if (t == 1 || t > 33) cnt = 16;
                else
                    for (int qy = 0; qy < 4; qy++)
                        for (int qx = 0; qx < 4; qx++)
                            cnt += tile_inside(t, (x - cx * 24) + (qx + 0.5f) * 0.25f, (y - cy * 24) + (qy + 0.5f) * 0.25f) ? 1 : 0;
                if (cnt) {
                    float cov = cnt * (1.f / 16.f);
                    int ta = (int)(cov * 255.f + 0.5f);
                    int tg = (int)(122.f * cov + 0.5f);
                    v = (tg * ta + v * (255 - ta)) >> 8;
                }
            }
            acc += wy * wx * (float)v;
        }
    }
    out[((size_t)env * GV_ROWS + r) * GV_COLS + c] = (uint8_t)fminf(fmaxf(rintf(acc), 0.f), 255.f);   // cvRound + saturate
}

// The whole gray canvas of one env range (NPlayHeadless.render() in grayscale mode: nsim_renderer.py:71-134, array of shape
// (600, 1056, 1), nplay_headless.py:144-156).  One workgroup per (env, canvas row).
__global__ __launch_bounds__(256) void npp_full_frame_kernel(KernelArgs a, int env0, uint8_t *out) {
    __shared__ Draw s_draw[GV_DRAW];
    __shared__ int s_n;
    const int e = blockIdx.x / 600, y = blockIdx.x - e * 600;
    const int env = env0 + e;
    if (env >= a.n) return;
    const LevelHdr &H = a.hdr[a.env_level[env]];
    const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
    if (threadIdx.x == 0) s_n = build_draw_list(a, H, env, px, py, -16.f, y - 16.f, 1056.f + 16.f, y + 1 + 16.f, s_draw, GV_DRAW);
    __syncthreads();
    const uint8_t *tiles = a.blob + H.off_tiles;
    uint8_t *row = out + ((size_t)e * 600 + y) * 1056;
    for (int x = threadIdx.x; x < 1056; x += blockDim.x) row[x] = (uint8_t)canvas_pixel(s_draw, s_n, tiles, x, y);
}

// switch_states (gym_environment/npp_environment.py:1782-1847): up to MAX_LOCKED_DOORS = 5 locked doors x [switch x, switch y,
// door x, door y, collected].  _extract_locked_door_positions looks for `segment.p1`, which GridSegmentLinear does not have
// (entities.py: x1, y1, x2, y2), so the "door" position falls back to the entity's xpos / ypos -- the switch position
// (entity_door_base.py:94-97).  Reproduced as is.
__global__ __launch_bounds__(256) void npp_switch_states_kernel(KernelArgs a, float *out) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= a.n) return;
    const LevelHdr &H = a.hdr[a.env_level[env]];
    const double *ex = reinterpret_cast<const double *>(a.blob + H.off_ent_x);
    const double *ey = reinterpret_cast<const double *>(a.blob + H.off_ent_y);
    float *o = out + (size_t)env * 25;
    for (int k = 0; k < 5; k++) {
        const int slot = H.locked_slots[k];
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (slot >= 0) {
            const double sx = ex[slot] / 1056.0, sy = ey[slot] / 600.0;
            const float fx = (float)(sx < 0.0 ? 0.0 : (sx > 1.0 ? 1.0 : sx)), fy = (float)(sy < 0.0 ? 0.0 : (sy > 1.0 ? 1.0 : sy));
            const uint32_t st = (a.ent_bits[(size_t)(slot >> 4) * a.n + env] >> ((slot & 15) * 2)) & 3u;
            v[0] = fx; v[1] = fy; v[2] = fx; v[3] = fy; v[4] = (st & 1u) ? 0.f : 1.f;
        }
        for (int j = 0; j < 5; j++) o[5 * k + j] = v[j];
    }
}

}  // namespace

hipError_t launch_full_frame(const KernelArgs &a, int env0, int count, uint8_t *d_out, hipStream_t s) {
    hipLaunchKernelGGL(npp_full_frame_kernel, dim3(count * 600), dim3(256), 0, s, a, env0, d_out);
    return hipGetLastError();
}

hipError_t launch_switch_states(const KernelArgs &a, float *d_out, hipStream_t s) {
    hipLaunchKernelGGL(npp_switch_states_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a, d_out);
    return hipGetLastError();
}

hipError_t launch_render(const KernelArgs &a, uint8_t *d_out, int centered, hipStream_t s) {
    hipLaunchKernelGGL(npp_render_kernel, dim3(a.n), dim3(256), 0, s, a, d_out, centered);
    return hipGetLastError();
}

hipError_t launch_global_view(const KernelArgs &a, uint8_t *d_out, hipStream_t s) {
    hipLaunchKernelGGL(npp_global_view_kernel, dim3(a.n * GV_ROWS), dim3(128), 0, s, a, d_out);
    return hipGetLastError();
}

}  // namespace npp
