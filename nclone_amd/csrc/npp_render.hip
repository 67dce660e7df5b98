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
        // the pixel square [x, x + 1] x [y, y + 1] against the disc: entirely outside (its nearest point is) or entirely
        // inside (its farthest corner is) without sampling -- only the pixels on the rim run the 16 samples
        const float ax = x - d.x, ay = y - d.y, r2 = d.r * d.r;
        const float nx = fmaxf(fmaxf(ax, -(ax + 1.f)), 0.f), ny = fmaxf(fmaxf(ay, -(ay + 1.f)), 0.f);
        if (nx * nx + ny * ny > r2) return 0;
        const float fx = fmaxf(fabsf(ax), fabsf(ax + 1.f)), fy = fmaxf(fabsf(ay), fabsf(ay + 1.f));
        if (fx * fx + fy * fy <= r2) return 16;
        float qx2[4], qy2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float qx = x + (k + 0.5f) * 0.25f - d.x, qy = y + (k + 0.5f) * 0.25f - d.y;
            qx2[k] = qx * qx; qy2[k] = qy * qy;
        }
#pragma unroll
        for (int sy = 0; sy < 4; sy++)
#pragma unroll
            for (int sx = 0; sx < 4; sx++) cnt += (qx2[sx] + qy2[sy] <= r2) ? 1 : 0;
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
        int nx = 0, ny = 0;   // the square is a product of two intervals: samples inside = columns inside x rows inside
#pragma unroll
        for (int k = 0; k < 4; k++) {
            nx += (fabsf(x + (k + 0.5f) * 0.25f - d.x) <= d.r) ? 1 : 0;
            ny += (fabsf(y + (k + 0.5f) * 0.25f - d.y) <= d.r) ? 1 : 0;
        }
        cnt = nx * ny;
    }
    return cnt;
}

// Tile layer, precomputed: coverage count (0..16 of the 4x4 sample grid) of every canvas pixel of every loaded level,
// u8 [n_levels][600][1056] in HBM (633 600 B per level: 324 MB for the 512-level set -- HBM is the cheap resource on this
// part, 16 point-in-polygon tests per pixel per frame are not).  Built once by npp_load_levels.
__global__ __launch_bounds__(256) void npp_tile_canvas_kernel(const LevelHdr *hdr, const unsigned char *blob, uint8_t *canvas) {
    const int y = blockIdx.x, lvl = blockIdx.y;
    const uint8_t *tiles = blob + hdr[lvl].off_tiles;
    uint8_t *row = canvas + ((size_t)lvl * 600 + y) * 1056;
    const int cy = y / 24;
    for (int x = threadIdx.x; x < 1056; x += blockDim.x) {
        const int cx = x / 24;
        const int t = tiles[cx * 25 + cy];
        int cnt = 0;
        if (t == 1 || t > 33) cnt = 16;
        else if (t)
            for (int sy = 0; sy < 4; sy++)
                for (int sx = 0; sx < 4; sx++)
                    cnt += tile_inside(t, (x - cx * 24) + (sx + 0.5f) * 0.25f, (y - cy * 24) + (sy + 0.5f) * 0.25f) ? 1 : 0;
        row[x] = (uint8_t)cnt;
    }
}

// entity layer of one canvas pixel: premultiplied gray + alpha over the drawables in draw order (cairo operator SOURCE)
__device__ inline void entity_layer(const Draw *draw, int nd, int x, int y, float &eg, float &ea) {
    for (int k = 0; k < nd; k++) {
        const int cnt = draw_cover(draw[k], x, y);
        if (cnt) {
            float cov = cnt * (1.f / 16.f);
            eg = eg * (1.f - cov) + draw[k].gray * cov;
            ea = ea * (1.f - cov) + cov;
        }
    }
}

// Gray value of a canvas pixel from its entity layer and its tile coverage count: background, entity layer (premultiplied
// gray + alpha), tile layer on top (nsim_renderer.py:71-134, 176-274)
__device__ inline int composite(float eg, float ea, int tile_cnt) {
    int v = 202;   // int((0.299*203 + 0.587*202 + 0.114*208)) background (nsim_renderer.py:75-79)
    int a8 = (int)(ea * 255.f + 0.5f);
    if (a8 > 0) v = ((int)(eg + 0.5f) * a8 + v * (255 - a8)) >> 8;
    if (tile_cnt) {
        float cov = tile_cnt * (1.f / 16.f);
        int ta = (int)(cov * 255.f + 0.5f);
        int tg = (int)(122.f * cov + 0.5f);   // luma(0x79, 0x79, 0x88) = 122, premultiplied
        v = (tg * ta + v * (255 - ta)) >> 8;
    }
    return v;
}

__device__ inline int canvas_pixel(const Draw *draw, int nd, const uint8_t *tile_cnt_canvas, int x, int y) {
    float eg = 0.f, ea = 0.f;
    entity_layer(draw, nd, x, y, eg, ea);
    return composite(eg, ea, tile_cnt_canvas[(size_t)y * 1056 + x]);
}

// extent of a drawable for culling: centre and half extent (conservative, + 1 px of anti-aliasing)
__device__ inline void draw_extent(const Draw &d, float &cx, float &cy, float &ex, float &ey) {
    if (d.shape == 1) {
        cx = 0.5f * (d.x + d.x2); cy = 0.5f * (d.y + d.y2);
        ex = 0.5f * fabsf(d.x2 - d.x) + d.r + 1.f; ey = 0.5f * fabsf(d.y2 - d.y) + d.r + 1.f;
    } else {
        cx = d.x; cy = d.y; ex = d.r + 1.f; ey = d.r + 1.f;
    }
}

// ---- the draw list, built by the whole workgroup ---------------------------------------------------------------------
// Drawables that can touch the window [wx0, wx1] x [wy0, wy1], in the reference's draw order (later ones overwrite:
// cairo operator SOURCE): closed door strokes (entity_renderer.py:63-97), then the active entities grouped by type
// (:100-150: discs of RADIUS, squares of SEMI_SIDE, oriented entities as a stroke of PLATFORMWIDTH across their normal),
// then the ninja.  Thread t of the workgroup evaluates candidate t of every chunk of blockDim.x candidates (its level
// table reads and its entity-bit read run in parallel with everybody else's); an ordered ballot / prefix compaction keeps
// the draw order.
struct DrawCtx {
    const KernelArgs *a;
    const LevelHdr *H;
    int env;
    float wx0, wy0, wx1, wy1;
    bool init;   // the level as it stands right after a reset: entity states from the level's init words, no movers, nothing
                 // repositioned (global_view's per-level static tables)
};

__device__ inline uint32_t ent_state_of(const KernelArgs &a, int env, int slot) {
    return (a.ent_bits[(size_t)(slot >> 4) * a.n + env] >> ((slot & 15) * 2)) & 3u;
}
__device__ inline uint32_t ent_state_ctx(const DrawCtx &c, int slot) {
    if (c.init) return (reinterpret_cast<const uint32_t *>(c.a->blob + c.H->off_init_words)[slot >> 4] >> ((slot & 15) * 2)) & 3u;
    return ent_state_of(*c.a, c.env, slot);
}

__device__ inline bool door_drawable(const DrawCtx &c, uint32_t d, Draw &out) {
    const KernelArgs &a = *c.a;
    const LevelHdr &H = *c.H;
    const uint32_t *meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_ent_meta);
    const double *doors = reinterpret_cast<const double *>(a.blob + H.off_doors);
    const int slot = (int)doors[5 * d + 4];
    const uint32_t st = ent_state_ctx(c, slot), kind = meta[slot] & 15u;
    // segment.active == door closed: locked = switch not collected, regular = bit 1, trap = switch collected
    const bool closed = kind == EK_LOCKED ? (st & 1u) != 0 : (kind == EK_DOOR_REG ? (st & 2u) != 0 : (st & 1u) == 0);
    if (!closed) return false;
    float x1 = (float)doors[5 * d], y1 = (float)doors[5 * d + 1], x2 = (float)doors[5 * d + 2], y2 = (float)doors[5 * d + 3];
    if (fmaxf(x1, x2) < c.wx0 || fminf(x1, x2) > c.wx1 || fmaxf(y1, y2) < c.wy0 || fminf(y1, y2) > c.wy1) return false;
    out = {x1, y1, 1.f, x2, y2, (float)luma(0x79, 0x79, 0x88), 1};   // DOORWIDTH 2
    return true;
}

__device__ inline bool entity_drawable(const DrawCtx &c, uint32_t k, Draw &out) {
    const KernelArgs &a = *c.a;
    const LevelHdr &H = *c.H;
    const int env = c.env;
    const uint16_t *order = reinterpret_cast<const uint16_t *>(a.blob + H.off_raster);
    const uint32_t ref = order[k];
    if (ref & 0x8000u) {   // a mover: position from the env's zoo block
        if (c.init || !(a.zoo && H.has_zoo)) return false;
        const double *zb = a.zoo + (size_t)env * a.zoo_words + ZOO_HEAD + (a.zoo_doors + 1) / 2;
        const uint32_t *mov_meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_mov_meta);
        const int m = (int)(ref & 0x7fffu);
        const float x = (float)zb[ZOO_MOV_WORDS * m], y = (float)zb[ZOO_MOV_WORDS * m + 1];
        if (x < c.wx0 || x > c.wx1 || y < c.wy0 || y > c.wy1) return false;
        const uint32_t mk = mov_meta[m] & 7u;
        if (mk == MK_DRONE) out = {x, y, 7.5f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};
        else if (mk == MK_MINI) out = {x, y, 4.f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};
        else if (mk == MK_BOUNCE) out = {x, y, 9.f, 0.f, 0.f, (float)luma(0xE3, 0xE3, 0xE5), 2};
        else if (mk == MK_THWUMP) out = {x, y, 9.f, 0.f, 0.f, (float)luma(0x83, 0x83, 0x84), 2};
        else if (mk == MK_BALL) out = {x, y, 5.f, 0.f, 0.f, (float)luma(0x15, 0xA7, 0xBD), 0};
        else out = {x, y, 8.f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};   // shove thwump: RADIUS wins over SEMI_SIDE
        return true;
    }
    const uint32_t *meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_ent_meta);
    const double *ex = reinterpret_cast<const double *>(a.blob + H.off_ent_x);
    const double *ey = reinterpret_cast<const double *>(a.blob + H.off_ent_y);
    const int slot = (int)ref;
    float x = (float)ex[slot], y = (float)ey[slot];
    if (!c.init && a.zoo && (slot == H.obs_switch || slot == H.obs_door)) {   // npp_set_entity_pos
        const double *hd = a.zoo + (size_t)env * a.zoo_words;
        const uint32_t ovr = reinterpret_cast<const uint32_t *>(hd + 3)[0];
        if (slot == H.obs_switch && (ovr & ZOO_OVR_SWITCH)) { x = (float)hd[4]; y = (float)hd[5]; }
        if (slot == H.obs_door && (ovr & ZOO_OVR_DOOR)) { x = (float)hd[6]; y = (float)hd[7]; }
    }
    if (x < c.wx0 || x > c.wx1 || y < c.wy0 || y > c.wy1) return false;
    const uint32_t mm = meta[slot], kind = mm & 15u, type = (mm >> 24) & 63u;
    const uint32_t st = ent_state_ctx(c, slot);
    if (kind == EK_MINE) {             // always active; radius follows the state
        out = {x, y, st == 0 ? 4.0f : (st == 1 ? 3.5f : 4.5f), 0.f, 0.f,
               type == 1 ? (float)luma(0x9E, 0x21, 0x26) : (float)luma(0xCE, 0x41, 0x46), 0};
        return true;
    }
    if (kind == EK_EXIT) {             // always active; dark blue (0, 0, 0.5) once the switch was hit
        out = {x, y, 12.f, 0.f, 0.f, st ? (float)luma(0, 0, 128) : (float)luma(0x83, 0x83, 0x84), 0};
        return true;
    }
    if (kind == EK_DOOR_REG) return false;   // entity_renderer.py:104-105
    if ((st & 1u) == 0) return false;        // collected gold / switches are inactive and not drawn
    if (kind == EK_GOLD) out = {x, y, 6.f, 0.f, 0.f, (float)luma(0xDB, 0xE1, 0x49), 0};
    else if (kind == EK_SWITCH) out = {x, y, 6.f, 0.f, 0.f, (float)luma(0x6D, 0x97, 0xC3), 0};
    else if (kind == EK_BOOST) out = {x, y, 6.f, 0.f, 0.f, (float)luma(0x66, 0x66, 0x66), 0};
    else if (kind == EK_LAUNCH || kind == EK_ONEWAY) {
        // _draw_oriented_entity: angle = atan2(nx, ny) + pi/2; end points (x +- sin(angle) R, y +- cos(angle) R)
        const uint32_t o = (mm >> 8) & 7u;
        const float dg = 0.70710678f;
        const int sx = (o == 0 || o == 1 || o == 7) ? 1 : ((o >= 3 && o <= 5) ? -1 : 0);
        const int sy = (o >= 1 && o <= 3) ? 1 : ((o >= 5) ? -1 : 0);
        const float nx = (o & 1) ? sx * dg : (float)sx, ny = (o & 1) ? sy * dg : (float)sy;
        const float R = kind == EK_LAUNCH ? 6.f : 12.f;
        out = {x + ny * R, y - nx * R, 1.5f, x - ny * R, y + nx * R,
               kind == EK_LAUNCH ? (float)luma(0x86, 0x87, 0x93) : (float)luma(0x66, 0x66, 0x66), 1};   // PLATFORMWIDTH 3
    } else out = {x, y, 5.f, 0.f, 0.f, 0.f, 0};   // locked / trap door switch: black
    return true;
}

// ordered append of one candidate per thread (thread order = list order); s_wc: one int per wavefront
__device__ inline void append_ordered(bool keep, const Draw &d, Draw *out, int cap, int *s_n, int *s_wc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned long long bal = __ballot(keep);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wc[wave] = __popcll(bal);
    __syncthreads();
    int base = *s_n;
    for (int w = 0; w < wave; w++) base += s_wc[w];
    if (keep && base + prefix < cap) out[base + prefix] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = *s_n;
        for (int w = 0; w < nw; w++) t += s_wc[w];
        *s_n = t < cap ? t : cap;
    }
    __syncthreads();
}

// Called by every thread of the workgroup.  On return *s_n drawables sit in out[] (visible to all threads).
__device__ void build_draw_list(const KernelArgs &a, const LevelHdr &H, int env, double px, double py, float wx0, float wy0,
                                float wx1, float wy1, Draw *out, int cap, int *s_n, int *s_wc) {
    DrawCtx c{&a, &H, env, wx0, wy0, wx1, wy1, false};
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    for (uint32_t d0 = 0; d0 < H.n_door; d0 += blockDim.x) {
        Draw d = {};
        const uint32_t k = d0 + threadIdx.x;
        const bool keep = k < H.n_door && door_drawable(c, k, d);
        append_ordered(keep, d, out, cap, s_n, s_wc);
    }
    const uint32_t n_draw = H.n_ent + H.n_mov;
    for (uint32_t k0 = 0; k0 < n_draw; k0 += blockDim.x) {
        Draw d = {};
        const uint32_t k = k0 + threadIdx.x;
        const bool keep = k < n_draw && entity_drawable(c, k, d);
        append_ordered(keep, d, out, cap, s_n, s_wc);
    }
    if (threadIdx.x == 0 && *s_n < cap) out[(*s_n)++] = {(float)px, (float)py, 10.f, 0.f, 0.f, 0.f, 0};   // the ninja, drawn last
    __syncthreads();
}

// The crop window of the player frame (observation_processor.py:219-231: rows sliced with player_x, columns with
// player_y -- the reference's axis swap -- unless `centered`), clipped to the canvas, centred in the 84 x 84 output.
struct Window { int row0, col0, h, w, top, left; };
__device__ inline Window frame_window(double px, double py, int centered) {
    const double rc = centered ? py : px, cc = centered ? px : py;
    int row0 = (int)(rc - 42), row1 = (int)(rc + 42), col0 = (int)(cc - 42), col1 = (int)(cc + 42);
    row0 = row0 < 0 ? 0 : row0; row1 = row1 > 600 ? 600 : row1;
    col0 = col0 < 0 ? 0 : col0; col1 = col1 > 1056 ? 1056 : col1;
    int h = row1 - row0, w = col1 - col0;
    h = h < 0 ? 0 : h; w = w < 0 ? 0 : w;
    if (h > FH) h = FH;
    if (w > FW) w = FW;
    return {row0, col0, h, w, (FH - h) / 2, (FW - w) / 2};
}

// Gray value of a canvas pixel that no entity touches, for every tile id and every pixel of the 24 x 24 cell:
// g_tile_gray[t][v][u] = composite(0, 0, coverage count of tile t at (u, v)).  19 584 bytes, the same for every level; built
// once per device (launch_tile_tables) and staged into LDS by the player_frame kernel, which therefore never reads the
// per-level coverage canvas.  g_gray_cnt inverts it (gray -> coverage count) for the few pixels an entity touches.
constexpr int TILE_TAB = 34 * 576;
__device__ __attribute__((aligned(16))) unsigned char g_tile_gray[TILE_TAB];
__device__ unsigned char g_gray_cnt[256];

__global__ __launch_bounds__(256) void npp_tile_tables_kernel() {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 17) g_gray_cnt[composite(0.f, 0.f, i)] = (unsigned char)i;   // the 17 grays are distinct (202 down to 121)
    if (i >= TILE_TAB) return;
    const int t = i / 576, v = (i % 576) / 24, u = i % 24;
    int cnt = 0;
    if (t == 1) cnt = 16;
    else if (t)
        for (int sy = 0; sy < 4; sy++)
            for (int sx = 0; sx < 4; sx++) cnt += tile_inside(t, u + (sx + 0.5f) * 0.25f, v + (sy + 0.5f) * 0.25f) ? 1 : 0;
    g_tile_gray[i] = (unsigned char)composite(0.f, 0.f, cnt);
}

// ---- player_frame ------------------------------------------------------------------------------------------------------
// One workgroup (4 wavefronts) per env.  What the profiles say (profiles/r02_render_*; tools/render_stamps.py): the kernel
// is bound by VALU issue -- 24 M wavefront-instructions per launch, half of the lanes idle in them -- because every frame
// holds the ninja (and often an exit door or mines) in its middle rows, and a wavefront whose 64 spans touch such a row
// used to run the coverage sampler for the few lanes that needed it.  Hence two passes:
//   pass 1  every lane classifies its 4-pixel spans.  A span over empty / solid tiles (the level's 1100-byte tile table is
//           staged in LDS) with no drawable's pixel box on it -- almost all spans -- is finished with a handful of integer
//           instructions and stored as one dword (a wavefront writes 256 contiguous bytes).  Any other span goes to a queue.
//   pass 2  the queued spans are shaded one PIXEL per lane (4 consecutive lanes per span, recombined with DPP-free shuffles):
//           the sampler runs on full wavefronts of pixels that need it.
// Wavefront 0 alone builds the window's draw list from the level's compact draw-order records (one 16-byte load + a window
// test per candidate; ballot / prefix compaction, no workgroup barrier inside) and ORs every kept drawable into a
// per-output-row bit mask; the other wavefronts stage the tile table and wait at ONE barrier.
constexpr int PF_MASK_WORDS = (MAX_DRAW + 63) / 64;   // 3
constexpr int PF_SPANS = (FW / 4) * FH;               // 1764
struct FrameLds {
    Draw draw[MAX_DRAW];
    short bx0[MAX_DRAW], bx1[MAX_DRAW];              // canvas pixel columns a drawable can touch (inclusive)
    short ry0[MAX_DRAW], ry1[MAX_DRAW];              // output rows a drawable can touch (inclusive, clipped to the window)
    unsigned long long rowmask[FH][PF_MASK_WORDS];
    unsigned short queue[PF_SPANS];
    int nq;
    int nd;
#ifdef NPP_RENDER_STAMPS
    unsigned long long bst[6];
#endif
    __attribute__((aligned(4))) unsigned char tiles[1100];   // the level's tile ids, cell (cx, cy) at cx * 25 + cy
    __attribute__((aligned(16))) unsigned char gray[TILE_TAB];   // g_tile_gray
};

static_assert(sizeof(FrameLds) <= 32768, "five workgroups per CU need <= 32 KB of LDS each");

// the drawable of a draw-order record (entity_renderer.py:100-150), given the entity's position and 2-bit state
__device__ inline bool rec_drawable(uint32_t info, float x, float y, uint32_t st, Draw &out) {
    const uint32_t kind = info & 15u, type = (info >> 4) & 63u;
    if (info & 0x8000u) {   // movers (kind = MoverKind)
        if (kind == MK_DRONE) out = {x, y, 7.5f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};
        else if (kind == MK_MINI) out = {x, y, 4.f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};
        else if (kind == MK_BOUNCE) out = {x, y, 9.f, 0.f, 0.f, (float)luma(0xE3, 0xE3, 0xE5), 2};
        else if (kind == MK_THWUMP) out = {x, y, 9.f, 0.f, 0.f, (float)luma(0x83, 0x83, 0x84), 2};
        else if (kind == MK_BALL) out = {x, y, 5.f, 0.f, 0.f, (float)luma(0x15, 0xA7, 0xBD), 0};
        else out = {x, y, 8.f, 0.f, 0.f, (float)luma(0x6E, 0xC9, 0xE0), 0};   // shove thwump: RADIUS wins over SEMI_SIDE
        return true;
    }
    if (kind == EK_MINE) {             // always active; radius follows the state
        out = {x, y, st == 0 ? 4.0f : (st == 1 ? 3.5f : 4.5f), 0.f, 0.f,
               type == 1 ? (float)luma(0x9E, 0x21, 0x26) : (float)luma(0xCE, 0x41, 0x46), 0};
        return true;
    }
    if (kind == EK_EXIT) {             // always active; dark blue (0, 0, 0.5) once the switch was hit
        out = {x, y, 12.f, 0.f, 0.f, st ? (float)luma(0, 0, 128) : (float)luma(0x83, 0x83, 0x84), 0};
        return true;
    }
    if (kind == EK_DOOR_REG) return false;   // entity_renderer.py:104-105
    if ((st & 1u) == 0) return false;        // collected gold / switches are inactive and not drawn
    if (kind == EK_GOLD) out = {x, y, 6.f, 0.f, 0.f, (float)luma(0xDB, 0xE1, 0x49), 0};
    else if (kind == EK_SWITCH) out = {x, y, 6.f, 0.f, 0.f, (float)luma(0x6D, 0x97, 0xC3), 0};
    else if (kind == EK_BOOST) out = {x, y, 6.f, 0.f, 0.f, (float)luma(0x66, 0x66, 0x66), 0};
    else if (kind == EK_LAUNCH || kind == EK_ONEWAY) {
        // _draw_oriented_entity: angle = atan2(nx, ny) + pi/2; end points (x +- sin(angle) R, y +- cos(angle) R)
        const uint32_t o = (info >> 10) & 7u;
        const float dg = 0.70710678f;
        const int sx = (o == 0 || o == 1 || o == 7) ? 1 : ((o >= 3 && o <= 5) ? -1 : 0);
        const int sy = (o >= 1 && o <= 3) ? 1 : ((o >= 5) ? -1 : 0);
        const float nx = (o & 1) ? sx * dg : (float)sx, ny = (o & 1) ? sy * dg : (float)sy;
        const float R = kind == EK_LAUNCH ? 6.f : 12.f;
        out = {x + ny * R, y - nx * R, 1.5f, x - ny * R, y + nx * R,
               kind == EK_LAUNCH ? (float)luma(0x86, 0x87, 0x93) : (float)luma(0x66, 0x66, 0x66), 1};   // PLATFORMWIDTH 3
    } else out = {x, y, 5.f, 0.f, 0.f, 0.f, 0};   // locked / trap door switch: black
    return true;
}

// The window's draw list, built by ONE wavefront (lane = 0..63): lane l evaluates "virtual candidate" l (+ 64 per round) of the
// sequence [closed door strokes | draw-order records | ninja] -- the reference's draw order -- and a ballot / prefix compaction
// keeps that order; no workgroup barrier inside.  Measured alternative (round 2): all four wavefronts on 256 candidates per
// round -- slower, a level has ~50 drawables, so three wavefronts only add instructions.  `entw` = the env's entity state word
// `lane` (loaded by the caller before anything else, so that the record -> state hop of a candidate is a cross-lane read
// instead of a dependent global load).  Fills L.draw / L.bx0 / L.bx1 / L.ry0 / L.ry1 / L.nd / L.nq.
// The header fields the player_frame kernel needs, read together at its top: taken where they are used, each became a scalar
// load of its own with its own wait -- five to six dependent L2 round trips on the builder's path (2 300 clocks).
struct FrameHdr {
    uint32_t off_draw_recs, n_door, n_draw, has_zoo, off_tiles;
    int32_t obs_switch, obs_door;
};

__device__ inline void frame_build(FrameLds &L, const KernelArgs &a, const LevelHdr &H, const FrameHdr &fh, int env, double px, double py,
                                   const Window &wd, int lane, uint32_t entw) {
    const float wx0 = wd.col0 - 16.f, wy0 = wd.row0 - 16.f, wx1 = wd.col0 + wd.w + 16.f, wy1 = wd.row0 + wd.h + 16.f;
    const uint4 *recs = reinterpret_cast<const uint4 *>(a.blob + fh.off_draw_recs);
    const uint32_t n_door = fh.n_door, n_draw = fh.n_draw, n_virt = n_door + n_draw + 1u;
    const double *zhead = a.zoo ? a.zoo + (size_t)env * a.zoo_words : nullptr;
    const uint32_t ovr = zhead ? reinterpret_cast<const uint32_t *>(zhead + 3)[0] : 0u;   // npp_set_entity_pos
    const bool words_in_lanes = a.n_words_max <= 64;
    int nd = 0;   // wavefront-uniform
#ifdef NPP_RENDER_STAMPS
    if (lane == 0) L.bst[0] = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t v0 = 0; v0 < n_virt; v0 += 64) {
        const uint32_t v = v0 + lane;
        bool keep = false;
        Draw d = {};
        uint4 rc = make_uint4(0, 0, 0, 0);
        const bool is_rec = v >= n_door && v < n_door + n_draw;
        if (is_rec) rc = recs[v - n_door];
        const uint32_t info = rc.z;
        const int slot = (int)(info >> 16);
        uint32_t sw = 0;   // the state word of the candidate's slot: every lane takes part in the cross-lane read
        if (words_in_lanes) sw = (uint32_t)__shfl((int)entw, (slot >> 4) & 63, 64);
#ifdef NPP_RENDER_STAMPS
        if (lane == 0 && v0 == 0) L.bst[1] = __builtin_amdgcn_s_memtime() + (sw & 0) + (rc.x & 0);
#endif
        if (v < n_door) {   // closed door strokes first (entity_renderer.py:63-97)
            DrawCtx c{&a, &H, env, wx0, wy0, wx1, wy1, false};
            keep = door_drawable(c, v, d);
        } else if (is_rec) {
            float x = __uint_as_float(rc.x), y = __uint_as_float(rc.y);
            bool live = true;
            if (info & 0x8000u) {   // a mover: position from the env's zoo block
                live = zhead != nullptr && fh.has_zoo;
                if (live) {
                    const double *zb = zhead + ZOO_HEAD + (a.zoo_doors + 1) / 2 + ZOO_MOV_WORDS * slot;
                    x = (float)zb[0]; y = (float)zb[1];
                }
            } else if (ovr) {
                if (slot == fh.obs_switch && (ovr & ZOO_OVR_SWITCH)) { x = (float)zhead[4]; y = (float)zhead[5]; }
                if (slot == fh.obs_door && (ovr & ZOO_OVR_DOOR)) { x = (float)zhead[6]; y = (float)zhead[7]; }
            }
            if (live && !(x < wx0 || x > wx1 || y < wy0 || y > wy1)) {
                uint32_t st = 1u;
                if (!(info & 0x8000u)) st = words_in_lanes ? (sw >> ((slot & 15) * 2)) & 3u : ent_state_of(a, env, slot);
                keep = rec_drawable(info, x, y, st, d);
            }
        } else if (v == n_door + n_draw) {   // the ninja, drawn last
            d = {(float)px, (float)py, 10.f, 0.f, 0.f, 0.f, 0};
            keep = true;
        }
        const unsigned long long bal = __ballot(keep);
#ifdef NPP_RENDER_STAMPS
        if (lane == 0 && v0 == 0) L.bst[2] = __builtin_amdgcn_s_memtime();
#endif
        if (bal == 0) continue;
        int pos = nd + __popcll(bal & ((1ull << lane) - 1ull));
        // a list that overflows keeps its head and the ninja (the last candidate of all)
        if (v == n_door + n_draw && pos > MAX_DRAW - 1) pos = MAX_DRAW - 1;
        if (keep && (pos < MAX_DRAW - 1 || v == n_door + n_draw)) {
            L.draw[pos] = d;
            // output rows / canvas columns this drawable can touch (extent + 1 px of anti-aliasing)
            float cx, cy, ex, ey;
            draw_extent(d, cx, cy, ex, ey);
            L.bx0[pos] = (short)((int)floorf(cx - ex) - 1);
            L.bx1[pos] = (short)((int)ceilf(cx + ex));
            int y0 = (int)floorf(cy - ey) - 1, y1 = (int)ceilf(cy + ey);
            int r0 = y0 - wd.row0 + wd.top, r1 = y1 - wd.row0 + wd.top;
            r0 = r0 < wd.top ? wd.top : r0;
            r1 = r1 > wd.top + wd.h - 1 ? wd.top + wd.h - 1 : r1;
            L.ry0[pos] = (short)r0;
            L.ry1[pos] = (short)r1;
        }
        nd += __popcll(bal);
    }
    nd = nd < MAX_DRAW ? nd : MAX_DRAW;
    if (lane == 0) { L.nd = nd; L.nq = 0; }
#ifdef NPP_RENDER_STAMPS
    if (lane == 0) L.bst[3] = __builtin_amdgcn_s_memtime();
#endif
    // per-output-row bit masks of the drawables: every row written exactly once (nothing to clear, no atomics)
    if (nd <= 64) {   // the usual case, one mask word: lane l does rows l and l + 64 off the same LDS broadcast reads
        unsigned long long m0 = 0ull, m1 = 0ull;
        const int r1 = lane + 64;
        for (int p = 0; p < nd; p++) {
            const int y0 = L.ry0[p], y1 = L.ry1[p];
            const unsigned long long bit = 1ull << p;
            m0 |= (lane >= y0 && lane <= y1) ? bit : 0ull;
            m1 |= (r1 >= y0 && r1 <= y1) ? bit : 0ull;
        }
        L.rowmask[lane][0] = m0; L.rowmask[lane][1] = 0ull; L.rowmask[lane][2] = 0ull;
        if (r1 < FH) { L.rowmask[r1][0] = m1; L.rowmask[r1][1] = 0ull; L.rowmask[r1][2] = 0ull; }
    } else {
        for (int r = lane; r < FH; r += 64) {
            unsigned long long m[PF_MASK_WORDS];
#pragma unroll
            for (int q = 0; q < PF_MASK_WORDS; q++) m[q] = 0ull;
            for (int p = 0; p < nd; p++) {   // LDS broadcast reads (same address in every lane)
                const bool hit = r >= L.ry0[p] && r <= L.ry1[p];
#pragma unroll
                for (int q = 0; q < PF_MASK_WORDS; q++)
                    if ((p >> 6) == q && hit) m[q] |= 1ull << (p & 63);
            }
#pragma unroll
            for (int q = 0; q < PF_MASK_WORDS; q++) L.rowmask[r][q] = m[q];
        }
    }
}

// gray of canvas pixel (x, y) without entities: the cell's tile id (glitched ids 34+ are solid) and the pixel inside the cell
__device__ inline uint32_t tile_gray(const FrameLds &L, uint32_t x, uint32_t y) {
    const uint32_t cx = x / 24u, cy = y / 24u;
    uint32_t t = L.tiles[cx * 25u + cy];
    t = t > 33u ? 1u : t;
    return L.gray[t * 576u + (y - cy * 24u) * 24u + (x - cx * 24u)];
}

// pass 1: the span's four pixels packed little-endian when no entity can touch it and it has no padding pixel; otherwise
// `slow` is set and the value is meaningless
__device__ inline uint32_t span_plain(const FrameLds &L, const Window &wd, int r, int c0, bool &slow) {
    slow = false;
    const int fr = r - wd.top;
    if (fr < 0 || fr >= wd.h) return 0;   // rows outside the window are padding: cv2.copyMakeBorder(..., value=0)
    const int y = wd.row0 + fr, xs = wd.col0 + c0 - wd.left;
    slow = true;
    if (c0 - wd.left < 0 || c0 + 3 - wd.left >= wd.w) return 0;   // padding pixels in the span
    for (int q = 0; q < PF_MASK_WORDS; q++) {
        unsigned long long m = L.rowmask[r][q];
        while (m) {
            const int k = q * 64 + __builtin_ctzll(m);
            m &= m - 1;
            if (L.bx1[k] >= xs && L.bx0[k] <= xs + 3) return 0;
        }
    }
    slow = false;
    const uint32_t cy = (uint32_t)y / 24u, v = (uint32_t)y - cy * 24u;
    const uint32_t cx0 = (uint32_t)xs / 24u, cx1 = (uint32_t)(xs + 3) / 24u;
    uint32_t t0 = L.tiles[cx0 * 25u + cy], t1 = L.tiles[cx1 * 25u + cy];
    t0 = t0 > 33u ? 1u : t0; t1 = t1 > 33u ? 1u : t1;
    const uint32_t b0 = t0 * 576u + v * 24u, b1 = t1 * 576u + v * 24u;
    const uint32_t u0 = (uint32_t)xs - cx0 * 24u;   // column of the span's first pixel inside its cell (0..23)
    uint32_t out = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t u = u0 + j;   // a column >= 24 lies in the next cell
        out |= (uint32_t)L.gray[u < 24u ? b0 + u : b1 + u - 24u] << (8 * j);
    }
    return out;
}

// pass 1 for full-width windows (wd.w == 84, hence wd.left == 0): one lane per (output row, 24-pixel cell strip).
// Dword j of an output row holds canvas pixels col0 + 4 j .. + 3; inside its cell those start at byte u = s (mod 4) with
// s = col0 % 4 for EVERY dword of the frame, so a strip's six owned dwords are six funnel shifts over its seven consecutive
// aligned dwords S0..S5 (its 24-byte gray row) and N0 (the first dword of the next cell's row): row, cell, tile id and the
// drawable test are paid once per six dwords instead of once per dword.  Strip k of a row owns dwords 6 k - m .. 6 k - m + 5,
// m = (col0 % 24) / 4 (those whose first pixel lies in the cell); five strips cover the 21 dwords.
__device__ inline void pass1_strips(FrameLds &L, const Window &wd, uint32_t *dst) {
    constexpr int DW_PER_ROW = FW / 4;
    const uint32_t sh = (uint32_t)wd.col0 & 3u;
    const int cxb = wd.col0 / 24;
    const int m = (wd.col0 - cxb * 24) >> 2;
    for (int item = threadIdx.x; item < FH * 5; item += blockDim.x) {
        const int r = item / 5, k = item - r * 5;
        const int fr = r - wd.top;
        if (fr < 0 || fr >= wd.h) {   // a padding row (cv2.copyMakeBorder(..., value=0)): its five lanes clear it
            for (int j = k; j < DW_PER_ROW; j += 5) dst[r * DW_PER_ROW + j] = 0;
            continue;
        }
        const int j0 = 6 * k - m;
        if (j0 >= DW_PER_ROW) continue;
        const uint32_t y = (uint32_t)(wd.row0 + fr), cy = y / 24u, v = y - cy * 24u;
        const uint32_t cx = (uint32_t)(cxb + k);            // <= 43: the strip owns a pixel of the window
        const uint32_t cxn = cx < 43u ? cx + 1u : 43u;
        uint32_t t0 = L.tiles[cx * 25u + cy], t1 = L.tiles[cxn * 25u + cy];
        t0 = t0 > 33u ? 1u : t0; t1 = t1 > 33u ? 1u : t1;
        const uint2 *g0 = reinterpret_cast<const uint2 *>(L.gray + t0 * 576u + v * 24u);   // 8-byte aligned
        const uint2 a0 = g0[0], a1 = g0[1], a2 = g0[2];
        const uint32_t S[7] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, *reinterpret_cast<const uint32_t *>(L.gray + t1 * 576u + v * 24u)};
        // dwords some drawable of this row can touch go to pass 2
        uint32_t slow = 0;
#pragma unroll
        for (int q = 0; q < PF_MASK_WORDS; q++) {
            unsigned long long mm = L.rowmask[r][q];
            while (mm) {
                const int p = q * 64 + __builtin_ctzll(mm);
                mm &= mm - 1;
                const int xa = L.bx0[p] - wd.col0, xb = L.bx1[p] - wd.col0;   // window columns, inclusive
                if (xb < 0 || xa > FW - 1) continue;
                int ja = ((xa < 0 ? 0 : xa) >> 2) - j0, jb = ((xb > FW - 1 ? FW - 1 : xb) >> 2) - j0;
                if (jb < 0 || ja > 5) continue;
                ja = ja < 0 ? 0 : ja; jb = jb > 5 ? 5 : jb;
                slow |= ((2u << jb) - 1u) & ~((1u << ja) - 1u);
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int j = j0 + i;
            if (j < 0 || j >= DW_PER_ROW) continue;
            const int q = r * DW_PER_ROW + j;
            if ((slow >> i) & 1u) L.queue[atomicAdd(&L.nq, 1)] = (unsigned short)q;
            else dst[q] = __builtin_amdgcn_alignbyte(S[i + 1], S[i], sh);
        }
    }
}

// pass 2: one pixel (r, c) of the output frame in full generality
__device__ inline uint32_t pixel_full(const FrameLds &L, const Window &wd, const unsigned char *gray_cnt, int r, int c) {
    const int fr = r - wd.top, fc = c - wd.left;
    if (fr < 0 || fr >= wd.h || fc < 0 || fc >= wd.w) return 0;
    const int y = wd.row0 + fr, x = wd.col0 + fc;
    const uint32_t g = tile_gray(L, (uint32_t)x, (uint32_t)y);
    float eg = 0.f, ea = 0.f;
    for (int q = 0; q < PF_MASK_WORDS; q++) {
        unsigned long long m = L.rowmask[r][q];
        while (m) {
            const int k = q * 64 + __builtin_ctzll(m);
            m &= m - 1;
            if (L.bx1[k] < x || L.bx0[k] > x) continue;
            const Draw &d = L.draw[k];
            const int cv = draw_cover(d, x, y);
            if (cv) {
                float cov = cv * (1.f / 16.f);
                eg = eg * (1.f - cov) + d.gray * cov;
                ea = ea * (1.f - cov) + cov;
            }
        }
    }
    if (ea == 0.f) return g;
    return (uint32_t)composite(eg, ea, (int)gray_cnt[g]);
}

#ifndef NPP_RENDER_OCC
#define NPP_RENDER_OCC 5
#endif
__global__ __launch_bounds__(256, NPP_RENDER_OCC) void npp_render_kernel(KernelArgs a, uint8_t *out, int centered) {
    __shared__ FrameLds L;
    if ((int)blockIdx.x >= a.n) return;
    // heavy-first: the envs whose frame took longest last time (many drawables in the window) are dispatched first
    const int env = a.wg_order ? (int)a.wg_order[blockIdx.x] : (int)blockIdx.x;
    if (a.phase && a.phase_id >= 0 && a.phase[env] != (uint8_t)a.phase_id) return;   // stepped by the other part of a split launch
    const unsigned long long pf_t0 = a.wg_cost ? __builtin_amdgcn_s_memtime() : 0ull;
#ifdef NPP_RENDER_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    const int lvl = __builtin_amdgcn_readfirstlane(a.env_level[env]);
    const LevelHdr &H = a.hdr[lvl];
    const FrameHdr fh = {H.off_draw_recs, H.n_door, H.n_ent + H.n_mov, H.has_zoo, H.off_tiles, H.obs_switch, H.obs_door};
    const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
    const Window wd = frame_window(px, py, centered);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)env * FW * FH);
    constexpr int DW_PER_ROW = FW / 4;   // 21
#ifdef NPP_RENDER_STAMPS
    const unsigned long long t1 = wd.h ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memtime();
#endif
    if (wd.h == 0 || wd.w == 0) {        // the window lies outside the canvas (axis swap with player_x > 642): all padding
        for (int q = threadIdx.x; q < DW_PER_ROW * FH; q += blockDim.x) dst[q] = 0;
        if (a.wg_cost && threadIdx.x == 0) a.wg_cost[env] = (uint32_t)(__builtin_amdgcn_s_memtime() - pf_t0);
        return;
    }
    // One wavefront builds the draw list while the other three stage the level's tile ids (1100 bytes; 64 envs per level keep
    // the source in L2) and the tile gray table (19 584 bytes, the same for every workgroup of the launch).  The builder rotates
    // with the env: the wavefronts of a workgroup sit on different SIMDs, and a fixed builder would pile the list builds of all
    // resident workgroups of a CU onto one of them (56.1 -> 51.4 us per 8192-env launch).
    const int bw = env & 3, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == bw) {
        uint32_t entw = 0;
        if (lane < a.n_words_max && a.n_words_max <= 64) entw = a.ent_bits[(size_t)lane * a.n + env];
        frame_build(L, a, H, fh, env, px, py, wd, lane, entw);
    } else {
        const int t = (wave - (wave > bw ? 1 : 0)) * 64 + lane;   // 0..191
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.blob + fh.off_tiles);
        uint32_t *dstw = reinterpret_cast<uint32_t *>(L.tiles);
        const uint4 *gs = reinterpret_cast<const uint4 *>(g_tile_gray);
        uint4 *gd = reinterpret_cast<uint4 *>(L.gray);
        for (int i = t; i < TILE_TAB / 16; i += 192) gd[i] = gs[i];
        for (int i = t; i < 275; i += 192) dstw[i] = src[i];
    }
#ifdef NPP_RENDER_STAMPS
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef NPP_RENDER_STAMPS
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
    if (wd.w == FW) {
        pass1_strips(L, wd, dst);   // pass 1
    } else {   // a window narrower than the frame (player_y < 42 under the axis swap): generic 4-pixel spans with padding columns
        for (int q = threadIdx.x; q < PF_SPANS; q += blockDim.x) {
            const int r = q / DW_PER_ROW, c0 = (q - r * DW_PER_ROW) * 4;
            bool slow;
            const uint32_t v = span_plain(L, wd, r, c0, slow);
            if (slow) L.queue[atomicAdd(&L.nq, 1)] = (unsigned short)q;
            else dst[q] = v;
        }
    }
    __syncthreads();
#ifdef NPP_RENDER_STAMPS
    const unsigned long long t3b = __builtin_amdgcn_s_memtime();
#endif
    const int npix = L.nq * 4;
    for (int i = threadIdx.x; i < npix; i += blockDim.x) {   // pass 2: lanes 4 m .. 4 m + 3 shade the pixels of one span
        const int q = L.queue[i >> 2], j = i & 3;
        const int r = q / DW_PER_ROW, c = (q - r * DW_PER_ROW) * 4 + j;
        uint32_t v = pixel_full(L, wd, g_gray_cnt, r, c) << (8 * j);
        v |= __shfl_xor(v, 1, 64);
        v |= __shfl_xor(v, 2, 64);
        if (j == 0) dst[q] = v;
    }
    if (a.wg_cost && threadIdx.x == 0) a.wg_cost[env] = (uint32_t)(__builtin_amdgcn_s_memtime() - pf_t0);   // wavefront 0's view of the frame
#ifdef NPP_RENDER_STAMPS
    __syncthreads();
    if (threadIdx.x == bw * 64) {   // diagnostic build only: phase durations (shader clocks, builder wavefront) over the frame's first bytes
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        dst[0] = (uint32_t)(t1 - t0); dst[1] = (uint32_t)(t2 - t1); dst[2] = (uint32_t)(t3 - t2); dst[3] = (uint32_t)(t4 - t3);
        dst[4] = (uint32_t)L.nd; dst[5] = H.n_ent + H.n_mov; dst[6] = (uint32_t)L.nq; dst[7] = (uint32_t)(t3b - t3);
        dst[8] = (uint32_t)(L.bst[0] - t1); dst[9] = (uint32_t)(L.bst[1] - L.bst[0]); dst[10] = (uint32_t)(L.bst[2] - L.bst[1]);
        dst[11] = (uint32_t)(L.bst[3] - L.bst[2]); dst[12] = (uint32_t)(t2 - L.bst[3]);
    }
#endif
}

// global_view (observation_processor.py:304-328): cv2.resize(frame, (RENDERED_VIEW_WIDTH = 100, RENDERED_VIEW_HEIGHT = 176),
// INTER_AREA) of the (600 rows x 1056 columns) gray frame.  The reference's constants are swapped (constants.py:18-19:
// "100 / 6", "1056 / 6"), so the frame is squashed anisotropically: 600 rows -> 176 (x 3.409) and 1056 columns -> 100
// (x 10.56).  Reproduced as is.  INTER_AREA with a non-integer factor is the area-weighted mean of the source pixels under
// each destination pixel, accumulated the way OpenCV's resizeArea does it: per source row y the horizontal sums
// hs[y][c] = sum_x wx * pixel (x ascending, float), then acc[r][c] = sum_y wy * hs[y][c] (y ascending), rounded to nearest.
//
// 633 600 source pixels per env and step is what the round-2 profile of config 5 showed this kernel spending 20 ms on (98 %
// of the step).  Almost all of them are the same every step: the tile layer never changes and an entity changes its picture
// only when its state does.  So the level's picture right after a reset is reduced ONCE per level (hs: f32[600][100], the
// final view: u8[176][100]; built with the same pixel function), and per env and step only the destination cells whose
// source rectangle meets a "dirty box" are recomputed: the ninja, every mover, every entity whose drawable differs from the
// one it had after the reset (old and new extent), every repositioned entity.  Inside a dirty cell the source rows no dirty
// box touches take their hs from the level table; the others are re-summed pixel by pixel over the current draw list.
// Because the tables hold exactly the partial sums the full computation would produce, the result is bit-identical to
// reducing the whole frame (tests/test_gpu_render.py compares both).
constexpr int GV_ROWS = 176, GV_COLS = 100, GV_DRAW = 224, GV_CELLS = GV_ROWS * GV_COLS;

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

struct AreaTab { int s1, s2, a, b; float wh, wm, wt; };   // source indices [a, b), weights head / middle / tail
__device__ inline AreaTab gv_col_tab(int c) {
    AreaTab t;
    area_tab(c, 1056.f / GV_COLS, 1056, t.s1, t.s2, t.wh, t.wm, t.wt);
    t.a = t.wh > 0.f ? t.s1 - 1 : t.s1; t.b = t.wt > 0.f ? t.s2 + 1 : t.s2;
    return t;
}
__device__ inline AreaTab gv_row_tab(int r) {
    AreaTab t;
    area_tab(r, 600.f / GV_ROWS, 600, t.s1, t.s2, t.wh, t.wm, t.wt);
    t.a = t.wh > 0.f ? t.s1 - 1 : t.s1; t.b = t.wt > 0.f ? t.s2 + 1 : t.s2;
    return t;
}
// weight of source index s.  Written on scalars: selecting among the struct's fields by value made the compiler spill the
// struct and index it in scratch memory (a dependent scratch load per pixel).
__device__ inline float tab_w3(int s, int s1, int s2, float wh, float wm, float wt) {
    float w = wm;
    w = s < s1 ? wh : w;
    w = s >= s2 ? wt : w;
    return w;
}
__device__ inline float tab_w(const AreaTab &t, int s) { return tab_w3(s, t.s1, t.s2, t.wh, t.wm, t.wt); }

// per-level static tables, pass 1: the level's picture right after a reset, pixel by pixel (gv_p, u8[600][1056]) and its
// horizontal sums (gv_h).  Grid (600 source rows, levels).
__global__ __launch_bounds__(256) void npp_gv_static_h_kernel(KernelArgs a, uint8_t *gv_p, float *gv_h) {
    __shared__ Draw s_draw[GV_DRAW];
    __shared__ int s_n;
    __shared__ int s_wc[4];
    __shared__ unsigned char s_row[1056];
    const int y = blockIdx.x, lvl = blockIdx.y;
    const LevelHdr &H = a.hdr[lvl];
    DrawCtx c{&a, &H, 0, -16.f, y - 16.f, 1056.f + 16.f, y + 1 + 16.f, true};
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t d0 = 0; d0 < H.n_door; d0 += blockDim.x) {
        Draw d = {};
        const uint32_t k = d0 + threadIdx.x;
        const bool keep = k < H.n_door && door_drawable(c, k, d);
        append_ordered(keep, d, s_draw, GV_DRAW, &s_n, s_wc);
    }
    const uint32_t n_draw = H.n_ent + H.n_mov;
    for (uint32_t k0 = 0; k0 < n_draw; k0 += blockDim.x) {
        Draw d = {};
        const uint32_t k = k0 + threadIdx.x;
        const bool keep = k < n_draw && entity_drawable(c, k, d);
        append_ordered(keep, d, s_draw, GV_DRAW, &s_n, s_wc);
    }
    const uint8_t *canvas = a.tile_canvas + (size_t)lvl * 600 * 1056;
    uint8_t *prow = gv_p + ((size_t)lvl * 600 + y) * 1056;
    for (int x = threadIdx.x; x < 1056; x += blockDim.x) {
        const unsigned char v = (unsigned char)canvas_pixel(s_draw, s_n, canvas, x, y);
        s_row[x] = v;
        prow[x] = v;
    }
    __syncthreads();
    const int col = threadIdx.x;
    if (col >= GV_COLS) return;
    const AreaTab tx = gv_col_tab(col);
    float hs = 0.f;
    for (int x = tx.a; x < tx.b; x++) hs += tab_w3(x, tx.s1, tx.s2, tx.wh, tx.wm, tx.wt) * (float)s_row[x];
    gv_h[((size_t)lvl * 600 + y) * GV_COLS + col] = hs;
}

// pass 2: the level's view from its hs table.  Grid (176 destination rows, levels).
__global__ __launch_bounds__(128) void npp_gv_static_v_kernel(const float *gv_h, uint8_t *gv_v) {
    const int r = blockIdx.x, lvl = blockIdx.y, col = threadIdx.x;
    if (col >= GV_COLS) return;
    const AreaTab ty = gv_row_tab(r);
    float acc = 0.f;
    for (int y = ty.a; y < ty.b; y++) acc += tab_w(ty, y) * gv_h[((size_t)lvl * 600 + y) * GV_COLS + col];
    gv_v[((size_t)lvl * GV_ROWS + r) * GV_COLS + col] = (uint8_t)fminf(fmaxf(rintf(acc), 0.f), 255.f);   // cvRound + saturate
}

#ifndef NPP_GV_Q   // (the three LDS budgets are build options for occupancy A/B runs; running out of any of them stays exact)
#define NPP_GV_Q 512
#endif
#ifndef NPP_GV_PATCH
#define NPP_GV_PATCH 3072
#endif
#ifndef NPP_GV_BOX_MAX
#define NPP_GV_BOX_MAX 192
#endif
constexpr int GV_Q = NPP_GV_Q, GV_WORDS = (GV_CELLS + 31) / 32;
constexpr int GV_PBOX = 32, GV_PATCH = NPP_GV_PATCH;   // dirty boxes whose pixels are composed up front into LDS patches, patch bytes
// The cell pass as a kernel of its own (round 2): npp_global_view_kernel's launch lasted as long as its HEAVIEST env (cell pass: 36 k
// clocks at the median, 250-380 k at the maximum), so an env with at most GV_XQ dirty cells now EXPORTS what the cell pass needs
// (boxes, patch rectangles, patches, the cell queue; the draw list only if some row may have to compose in place) to a per-env
// scratch block in HBM, and npp_gv_cells_kernel runs GV_XWAVES wavefronts per env, wavefront j taking the queue slices j, j + GV_XWAVES,
// ... of GV_ITEM_CELLS cells: the cells of a heavy env are recomputed by several wavefronts at once and the wavefronts of light envs
// retire after one header read.  (First cut: a global work list with one atomic per item -- 18 k same-address atomics took longer,
// 280 us, than the cell pass they distributed.)  Same arithmetic, same order per cell.
// MEASURED AND NOT SHIPPED (-DNPP_GV_SPLIT builds it): 8192 envs on the door levels, fused 223 us; split with one wavefront per env
// 115 + 122 us, with four 115 + 196 us (32 768 workgroups: ~60 cycles of dispatch each per XCD).  Both halves stay bound by their
// slowest wavefront -- list + patches up to 149 k clocks, cells up to 202 k (an env with 10 dirty boxes and 110 dirty cells on
// `switch-simple`) -- and the heavy envs sit late in the grid, so two kernels pay two tails.
#ifndef NPP_GV_XWAVES
#define NPP_GV_XWAVES 4
#endif
constexpr int GV_XQ = 1024, GV_ITEM_CELLS = 32, GV_BOX_MAX = NPP_GV_BOX_MAX, GV_XWAVES = NPP_GV_XWAVES;
constexpr int XS_HDR = 0, XS_BOX = 16, XS_PRECT = XS_BOX + GV_BOX_MAX * 8, XS_POFF = XS_PRECT + GV_PBOX * 8, XS_QUEUE = XS_POFF + GV_PBOX * 2,
              XS_PATCH = XS_QUEUE + GV_XQ * 2, XS_DRAW = XS_PATCH + GV_PATCH, XS_CBOX = XS_DRAW + 224 * 28, XS_END = XS_CBOX + 224 * 4;
static_assert(XS_DRAW % 16 == 0 && XS_PATCH % 4 == 0 && XS_END <= GV_XSTRIDE, "scratch block layout");
// LDS of one env (dynamic: the draw list and the box list are sized for the level SET -- the largest number of draw records of
// a loaded level -- so that ordinary sets leave room for 12 wavefronts per CU instead of 8)
struct GvLds {
    Draw *draw;                   // [draw_cap]
    uchar4 *cbox;                 // [draw_cap] destination cells a drawable can touch: row0, row1, col0, col1 (inclusive, conservative)
    short4 *box;                  // [box_cap] dirty boxes: the canvas pixels x0..x1, y0..y1 (inclusive) a changed drawable can cover
    uint32_t *dirty;              // [GV_WORDS] destination cells to recompute
    unsigned short *queue;        // [GV_Q]
    short4 *prect;                // [GV_PBOX] pixel rectangle of a patched box: x0, y0, width, height; width 0 = no patch
    unsigned short *poff;         // [GV_PBOX] its first byte in patch[]
    unsigned char *patch;         // [GV_PATCH]
    int *ctr;                     // [0] boxes pushed, [1] cells queued, [2] cells left for another round
    int draw_cap, box_cap;
};
__host__ __device__ inline int gv_box_cap(int draw_cap) { return 2 * draw_cap + 2 < GV_BOX_MAX ? 2 * draw_cap + 2 : GV_BOX_MAX; }
__host__ __device__ inline size_t gv_lds_bytes(int draw_cap) {
    const int dc4 = (draw_cap + 3) & ~3;
    return ((size_t)dc4 * sizeof(Draw) + (size_t)dc4 * 4 + (size_t)gv_box_cap(draw_cap) * 8 + GV_WORDS * 4 + GV_Q * 2 + GV_PBOX * 8 + GV_PBOX * 2 +
           GV_PATCH + 16 + 15) & ~(size_t)15;
}
__device__ inline GvLds gv_lds_layout(unsigned char *base, int draw_cap) {
    GvLds L;
    const int dc4 = (draw_cap + 3) & ~3;   // sizeof(Draw) = 28: four of them keep 16-byte alignment
    L.draw_cap = draw_cap; L.box_cap = gv_box_cap(draw_cap);
    L.draw = reinterpret_cast<Draw *>(base); base += (size_t)dc4 * sizeof(Draw);
    L.cbox = reinterpret_cast<uchar4 *>(base); base += (size_t)dc4 * 4;
    L.box = reinterpret_cast<short4 *>(base); base += (size_t)L.box_cap * 8;
    L.dirty = reinterpret_cast<uint32_t *>(base); base += GV_WORDS * 4;
    L.prect = reinterpret_cast<short4 *>(base); base += GV_PBOX * 8;
    L.ctr = reinterpret_cast<int *>(base); base += 16;
    L.queue = reinterpret_cast<unsigned short *>(base); base += GV_Q * 2;
    L.poff = reinterpret_cast<unsigned short *>(base); base += GV_PBOX * 2;
    L.patch = base;
    return L;
}

__device__ inline bool draw_differs(const Draw &p, const Draw &q) {
    return p.x != q.x || p.y != q.y || p.r != q.r || p.x2 != q.x2 || p.y2 != q.y2 || p.gray != q.gray || p.shape != q.shape;
}
// The canvas pixels a drawable can cover: pixel p spans [p, p + 1), every coverage sample lies inside its pixel, and a sample is
// covered only within r of the disc centre / square centre / stroke axis -- so pixels floor(c - e) .. floor(c + e) with e the
// half extent WITHOUT the anti-aliasing margin draw_extent() adds for culling.
__device__ inline void gv_push_box(const GvLds &L, const Draw &d) {
    float cx, cy, ex, ey;
    draw_extent(d, cx, cy, ex, ey);
    ex -= 1.f; ey -= 1.f;
    int x0 = (int)floorf(cx - ex), x1 = (int)floorf(cx + ex), y0 = (int)floorf(cy - ey), y1 = (int)floorf(cy + ey);
    x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; x1 = x1 > 1055 ? 1055 : x1; y1 = y1 > 599 ? 599 : y1;
    if (x1 < x0 || y1 < y0) return;   // nothing of it on the canvas
    const int i = atomicAdd(&L.ctr[0], 1);
    if (i < L.box_cap) L.box[i] = make_short4((short)x0, (short)y0, (short)x1, (short)y1);
}
// destination rows / columns whose source range can meet [lo, hi] (one cell of slack on each side)
__device__ inline void gv_cell_range(float lo, float hi, float scale, int n, int &i0, int &i1) {
    i0 = (int)floorf(lo / scale) - 1; i1 = (int)floorf(hi / scale) + 1;
    i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);
    i1 = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
}
__device__ inline uchar4 gv_cell_box(const Draw &d) {
    float cx, cy, ex, ey;
    draw_extent(d, cx, cy, ex, ey);
    int r0, r1, c0, c1;
    gv_cell_range(cy - ey, cy + ey, 600.f / GV_ROWS, GV_ROWS, r0, r1);
    gv_cell_range(cx - ex, cx + ex, 1056.f / GV_COLS, GV_COLS, c0, c1);
    return make_uchar4((unsigned char)r0, (unsigned char)r1, (unsigned char)c0, (unsigned char)c1);
}
__device__ inline void wave_sync() {   // LDS traffic of one wavefront: order it, no s_barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One destination cell of the view, recomputed by the 8 lanes of a group (one source row per lane, then an ordered accumulation
// through the group's lanes): shared by the in-place cell pass of npp_global_view_kernel and by npp_gv_cells_kernel.  Every lane of
// the calling wavefront whose group has a cell calls it (the shuffles stay inside a group).
__device__ inline void gv_cell(const GvLds &L, int nd, int nb, const uint8_t *canvas, const uint8_t *pstat, const float *hrow,
                               uint8_t *out_env, int cell, int lane) {
    const int sub = lane & 7;
    const int r = cell / GV_COLS, c = cell - r * GV_COLS;
    const AreaTab tx = gv_col_tab(c), ty = gv_row_tab(r);
    const int tx_s1 = tx.s1, tx_s2 = tx.s2, ty_s1 = ty.s1, ty_s2 = ty.s2;
    const float tx_wh = tx.wh, tx_wm = tx.wm, tx_wt = tx.wt, ty_wh = ty.wh, ty_wm = ty.wm, ty_wt = ty.wt;
    const int y = ty.a + sub;   // this lane's source row (a destination row spans at most 6)
    // issued before the mask work so that their latency is hidden: the row's static hs and its static picture slice
    // (at most 12 pixels -> four aligned dwords; the table is padded by 16 bytes).  (Issuing them a whole cell ahead from the caller's
    // loop was measured: 133 -> 137 us, the extra live registers cost more than the latency.)
    const int xb0 = tx.a & ~3;
    const int yl = y < 599 ? y : 599;
    const float hs_static = hrow[(size_t)yl * GV_COLS + c];
    const uint32_t *pp = reinterpret_cast<const uint32_t *>(pstat + (size_t)yl * 1056 + xb0);   // dword aligned
    const uint4 pw = make_uint4(pp[0], pp[1], pp[2], pp[3]);
    // the drawables that can touch this cell -- needed only by row slices that compose pixels in place (dirty boxes beyond
    // the patch budget: a crowd of movers), so the scan runs only when some lane of the wavefront asks for it (it was
    // 20 % of the cell pass when it ran for every cell): the group's lanes share the scan of the cell boxes (four per
    // 16-byte LDS read), then OR their masks together
    unsigned long long nm[4] = {0ull, 0ull, 0ull, 0ull};
    auto scan_cell_masks = [&]() {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned long long m = 0ull;
            if (w * 64 >= nd) { nm[w] = 0ull; continue; }   // wavefront-uniform
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int chunk = w * 16 + h * 8 + sub;   // drawables 4 chunk .. 4 chunk + 3
                if (chunk * 4 < nd) {
                    const uint4 cb = reinterpret_cast<const uint4 *>(L.cbox)[chunk];
                    const uint32_t q[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int br0 = q[j] & 255u, br1 = (q[j] >> 8) & 255u, bc0 = (q[j] >> 16) & 255u, bc1 = q[j] >> 24;
                        if (r >= br0 && r <= br1 && c >= bc0 && c <= bc1) m |= 1ull << ((h * 8 + sub) * 4 + j);
                    }
                }
            }
            uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
#pragma unroll
            for (int sft = 1; sft < 8; sft <<= 1) {
                lo |= (uint32_t)__shfl_xor((int)lo, sft, 64);
                hi |= (uint32_t)__shfl_xor((int)hi, sft, 64);
            }
            nm[w] = ((unsigned long long)hi << 32) | lo;
        }
    };
    float hs = 0.f;
    // the part of this row slice inside dirty boxes: only those pixels are composed again, the others come from the
    // level's static picture (a pixel outside every dirty box is covered by the same drawables as after the reset, in
    // the same order)
    const bool row_live = y < ty.b;
    int hx0 = 4096, hx1 = -4096;
    // patched boxes on this row slice: up to four are remembered (two were: a row slice under three boxes -- ninja, switch and door
    // on a door level -- then composed every pixel in place, and one such env set the duration of the launch)
    int pn = 0, pi0 = 0, pi1 = 0, pi2 = 0, pi3 = 0;
    bool inline_compose = false;
    if (row_live) {
        // four boxes per trip: their LDS reads (box, patch rectangle) are issued together instead of one dependent pair per box
        for (int b0 = 0; b0 < nb; b0 += 4) {
            short4 bxs[4], prs[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int b = b0 + u < nb ? b0 + u : nb - 1;
                bxs[u] = L.box[b];
                prs[u] = L.prect[b < GV_PBOX ? b : 0];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int b = b0 + u;
                const short4 bx = bxs[u];
                if (b < nb && bx.z >= tx.a && bx.x < tx.b && bx.w >= y && bx.y <= y) {
                    hx0 = hx0 < bx.x ? hx0 : bx.x; hx1 = hx1 > bx.z ? hx1 : bx.z;
                    if (b < GV_PBOX && prs[u].z > 0) {
                        if (pn == 0) pi0 = b;
                        else if (pn == 1) pi1 = b;
                        else if (pn == 2) pi2 = b;
                        else if (pn == 3) pi3 = b;
                        else inline_compose = true;
                        pn += 1;
                    } else {
                        inline_compose = true;
                    }
                }
            }
        }
    }
    if (__any(inline_compose)) scan_cell_masks();   // whole wavefront: the scan shuffles across the group's lanes
    if (row_live) {
        if (hx1 < hx0) {
            hs = hs_static;
        } else {
            // x range [rx, rx + rw) and patch byte of column 0 of this row, per remembered box (rw = 0: not on this row)
            int rx0 = 0, rw0 = 0, ro0 = 0, rx1 = 0, rw1 = 0, ro1 = 0, rx2 = 0, rw2 = 0, ro2 = 0, rx3 = 0, rw3 = 0, ro3 = 0;
#define GV_ROW_RECT(I, RX, RW, RO)                                                                                   \
    if (pn > I) {                                                                                                    \
        const short4 rr = L.prect[pi##I];                                                                            \
        RX = rr.x; RW = (y < rr.y || y >= rr.y + rr.w) ? 0 : rr.z; RO = L.poff[pi##I] + (y - rr.y) * rr.z - rr.x;   \
    }
            GV_ROW_RECT(0, rx0, rw0, ro0) GV_ROW_RECT(1, rx1, rw1, ro1) GV_ROW_RECT(2, rx2, rw2, ro2) GV_ROW_RECT(3, rx3, rw3, ro3)
#undef GV_ROW_RECT
            if (!inline_compose) {
#ifndef NPP_GV_NO_UNROLL_X
                // a slice is 11 or 12 pixels: a fixed trip count lets the patch reads of all of them be in flight together (the sum
                // stays in x order)
                int pixv[12];
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const int x = tx.a + i;
                    const int o = x - xb0, sh = (o & 3) * 8;
                    int pix = (int)(((o < 4 ? pw.x : (o < 8 ? pw.y : (o < 12 ? pw.z : pw.w))) >> sh) & 0xffu);
                    int pa = -1;
                    if (x >= rx0 && x < rx0 + rw0) pa = ro0 + x;
                    else if (x >= rx1 && x < rx1 + rw1) pa = ro1 + x;
                    else if (x >= rx2 && x < rx2 + rw2) pa = ro2 + x;
                    else if (x >= rx3 && x < rx3 + rw3) pa = ro3 + x;
                    const int pv = L.patch[pa < 0 ? 0 : pa];
                    pixv[i] = pa < 0 ? pix : pv;
                }
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const int x = tx.a + i;
                    if (x < tx.b) hs += tab_w3(x, tx_s1, tx_s2, tx_wh, tx_wm, tx_wt) * (float)pixv[i];
                }
#else
                for (int x = tx.a; x < tx.b; x++) {
                    const int o = x - xb0, sh = (o & 3) * 8;
                    int pix = (int)(((o < 4 ? pw.x : (o < 8 ? pw.y : (o < 12 ? pw.z : pw.w))) >> sh) & 0xffu);
                    // every patch holds the finished pixel, so whichever covers x will do
                    if (x >= rx0 && x < rx0 + rw0) pix = L.patch[ro0 + x];
                    else if (x >= rx1 && x < rx1 + rw1) pix = L.patch[ro1 + x];
                    else if (x >= rx2 && x < rx2 + rw2) pix = L.patch[ro2 + x];
                    else if (x >= rx3 && x < rx3 + rw3) pix = L.patch[ro3 + x];
                    hs += tab_w3(x, tx_s1, tx_s2, tx_wh, tx_wm, tx_wt) * (float)pix;
                }
#endif
            } else {
                const uint32_t *cp = reinterpret_cast<const uint32_t *>(canvas + (size_t)y * 1056 + xb0);
                const uint4 cw = make_uint4(cp[0], cp[1], cp[2], cp[3]);
                for (int x = tx.a; x < tx.b; x++) {
                    const int o = x - xb0, sh = (o & 3) * 8;
                    int pix = (int)(((o < 4 ? pw.x : (o < 8 ? pw.y : (o < 12 ? pw.z : pw.w))) >> sh) & 0xffu);
                    if (x >= hx0 && x <= hx1) {
                        float eg = 0.f, ea = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            unsigned long long m = nm[w];
                            while (m) {
                                const Draw &d = L.draw[w * 64 + __builtin_ctzll(m)];
                                m &= m - 1;
                                const int cnt = draw_cover(d, x, y);
                                if (cnt) {
                                    float cov = cnt * (1.f / 16.f);
                                    eg = eg * (1.f - cov) + d.gray * cov;
                                    ea = ea * (1.f - cov) + cov;
                                }
                            }
                        }
                        pix = composite(eg, ea, (int)(((o < 4 ? cw.x : (o < 8 ? cw.y : (o < 12 ? cw.z : cw.w))) >> sh) & 0xffu));
                    }
                    hs += tab_w3(x, tx_s1, tx_s2, tx_wh, tx_wm, tx_wt) * (float)pix;
                }
            }
        }
    }
    // ordered vertical accumulation (y ascending) through the group's lanes
    float acc = 0.f;
    const int rows = ty.b - ty.a;
#pragma unroll
    for (int jr = 0; jr < 8; jr++) {
        const float h = __shfl(hs, (lane & ~7) + jr, 64);
        if (jr < rows) acc += tab_w3(ty.a + jr, ty_s1, ty_s2, ty_wh, ty_wm, ty_wt) * h;
    }
    if (sub == 0) out_env[cell] = (uint8_t)fminf(fmaxf(rintf(acc), 0.f), 255.f);   // cvRound + saturate
}

// Occupancy (round 2): 3 wavefronts per SIMD (136 VGPRs, ~12 KB of LDS per env on the door levels).  Capping the registers at
// 128 and trimming the LDS for 4 per SIMD was measured and dropped (246 -> 288 us inside the config-5 step): the launch lasts as
// long as its heaviest env (cells phase p50 36 k clocks, max 250-380 k), not as long as the average one.
#ifndef NPP_GV_WAVES
#define NPP_GV_WAVES 3
#endif
// wavefronts (= envs) per workgroup: the wavefronts of a workgroup never synchronise with each other, a workgroup only bundles
// them for the dispatcher
#ifndef NPP_GV_WPB
#define NPP_GV_WPB 1
#endif
constexpr int GV_WPB = NPP_GV_WPB;
// One WAVEFRONT per env (the per-env work is small and serial phases dominate: no workgroup barriers, ~20 KB of LDS, many envs
// in flight per CU).  Phases: copy the level's view; build the current draw list from the level's compact draw-order records
// and collect the dirty boxes; mark + queue the dirty destination cells; recompute them, 8 lanes per cell (one source row per
// lane, then an ordered accumulation through lane 0 of the group).
__global__ __launch_bounds__(64 * GV_WPB, NPP_GV_WAVES) void npp_global_view_kernel(KernelArgs a, int draw_cap, const uint8_t *gv_p, const float *gv_h,
                                                              const uint8_t *gv_v, uint8_t *out, unsigned char *xscr,
                                                              const uint32_t *order, uint32_t *cost) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gv_lds[];
    const int wv = threadIdx.x >> 6;
    const GvLds L = gv_lds_layout(gv_lds + (size_t)wv * gv_lds_bytes(draw_cap), draw_cap);
    const int slot = blockIdx.x * GV_WPB + wv, lane = threadIdx.x & 63;
    if (slot >= a.n) return;
    // heavy envs first: `order` lists the envs by the clocks their wavefront took in an earlier launch (npp_gv_order_kernel)
    const int env = order ? (int)order[slot] : slot;
    if (a.phase && a.phase_id >= 0 && a.phase[env] != (uint8_t)a.phase_id) return;   // stepped by the other part of a split launch
    const unsigned long long cost_t0 = __builtin_amdgcn_s_memtime();
#ifdef NPP_GV_STATS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int stat_nq = 0;
#endif
    const int lvl = __builtin_amdgcn_readfirstlane(a.env_level[env]);
    const LevelHdr &H = a.hdr[lvl];
    const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
    // issued here, consumed by the draw-list walk after the copy of the view: the first 128 draw records (one per lane and pass) and
    // the env's entity state words next to the level's init words, one word per lane (the walk then reads a record's two state codes
    // with a cross-lane read instead of two dependent global loads)
    const uint4 *recs = reinterpret_cast<const uint4 *>(a.blob + H.off_draw_recs);
    const uint32_t *init_words = reinterpret_cast<const uint32_t *>(a.blob + H.off_init_words);
    const uint32_t n_draw = H.n_ent + H.n_mov;
    const bool words_in_lanes = H.n_words <= 64u;
    uint4 rc_pre0 = make_uint4(0u, 0u, 0u, 0u), rc_pre1 = rc_pre0;
    if ((uint32_t)lane < n_draw) rc_pre0 = recs[lane];
    if ((uint32_t)lane + 64u < n_draw) rc_pre1 = recs[lane + 64];
    uint32_t w_now = 0u, w_init = 0u;
    if (words_in_lanes && (uint32_t)lane < H.n_words) {
        w_now = a.ent_bits[(size_t)lane * a.n + env];
        w_init = init_words[lane];
    }
    for (int i = lane; i < GV_WORDS; i += 64) L.dirty[i] = 0u;
    if (lane == 0) { L.ctr[0] = 0; L.ctr[1] = 0; L.ctr[2] = 0; }
    {   // the level's view, to be patched below
        const uint8_t *src = gv_v + (size_t)lvl * GV_CELLS;
        uint8_t *dst = out + (size_t)env * GV_CELLS;
        if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            static_assert(GV_CELLS / 16 == 17 * 64 + 12, "view copied as 17 full wavefront rows of uint4 + 12");
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src) + lane;
            uint4 *d4 = reinterpret_cast<uint4 *>(dst) + lane;
#define GV_COPY6(o)                                                                                                        \
    {                                                                                                                      \
        const uint4 v0 = s4[(o) * 64], v1 = s4[(o + 1) * 64], v2 = s4[(o + 2) * 64], v3 = s4[(o + 3) * 64],               \
                    v4 = s4[(o + 4) * 64], v5 = s4[(o + 5) * 64];                                                          \
        d4[(o) * 64] = v0; d4[(o + 1) * 64] = v1; d4[(o + 2) * 64] = v2; d4[(o + 3) * 64] = v3;                            \
        d4[(o + 4) * 64] = v4; d4[(o + 5) * 64] = v5;                                                                      \
    }
            GV_COPY6(0) GV_COPY6(6)   // six loads in flight per lane
            {
                const uint4 v0 = s4[12 * 64], v1 = s4[13 * 64], v2 = s4[14 * 64], v3 = s4[15 * 64], v4 = s4[16 * 64];
                uint4 v5 = v0;
                if (lane < 12) v5 = s4[17 * 64];
                d4[12 * 64] = v0; d4[13 * 64] = v1; d4[14 * 64] = v2; d4[15 * 64] = v3; d4[16 * 64] = v4;
                if (lane < 12) d4[17 * 64] = v5;
            }
#undef GV_COPY6
        } else {
            for (int i = lane; i < GV_CELLS; i += 64) dst[i] = src[i];
        }
    }
    wave_sync();
#ifdef NPP_GV_STATS
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- current draw list (draw order) + dirty boxes
    const float wx0 = -16.f, wy0 = -16.f, wx1 = 1056.f + 16.f, wy1 = 600.f + 16.f;
    int nd = 0;
    auto append = [&](bool keep, const Draw &d) {
        const unsigned long long bal = __ballot(keep);
        const int pos = nd + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep && pos < L.draw_cap) L.draw[pos] = d;   // (its cell box is computed later, and only if some row slice composes in place)
        nd += __popcll(bal);
    };
    if (H.n_door) {
        DrawCtx cc{&a, &H, env, wx0, wy0, wx1, wy1, false};
        DrawCtx ci = cc;
        ci.init = true;
        for (uint32_t d0 = 0; d0 < H.n_door; d0 += 64) {
            Draw d = {}, di = {};
            const uint32_t k = d0 + lane;
            const bool kc = k < H.n_door && door_drawable(cc, k, d);
            // a door's stroke depends on its slot's state only: same state as after the reset = same drawable, no dirty box
            bool same = true;
            if (k < H.n_door) {
                const int slot = (int)reinterpret_cast<const double *>(a.blob + H.off_doors)[5 * k + 4];
                same = ent_state_ctx(cc, slot) == ent_state_ctx(ci, slot);
            }
            if (!same) {
                const bool ki = door_drawable(ci, k, di);
                if (kc != ki || (kc && draw_differs(d, di))) {
                    if (kc) gv_push_box(L, d);
                    if (ki) gv_push_box(L, di);
                }
            }
            append(kc, d);
        }
    }
    {
        const double *zhead = a.zoo ? a.zoo + (size_t)env * a.zoo_words : nullptr;
        const uint32_t ovr = zhead ? reinterpret_cast<const uint32_t *>(zhead + 3)[0] : 0u;   // npp_set_entity_pos
        for (uint32_t k0 = 0; k0 < n_draw; k0 += 64) {
            const uint32_t k = k0 + lane;
            bool kc = false, ki = false;
            Draw d = {}, di = {};
            uint4 rc = k0 == 0 ? rc_pre0 : rc_pre1;
            if (k0 >= 128 && k < n_draw) rc = recs[k];
            // both state codes of the record's slot through cross-lane reads of the word lanes (every lane takes part)
            uint32_t sh_now = 0u, sh_init = 0u;
            if (words_in_lanes) {
                const int wl = (k < n_draw && !(rc.z & 0x8000u)) ? (int)((rc.z >> 16) >> 4) : 0;
                sh_now = (uint32_t)__shfl((int)w_now, wl, 64);
                sh_init = (uint32_t)__shfl((int)w_init, wl, 64);
            }
            if (k < n_draw) {
                const float x0 = __uint_as_float(rc.x), y0 = __uint_as_float(rc.y);
                float x = x0, y = y0;
                const uint32_t info = rc.z;
                const int slot = (int)(info >> 16);
                const bool mover = (info & 0x8000u) != 0;
                bool live = true;
                if (mover) {   // position from the env's zoo block
                    live = zhead != nullptr && H.has_zoo;
                    if (live) {
                        const double *zb = zhead + ZOO_HEAD + (a.zoo_doors + 1) / 2 + ZOO_MOV_WORDS * slot;
                        x = (float)zb[0]; y = (float)zb[1];
                    }
                } else if (ovr) {
                    if (slot == H.obs_switch && (ovr & ZOO_OVR_SWITCH)) { x = (float)zhead[4]; y = (float)zhead[5]; }
                    if (slot == H.obs_door && (ovr & ZOO_OVR_DOOR)) { x = (float)zhead[6]; y = (float)zhead[7]; }
                }
                const uint32_t st_now = mover ? 1u : (words_in_lanes ? (sh_now >> ((slot & 15) * 2)) & 3u : ent_state_of(a, env, slot));
                const uint32_t st_init = words_in_lanes && !mover ? (sh_init >> ((slot & 15) * 2)) & 3u
                                                                  : (init_words[slot >> 4] >> ((slot & 15) * 2)) & 3u;
                if (live && !(x < wx0 || x > wx1 || y < wy0 || y > wy1)) kc = rec_drawable(info, x, y, st_now, d);
                // the usual record: a static entity in the state and at the place it has after a reset -> the drawable IS the
                // init drawable, no dirty box, one evaluation instead of two (most of a level's mines, all its untouched gold)
                if (mover || st_now != st_init || x != x0 || y != y0) {
                    if (!mover && !(x0 < wx0 || x0 > wx1 || y0 < wy0 || y0 > wy1)) ki = rec_drawable(info, x0, y0, st_init, di);
                    if (kc != ki || (kc && draw_differs(d, di))) {
                        if (kc) gv_push_box(L, d);
                        if (ki) gv_push_box(L, di);
                    }
                }
            }
            append(kc, d);
        }
    }
    nd = nd < L.draw_cap - 1 ? nd : L.draw_cap - 1;
    if (lane == 0) {   // the ninja, drawn last
        const Draw nj = {(float)px, (float)py, 10.f, 0.f, 0.f, 0.f, 0};
        L.draw[nd] = nj;
        gv_push_box(L, nj);
    }
    nd += 1;
    wave_sync();
    if (L.ctr[0] > L.box_cap) {   // more dirty boxes than the list holds (a crowd of movers): everything is dirty, composed in the
        wave_sync();              // cell pass -- slow and exact
        if (lane == 0) { L.ctr[0] = 1; L.box[0] = make_short4(0, 0, 1055, 599); }
        wave_sync();
    }
    const int nb = L.ctr[0];
#ifdef NPP_GV_STATS
    const unsigned long long t1b = __builtin_amdgcn_s_memtime();   // draw list + dirty boxes done; the patches follow
#endif
    const uint8_t *canvas = a.tile_canvas + (size_t)lvl * 600 * 1056;
    // ---- the pixels inside the dirty boxes, composed ONE PER LANE into LDS patches (full wavefronts run the coverage sampler;
    //      the cell pass below then only sums bytes).  Boxes beyond the patch budget (many movers) are composed in the cell pass.
    const int npb = nb < GV_PBOX ? nb : GV_PBOX;
    if (lane == 0) {
        int used = 0;
        for (int b = 0; b < npb; b++) {
            const short4 bx = L.box[b];
            const int w = bx.z - bx.x + 1, h = bx.w - bx.y + 1;
            if (used + w * h <= GV_PATCH) {
                L.prect[b] = make_short4(bx.x, bx.y, (short)w, (short)h);
                L.poff[b] = (unsigned short)used;
                used += w * h;
            } else {
                L.prect[b] = make_short4(0, 0, 0, 0);   // width 0: no patch, composed in the cell pass
                L.poff[b] = 0;
            }
        }
        L.ctr[3] = used;
    }
    wave_sync();
    // may a row slice have to compose pixels in place (a dirty box without a patch, or five boxes on one slice)?  Only then does the
    // cell pass look at the drawables' cell boxes -- computed here, for the whole list, instead of at every append (round 3)
    const bool may_inline = nb >= 5 || nb > npb || __any(lane < npb && L.prect[lane < npb ? lane : 0].z == 0);
    if (may_inline) {
        for (int k = lane; k < nd; k += 64) L.cbox[k] = gv_cell_box(L.draw[k]);
        for (int k = nd + lane; k < ((nd + 3) & ~3); k += 64) L.cbox[k] = make_uchar4(255, 0, 255, 0);   // pad the last group of four: empty
    }
    for (int b = 0; b < npb; b++) {
        const short4 pr = L.prect[b];
        if (pr.z == 0) continue;
        // drawables that can touch the rectangle: the same set for every lane (scalar masks)
        unsigned long long bm[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            bool hit = false;
            const int k = w * 64 + lane;
            if (k < nd) {
                float cx, cy, ex, ey;
                draw_extent(L.draw[k], cx, cy, ex, ey);
                hit = cx + ex >= pr.x && cx - ex <= pr.x + pr.z && cy + ey >= pr.y && cy - ey <= pr.y + pr.w;
            }
            bm[w] = __ballot(hit);
        }
        const int np = pr.z * pr.w, pbase = L.poff[b];
        // row of pixel i = floor(i / width) without an integer division per pixel: (i + 0.5) / width lies at least 0.5 / width away
        // from every integer, far more than the rounding error of the float product (i < 3072) for the widths drawables have; a
        // wider rectangle keeps the division
        const float rw = 1.f / (float)pr.z;
        const bool narrow = pr.z <= 128;
        for (int i0 = lane; i0 < np; i0 += 128) {   // two pixels per lane and trip: their canvas loads overlap
            int xs[2], ys[2], cn[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + 64 * u < np ? i0 + 64 * u : i0;
                const int yy = narrow ? (int)(((float)i + 0.5f) * rw) : i / pr.z;
                xs[u] = pr.x + (i - yy * pr.z); ys[u] = pr.y + yy;
                cn[u] = canvas[(size_t)ys[u] * 1056 + xs[u]];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float eg = 0.f, ea = 0.f;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    unsigned long long m = bm[w];
                    while (m) {
                        const Draw &d = L.draw[w * 64 + __builtin_ctzll(m)];
                        m &= m - 1;
                        const int cnt = draw_cover(d, xs[u], ys[u]);
                        if (cnt) {
                            float cov = cnt * (1.f / 16.f);
                            eg = eg * (1.f - cov) + d.gray * cov;
                            ea = ea * (1.f - cov) + cov;
                        }
                    }
                }
                if (i0 + 64 * u < np) L.patch[pbase + i0 + 64 * u] = (unsigned char)composite(eg, ea, cn[u]);
            }
        }
    }
#ifdef NPP_GV_STATS
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- destination cells under the dirty boxes
    const float sx = 1056.f / GV_COLS, sy = 600.f / GV_ROWS;
    for (int b = 0; b < nb; b++) {   // the lanes share one box's candidate cells (a box covers a few rows x a few columns)
        const short4 bx = L.box[b];
        int c0 = (int)floorf(bx.x / sx) - 1, c1 = (int)floorf(bx.z / sx) + 1, r0 = (int)floorf(bx.y / sy) - 1, r1 = (int)floorf(bx.w / sy) + 1;
        c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
        c1 = c1 > GV_COLS - 1 ? GV_COLS - 1 : c1; r1 = r1 > GV_ROWS - 1 ? GV_ROWS - 1 : r1;
        const int nc = c1 - c0 + 1, cells = nc * (r1 - r0 + 1);
        for (int i = lane; i < cells; i += 64) {
            const int r = r0 + i / nc, c = c0 + i % nc;
            const AreaTab ty = gv_row_tab(r), tx = gv_col_tab(c);
            if (bx.w >= ty.a && bx.y < ty.b && bx.z >= tx.a && bx.x < tx.b) {   // source pixels [a, b) meet the box
                const int cell = r * GV_COLS + c;
                atomicOr(&L.dirty[cell >> 5], 1u << (cell & 31));
            }
        }
    }
    wave_sync();
    const uint8_t *pstat = gv_p + (size_t)lvl * 600 * 1056;
    const float *hrow = gv_h + (size_t)lvl * 600 * GV_COLS;
    const int grp = lane >> 3;
#ifdef NPP_GV_STATS
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
    {   // ---- few dirty cells (the usual case): hand the cell pass to npp_gv_cells_kernel
        int tot = 0;
        for (int w = lane; w < GV_WORDS; w += 64) tot += __popc(L.dirty[w]);
#pragma unroll
        for (int sft = 32; sft; sft >>= 1) tot += __shfl_xor(tot, sft, 64);
        if (xscr != nullptr && tot <= GV_XQ) {
            unsigned char *x = xscr + (size_t)env * GV_XSTRIDE;
            unsigned short *xq = reinterpret_cast<unsigned short *>(x + XS_QUEUE);
            for (int w = lane; w < GV_WORDS; w += 64) {
                uint32_t m = L.dirty[w];
                while (m) {
                    const int bit = __builtin_ctz(m);
                    m &= m - 1;
                    xq[atomicAdd(&L.ctr[1], 1)] = (unsigned short)(w * 32 + bit);
                }
            }
            const int used = L.ctr[3];
            // may a row have to compose in place?  (a dirty box without a patch, or five boxes on one row slice)
            const bool inl = may_inline;
            if (lane == 0) *reinterpret_cast<int4 *>(x + XS_HDR) = make_int4(nd, nb, tot, used | (inl ? (int)0x80000000 : 0));
            {
                uint2 *xb = reinterpret_cast<uint2 *>(x + XS_BOX);
                const uint2 *lb = reinterpret_cast<const uint2 *>(L.box);
                for (int i = lane; i < nb; i += 64) xb[i] = lb[i];
                uint2 *xr = reinterpret_cast<uint2 *>(x + XS_PRECT);
                const uint2 *lr = reinterpret_cast<const uint2 *>(L.prect);
                unsigned short *xo = reinterpret_cast<unsigned short *>(x + XS_POFF);
                for (int i = lane; i < npb; i += 64) { xr[i] = lr[i]; xo[i] = L.poff[i]; }
                uint32_t *xp = reinterpret_cast<uint32_t *>(x + XS_PATCH);
                const uint32_t *lp = reinterpret_cast<const uint32_t *>(L.patch);
                for (int i = lane; i < (used + 3) / 4; i += 64) xp[i] = lp[i];
                if (inl) {
                    uint32_t *xd = reinterpret_cast<uint32_t *>(x + XS_DRAW);
                    const uint32_t *ld = reinterpret_cast<const uint32_t *>(L.draw);
                    for (int i = lane; i < nd * 7; i += 64) xd[i] = ld[i];
                    uint32_t *xc = reinterpret_cast<uint32_t *>(x + XS_CBOX);
                    const uint32_t *lc = reinterpret_cast<const uint32_t *>(L.cbox);
                    for (int i = lane; i < ((nd + 3) & ~3); i += 64) xc[i] = lc[i];
                }
            }
#ifdef NPP_GV_STATS
            wave_sync();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) {   // diagnostic build only: the stamps of this kernel's phases over the view's first bytes (a cell of the
                               // first row that npp_gv_cells_kernel recomputes overwrites one of them)
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                uint32_t *dbg = reinterpret_cast<uint32_t *>(out + (size_t)env * GV_CELLS);
                dbg[0] = (uint32_t)(t1 - t0); dbg[1] = (uint32_t)(t2 - t1); dbg[2] = (uint32_t)(t3 - t2); dbg[3] = (uint32_t)(t4 - t3);
                dbg[4] = (uint32_t)nd; dbg[5] = (uint32_t)nb; dbg[6] = (uint32_t)tot; dbg[7] = H.n_ent + H.n_mov;
                dbg[8] = (uint32_t)(t1b - t1);
                for (int k = 9; k < 14; k++) dbg[k] = 0u;
            }
#endif
            return;
        }
        if (xscr != nullptr && lane == 0) *reinterpret_cast<int4 *>(xscr + (size_t)env * GV_XSTRIDE + XS_HDR) = make_int4(0, 0, 0, 0);   // nothing exported
    }
    // ---- many dirty cells (a crowd of movers, "everything dirty"): the cell pass runs in place
    // the dword copy of the view must have landed before single bytes of it are overwritten
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    for (;;) {
        // queue (up to GV_Q of) the dirty cells; a cell that does not fit keeps its bit for the next round
        for (int w = lane; w < GV_WORDS; w += 64) {
            uint32_t m = L.dirty[w], left = 0u;
            while (m) {
                const int bit = __builtin_ctz(m);
                m &= m - 1;
                const int slot = atomicAdd(&L.ctr[1], 1);
                if (slot < GV_Q) L.queue[slot] = (unsigned short)(w * 32 + bit);
                else { left |= 1u << bit; L.ctr[2] = 1; }
            }
            L.dirty[w] = left;
        }
        wave_sync();
        const int nq = L.ctr[1] < GV_Q ? L.ctr[1] : GV_Q;
        const int more = L.ctr[2];
#ifdef NPP_GV_STATS
        stat_nq += nq;
#endif
        for (int qi = grp; qi < nq; qi += 8) gv_cell(L, nd, nb, canvas, pstat, hrow, out + (size_t)env * GV_CELLS, L.queue[qi], lane);
        if (!more) break;
        wave_sync();
        if (lane == 0) { L.ctr[1] = 0; L.ctr[2] = 0; }
        wave_sync();
    }
#ifdef NPP_GV_STATS
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {   // diagnostic build only: phase durations (100 MHz ticks) and counts over the view's first bytes
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        uint32_t *dbg = reinterpret_cast<uint32_t *>(out + (size_t)env * GV_CELLS);
        dbg[0] = (uint32_t)(t1 - t0); dbg[1] = (uint32_t)(t2 - t1); dbg[2] = (uint32_t)(t3 - t2); dbg[3] = (uint32_t)(t4 - t3);
        dbg[4] = (uint32_t)nd; dbg[5] = (uint32_t)nb; dbg[6] = (uint32_t)stat_nq; dbg[7] = H.n_ent + H.n_mov;
        dbg[8] = (uint32_t)(t1b - t1);
        for (int k = 9; k < 14; k++) dbg[k] = 0u;
    }
#endif
    if (cost && lane == 0) cost[env] = (uint32_t)(__builtin_amdgcn_s_memtime() - cost_t0);
}

// Launch order of npp_global_view_kernel: envs binned by the clocks of their last measured wavefront (128 logarithmic bins, four per
// octave), heaviest bin first.  The launch lasts about as long as its slowest wavefronts; in env order those sit anywhere in the grid
// (the cost follows the level and the env's state, both of which change slowly), listed first they start at time zero: 223 -> 141 us
// on the door levels.  (A two-group variant -- above 1.5 x the mean first, both groups in env order -- was stable but weaker, 164 us,
// and its serial scan cost 21 us.)  One workgroup; a pure scheduling aid: every env appears exactly once whatever the costs are, and
// the order inside a bin is whatever the atomics make it.
__device__ inline int gv_cost_bin(uint32_t c) {
    if (c < 16u) return 0;
    const int msb = 31 - __builtin_clz(c);                 // 4 .. 31
    return (msb - 4) * 4 + (int)((c >> (msb - 2)) & 3u) + 1;   // 1 .. 112
}
__global__ __launch_bounds__(1024) void npp_gv_order_kernel(const uint32_t *cost, uint32_t *order, int n, int fold) {
    __shared__ int hist[128];
    __shared__ int base[128];
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += blockDim.x) atomicAdd(&hist[gv_cost_bin(cost[e])], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 127; b >= 0; b--) { base[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    // `fold` > 0 (npp_step, two workgroups per CU): positions fold .. 2 fold - 1 are filled backwards, so that if the dispatcher gives
    // workgroup fold + k the second slot of workgroup k's CU the heaviest workgroups share their CU with the lightest ones
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        int pos = atomicAdd(&base[gv_cost_bin(cost[e])], 1);
        if (fold > 0 && pos >= fold && pos < 2 * fold) {
            const int hi = (2 * fold < n ? 2 * fold : n) - 1;
            pos = fold + (hi - pos);
        }
        order[pos] = (uint32_t)e;
    }
}

// The cell pass of the envs that exported it: GV_XWAVES wavefronts per env; wavefront j stages the env's boxes / patches from its
// scratch block once and runs the same per-cell code as the in-place pass on the queue slices j, j + GV_XWAVES, ...
struct GvCellsLds {
    Draw draw[224];
    uchar4 cbox[224];
    short4 box[GV_BOX_MAX];
    short4 prect[GV_PBOX];
    unsigned short poff[GV_PBOX];
    unsigned short queue[GV_ITEM_CELLS];
    __attribute__((aligned(4))) unsigned char patch[GV_PATCH];
};
__global__ __launch_bounds__(64 * GV_WPB, NPP_GV_WAVES) void npp_gv_cells_kernel(KernelArgs a, const uint8_t *gv_p, const float *gv_h,
                                                                         const unsigned char *xscr, uint8_t *out) {
    __shared__ __attribute__((aligned(16))) GvCellsLds SS[GV_WPB];
    GvCellsLds &S = SS[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63, grp = lane >> 3;
    const int gw = blockIdx.x * GV_WPB + (threadIdx.x >> 6);
    const int env = gw / GV_XWAVES, j = gw - env * GV_XWAVES;
    if (env >= a.n) return;
#ifdef NPP_GV_STATS
    const unsigned long long k2t0 = __builtin_amdgcn_s_memtime();
#endif
    const unsigned char *x = xscr + (size_t)env * GV_XSTRIDE;
    const int4 hd = *reinterpret_cast<const int4 *>(x + XS_HDR);
    const int nd = hd.x, nb = hd.y, nq = hd.z, used = hd.w & 0xffff;
    if (j * GV_ITEM_CELLS >= nq) return;   // nothing for this wavefront (also: the env ran its cell pass in place, nq = 0)
    const bool inl = hd.w < 0;
    GvLds L;
    L.draw = S.draw; L.cbox = S.cbox; L.box = S.box; L.prect = S.prect; L.poff = S.poff; L.queue = S.queue; L.patch = S.patch;
    L.dirty = nullptr; L.ctr = nullptr; L.draw_cap = 224; L.box_cap = GV_BOX_MAX;
    const int npb = nb < GV_PBOX ? nb : GV_PBOX;
    {
        const uint2 *xb = reinterpret_cast<const uint2 *>(x + XS_BOX);
        uint2 *lb = reinterpret_cast<uint2 *>(S.box);
        for (int i = lane; i < nb; i += 64) lb[i] = xb[i];
        const uint2 *xr = reinterpret_cast<const uint2 *>(x + XS_PRECT);
        uint2 *lr = reinterpret_cast<uint2 *>(S.prect);
        const unsigned short *xo = reinterpret_cast<const unsigned short *>(x + XS_POFF);
        for (int i = lane; i < npb; i += 64) { lr[i] = xr[i]; S.poff[i] = xo[i]; }
        const uint32_t *xp = reinterpret_cast<const uint32_t *>(x + XS_PATCH);
        uint32_t *lp = reinterpret_cast<uint32_t *>(S.patch);
        for (int i = lane; i < (used + 3) / 4; i += 64) lp[i] = xp[i];
        if (inl) {
            const uint32_t *xd = reinterpret_cast<const uint32_t *>(x + XS_DRAW);
            uint32_t *ld = reinterpret_cast<uint32_t *>(S.draw);
            for (int i = lane; i < nd * 7; i += 64) ld[i] = xd[i];
            const uint32_t *xc = reinterpret_cast<const uint32_t *>(x + XS_CBOX);
            uint32_t *lc = reinterpret_cast<uint32_t *>(S.cbox);
            for (int i = lane; i < ((nd + 3) & ~3); i += 64) lc[i] = xc[i];
        }
    }
    const int lvl = __builtin_amdgcn_readfirstlane(a.env_level[env]);
    const uint8_t *canvas = a.tile_canvas + (size_t)lvl * 600 * 1056;
    const uint8_t *pstat = gv_p + (size_t)lvl * 600 * 1056;
    const float *hrow = gv_h + (size_t)lvl * 600 * GV_COLS;
    const unsigned short *xq = reinterpret_cast<const unsigned short *>(x + XS_QUEUE);
    for (int q0 = j * GV_ITEM_CELLS; q0 < nq; q0 += GV_XWAVES * GV_ITEM_CELLS) {
        const int ncell = nq - q0 < GV_ITEM_CELLS ? nq - q0 : GV_ITEM_CELLS;
        wave_sync();   // the previous slice's cells are done with the queue
        if (lane < ncell) S.queue[lane] = xq[q0 + lane];
        wave_sync();
        for (int qi = grp; qi < ncell; qi += 8) gv_cell(L, nd, nb, canvas, pstat, hrow, out + (size_t)env * GV_CELLS, S.queue[qi], lane);
    }
#ifdef NPP_GV_STATS
    wave_sync();
    if (lane == 0 && j == 0) {   // diagnostic build only: this wavefront's duration, dirty cells, in-place flag
        uint32_t *dbg = reinterpret_cast<uint32_t *>(out + (size_t)env * GV_CELLS);
        dbg[8] = (uint32_t)(__builtin_amdgcn_s_memtime() - k2t0); dbg[9] = (uint32_t)nq; dbg[10] = inl ? 1u : 0u; dbg[11] = (uint32_t)used;
    }
#endif
}

// The whole gray canvas of one env range (NPlayHeadless.render() in grayscale mode: nsim_renderer.py:71-134, array of shape
// (600, 1056, 1), nplay_headless.py:144-156).  One workgroup per (env, canvas row).
__global__ __launch_bounds__(256) void npp_full_frame_kernel(KernelArgs a, int env0, uint8_t *out) {
    __shared__ Draw s_draw[GV_DRAW];
    __shared__ int s_n;
    __shared__ int s_wc[4];
    const int e = blockIdx.x / 600, y = blockIdx.x - e * 600;
    const int env = env0 + e;
    if (env >= a.n) return;
    const int lvl = a.env_level[env];
    const LevelHdr &H = a.hdr[lvl];
    const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
    build_draw_list(a, H, env, px, py, -16.f, y - 16.f, 1056.f + 16.f, y + 1 + 16.f, s_draw, GV_DRAW, &s_n, s_wc);
    const uint8_t *canvas = a.tile_canvas + (size_t)lvl * 600 * 1056;
    uint8_t *row = out + ((size_t)e * 600 + y) * 1056;
    for (int x = threadIdx.x; x < 1056; x += blockDim.x) row[x] = (uint8_t)canvas_pixel(s_draw, s_n, canvas, x, y);
}

// switch_states (gym_environment/npp_environment.py:1782-1847): up to MAX_LOCKED_DOORS = 5 locked doors x [switch x, switch y,
// door x, door y, collected].  _extract_locked_door_positions looks for `segment.p1`, which GridSegmentLinear does not have
// (entities.py: x1, y1, x2, y2), so the "door" position falls back to the entity's xpos / ypos -- the switch position
// (entity_door_base.py:94-97).  Reproduced as is.
__global__ __launch_bounds__(256) void npp_switch_states_kernel(KernelArgs a, float *out) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= a.n) return;
    if (a.phase && a.phase_id >= 0 && a.phase[env] != (uint8_t)a.phase_id) return;
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

hipError_t launch_global_view(const KernelArgs &a, int max_records, const uint8_t *gv_p, const float *gv_h, const uint8_t *gv_v,
                              uint8_t *d_out, unsigned char *xscr, uint32_t *order, uint32_t *cost, int reorder, hipStream_t s) {
    int cap = max_records + 1;   // + the ninja
    cap = cap < 16 ? 16 : (cap > GV_DRAW ? GV_DRAW : cap);
#ifndef NPP_GV_SPLIT
    xscr = nullptr;   // shipped: the whole cell pass inside the first kernel (the split variant is an A/B build, see above)
#endif
    if (order && reorder) hipLaunchKernelGGL(npp_gv_order_kernel, dim3(1), dim3(1024), 0, s, cost, order, a.n, 0);
    hipLaunchKernelGGL(npp_global_view_kernel, dim3((a.n + GV_WPB - 1) / GV_WPB), dim3(64 * GV_WPB), GV_WPB * gv_lds_bytes(cap), s, a, cap, gv_p, gv_h,
                       gv_v, d_out, xscr, order, cost);
    if (xscr)
        hipLaunchKernelGGL(npp_gv_cells_kernel, dim3((a.n * GV_XWAVES + GV_WPB - 1) / GV_WPB), dim3(64 * GV_WPB), 0, s, a, gv_p, gv_h, xscr, d_out);
    return hipGetLastError();
}

// One wavefront that does nothing for `ticks` of the 100 MHz wall clock (bounded): the observation overlap's stream calibration launches
// two of them on two streams and looks at whether they ran side by side (npp_capi.cpp: streams_overlap).
__global__ __launch_bounds__(64) void npp_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    ticks = ticks > 100000 ? 100000 : ticks;   // at most 1 ms, whatever the caller asks for
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
hipError_t launch_spin(long long ticks, hipStream_t s) {
    hipLaunchKernelGGL(npp_spin_kernel, dim3(1), dim3(64), 0, s, ticks);
    return hipGetLastError();
}

// Observation overlap: which part of a split step launch steps an env, from the heavy-first workgroup order and the cut positions
// (order entries [0, e1) -> the last part, [e1, e2) -> the one before, ... the tail -> part 0).  Written BEFORE the parts are launched
// (whenever the order or the cuts change), never by the step kernels themselves: an observation kernel of one part filters on the
// bytes of ALL envs while the other parts may still be stepping.
__global__ __launch_bounds__(256) void npp_phase_kernel(const uint32_t *order, int blocks, int epb, int n, int e1, int e2, int e3, int parts,
                                                        uint8_t *phase) {
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= blocks) return;
    const int seg = (pos >= e1 ? 1 : 0) + (pos >= e2 ? 1 : 0) + (pos >= e3 ? 1 : 0);   // cuts that lie at or before this entry
    const uint8_t q = (uint8_t)(parts - 1 - seg);
    const int env0 = (int)order[pos] * epb;
    for (int j = 0; j < epb && env0 + j < n; j++) phase[env0 + j] = q;
}
hipError_t launch_phase_assign(const uint32_t *order, int blocks, int epb, int n, const int *edge, int parts, uint8_t *phase, hipStream_t s) {
    const int big = 0x7fffffff;
    hipLaunchKernelGGL(npp_phase_kernel, dim3((blocks + 255) / 256), dim3(256), 0, s, order, blocks, epb, n, parts > 1 ? edge[1] : big,
                       parts > 2 ? edge[2] : big, parts > 3 ? edge[3] : big, parts, phase);
    return hipGetLastError();
}

hipError_t launch_cost_order(const uint32_t *cost, uint32_t *order, int n, int fold, hipStream_t s) {
    hipLaunchKernelGGL(npp_gv_order_kernel, dim3(1), dim3(1024), 0, s, cost, order, n, fold);
    return hipGetLastError();
}

hipError_t launch_gv_static(const KernelArgs &a, int n_levels, uint8_t *gv_p, float *gv_h, uint8_t *gv_v, hipStream_t s) {
    hipLaunchKernelGGL(npp_gv_static_h_kernel, dim3(600, n_levels), dim3(256), 0, s, a, gv_p, gv_h);
    hipLaunchKernelGGL(npp_gv_static_v_kernel, dim3(GV_ROWS, n_levels), dim3(128), 0, s, gv_h, gv_v);
    return hipGetLastError();
}

hipError_t launch_tile_tables(hipStream_t s) {
    hipLaunchKernelGGL(npp_tile_tables_kernel, dim3((TILE_TAB + 255) / 256), dim3(256), 0, s);
    return hipGetLastError();
}

hipError_t launch_tile_canvas(const LevelHdr *d_hdr, const unsigned char *d_blob, uint8_t *d_canvas, int n_levels, hipStream_t s) {
    hipLaunchKernelGGL(npp_tile_canvas_kernel, dim3(600, n_levels), dim3(256), 0, s, d_hdr, d_blob, d_canvas);
    return hipGetLastError();
}

}  // namespace npp
