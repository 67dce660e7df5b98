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
constexpr int MAX_DRAW = 96;   // drawables kept per window (discs + strokes)

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

struct Draw {
    float x, y, r;      // disc centre/radius, or stroke p1 + (r < 0)
    float x2, y2;       // stroke p2
    float gray;         // luma of the fill colour
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
        // drawables that can touch the window, in draw order (later ones overwrite: cairo operator SOURCE)
        int n = 0;
        const float wx0 = col0 - 16.f, wx1 = col0 + w + 16.f, wy0 = row0 - 16.f, wy1 = row0 + h + 16.f;
        const double *doors = reinterpret_cast<const double *>(a.blob + H.off_doors);
        for (uint32_t d = 0; d < H.n_door && n < MAX_DRAW; d++) {
            int slot = (int)doors[5 * d + 4];
            uint32_t st = (a.ent_bits[(size_t)(slot >> 4) * a.n + env] >> ((slot & 15) * 2)) & 3u;
            if (st == 0) continue;   // switch collected -> door open -> segment inactive
            float x1 = (float)doors[5 * d], y1 = (float)doors[5 * d + 1], x2 = (float)doors[5 * d + 2], y2 = (float)doors[5 * d + 3];
            if (fmaxf(x1, x2) < wx0 || fminf(x1, x2) > wx1 || fmaxf(y1, y2) < wy0 || fminf(y1, y2) > wy1) continue;
            s_draw[n++] = {x1, y1, -1.f, x2, y2, (float)luma(0x79, 0x79, 0x88)};
        }
        const uint16_t *order = reinterpret_cast<const uint16_t *>(a.blob + H.off_raster);
        const double *ex = reinterpret_cast<const double *>(a.blob + H.off_ent_x);
        const double *ey = reinterpret_cast<const double *>(a.blob + H.off_ent_y);
        const uint32_t *meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_ent_meta);
        for (uint32_t k = 0; k < H.n_ent && n < MAX_DRAW; k++) {
            int slot = order[k];
            float x = (float)ex[slot], y = (float)ey[slot];
            if (x < wx0 || x > wx1 || y < wy0 || y > wy1) continue;
            uint32_t m = meta[slot], kind = m & 15u, type = (m >> 24) & 63u;
            uint32_t st = (a.ent_bits[(size_t)(slot >> 4) * a.n + env] >> ((slot & 15) * 2)) & 3u;
            float r, g;
            if (kind == EK_MINE) {             // always active; radius follows the state
                r = st == 0 ? 4.0f : (st == 1 ? 3.5f : 4.5f);
                g = type == 1 ? (float)luma(0x9E, 0x21, 0x26) : (float)luma(0xCE, 0x41, 0x46);
            } else if (kind == EK_EXIT) {      // always active; dark blue (0, 0, 0.5) once the switch was hit
                r = 12.f;
                g = st ? (float)luma(0, 0, 128) : (float)luma(0x83, 0x83, 0x84);
            } else {
                if (st == 0) continue;         // collected gold / switches are inactive and not drawn
                if (kind == EK_GOLD) { r = 6.f; g = (float)luma(0xDB, 0xE1, 0x49); }
                else if (kind == EK_SWITCH) { r = 6.f; g = (float)luma(0x6D, 0x97, 0xC3); }
                else { r = 5.f; g = 0.f; }     // locked-door switch: black
            }
            s_draw[n++] = {x, y, r, 0.f, 0.f, g};
        }
        if (n < MAX_DRAW) s_draw[n++] = {(float)px, (float)py, 10.f, 0.f, 0.f, 0.f};   // the ninja, drawn last
        s_n = n;
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
        if (fr >= 0 && fr < h && fc >= 0 && fc < w) {
            const int y = row0 + fr, x = col0 + fc;   // canvas pixel
            // ---- entity layer (premultiplied gray + alpha), 4x4 supersampling
            float eg = 0.f, ea = 0.f;
            for (int k = 0; k < nd; k++) {
                const Draw d = s_draw[k];
                int cnt = 0;
                if (d.r >= 0.f) {
                    float ddx = (x + 0.5f) - d.x, ddy = (y + 0.5f) - d.y, lim = d.r + 1.f;
                    if (ddx * ddx + ddy * ddy > lim * lim) continue;
                    for (int sy = 0; sy < 4; sy++)
                        for (int sx = 0; sx < 4; sx++) {
                            float qx = x + (sx + 0.5f) * 0.25f - d.x, qy = y + (sy + 0.5f) * 0.25f - d.y;
                            cnt += (qx * qx + qy * qy <= d.r * d.r) ? 1 : 0;
                        }
                } else {
                    // stroke of width 2 with butt caps around the segment p1-p2
                    float vx = d.x2 - d.x, vy = d.y2 - d.y, len = sqrtf(vx * vx + vy * vy);
                    if (len <= 0.f) continue;
                    vx /= len; vy /= len;
                    for (int sy = 0; sy < 4; sy++)
                        for (int sx = 0; sx < 4; sx++) {
                            float qx = x + (sx + 0.5f) * 0.25f - d.x, qy = y + (sy + 0.5f) * 0.25f - d.y;
                            float al = qx * vx + qy * vy, pe = qx * vy - qy * vx;
                            cnt += (al >= 0.f && al <= len && fabsf(pe) <= 1.f) ? 1 : 0;
                        }
                }
                if (cnt) {
                    float cov = cnt * (1.f / 16.f);
                    eg = eg * (1.f - cov) + d.gray * cov;
                    ea = ea * (1.f - cov) + cov;
                }
            }
            int v = 202;   // int((0.299*203 + 0.587*202 + 0.114*208)) background (nsim_renderer.py:75-79)
            int a8 = (int)(ea * 255.f + 0.5f);
            if (a8 > 0) v = ((int)(eg + 0.5f) * a8 + v * (255 - a8)) >> 8;
            // ---- tile layer on top
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
            val = (uint8_t)v;
        }
        dst[p] = val;
    }
}

}  // namespace

hipError_t launch_render(const KernelArgs &a, uint8_t *d_out, int centered, hipStream_t s) {
    hipLaunchKernelGGL(npp_render_kernel, dim3(a.n), dim3(256), 0, s, a, d_out, centered);
    return hipGetLastError();
}

}  // namespace npp
