// npp_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the batched N++ tick.
//
// One wavefront lane per environment; one 64-lane wavefront per workgroup.  The ninja state lives in registers
// for the whole step (frame_skip ticks per launch), SoA planes in HBM are read and written once per launch with
// coalesced 512-byte wave accesses.  A workgroup whose 64 envs play the same level stages that level's packed
// collision table (CSR over cells + 16-bit segments + 8-bit cell bounds, ~11 KB) into LDS; per-env entity bits
// (2 bits per entity) are kept in LDS for the launch as well.  Observations are assembled in LDS and written
// with contiguous wave stores.  No MFMA: there is no dense contraction on this path (fp64 scalar chains).
//
// Arithmetic contract: IEEE fp64, no fused contraction (the reference is CPython float arithmetic); every
// comparison keeps the reference's strictness and operand order.  Reference citations are file:line in
// /root/reference/nclone/.
#include <hip/hip_runtime.h>

#include "npp_internal.hpp"
#include "npp_level.hpp"

#pragma clang fp contract(off)

namespace npp {
namespace {

// constants/physics_constants.py:11-60,282-344
constexpr double NINJA_RADIUS = 10.0;
constexpr double GRAVITY_FALL = 0.06666666666666665;
constexpr double GRAVITY_JUMP = 0.01111111111111111;
constexpr double GROUND_ACCEL = 0.06666666666666665;
constexpr double AIR_ACCEL = 0.04444444444444444;
constexpr double DRAG_REGULAR = 0.9933221725495059;
constexpr double DRAG_SLOW = 0.8617738760127536;
constexpr double FRICTION_GROUND = 0.9459290248857720;
constexpr double FRICTION_GROUND_SLOW = 0.8617738760127536;
constexpr double FRICTION_WALL = 0.9113380468927672;
constexpr double MAX_HOR_SPEED = 3.333;
constexpr int MAX_JUMP_DURATION = 45;
constexpr double MAX_SURVIVABLE_IMPACT = 6.0;
constexpr double TWO_THIRDS = 2.0 / 3;
// segment direction constants: sqrt(wx^2+wy^2) and w/len for the only shapes tiles produce
constexpr double LEN45 = 0x1.0f876ccdf6cd9p+5;   // sqrt(1152)
constexpr double DIR45 = 0x1.6a09e667f3bcdp-1;   // 24 / sqrt(1152)
constexpr double LEN26 = 0x1.ad5336963eefcp+4;   // sqrt(720)
constexpr double DIR26L = 0x1.c9f25c5bfedd9p-1;  // 24 / sqrt(720)
constexpr double DIR26S = 0x1.c9f25c5bfedd9p-2;  // 12 / sqrt(720)

#define DEV __device__ __forceinline__

struct Nj {
    double x, y, vx, vy, vxo, vyo, fnx, fny, cnx, cny;
    int state, airborn, airborn_old, walled, wn, jio, hor, jump, gjump, dslow;
    int jbuf, fbuf, wbuf, lbuf, cause, timpact;
    int jdur, fcount, ccount, pstate, fair, scf, frame, gold, doors, pcell;
};

struct Lv {
    const uint16_t *seg_start;
    const uint16_t *ent_start;
    const uint8_t *bounds;
    const uint16_t *segs;
    const double *ent_x;
    const double *ent_y;
    const uint32_t *ent_meta;
    const uint32_t *init_words;
    uint32_t n_think, n_words;
    int obs_switch, obs_door;
    double spawn_x, spawn_y, sw_x, sw_y, door_x, door_y;
};

DEV double sq(double v) { return v * v; }
DEV double dabs(double v) { return __builtin_fabs(v); }
DEV double dsqrt(double v) { return __builtin_sqrt(v); }
DEV double pymin(double a, double b) { return b < a ? b : a; }  // Python min(a, b)
DEV double pymax(double a, double b) { return b > a ? b : a; }  // Python max(a, b)

DEV int cell_coord(double p, int hi) {
    double q = __builtin_floor(p / 24.0);
    q = q < 0.0 ? 0.0 : q;          // also maps NaN to 0
    q = q > (double)hi ? (double)hi : q;
    return (int)q;
}

// ---- state planes <-> registers -------------------------------------------------------------------------------
DEV void load_state(const KernelArgs &a, int e, Nj &n) {
    const size_t N = (size_t)a.n;
    n.x = a.f64[F_X * N + e];     n.y = a.f64[F_Y * N + e];
    n.vx = a.f64[F_VX * N + e];   n.vy = a.f64[F_VY * N + e];
    n.vxo = a.f64[F_VXO * N + e]; n.vyo = a.f64[F_VYO * N + e];
    n.fnx = a.f64[F_FNX * N + e]; n.fny = a.f64[F_FNY * N + e];
    n.cnx = a.f64[F_CNX * N + e]; n.cny = a.f64[F_CNY * N + e];
    uint32_t A = a.u32[U_A * N + e], B = a.u32[U_B * N + e], C = a.u32[U_C * N + e], D = a.u32[U_D * N + e],
             E = a.u32[U_E * N + e];
    n.state = A & 15; n.airborn = (A >> 4) & 1; n.airborn_old = (A >> 5) & 1; n.walled = (A >> 6) & 1;
    n.wn = (int)((A >> 7) & 3) - 1; n.jio = (A >> 9) & 1; n.hor = (int)((A >> 10) & 3) - 1; n.jump = (A >> 12) & 1;
    n.gjump = (A >> 13) & 1; n.dslow = (A >> 14) & 1;
    n.jbuf = (int)((A >> 15) & 7) - 1; n.fbuf = (int)((A >> 18) & 7) - 1; n.wbuf = (int)((A >> 21) & 7) - 1;
    n.lbuf = (int)((A >> 24) & 7) - 1; n.cause = (A >> 27) & 3; n.timpact = (A >> 29) & 1;
    n.jdur = B & 63; n.fcount = (B >> 6) & 255; n.ccount = (B >> 14) & 255; n.pstate = (B >> 22) & 15;
    n.fair = C & 0xffff; n.scf = C >> 16;
    n.frame = D & 0xffff; n.gold = (D >> 16) & 255; n.doors = D >> 24;
    n.pcell = E & 0xffff;
}

DEV int sat(int v, int hi) { return v > hi ? hi : v; }

DEV void store_state(const KernelArgs &a, int e, const Nj &n) {
    const size_t N = (size_t)a.n;
    a.f64[F_X * N + e] = n.x;     a.f64[F_Y * N + e] = n.y;
    a.f64[F_VX * N + e] = n.vx;   a.f64[F_VY * N + e] = n.vy;
    a.f64[F_VXO * N + e] = n.vxo; a.f64[F_VYO * N + e] = n.vyo;
    a.f64[F_FNX * N + e] = n.fnx; a.f64[F_FNY * N + e] = n.fny;
    a.f64[F_CNX * N + e] = n.cnx; a.f64[F_CNY * N + e] = n.cny;
    uint32_t A = (uint32_t)n.state | (n.airborn << 4) | (n.airborn_old << 5) | (n.walled << 6) | ((n.wn + 1) << 7) |
                 (n.jio << 9) | ((n.hor + 1) << 10) | (n.jump << 12) | (n.gjump << 13) | (n.dslow << 14) |
                 ((n.jbuf + 1) << 15) | ((n.fbuf + 1) << 18) | ((n.wbuf + 1) << 21) | ((n.lbuf + 1) << 24) |
                 (n.cause << 27) | (n.timpact << 29);
    uint32_t B = (uint32_t)sat(n.jdur, 63) | (sat(n.fcount, 255) << 6) | (sat(n.ccount, 255) << 14) | (n.pstate << 22);
    uint32_t C = (uint32_t)sat(n.fair, 0xffff) | ((uint32_t)sat(n.scf, 0xffff) << 16);
    uint32_t D = (uint32_t)sat(n.frame, 0xffff) | (sat(n.gold, 255) << 16) | ((uint32_t)sat(n.doors, 255) << 24);
    uint32_t E = (uint32_t)n.pcell;
    a.u32[U_A * N + e] = A; a.u32[U_B * N + e] = B; a.u32[U_C * N + e] = C; a.u32[U_D * N + e] = D;
    a.u32[U_E * N + e] = E;
}

// Ninja.__init__ / reset_state (ninja.py:80-196,1288-1394)
DEV void spawn_state(const Lv &lv, Nj &n) {
    n.x = lv.spawn_x; n.y = lv.spawn_y; n.vx = 0; n.vy = 0; n.vxo = 0; n.vyo = 0;
    n.fnx = 0; n.fny = -1; n.cnx = 0; n.cny = 1;
    n.state = 0; n.airborn = 0; n.airborn_old = 0; n.walled = 0; n.wn = 0; n.jio = 0; n.hor = 0; n.jump = 0;
    n.gjump = 0; n.dslow = 0; n.jbuf = -1; n.fbuf = -1; n.wbuf = -1; n.lbuf = -1; n.cause = 0; n.timpact = 0;
    n.jdur = 0; n.fcount = 0; n.ccount = 0; n.pstate = 0; n.fair = 0; n.scf = 0; n.frame = 0; n.gold = 0; n.doors = 0;
    n.pcell = cell_coord(n.x, 43) * 25 + cell_coord(n.y, 24);
}

// ---- entity bits in LDS: word w of lane l at ew[w * 64 + l] ----------------------------------------------------
DEV uint32_t ent_get(const uint32_t *ew, int lane, int slot) {
    return (ew[(slot >> 4) * BLOCK + lane] >> ((slot & 15) * 2)) & 3u;
}
DEV void ent_set(uint32_t *ew, int lane, int slot, uint32_t v) {
    uint32_t *p = &ew[(slot >> 4) * BLOCK + lane];
    int sh = (slot & 15) * 2;
    *p = (*p & ~(3u << sh)) | (v << sh);
}

// ---- time-of-intersection primitives (physics.py:247-314) -----------------------------------------------------
DEV double toi_circle_point(double px, double py, double vx, double vy, double vel_sq, double a, double b, double radius) {
    double dx = px - a, dy = py - b;
    double dist_sq = sq(dx) + sq(dy);
    double dot_prod = dx * vx + dy * vy;
    double rr = sq(radius);
    if (dist_sq - rr > 0) {
        double radicand = sq(dot_prod) - vel_sq * (dist_sq - rr);
        if (vel_sq > 0.0001 && dot_prod < 0 && radicand >= 0) return (-dot_prod - dsqrt(radicand)) / vel_sq;
        return 1;
    }
    return 0;
}

// linear body; (wxu, wyu) = segment vector in units of 12 px
DEV double toi_circle_lineseg(double px, double py, double dx, double dy, double a1, double b1, int wxu, int wyu, double radius) {
    int ax = wxu < 0 ? -wxu : wxu, ay = wyu < 0 ? -wyu : wyu;
    double seg_len, mx, my;
    if (ay == 0 && ax == 1) { seg_len = 12.0; mx = 1.0; my = 0.0; }
    else if (ax == 0 && ay == 1) { seg_len = 12.0; mx = 0.0; my = 1.0; }
    else if (ax == 2 && ay == 2) { seg_len = LEN45; mx = DIR45; my = DIR45; }
    else if (ax == 2 && ay == 1) { seg_len = LEN26; mx = DIR26L; my = DIR26S; }
    else if (ax == 1 && ay == 2) { seg_len = LEN26; mx = DIR26S; my = DIR26L; }
    else {
        double wx = 12.0 * ax, wy = 12.0 * ay;
        seg_len = dsqrt(sq(wx) + sq(wy));
        mx = wx / seg_len; my = wy / seg_len;
    }
    double nx = wxu < 0 ? -mx : mx, ny = wyu < 0 ? -my : my;
    double normal_proj = (px - a1) * ny - (py - b1) * nx;
    double hor_proj = (px - a1) * nx + (py - b1) * ny;
    if (dabs(normal_proj) >= radius) {
        double dir = dx * ny - dy * nx;
        if (dir * normal_proj < 0) {
            double t = pymin((dabs(normal_proj) - radius) / dabs(dir), 1.0);
            double hor_proj2 = hor_proj + t * (dx * nx + dy * ny);
            if (0 <= hor_proj2 && hor_proj2 <= seg_len) return t;
        }
    } else {
        if (0 <= hor_proj && hor_proj <= seg_len) return 0;
    }
    return 1;
}

DEV double toi_circle_arc(double px, double py, double vx, double vy, double vel_sq, double a, double b, double hor, double ver,
                          double radius_circle) {
    double dx = px - a, dy = py - b;
    double dist_sq = sq(dx) + sq(dy);
    double dot_prod = dx * vx + dy * vy;
    double radius1 = 24.0 + radius_circle, radius2 = 24.0 - radius_circle;
    double t = 1;
    if (dist_sq > sq(radius1)) {
        double radicand = sq(dot_prod) - vel_sq * (dist_sq - sq(radius1));
        if (vel_sq > 0.0001 && dot_prod < 0 && radicand >= 0) t = (-dot_prod - dsqrt(radicand)) / vel_sq;
    } else if (dist_sq < sq(radius2)) {
        double radicand = sq(dot_prod) - vel_sq * (dist_sq - sq(radius2));
        if (vel_sq > 0.0001) t = pymin((-dot_prod + dsqrt(radicand)) / vel_sq, 1.0);
    } else {
        t = 0;
    }
    if ((dx + t * vx) * hor > 0 && (dy + t * vy) * ver > 0) return t;
    return 1;
}

// GridSegment*.intersect_with_ray (entities.py:82-96,180-203)
DEV double seg_toi(uint32_t s, int xc, int yc, double px, double py, double dx, double dy, double vel_sq, double radius) {
    double ox = 24.0 * xc, oy = 24.0 * yc;
    double t1, t2, t3;
    if ((s & 1u) == 0) {
        int ax = (s >> 2) & 3, ay = (s >> 4) & 3, bx = (s >> 6) & 3, by = (s >> 8) & 3;
        double x1 = ox + 12.0 * ax, y1 = oy + 12.0 * ay, x2 = ox + 12.0 * bx, y2 = oy + 12.0 * by;
        t1 = toi_circle_point(px, py, dx, dy, vel_sq, x1, y1, radius);
        t2 = toi_circle_point(px, py, dx, dy, vel_sq, x2, y2, radius);
        t3 = toi_circle_lineseg(px, py, dx, dy, x1, y1, bx - ax, by - ay, radius);
    } else {
        double cx = ox + 12.0 * ((s >> 2) & 3), cy = oy + 12.0 * ((s >> 4) & 3);
        double hor = ((s >> 6) & 1) ? 1.0 : -1.0, ver = ((s >> 7) & 1) ? 1.0 : -1.0;
        t1 = toi_circle_point(px, py, dx, dy, vel_sq, cx + 24.0 * hor, cy, radius);
        t2 = toi_circle_point(px, py, dx, dy, vel_sq, cx, cy + 24.0 * ver, radius);
        t3 = toi_circle_arc(px, py, dx, dy, vel_sq, cx, cy, hor, ver, radius);
    }
    double r = t1;
    if (t2 < r) r = t2;
    if (t3 < r) r = t3;
    return r;
}

// GridSegment*.get_closest_point (entities.py:43-59,127-157); returns is_back_facing; also yields the
// segment's AABB (entities.py:36-41,119-125)
DEV bool seg_closest(uint32_t s, int xc, int yc, double px, double py, double &a, double &b) {
    double ox = 24.0 * xc, oy = 24.0 * yc;
    if ((s & 1u) == 0) {
        int ax = (s >> 2) & 3, ay = (s >> 4) & 3, bx = (s >> 6) & 3, by = (s >> 8) & 3;
        double x1 = ox + 12.0 * ax, y1 = oy + 12.0 * ay;
        double wx = 12.0 * (bx - ax), wy = 12.0 * (by - ay);
        double dx = px - x1, dy = py - y1;
        double u = (dx * wx + dy * wy) / (sq(wx) + sq(wy));
        u = pymax(u, 0.0);
        u = pymin(u, 1.0);
        a = x1 + u * wx;
        b = y1 + u * wy;
        return dy * wx - dx * wy < 0;   // tile segments are always oriented
    }
    double cx = ox + 12.0 * ((s >> 2) & 3), cy = oy + 12.0 * ((s >> 4) & 3);
    double hor = ((s >> 6) & 1) ? 1.0 : -1.0, ver = ((s >> 7) & 1) ? 1.0 : -1.0;
    bool convex = (s >> 8) & 1;
    double dx = px - cx, dy = py - cy;
    bool back = false;
    if (dx * hor > 0 && dy * ver > 0) {
        double dist = dsqrt(sq(dx) + sq(dy));
        if (dist == 0) {
            if (dx * hor > dy * ver) { a = cx + 24.0 * hor; b = cy; }
            else { a = cx; b = cy + 24.0 * ver; }
            return false;
        }
        a = cx + 24.0 * dx / dist;
        b = cy + 24.0 * dy / dist;
        back = convex ? (dist < 24.0) : (dist > 24.0);
    } else {
        if (dx * hor > dy * ver) { a = cx + 24.0 * hor; b = cy; }
        else { a = cx; b = cy + 24.0 * ver; }
    }
    return back;
}

DEV void seg_aabb(uint32_t s, int xc, int yc, double &x0, double &y0, double &x1, double &y1) {
    int ux0, uy0, ux1, uy1;
    if ((s & 1u) == 0) {
        int ax = (s >> 2) & 3, ay = (s >> 4) & 3, bx = (s >> 6) & 3, by = (s >> 8) & 3;
        ux0 = ax < bx ? ax : bx; ux1 = ax < bx ? bx : ax;
        uy0 = ay < by ? ay : by; uy1 = ay < by ? by : ay;
    } else {
        int cx = (s >> 2) & 3, cy = (s >> 4) & 3;
        int hx = cx + (((s >> 6) & 1) ? 2 : -2), vy = cy + (((s >> 7) & 1) ? 2 : -2);
        ux0 = cx < hx ? cx : hx; ux1 = cx < hx ? hx : cx;
        uy0 = cy < vy ? cy : vy; uy1 = cy < vy ? vy : cy;
    }
    x0 = 24.0 * xc + 12.0 * ux0; x1 = 24.0 * xc + 12.0 * ux1;
    y0 = 24.0 * yc + 12.0 * uy0; y1 = 24.0 * yc + 12.0 * uy1;
}

// SpatialSegmentIndex.query_region cell filter (utils/spatial_segment_index.py:140-156, inclusive test :186-188)
DEV bool cell_passes(const Lv &lv, int c, int xc, int yc, double qx0, double qy0, double qx1, double qy1) {
    uint32_t cb = lv.bounds[c];
    double bx0 = 24.0 * xc + 12.0 * (cb & 3), by0 = 24.0 * yc + 12.0 * ((cb >> 2) & 3);
    double bx1 = 24.0 * xc + 12.0 * ((cb >> 4) & 3), by1 = 24.0 * yc + 12.0 * ((cb >> 6) & 3);
    return !(qx1 < bx0 || qx0 > bx1 || qy1 < by0 || qy0 > by1);
}

// sweep_circle_vs_tiles (physics.py:104-128)
DEV double sweep_circle_vs_tiles(const Lv &lv, double xo, double yo, double dx, double dy, double radius) {
    double xn = xo + dx, yn = yo + dy;
    double width = radius + 1;
    double qx0 = (xo < xn ? xo : xn) - width, qy0 = (yo < yn ? yo : yn) - width;
    double qx1 = (xo > xn ? xo : xn) + width, qy1 = (yo > yn ? yo : yn) + width;
    int c0x = cell_coord(qx0, 43), c1x = cell_coord(qx1, 43), c0y = cell_coord(qy0, 24), c1y = cell_coord(qy1, 24);
    double vel_sq = sq(dx) + sq(dy);
    double shortest = 1;
    for (int xc = c0x; xc <= c1x; xc++)
        for (int yc = c0y; yc <= c1y; yc++) {
            int c = xc * 25 + yc;
            int s0 = lv.seg_start[c], s1 = lv.seg_start[c + 1];
            if (s0 == s1) continue;
            if (!cell_passes(lv, c, xc, yc, qx0, qy0, qx1, qy1)) continue;
            for (int i = s0; i < s1; i++) {
                double t = seg_toi(lv.segs[i], xc, yc, xo, yo, dx, dy, vel_sq, radius);
                if (t == 0) return 0;
                if (t < shortest) shortest = t;
            }
        }
    return shortest;
}

// Ninja.collide_vs_tiles (ninja.py:269-379)
DEV void collide_vs_tiles(const Lv &lv, Nj &n, double xold, double yold, double &fnsx, double &fnsy, double &cnsx, double &cnsy) {
    double dx = n.x - xold, dy = n.y - yold;
    double time = sweep_circle_vs_tiles(lv, xold, yold, dx, dy, NINJA_RADIUS * 0.5);
    n.x = xold + time * dx;
    n.y = yold + time * dy;
    // the segment list is gathered ONCE at the post-sweep position (ninja.py:282-285): remember which cells passed
    double gx0 = n.x - NINJA_RADIUS, gy0 = n.y - NINJA_RADIUS, gx1 = n.x + NINJA_RADIUS, gy1 = n.y + NINJA_RADIUS;
    int c0x = cell_coord(gx0, 43), c1x = cell_coord(gx1, 43), c0y = cell_coord(gy0, 24), c1y = cell_coord(gy1, 24);
    uint32_t cmask = 0;
    for (int xc = c0x; xc <= c1x; xc++)
        for (int yc = c0y; yc <= c1y; yc++) {
            int c = xc * 25 + yc;
            if (lv.seg_start[c] == lv.seg_start[c + 1]) continue;
            if (cell_passes(lv, c, xc, yc, gx0, gy0, gx1, gy1)) cmask |= 1u << ((xc - c0x) * 4 + (yc - c0y));
        }
    if (cmask == 0) return;
    double xpos = n.x, ypos = n.y, xspeed = n.vx, yspeed = n.vy;
    for (int it = 0; it < 32; it++) {
        // get_single_closest_point (physics.py:131-180)
        double shortest = __builtin_inf();
        int result = 0;
        double ca = 0, cb = 0;
        double qx0 = xpos - NINJA_RADIUS, qy0 = ypos - NINJA_RADIUS, qx1 = xpos + NINJA_RADIUS, qy1 = ypos + NINJA_RADIUS;
        for (int xc = c0x; xc <= c1x; xc++)
            for (int yc = c0y; yc <= c1y; yc++) {
                if (!((cmask >> ((xc - c0x) * 4 + (yc - c0y))) & 1u)) continue;
                int c = xc * 25 + yc;
                int s0 = lv.seg_start[c], s1 = lv.seg_start[c + 1];
                for (int i = s0; i < s1; i++) {
                    uint32_t s = lv.segs[i];
                    double bx0, by0, bx1, by1;
                    seg_aabb(s, xc, yc, bx0, by0, bx1, by1);
                    if (bx1 < qx0 || bx0 > qx1 || by1 < qy0 || by0 > qy1) continue;
                    double a, b;
                    bool back = seg_closest(s, xc, yc, xpos, ypos, a, b);
                    double distance_sq = sq(xpos - a) + sq(ypos - b);
                    if (!back) distance_sq -= 0.1;
                    if (distance_sq < shortest) {
                        shortest = distance_sq;
                        ca = a; cb = b;
                        result = back ? -1 : 1;
                    }
                }
            }
        if (result == 0) break;
        dx = xpos - ca;
        dy = ypos - cb;
        if (dabs(dx) <= 0.0000001) {   // band-aid constants of the reference (ninja.py:313-318)
            dx = 0;
            if (xpos == 50.51197510492316 || xpos == 49.23232124849253) dx = -0x1p-47;
            if (xpos == 49.153536108584795) dx = 0x1p-47;
        }
        double dist_sq = dx * dx + dy * dy;
        if (dist_sq < 1e-16) break;
        double dist = dsqrt(dist_sq);
        double depen_len = NINJA_RADIUS - dist * result;
        if (depen_len < 0.0000001) break;
        double inv_dist = 1.0 / dist;
        double norm_dx = dx * inv_dist, norm_dy = dy * inv_dist;
        xpos += norm_dx * depen_len;
        ypos += norm_dy * depen_len;
        double dot_product = xspeed * dx + yspeed * dy;
        if (dot_product < 0) {
            double cross_product = xspeed * dy - yspeed * dx;
            double inv_dist_sq = inv_dist * inv_dist;
            xspeed = cross_product * inv_dist_sq * dy;
            yspeed = cross_product * inv_dist_sq * (-dx);
        }
        if (dy >= -0.0001) { n.ccount += 1; cnsx += norm_dx; cnsy += norm_dy; }
        else { n.fcount += 1; fnsx += norm_dx; fnsy += norm_dy; }
    }
    n.x = xpos; n.y = ypos; n.vx = xspeed; n.vy = yspeed;
}

// overlap_circle_vs_circle (physics.py:204-207) with an exact-safe early reject
DEV bool overlaps(double ex, double ey, double rsum, double px, double py) {
    double dx = ex - px, dy = ey - py;
    if (dabs(dx) > rsum + 1.0 || dabs(dy) > rsum + 1.0) return false;
    return dsqrt(sq(dx) + sq(dy)) < rsum;
}

DEV bool valid_target(int state) { return !(state == 6 || state == 8 || state == 9); }  // ninja.py:1272
DEV void ninja_kill(Nj &n, int cause) {   // ninja.py:1253-1270
    if (n.state < 6) { n.cause = cause; if (n.state == 3) n.gjump = 0; n.state = 7; }
}
DEV void ninja_win(Nj &n) {                // ninja.py:1246-1251
    if (n.state < 6) { if (n.state == 3) n.gjump = 0; n.state = 8; }
}
DEV double mine_radius(uint32_t st) { return st == 0 ? 4.0 : (st == 1 ? 3.5 : 4.5); }

// EntityToggleMine.think for every mine that can change (entity_toggle_mine.py:90-118).  The reference visits all
// mines each tick; only mines within one cell of the ninja now or at the previous think can change state
// (overlap radius <= 14.5 px < 24 px), and visiting a superset is harmless, so the bounding box of the two 3x3
// neighbourhoods is scanned.
DEV void think_mines(const Lv &lv, Nj &n, uint32_t *ew, int lane) {
    int ccx = cell_coord(n.x, 43), ccy = cell_coord(n.y, 24);
    int pcx = n.pcell / 25, pcy = n.pcell - pcx * 25;
    n.pcell = ccx * 25 + ccy;
    if (lv.n_think == 0) return;
    bool vt = valid_target(n.state);
    if (!vt && n.state != 6) return;
    int x0 = (ccx < pcx ? ccx : pcx) - 1, x1 = (ccx > pcx ? ccx : pcx) + 1;
    int y0 = (ccy < pcy ? ccy : pcy) - 1, y1 = (ccy > pcy ? ccy : pcy) + 1;
    x0 = x0 < 0 ? 0 : x0; x1 = x1 > 43 ? 43 : x1; y0 = y0 < 0 ? 0 : y0; y1 = y1 > 24 ? 24 : y1;
    for (int xc = x0; xc <= x1; xc++) {
        int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int i = i0; i < i1; i++) {
            if ((lv.ent_meta[i] & 15u) != EK_MINE) continue;
            uint32_t st = ent_get(ew, lane, i);
            if (vt) {
                if (st == 1) { if (overlaps(lv.ent_x[i], lv.ent_y[i], 3.5 + NINJA_RADIUS, n.x, n.y)) ent_set(ew, lane, i, 2); }
                else if (st == 2) { if (!overlaps(lv.ent_x[i], lv.ent_y[i], 4.5 + NINJA_RADIUS, n.x, n.y)) ent_set(ew, lane, i, 0); }
            } else if (st == 2) {
                ent_set(ew, lane, i, 1);
            }
        }
    }
}

// logical collisions of post_collision (ninja.py:388-420) over the 3x3 neighbourhood gathered x-major
// (physics.py:79-101); an exit door added to the grid by its switch this tick is not in the snapshot.
DEV void logical_collisions(const Lv &lv, Nj &n, uint32_t *ew, int lane) {
    int cx = cell_coord(n.x, 43), cy = cell_coord(n.y, 24);
    int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < 43 ? cx + 1 : 43, y0 = cy > 0 ? cy - 1 : 0, y1 = cy < 24 ? cy + 1 : 24;
    int pend0 = -1, pend1 = -1;
    for (int xc = x0; xc <= x1; xc++) {
        int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int i = i0; i < i1; i++) {
            uint32_t meta = lv.ent_meta[i];
            uint32_t kind = meta & 15u;
            uint32_t st = ent_get(ew, lane, i);
            double ex = lv.ent_x[i], ey = lv.ent_y[i];
            if (kind == EK_MINE) {   // entity_toggle_mine.py:120-128
                if (valid_target(n.state) && st == 0 && overlaps(ex, ey, 4.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(ew, lane, i, 1);
                    ninja_kill(n, 1);
                }
            } else if (st == 0) {
                continue;   // inactive (or exit door not yet in the grid)
            } else if (kind == EK_GOLD) {   // entity_gold.py:66-74
                if (n.state != 8 && overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) { n.gold += 1; ent_set(ew, lane, i, 0); }
            } else if (kind == EK_EXIT) {   // entity_exit.py:66-74
                if (overlaps(ex, ey, 12.0 + NINJA_RADIUS, n.x, n.y)) ninja_win(n);
            } else if (kind == EK_SWITCH) { // entity_exit_switch.py:67-129
                if (overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(ew, lane, i, 0);
                    int door = (int)((meta >> 8) & 0xffffu);
                    if (pend0 < 0) pend0 = door; else pend1 = door;
                }
            } else if (kind == EK_LOCKED) { // entity_door_locked.py:54-67
                if (overlaps(ex, ey, 5.0 + NINJA_RADIUS, n.x, n.y)) { n.doors += 1; ent_set(ew, lane, i, 0); }
            }
        }
    }
    if (pend0 >= 0) ent_set(ew, lane, pend0, 1);
    if (pend1 >= 0) ent_set(ew, lane, pend1, 1);
}

// Ninja.post_collision (ninja.py:381-537)
DEV void post_collision(const Lv &lv, Nj &n, uint32_t *ew, int lane, double fnsx, double fnsy, double cnsx, double cnsy) {
    logical_collisions(lv, n, ew, lane);
    // wall probe (ninja.py:424-441)
    double wall_normal = 0;
    const double rad = NINJA_RADIUS + 0.1;
    double qx0 = n.x - rad, qy0 = n.y - rad, qx1 = n.x + rad, qy1 = n.y + rad;
    int c0x = cell_coord(qx0, 43), c1x = cell_coord(qx1, 43), c0y = cell_coord(qy0, 24), c1y = cell_coord(qy1, 24);
    for (int xc = c0x; xc <= c1x; xc++)
        for (int yc = c0y; yc <= c1y; yc++) {
            int c = xc * 25 + yc;
            int s0 = lv.seg_start[c], s1 = lv.seg_start[c + 1];
            if (s0 == s1) continue;
            if (!cell_passes(lv, c, xc, yc, qx0, qy0, qx1, qy1)) continue;
            for (int i = s0; i < s1; i++) {
                double a, b;
                seg_closest(lv.segs[i], xc, yc, n.x, n.y, a, b);
                double dx = n.x - a, dy = n.y - b;
                if (dabs(dy) < 0.00001) {
                    double dist = dsqrt(sq(dx) + sq(dy));
                    if (0 < dist && dist <= rad) wall_normal += dx / dist;
                }
            }
        }
    n.airborn_old = n.airborn;
    n.airborn = 1;
    n.walled = 0;
    if (wall_normal != 0) { n.walled = 1; n.wn = wall_normal > 0 ? 1 : -1; }   // wall_normal / abs(wall_normal)
    if (n.fcount > 0) {
        n.airborn = 0;
        double floor_scalar = dsqrt(sq(fnsx) + sq(fnsy));
        if (floor_scalar == 0) { n.fnx = 0; n.fny = -1; }
        else { n.fnx = fnsx / floor_scalar; n.fny = fnsy / floor_scalar; }
        if (n.state != 8 && n.airborn_old) {
            double impact_vel = -(n.fnx * n.vxo + n.fny * n.vyo);
            if (impact_vel > MAX_SURVIVABLE_IMPACT - 4.0 / 3 * dabs(n.fny)) {
                n.vx = n.vxo; n.vy = n.vyo;
                ninja_kill(n, 2);
                n.timpact = 1;
            }
        }
    }
    n.fair = n.airborn ? n.fair + 1 : 0;
    if (n.ccount > 0) {
        double ceiling_scalar = dsqrt(sq(cnsx) + sq(cnsy));
        if (ceiling_scalar == 0) { n.cnx = 0; n.cny = 1; }
        else { n.cnx = cnsx / ceiling_scalar; n.cny = cnsy / ceiling_scalar; }
        if (n.state != 8) {
            double impact_vel = -(n.cnx * n.vxo + n.cny * n.vyo);
            if (impact_vel > MAX_SURVIVABLE_IMPACT - 4.0 / 3 * dabs(n.cny)) {
                n.vx = n.vxo; n.vy = n.vyo;
                ninja_kill(n, 2);
                n.timpact = 1;
            }
        }
    }
}

DEV void floor_jump(Nj &n) {   // ninja.py:539-579
    n.jbuf = -1; n.fbuf = -1; n.lbuf = -1;
    n.state = 3;
    n.gjump = 1;
    double jx, jy;
    if (n.fnx == 0) { jx = 0; jy = -2; }
    else {
        double dx = n.fnx, dy = n.fny;
        if (n.vx * dx >= 0) {
            if (n.vx * n.hor >= 0) { jx = TWO_THIRDS * dx; jy = 2 * dy; }
            else { jx = 0; jy = -1.4; }
        } else {
            if (n.vx * n.hor > 0) { jx = 0; jy = -1.4; }
            else { n.vx = 0; jx = TWO_THIRDS * dx; jy = 2 * dy; }
        }
    }
    if (n.vy > 0) n.vy = 0;
    n.vx += jx; n.vy += jy; n.x += jx; n.y += jy;
    n.jdur = 0;
}

DEV void wall_jump(Nj &n) {    // ninja.py:581-608
    double jx, jy;
    if (n.hor * n.wn < 0 && n.state == 5) { jx = TWO_THIRDS; jy = -1; }
    else { jx = 1; jy = -1.4; }
    n.state = 3;
    n.gjump = 1;
    double wn = (double)n.wn;
    if (n.vx * wn < 0) n.vx = 0;
    if (n.vy > 0) n.vy = 0;
    n.vx += jx * wn; n.vy += jy; n.x += jx * wn; n.y += jy;
    n.jbuf = -1; n.wbuf = -1; n.lbuf = -1;
    n.jdur = 0;
}

// Ninja.think (ninja.py:849-1059)
DEV void ninja_think(Nj &n) {
    if (n.state != n.pstate) { n.scf = 0; n.pstate = n.state; } else { n.scf += 1; }
    bool new_jump_check = n.jump ? (n.jio == 0) : false;
    n.jio = n.jump;
    n.lbuf = (-1 < n.lbuf && n.lbuf < 3) ? n.lbuf + 1 : -1;
    n.jbuf = (-1 < n.jbuf && n.jbuf < 5) ? n.jbuf + 1 : -1;
    bool in_jump_buffer = -1 < n.jbuf && n.jbuf < 5;
    n.wbuf = (-1 < n.wbuf && n.wbuf < 5) ? n.wbuf + 1 : -1;
    bool in_wall_buffer = -1 < n.wbuf && n.wbuf < 5;
    n.fbuf = (-1 < n.fbuf && n.fbuf < 5) ? n.fbuf + 1 : -1;
    bool in_floor_buffer = -1 < n.fbuf && n.fbuf < 5;
    if (new_jump_check && n.airborn) n.jbuf = 0;
    if (n.walled) n.wbuf = 0;
    if (!n.airborn) n.fbuf = 0;
    if (n.state == 6 || n.state == 9) return;
    if (n.state == 7) { n.state = 6; return; }
    if (n.state == 8) { n.dslow = n.airborn ? 0 : 1; return; }
    if (!n.airborn) {
        double xspeed_new = n.vx + GROUND_ACCEL * n.hor;
        if (dabs(xspeed_new) < MAX_HOR_SPEED) n.vx = xspeed_new;
        if (n.state > 2) {
            if (n.state == 3) n.gjump = 0;
            n.state = (n.vx * n.hor <= 0) ? 2 : 1;
        }
        if (!in_jump_buffer && !new_jump_check) {
            if (n.state == 2) {
                double projection = dabs(n.vy * n.fnx - n.vx * n.fny);
                if (n.hor * projection * n.vx > 0) { n.state = 1; return; }
                if (projection < 0.1 && n.fnx == 0) { n.state = 0; return; }
                if (n.vy < 0 && n.fnx != 0) {
                    double speed_scalar = dsqrt(sq(n.vx) + sq(n.vy));
                    double fric_force = dabs(n.vx * (1 - FRICTION_GROUND) * n.fny);
                    double fric_force2 = speed_scalar - fric_force * sq(n.fny);
                    n.vx = n.vx / speed_scalar * fric_force2;
                    n.vy = n.vy / speed_scalar * fric_force2;
                    return;
                }
                n.vx *= FRICTION_GROUND;
                return;
            }
            if (n.state == 1) {
                double projection = dabs(n.vy * n.fnx - n.vx * n.fny);
                if (n.hor * projection * n.vx > 0) {
                    if (n.hor * n.fnx >= 0) return;
                    if (dabs(xspeed_new) < MAX_HOR_SPEED) {
                        double boost = GROUND_ACCEL / 2 * n.hor;
                        double xboost = boost * n.fny * n.fny;
                        double yboost = boost * n.fny * -n.fnx;
                        n.vx += xboost;
                        n.vy += yboost;
                    }
                    return;
                }
                n.state = 2;
            } else {
                if (n.hor) { n.state = 1; return; }
                double projection = dabs(n.vy * n.fnx - n.vx * n.fny);
                if (projection < 0.1) { n.vx *= FRICTION_GROUND_SLOW; return; }
                n.state = 2;
            }
            return;
        }
        floor_jump(n);
        return;
    }
    double xspeed_new = n.vx + AIR_ACCEL * n.hor;
    if (dabs(xspeed_new) < MAX_HOR_SPEED) n.vx = xspeed_new;
    if (n.state < 3) { n.state = 4; return; }
    if (n.state == 3) {
        n.jdur += 1;
        if (!n.jump || n.jdur > MAX_JUMP_DURATION) { n.gjump = 0; n.state = 4; return; }
    }
    if (in_jump_buffer || new_jump_check) {
        if (n.walled || in_wall_buffer) { wall_jump(n); return; }
        if (in_floor_buffer) { floor_jump(n); return; }
        // launch-pad jump (ninja.py:1043-1045) needs a launch pad, which the accelerated path does not simulate
    }
    if (!n.walled) {
        if (n.state == 5) n.state = 4;
    } else if (n.state == 5) {
        if (n.hor * n.wn <= 0) n.vy *= FRICTION_WALL;
        else n.state = 4;
    } else if (n.vy > 0 && n.hor * n.wn < 0) {
        if (n.state == 3) n.gjump = 0;
        n.state = 5;
    }
}

// Simulator.tick (nsim.py:221-292)
DEV void sim_tick(const Lv &lv, Nj &n, uint32_t *ew, int lane, int hor, int jump) {
    n.frame += 1;
    n.hor = hor;
    n.jump = jump;
    think_mines(lv, n, ew, lane);
    if (n.state == 9) return;
    if (n.state != 6) {
        // integrate (ninja.py:198-206)
        double drag = n.dslow ? DRAG_SLOW : DRAG_REGULAR;
        n.vx *= drag;
        n.vy *= drag;
        n.vy += n.gjump ? GRAVITY_JUMP : GRAVITY_FALL;
        double xold = n.x, yold = n.y;
        n.x += n.vx;
        n.y += n.vy;
        // pre_collision (ninja.py:208-222)
        n.vxo = n.vx; n.vyo = n.vy;
        n.fcount = 0; n.ccount = 0;
        double fnsx = 0, fnsy = 0, cnsx = 0, cnsy = 0;
        // 4 substeps (nsim.py:263-267); collide_vs_objects has nothing physical to hit on this path
        for (int k = 0; k < 4; k++) collide_vs_tiles(lv, n, xold, yold, fnsx, fnsy, cnsx, cnsy);
        post_collision(lv, n, ew, lane, fnsx, fnsy, cnsx, cnsy);
    }
    ninja_think(n);
}

// ---- observations ---------------------------------------------------------------------------------------------
// get_ninja_state (nplay_headless.py:735-924) + time_remaining (base_environment.py:2811-2829); fp64 then f32 cast
DEV void write_game_state(const Nj &n, int limit, float *o /* stride 1 */) {
    double vmag = dsqrt(sq(n.vx) + sq(n.vy));
    o[0] = (float)(pymin(vmag / (MAX_HOR_SPEED * 2), 1.0) * 2 - 1);
    bool mv = vmag > 1e-6;
    o[1] = (float)(mv ? n.vx / vmag : 0.0);
    o[2] = (float)(mv ? n.vy / vmag : 0.0);
    o[3] = (n.state <= 2) ? 1.f : -1.f;
    o[4] = (n.state == 3 || n.state == 4) ? 1.f : -1.f;
    o[5] = (n.state == 5) ? 1.f : -1.f;
    o[6] = (n.state >= 6 && n.state <= 9) ? 1.f : -1.f;
    o[7] = n.airborn ? 1.f : -1.f;
    o[8] = (float)n.hor;
    o[9] = n.jump ? 1.f : -1.f;
    o[10] = (float)(((n.jbuf > 0 ? n.jbuf : 0) / 5.0) * 2 - 1);
    o[11] = (float)(((n.fbuf > 0 ? n.fbuf : 0) / 5.0) * 2 - 1);
    o[12] = (float)(((n.wbuf > 0 ? n.wbuf : 0) / 5.0) * 2 - 1);
    o[13] = (float)((n.fcount < 1 ? n.fcount : 1) * 2 - 1);
    o[14] = -1.f;                       // wall_count is never incremented in the reference (ninja.py:165,213)
    o[15] = (float)((n.ccount < 1 ? n.ccount : 1) * 2 - 1);
    o[16] = (float)(dsqrt(sq(n.fnx) + sq(n.fny)) * 2 - 1);
    o[17] = 0.f;                        // needs wall_count > 0
    o[18] = (float)n.fny;
    double g = n.gjump ? GRAVITY_JUMP : GRAVITY_FALL;
    o[19] = (float)((g - GRAVITY_JUMP) / (GRAVITY_FALL - GRAVITY_JUMP) * 2 - 1);
    o[20] = n.walled ? 1.f : -1.f;
    o[21] = (float)n.fnx;
    o[22] = (float)n.cnx;
    o[23] = (float)n.cny;
    double drag = n.dslow ? DRAG_SLOW : DRAG_REGULAR;
    o[24] = (float)((drag - DRAG_SLOW) / (DRAG_REGULAR - DRAG_SLOW) * 2 - 1);
    o[25] = (float)((FRICTION_GROUND - FRICTION_GROUND_SLOW) / (FRICTION_GROUND - FRICTION_GROUND_SLOW) * 2 - 1);
    o[26] = (float)pymax(-1.0, pymin(1.0, (n.vx - n.vxo) / MAX_HOR_SPEED));
    o[27] = (float)pymax(-1.0, pymin(1.0, (n.vy - n.vyo) / MAX_HOR_SPEED));
    o[28] = (float)(pymin(vmag / (MAX_HOR_SPEED * 1.5), 1.0) * 2 - 1);
    o[29] = (float)(pymin(n.fair / 60.0, 1.0) * 2 - 1);
    o[30] = (float)(pymin(n.jdur / (double)MAX_JUMP_DURATION, 1.0) * 2 - 1);
    o[31] = (float)(pymin(n.scf / 30.0, 1.0) * 2 - 1);
    double ke = 0.5 * (sq(n.vx) + sq(n.vy));
    o[32] = (float)(pymin(ke / sq(MAX_HOR_SPEED), 1.0) * 2 - 1);
    o[33] = (float)((n.y / 600.0) * 2 - 1);
    double fm = dsqrt(sq(g) + sq(!n.airborn ? GROUND_ACCEL : AIR_ACCEL));
    o[34] = (float)(pymin(fm / 0.1, 1.0) * 2 - 1);
    double pke = 0.5 * (sq(n.vxo) + sq(n.vyo));
    o[35] = (float)pymax(-1.0, pymin(1.0, (ke - pke) / pymax(ke + 0.01, 0.01)));
    o[36] = (float)(pymin(n.fcount / 5.0, 1.0) * 2 - 1);
    o[37] = -1.f;                       // min(wall_count / 3, 1) * 2 - 1 with wall_count == 0
    o[38] = (float)(atan2(n.fny, n.fnx) / 3.141592653589793);
    o[39] = n.walled ? (float)n.wn : 0.f;
    o[40] = (float)(limit <= 0 ? 1.0 : pymax(0.0, (double)(limit - n.frame) / (double)limit));
}

// Ninja.get_valid_action_mask (ninja.py:628-839), path-direction masking inert
DEV uint32_t action_mask_bits(const Nj &n) {
    uint32_t mask = 0x3f;
    bool has_active_buffer = (-1 < n.jbuf && n.jbuf < 5) || (-1 < n.fbuf && n.fbuf < 5) || (-1 < n.wbuf && n.wbuf < 5) ||
                             (-1 < n.lbuf && n.lbuf < 4);
    if (n.airborn && n.state != 3 && n.jump != 0 && !has_active_buffer) mask &= ~(8u | 16u | 32u);
    if (n.walled) {
        bool masks_dir = !n.airborn || n.state == 5 || !(n.vy >= 0);
        if (masks_dir) { if (n.wn > 0) mask &= ~2u; else if (n.wn < 0) mask &= ~4u; }
    }
    if (!mask) mask = 1;
    return mask;
}

// contiguous wave store of per-lane rows of `width` 4-byte words staged at stage[lane * width + k]
DEV void wave_store_rows(const uint32_t *stage, uint32_t *dst_block, int width, int lane, int n_valid) {
    int total = n_valid * width;
    for (int j = lane; j < total; j += BLOCK) dst_block[j] = stage[j];
}

template <bool LDS_LEVEL>
DEV void run(const KernelArgs &a, unsigned char *smem) {
    const int lane = threadIdx.x;
    const int env0 = blockIdx.x * BLOCK;
    const int env = env0 + lane;
    const bool valid = env < a.n;
    const int e = valid ? env : a.n - 1;
    const int n_valid = (a.n - env0) < BLOCK ? (a.n - env0) : BLOCK;

    uint32_t *ew = reinterpret_cast<uint32_t *>(smem + a.lds_hot_cap);
    uint32_t *stage = ew + (size_t)a.n_words_max * BLOCK;

    const int lvl = a.env_level[e];
    const LevelHdr &H = a.hdr[lvl];
    Lv lv;
    const unsigned char *hot = a.blob + H.off_hot;
    if (LDS_LEVEL) {
        // all 64 envs of this workgroup play level `lvl`: stage its collision table into LDS with 16-byte loads
        const uint4 *src = reinterpret_cast<const uint4 *>(hot);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const int nvec = (int)(H.hot_bytes >> 4);
        for (int i = lane; i < nvec; i += BLOCK) dst[i] = src[i];
        hot = smem;
    }
    lv.seg_start = reinterpret_cast<const uint16_t *>(hot + HOT_SEG_START);
    lv.ent_start = reinterpret_cast<const uint16_t *>(hot + HOT_ENT_START);
    lv.bounds = reinterpret_cast<const uint8_t *>(hot + HOT_BOUNDS);
    lv.segs = reinterpret_cast<const uint16_t *>(hot + HOT_SEGS);
    lv.ent_x = reinterpret_cast<const double *>(a.blob + H.off_ent_x);
    lv.ent_y = reinterpret_cast<const double *>(a.blob + H.off_ent_y);
    lv.ent_meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_ent_meta);
    lv.init_words = reinterpret_cast<const uint32_t *>(a.blob + H.off_init_words);
    lv.n_think = H.n_think; lv.n_words = H.n_words;
    lv.obs_switch = H.obs_switch; lv.obs_door = H.obs_door;
    lv.spawn_x = H.spawn_x; lv.spawn_y = H.spawn_y;
    lv.sw_x = H.sw_x; lv.sw_y = H.sw_y; lv.door_x = H.door_x; lv.door_y = H.door_y;

    Nj n;
    load_state(a, e, n);
    const int nw = (int)lv.n_words;
    for (int w = 0; w < nw; w++) ew[w * BLOCK + lane] = a.ent_bits[(size_t)w * a.n + e];
    __syncthreads();

    const int limit = a.trunc_limit[e];
    uint32_t flags = 0;
    int executed = 0;
    float reward = 0.f;
    const bool had_switch = lv.obs_switch >= 0 && ent_get(ew, lane, lv.obs_switch) == 0;

    if (a.mode == 0) {
        // NppEnvironment.step frame-skip loop (base_environment.py:524-609)
        const int act = a.n_ticks > 0 ? a.inputs[e] : 0;
        const int hor = (act == 1 || act == 4) ? -1 : ((act == 2 || act == 5) ? 1 : 0);   // :366-402
        const int jump = act >= 3 ? 1 : 0;
        // an env that is already terminal (no auto-reset) is not stepped again until the caller resets it
        bool live = valid && !(n.state == 8 || n.state == 6 || n.state == 7);
        for (int t = 0; t < a.n_ticks; t++) {
            if (live) {
                sim_tick(lv, n, ew, lane, hor, jump);
                executed++;
                if (n.state == 8 || n.state == 6 || n.state == 7) live = false;
            }
            if (!__any(live)) break;
        }
    } else {
        // NPlayHeadless.tick driven by replay bytes (replay/replay_executor.py:61-84)
        for (int t = 0; t < a.n_ticks; t++) {
            const int b = a.inputs[(size_t)t * a.n + e];
            const int l = (b >> 2) & 1, r = (b >> 1) & 1;
            const int hor = (l && r) ? 0 : (l ? -1 : (r ? 1 : 0));
            if (valid) sim_tick(lv, n, ew, lane, hor, b & 1);
            executed++;
        }
    }

    const bool sw_now = lv.obs_switch >= 0 ? ent_get(ew, lane, lv.obs_switch) == 0 : true;   // nplay_headless.py:566-576
    if (n.state == 8) flags |= 1u;
    if (n.state == 6 || n.state == 7) flags |= 2u;
    if (sw_now) flags |= 4u;
    if (n.cause == 1) flags |= 16u;
    if (n.cause == 2) flags |= 32u;
    bool done = (flags & 3u) != 0;
    const bool stepping = a.mode == 0 && a.n_ticks > 0;
    if (stepping && !done && n.frame >= limit) { flags |= 8u; done = true; }   // truncation_checker.py:46-77
    // sparse terminal reward: completion 200, switch 100, death -30, scaled by 0.1 (reward_constants.py:71,115,148,212)
    if (flags & 1u) reward += 20.f;
    if (flags & 2u) reward -= 3.f;
    if (sw_now && !had_switch && lv.obs_switch >= 0) reward += 10.f;

    if (valid) {
        if (a.out.flags) a.out.flags[env] = (uint8_t)flags;
        if (a.out.reward) a.out.reward[env] = reward;
        if (a.out.frames) a.out.frames[env] = (uint16_t)executed;
    }

    const bool do_reset = a.autoreset && stepping && done;
    if (a.out.terminal_state && do_reset && valid) write_game_state(n, limit, a.out.terminal_state + (size_t)env * 41);
    if (do_reset) {
        spawn_state(lv, n);
        for (int w = 0; w < nw; w++) ew[w * BLOCK + lane] = lv.init_words[w];
    }

    // observations, assembled in LDS then stored as contiguous wave writes
    if (a.out.game_state) {
        float *row = reinterpret_cast<float *>(stage) + lane * 41;
        write_game_state(n, limit, row);
        __syncthreads();
        wave_store_rows(stage, reinterpret_cast<uint32_t *>(a.out.game_state + (size_t)env0 * 41), 41, lane, n_valid);
        __syncthreads();
    }
    if (a.out.entity_pos) {
        float *row = reinterpret_cast<float *>(stage) + lane * 6;
        row[0] = (float)(n.x / 1056.0); row[1] = (float)(n.y / 600.0);
        row[2] = (float)(lv.sw_x / 1056.0); row[3] = (float)(lv.sw_y / 600.0);
        row[4] = (float)(lv.door_x / 1056.0); row[5] = (float)(lv.door_y / 600.0);
        __syncthreads();
        wave_store_rows(stage, reinterpret_cast<uint32_t *>(a.out.entity_pos + (size_t)env0 * 6), 6, lane, n_valid);
        __syncthreads();
    }
    if (a.out.action_mask && valid) {
        uint32_t m = action_mask_bits(n);
        int8_t *row = a.out.action_mask + (size_t)env * 6;
        for (int k = 0; k < 6; k++) row[k] = (int8_t)((m >> k) & 1u);
    }

    if (valid) {
        store_state(a, env, n);
        for (int w = 0; w < nw; w++) a.ent_bits[(size_t)w * a.n + env] = ew[w * BLOCK + lane];
    }
}

__global__ __launch_bounds__(BLOCK) void npp_step_kernel(KernelArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int env = blockIdx.x * BLOCK + threadIdx.x;
    const int e = env < a.n ? env : a.n - 1;
    const int lvl = a.env_level[e];
    const int lvl0 = __builtin_amdgcn_readfirstlane(lvl);
    const bool uniform = __all(lvl == lvl0) && a.hdr[lvl0].fits_lds;
    if (uniform) run<true>(a, smem);
    else run<false>(a, smem);
}

// Simulator.reset / fast_reset (nsim.py:62-140) for masked envs
__global__ __launch_bounds__(BLOCK) void npp_reset_kernel(KernelArgs a) {
    const int env = blockIdx.x * BLOCK + threadIdx.x;
    if (env >= a.n) return;
    if (a.reset_mask && a.reset_mask[env] == 0) return;
    const LevelHdr &H = a.hdr[a.env_level[env]];
    Lv lv;
    lv.spawn_x = H.spawn_x; lv.spawn_y = H.spawn_y;
    Nj n;
    spawn_state(lv, n);
    store_state(a, env, n);
    const uint32_t *init = reinterpret_cast<const uint32_t *>(a.blob + H.off_init_words);
    for (uint32_t w = 0; w < H.n_words; w++) a.ent_bits[(size_t)w * a.n + env] = init[w];
}

}  // namespace

hipError_t launch_step(const KernelArgs &a, hipStream_t s) {
    const int blocks = (a.n + BLOCK - 1) / BLOCK;
    const size_t lds = lds_bytes(a.lds_hot_cap, a.n_words_max);
    hipLaunchKernelGGL(npp_step_kernel, dim3(blocks), dim3(BLOCK), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_reset(const KernelArgs &a, hipStream_t s) {
    const int blocks = (a.n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(npp_reset_kernel, dim3(blocks), dim3(BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_render(const KernelArgs &, uint8_t *, hipStream_t) { return hipErrorNotSupported; }

}  // namespace npp
