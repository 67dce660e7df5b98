// npp_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the batched N++ tick.
//
// G wavefront lanes cooperate on one environment (G = 1..64, chosen per launch from the env count so that the
// grid fills the 1024 SIMDs of the chip): the lanes of a group hold the same ninja state in registers, split the
// per-segment work of every region query between them and combine with DPP butterflies.  The state lives in
// registers for the whole step (frame_skip ticks per launch); SoA planes in HBM are read and written once per
// launch.  A workgroup whose envs all play the same level stages that level's packed collision table (CSR over
// cells + 16-bit segments + 8-bit cell bounds, ~11 KB) into LDS; per-env entity bits (2 bits per entity) are kept
// in LDS for the launch as well.  Observations are assembled in LDS and written with contiguous stores.
// No MFMA: there is no dense contraction on this path (fp64 scalar chains).
//
// Arithmetic contract: IEEE fp64, no fused contraction (the reference is CPython float arithmetic); every
// comparison keeps the reference's strictness and operand order.  Reference citations are file:line in
// /root/reference/nclone/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include <type_traits>

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

// Diagnostic build only (-DNPP_STAMPS, tools/stamp_profile.py): per-phase shader-clock totals accumulated by lane 0
// of every wavefront.  The shipped library is built without it (no stamp executes).
#ifdef NPP_STAMPS
constexpr int N_STAMP = 12;
__device__ unsigned long long g_stamps[N_STAMP * 16384];   // [wave][phase], written once per wave per launch
struct Stamps { unsigned long long acc[N_STAMP]; unsigned long long t0; };
#define STAMP_ARG , Stamps &st
#define STAMP_PASS , st
#define STAMP_INIT st.t0 = __builtin_amdgcn_s_memtime()
#define STAMP(i)                                                       \
    do {                                                               \
        unsigned long long _t1 = __builtin_amdgcn_s_memtime();         \
        st.acc[i] += _t1 - st.t0;                                      \
        st.t0 = __builtin_amdgcn_s_memtime();                          \
    } while (0)
#else
#define STAMP_ARG
#define STAMP_PASS
#define STAMP_INIT do { } while (0)
#define STAMP(i) do { } while (0)
#endif

struct Nj {
    double x, y, vx, vy, vxo, vyo, fnx, fny, cnx, cny;
    // (where the cached mine overlay of spatial_context was computed lives in the F_SCX / F_SCY state planes only: the one place
    // that reads or writes it is write_spatial_context, so the kernel does not carry it through the ticks)
    // counters with a wide range stay plain ints (they saturate when stored, never wrap); everything with a handful of values is
    // a bit field: two 32-bit storage units instead of twenty registers per lane.  The kernel is register-bound at two wavefronts
    // per SIMD ("Registers and occupancy"), and these fields are dead weight inside the collision loops.
    int fcount, ccount, fair, scf, frame, gold, doors;
    int work;   // depenetration iterations applied since the step began (npp_step_out.d_work; not part of the state)
    int fastord;   // bit 0: set after a Simulator.fast_reset -- the cell lists are in entity_dic order (nsim.py:124-140);
                   // bit 1: the level was assigned and no Simulator.reset has run since (the state of a fresh NppEnvironment after
                   // __init__'s load_map, base_environment.py:318): the next automatic reset is a FULL one, as in the reference, whose
                   // first reset() finds _last_reset_map_name None and reloads the map (npp_environment.py:518-557);
                   // bits 2-14: episode counter mod 8192, bumped by every reset (the reachability kernel drops the env's per-episode
                   // path-distance cache when it changes: reachability_mixin.py:67-70 clears the calculator at every reset)
    unsigned state : 4, airborn : 1, airborn_old : 1, walled : 1, jio : 1, jump : 1, gjump : 1, dslow : 1;
    int wn : 2, hor : 2, jbuf : 4, fbuf : 4, wbuf : 4, lbuf : 4;            // wn, hor in {-1, 0, 1}; buffers in -1 .. 5
    unsigned cause : 2, timpact : 1, jdur : 7, pstate : 4, scvalid : 1, pcell : 11;   // jdur <= 46, pcell < 1100
};
DEV int next_episode(int fastord) { return (((fastord >> 2) + 1) & 0x1fff) << 2; }

struct Lv {
    const uint16_t *seg_start;
    const uint16_t *ent_start;
    const uint8_t *bounds;
    const uint16_t *segs;
    const double *ent_x;
    const double *ent_y;
    const uint32_t *ent_meta;
    const uint32_t *init_words;
    const uint16_t *perm;   // CSR walk position -> slot: identity after Simulator.reset, entity_dic order after a fast reset
    uint32_t n_think, n_words;
    int obs_switch, obs_door;
    double spawn_x, spawn_y, sw_x, sw_y, door_x, door_y;
    double *spill;   // the env's LDS spill row (npp_internal.hpp: LDS_SPILL_BYTES): doubles [0, 8) parked ninja doubles, [8, 12) parked
                     // ninja counters (as ints), [12, ..) DepenIO
};

// the three collision tables, passed BY VALUE to the out-of-line fallbacks so that nothing the hot loop touches
// has to live in (scratch) memory
struct TileRefs {
    const uint16_t *seg_start;
    const uint16_t *segs;
    const uint8_t *bounds;
};

DEV double sq(double v) { return v * v; }
DEV double dabs(double v) { return __builtin_fabs(v); }
DEV double dsqrt(double v) { return __builtin_sqrt(v); }
DEV double pymin(double a, double b) { return b < a ? b : a; }  // Python min(a, b)
DEV double pymax(double a, double b) { return b > a ? b : a; }  // Python max(a, b)

// floor(p / 12) as an integer, exactly, without an fp64 division: the reciprocal product is off by at most one
// and is fixed up with exact comparisons (12 * k is exact).  floor(fl(p / 24)) of the reference equals the true
// floor(p / 24) for every double p (a p just below 24 k is at least 16 ulp(k) below it, so p / 24 cannot round up
// to k), and floor(p / 24) == floor(floor(p / 12) / 2).
DEV int floor12(double p) {
    double q = __builtin_floor(p * (1.0 / 12.0));
    q = q < -1.0e6 ? -1.0e6 : q;    // also maps NaN down
    q = q > 1.0e6 ? 1.0e6 : q;
    int k = (int)q;
    double lo = 12.0 * k;
    k = (lo > p) ? k - 1 : ((lo + 12.0 <= p) ? k + 1 : k);
    return k;
}
DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
DEV int cell_coord(double p, int hi) { return clampi(floor12(p) >> 1, 0, hi); }   // clamp(floor(p / 24), 0, hi)

// a / b for a constant b with y = RN(1 / b): Markstein's correction gives the correctly rounded quotient
// (checked against IEEE division on 1.2e9 operands for b in {144, 720, 1152}; tests/test_gpu_parity.py re-checks
// the whole path bit-for-bit against the oracle's true divisions).
DEV double div_const(double a, double b, double y) {
    double q = a * y;
    double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, y, q);
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
    n.scvalid = (E >> 16) & 1;
    n.fastord = (E >> 17) & 0x7fff;
    n.work = 0;
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
    uint32_t E = (uint32_t)n.pcell | ((uint32_t)n.scvalid << 16) | ((uint32_t)n.fastord << 17);
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
    n.scvalid = 0;   // reset_mine_overlay_cache (npp_environment.py:569-571); the anchor planes are rewritten with the next overlay
    n.fastord = 0; n.work = 0;
}

// ---- entity bits in LDS: word w of the env at w[w * stride] (stride = envs per workgroup) ----------------------------
struct EntBits {
    uint32_t *w;
    int stride;
};
DEV uint32_t ent_get(EntBits eb, int slot) { return (eb.w[(slot >> 4) * eb.stride] >> ((slot & 15) * 2)) & 3u; }
DEV void ent_set(EntBits eb, int slot, uint32_t v) {
    uint32_t *p = &eb.w[(slot >> 4) * eb.stride];
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
        // seg_lensq is 144 (axis aligned), 1152 (45 degrees) or 720 (the two gentle/steep slopes)
        int aw = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);   // 1, 8 or 5 in units of 144
        double num = dx * wx + dy * wy;
        double u = aw == 1 ? div_const(num, 144.0, 1.0 / 144.0)
                 : (aw == 8 ? div_const(num, 1152.0, 1.0 / 1152.0)
                 : (aw == 5 ? div_const(num, 720.0, 1.0 / 720.0) : num / (sq(wx) + sq(wy))));
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

// ---- cooperative groups of G lanes per environment -------------------------------------------------------------
// All G lanes of a group hold the same ninja state and execute the same scalar code redundantly; they split up only
// the per-segment work of a region query (lane r takes segments r, r+G, ... of the query's flat order) and combine
// with DPP butterflies.  Order-dependent semantics of the reference are preserved exactly: minima are
// order-independent, "first wins" ties (physics.py:176 strict <) are broken by the flat query index, and the wall
// probe's sum (ninja.py:441) is accumulated in query order.
// The butterfly partner always lies in the same lane group, whose lanes execute together, so it is always active:
// bound_ctrl lets the compiler fold the move into the consuming instruction instead of keeping a copy for `old`.
template <int CTRL> DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> DEV double dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = dpp_i<CTRL>(lo);
    hi = dpp_i<CTRL>(hi);
    return __hiloint2double(hi, lo);
}
// value held by the butterfly partner at distance STEP (1,2: quad_perm; 4: row_half_mirror; 8: row_mirror)
template <int STEP> DEV int partner_i(int v) {
    if constexpr (STEP == 1) return dpp_i<0xB1>(v);
    else if constexpr (STEP == 2) return dpp_i<0x4E>(v);
    else if constexpr (STEP == 4) return dpp_i<0x141>(v);
    else if constexpr (STEP == 8) return dpp_i<0x140>(v);
    else return __shfl_xor(v, STEP, 64);
}
template <int STEP> DEV double partner_d(double v) {
    if constexpr (STEP == 1) return dpp_d<0xB1>(v);
    else if constexpr (STEP == 2) return dpp_d<0x4E>(v);
    else if constexpr (STEP == 4) return dpp_d<0x141>(v);
    else if constexpr (STEP == 8) return dpp_d<0x140>(v);
    else return __shfl_xor(v, STEP, 64);
}

// v_min_f64 without the canonicalising v_max_f64 x, x, x that llvm.minnum puts in front of every operand it cannot prove
// quiet (anything that went through a select or a DPP move): the operands here are never NaN (+inf / 1 mark "none"), and each
// such instruction is ~7 cycles of the single wavefront's issue cadence inside the depenetration loop (five per iteration).
DEV double min_nonan(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int G, int STEP = 1> DEV double group_min(double t) {
    if constexpr (STEP < G) {
        t = min_nonan(t, partner_d<STEP>(t));
        return group_min<G, STEP * 2>(t);
    } else {
        return t;
    }
}

struct Best {
    double key;   // biased squared distance (physics.py:171-176)
    int idx;      // flat query index << 8 | evaluating lane's rank in its group << 1 | is_back_facing
    double a, b;  // closest point
};

template <int G, int STEP = 1> DEV int group_min_i(int v) {
    if constexpr (STEP < G) {
        int o = partner_i<STEP>(v);
        v = o < v ? o : v;
        return group_min_i<G, STEP * 2>(v);
    } else {
        return v;
    }
}

// Group-wide "first wins" argmin (physics.py:176 strict <): smallest key, ties broken by the smallest flat query
// index.  Two cheap butterflies (key, then index among the lanes holding the minimum key) and one cross-lane fetch
// of the winner's closest point; every lane of the group ends up with the same m.
template <int G> DEV void group_argmin(Best &m) {
    if constexpr (G > 1) {
        const double kmin = group_min<G>(m.key);
        const int cand = (m.idx != 0x7fffffff && m.key == kmin) ? m.idx : 0x7fffffff;
        const int imin = group_min_i<G>(cand);
        const int src = (((threadIdx.x & 63) & ~(G - 1)) + ((imin >> 1) & 63)) << 2;   // winner lane, byte address
        int alo = __builtin_amdgcn_ds_bpermute(src, __double2loint(m.a)), ahi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(m.a));
        int blo = __builtin_amdgcn_ds_bpermute(src, __double2loint(m.b)), bhi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(m.b));
        m.key = kmin;
        m.idx = imin;
        m.a = __hiloint2double(ahi, alo);
        m.b = __hiloint2double(bhi, blo);
    }
}

// SpatialSegmentIndex.query_region cell filter (utils/spatial_segment_index.py:140-156, inclusive test :186-188)
DEV bool cell_passes(uint32_t cb, int xc, int yc, double qx0, double qy0, double qx1, double qy1) {
    double bx0 = 24.0 * xc + 12.0 * (cb & 3), by0 = 24.0 * yc + 12.0 * ((cb >> 2) & 3);
    double bx1 = 24.0 * xc + 12.0 * ((cb >> 4) & 3), by1 = 24.0 * yc + 12.0 * ((cb >> 6) & 3);
    return !(qx1 < bx0 || qx0 > bx1 || qy1 < by0 || qy0 > by1);
}

// ---- per-tick candidate registers (fast path) ------------------------------------------------------------------
// Once per tick every lane group gathers ALL segments of the cells around the ninja's path (old position -> new
// position, inflated by the largest query radius) and keeps them decoded in registers: lane r holds candidates
// r, r + G, ... of the region's x-major order (K slots per lane).  A region query of the reference (sweep box,
// depenetration gather box, wall-probe box) whose clamped cell range lies inside the gathered range is then
// answered from registers by applying the reference's own filters per candidate (cell range, inclusive cell-bounds
// test, per-segment AABB test) -- the surviving candidates in flat order are exactly the reference's list.  A query
// that leaves the region (or a region with more segments than lanes x slots) falls back to the LDS walk below.
// All filters are evaluated in doubles on exact quantities (multiples of 12 px), so no integer cell arithmetic is
// needed on the fast path.
template <int G> struct KSlots { static constexpr int value = G >= 32 ? 1 : (G >= 16 ? 2 : 4); };
// Build variants of the G = 16 plain (non-zoo) step kernels, chosen per handle by npp_step's autotuner (npp_capi.cpp) because no
// one of them wins everywhere (8192 envs, MI355X, end of round 2):
//   V = 0  2 wavefronts per SIMD, 2 candidate slots per lane   headline 75.9 M env-steps/s, door levels 32.3 M, mine levels 97.6 M
//   V = 1  2 wavefronts per SIMD, 1 candidate slot per lane    headline 83.0 M (63 instead of 196 spilled VGPRs), mine levels 100.9 M, but
//                                                               door levels 25.5 M: their creases gather more than 16 segments and
//                                                               every iteration then takes the LDS fallback
//   V = 2  1 wavefront per SIMD (no register cap), 2 slots      headline 68.3 M, door levels 34.0 M (one long chain per launch: the
//                                                               fastest single chain wins)
// All variants give the same bits (tests/test_gpu_parity.py).  Other G and the zoo kernels exist as V = 0 only.
template <int G, bool ZOO, int V> struct VariantK { static constexpr int value = (G == 16 && !ZOO && V == 1) ? 1 : KSlots<G>::value; };

DEV double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// A query box plus copies clamped to the map: clamp(floor(q / 24), 0, 43) == floor(clamp(q, 0, 1032) / 24) and
// likewise with 576 for y (utils/spatial_segment_index.py:131-134), so "cell xc is inside the query's clamped cell
// range" is 24 xc <= clamp(qx1) && 24 (xc + 1) > clamp(qx0).
struct QBox {
    double x0, y0, x1, y1;
    double cx0, cy0, cx1, cy1;
};
DEV QBox make_qbox(double x0, double y0, double x1, double y1) {
    QBox q;
    q.x0 = x0; q.y0 = y0; q.x1 = x1; q.y1 = y1;
    q.cx0 = clampd(x0, 0.0, 1032.0); q.cx1 = clampd(x1, 0.0, 1032.0);
    q.cy0 = clampd(y0, 0.0, 576.0); q.cy1 = clampd(y1, 0.0, 576.0);
    return q;
}

// The per-segment AABB test of get_single_closest_point (physics.py:150-167) against the ninja's box [x - 10, x + 10] x [y - 10, y + 10],
// as exact thresholds on x and y: segment bounds are multiples of 12 in [0, 1056], the reference rounds x - 10 and x + 10 before it
// compares, and rounding is monotone, so "not (b1 < fl(x - 10))" is "x <= b1 + 10" (b1 + 10 is exact and its ulp is at least b1's) and
// "not (b0 > fl(x + 10))" is "x >= AABB_LO[b0 / 12]", the smallest double whose rounded sum reaches b0 (it lies below b0 - 10 where b0 - 10
// sits in a lower binade than b0: 8 of the 89 entries).  Generated and checked against the rounded sums on 40 doubles either side of every
// threshold (tests/test_host_cpu.py repeats the check on the table below); saves the four fp64 additions per candidate and iteration.
__constant__ double AABB_LO[89] = {
    -0x1.4000000000000p+3 /* 0: fl(x + 10) >= 0 */,
    0x1.ffffffffffffcp+0 /* 1: fl(x + 10) >= 12 */,
    0x1.bffffffffffffp+3 /* 2: fl(x + 10) >= 24 */,
    0x1.9ffffffffffffp+4 /* 3: fl(x + 10) >= 36 */,
    0x1.3000000000000p+5 /* 4: fl(x + 10) >= 48 */,
    0x1.9000000000000p+5 /* 5: fl(x + 10) >= 60 */,
    0x1.effffffffffffp+5 /* 6: fl(x + 10) >= 72 */,
    0x1.2800000000000p+6 /* 7: fl(x + 10) >= 84 */,
    0x1.5800000000000p+6 /* 8: fl(x + 10) >= 96 */,
    0x1.8800000000000p+6 /* 9: fl(x + 10) >= 108 */,
    0x1.b800000000000p+6 /* 10: fl(x + 10) >= 120 */,
    0x1.e7fffffffffffp+6 /* 11: fl(x + 10) >= 132 */,
    0x1.0c00000000000p+7 /* 12: fl(x + 10) >= 144 */,
    0x1.2400000000000p+7 /* 13: fl(x + 10) >= 156 */,
    0x1.3c00000000000p+7 /* 14: fl(x + 10) >= 168 */,
    0x1.5400000000000p+7 /* 15: fl(x + 10) >= 180 */,
    0x1.6c00000000000p+7 /* 16: fl(x + 10) >= 192 */,
    0x1.8400000000000p+7 /* 17: fl(x + 10) >= 204 */,
    0x1.9c00000000000p+7 /* 18: fl(x + 10) >= 216 */,
    0x1.b400000000000p+7 /* 19: fl(x + 10) >= 228 */,
    0x1.cc00000000000p+7 /* 20: fl(x + 10) >= 240 */,
    0x1.e400000000000p+7 /* 21: fl(x + 10) >= 252 */,
    0x1.fbfffffffffffp+7 /* 22: fl(x + 10) >= 264 */,
    0x1.0a00000000000p+8 /* 23: fl(x + 10) >= 276 */,
    0x1.1600000000000p+8 /* 24: fl(x + 10) >= 288 */,
    0x1.2200000000000p+8 /* 25: fl(x + 10) >= 300 */,
    0x1.2e00000000000p+8 /* 26: fl(x + 10) >= 312 */,
    0x1.3a00000000000p+8 /* 27: fl(x + 10) >= 324 */,
    0x1.4600000000000p+8 /* 28: fl(x + 10) >= 336 */,
    0x1.5200000000000p+8 /* 29: fl(x + 10) >= 348 */,
    0x1.5e00000000000p+8 /* 30: fl(x + 10) >= 360 */,
    0x1.6a00000000000p+8 /* 31: fl(x + 10) >= 372 */,
    0x1.7600000000000p+8 /* 32: fl(x + 10) >= 384 */,
    0x1.8200000000000p+8 /* 33: fl(x + 10) >= 396 */,
    0x1.8e00000000000p+8 /* 34: fl(x + 10) >= 408 */,
    0x1.9a00000000000p+8 /* 35: fl(x + 10) >= 420 */,
    0x1.a600000000000p+8 /* 36: fl(x + 10) >= 432 */,
    0x1.b200000000000p+8 /* 37: fl(x + 10) >= 444 */,
    0x1.be00000000000p+8 /* 38: fl(x + 10) >= 456 */,
    0x1.ca00000000000p+8 /* 39: fl(x + 10) >= 468 */,
    0x1.d600000000000p+8 /* 40: fl(x + 10) >= 480 */,
    0x1.e200000000000p+8 /* 41: fl(x + 10) >= 492 */,
    0x1.ee00000000000p+8 /* 42: fl(x + 10) >= 504 */,
    0x1.f9fffffffffffp+8 /* 43: fl(x + 10) >= 516 */,
    0x1.0300000000000p+9 /* 44: fl(x + 10) >= 528 */,
    0x1.0900000000000p+9 /* 45: fl(x + 10) >= 540 */,
    0x1.0f00000000000p+9 /* 46: fl(x + 10) >= 552 */,
    0x1.1500000000000p+9 /* 47: fl(x + 10) >= 564 */,
    0x1.1b00000000000p+9 /* 48: fl(x + 10) >= 576 */,
    0x1.2100000000000p+9 /* 49: fl(x + 10) >= 588 */,
    0x1.2700000000000p+9 /* 50: fl(x + 10) >= 600 */,
    0x1.2d00000000000p+9 /* 51: fl(x + 10) >= 612 */,
    0x1.3300000000000p+9 /* 52: fl(x + 10) >= 624 */,
    0x1.3900000000000p+9 /* 53: fl(x + 10) >= 636 */,
    0x1.3f00000000000p+9 /* 54: fl(x + 10) >= 648 */,
    0x1.4500000000000p+9 /* 55: fl(x + 10) >= 660 */,
    0x1.4b00000000000p+9 /* 56: fl(x + 10) >= 672 */,
    0x1.5100000000000p+9 /* 57: fl(x + 10) >= 684 */,
    0x1.5700000000000p+9 /* 58: fl(x + 10) >= 696 */,
    0x1.5d00000000000p+9 /* 59: fl(x + 10) >= 708 */,
    0x1.6300000000000p+9 /* 60: fl(x + 10) >= 720 */,
    0x1.6900000000000p+9 /* 61: fl(x + 10) >= 732 */,
    0x1.6f00000000000p+9 /* 62: fl(x + 10) >= 744 */,
    0x1.7500000000000p+9 /* 63: fl(x + 10) >= 756 */,
    0x1.7b00000000000p+9 /* 64: fl(x + 10) >= 768 */,
    0x1.8100000000000p+9 /* 65: fl(x + 10) >= 780 */,
    0x1.8700000000000p+9 /* 66: fl(x + 10) >= 792 */,
    0x1.8d00000000000p+9 /* 67: fl(x + 10) >= 804 */,
    0x1.9300000000000p+9 /* 68: fl(x + 10) >= 816 */,
    0x1.9900000000000p+9 /* 69: fl(x + 10) >= 828 */,
    0x1.9f00000000000p+9 /* 70: fl(x + 10) >= 840 */,
    0x1.a500000000000p+9 /* 71: fl(x + 10) >= 852 */,
    0x1.ab00000000000p+9 /* 72: fl(x + 10) >= 864 */,
    0x1.b100000000000p+9 /* 73: fl(x + 10) >= 876 */,
    0x1.b700000000000p+9 /* 74: fl(x + 10) >= 888 */,
    0x1.bd00000000000p+9 /* 75: fl(x + 10) >= 900 */,
    0x1.c300000000000p+9 /* 76: fl(x + 10) >= 912 */,
    0x1.c900000000000p+9 /* 77: fl(x + 10) >= 924 */,
    0x1.cf00000000000p+9 /* 78: fl(x + 10) >= 936 */,
    0x1.d500000000000p+9 /* 79: fl(x + 10) >= 948 */,
    0x1.db00000000000p+9 /* 80: fl(x + 10) >= 960 */,
    0x1.e100000000000p+9 /* 81: fl(x + 10) >= 972 */,
    0x1.e700000000000p+9 /* 82: fl(x + 10) >= 984 */,
    0x1.ed00000000000p+9 /* 83: fl(x + 10) >= 996 */,
    0x1.f300000000000p+9 /* 84: fl(x + 10) >= 1008 */,
    0x1.f900000000000p+9 /* 85: fl(x + 10) >= 1020 */,
    0x1.fefffffffffffp+9 /* 86: fl(x + 10) >= 1032 */,
    0x1.0280000000000p+10 /* 87: fl(x + 10) >= 1044 */,
    0x1.0580000000000p+10 /* 88: fl(x + 10) >= 1056 */
};

template <int K> struct Cand {
    double rx0, ry0, rx1, ry1;   // gathered cell range in pixels: [24 c0, 24 (c1 + 1))
    bool ok;
    uint32_t s[K];               // packed segment (bits 0-15, incl. cell y) | cell x << 16 | valid << 31
    double x1[K], y1[K], x2[K], y2[K];       // linear: end points; arc: centre, p_hor.x, p_ver.y
    double wx[K], wy[K], l2[K], rl2[K];      // linear: segment vector, |w|^2 and RN(1 / |w|^2)
    double ox[K], oy[K];                     // origin of the owning cell
    double bx0[K], by0[K], bx1[K], by1[K];   // the owning cell's bounds over its segments (:86-105)
    double tx0[K], tx1[K], ty0[K], ty1[K];   // the segment's AABB as thresholds on the ninja position (AABB_LO above): tx0 <= x <= tx1, ...
};

template <int G, int K>
DEV void cand_gather(const Lv &lv, int r, double qx0, double qy0, double qx1, double qy1, Cand<K> &c) {
    const int c0x = cell_coord(qx0, 43), c1x = cell_coord(qx1, 43), c0y = cell_coord(qy0, 24), c1y = cell_coord(qy1, 24);
    c.rx0 = 24.0 * c0x; c.rx1 = 24.0 * (c1x + 1); c.ry0 = 24.0 * c0y; c.ry1 = 24.0 * (c1y + 1);
    const int ncol = c1x - c0x + 1;
    int a0 = lv.seg_start[c0x * 25 + c0y], n0 = lv.seg_start[c0x * 25 + c1y + 1] - a0;
    int a1 = 0, n1 = 0, a2 = 0, n2 = 0;
    if (ncol > 1) { a1 = lv.seg_start[(c0x + 1) * 25 + c0y]; n1 = lv.seg_start[(c0x + 1) * 25 + c1y + 1] - a1; }
    if (ncol > 2) { a2 = lv.seg_start[(c0x + 2) * 25 + c0y]; n2 = lv.seg_start[(c0x + 2) * 25 + c1y + 1] - a2; }
    const int total = n0 + n1 + n2;
    c.ok = ncol <= 3 && total <= G * K;
#pragma unroll
    for (int k = 0; k < K; k++) {
        int f = k * G + r;
        c.s[k] = 0;
        c.x1[k] = 0; c.y1[k] = 0; c.x2[k] = 0; c.y2[k] = 0; c.wx[k] = 0; c.wy[k] = 0; c.l2[k] = 1; c.rl2[k] = 1;
        c.ox[k] = 0; c.oy[k] = 0; c.bx0[k] = 0; c.by0[k] = 0; c.bx1[k] = 0; c.by1[k] = 0;
        c.tx0[k] = 0; c.tx1[k] = 0; c.ty0[k] = 0; c.ty1[k] = 0;
        if (c.ok && f < total) {
            int xc = c0x, i = a0 + f;
            if (f >= n0) { xc += 1; i = a1 + (f - n0); }
            if (f >= n0 + n1) { xc += 1; i = a2 + (f - n0 - n1); }
            uint32_t s = lv.segs[i];
            int yc = s >> 11;
            uint32_t cb = lv.bounds[xc * 25 + yc];
            c.s[k] = s | ((uint32_t)xc << 16) | 0x80000000u;
            double ox = 24.0 * xc, oy = 24.0 * yc;
            c.ox[k] = ox; c.oy[k] = oy;
            c.bx0[k] = ox + 12.0 * (cb & 3); c.by0[k] = oy + 12.0 * ((cb >> 2) & 3);
            c.bx1[k] = ox + 12.0 * ((cb >> 4) & 3); c.by1[k] = oy + 12.0 * ((cb >> 6) & 3);
            c.x1[k] = ox + 12.0 * ((s >> 2) & 3); c.y1[k] = oy + 12.0 * ((s >> 4) & 3);
            if ((s & 1u) == 0) {
                c.x2[k] = ox + 12.0 * ((s >> 6) & 3); c.y2[k] = oy + 12.0 * ((s >> 8) & 3);
                int wxu = (int)((s >> 6) & 3) - (int)((s >> 2) & 3), wyu = (int)((s >> 8) & 3) - (int)((s >> 4) & 3);
                int aw = wxu * wxu + wyu * wyu;   // |w|^2 / 144: 1 axis aligned, 8 at 45 degrees, 5 for the 1:2 slopes
                c.wx[k] = 12.0 * wxu; c.wy[k] = 12.0 * wyu;
                c.l2[k] = aw == 1 ? 144.0 : (aw == 8 ? 1152.0 : 720.0);
                c.rl2[k] = aw == 1 ? 1.0 / 144.0 : (aw == 8 ? 1.0 / 1152.0 : 1.0 / 720.0);
                if (!(aw == 1 || aw == 8 || aw == 5)) c.ok = false;   // never produced by the tile tables
            } else {
                c.x2[k] = c.x1[k] + (((s >> 6) & 1) ? 24.0 : -24.0);   // p_hor.x (entities.py:113)
                c.y2[k] = c.y1[k] + (((s >> 7) & 1) ? 24.0 : -24.0);   // p_ver.y (entities.py:114)
            }
            {   // AABB of the segment in units of 12 px (entities.py:36-41,119-125): min / max of (x1, x2) and of (y1, y2)
                const int ux1 = 2 * xc + (int)((s >> 2) & 3), uy1 = 2 * yc + (int)((s >> 4) & 3);
                int ux2, uy2;
                if ((s & 1u) == 0) { ux2 = 2 * xc + (int)((s >> 6) & 3); uy2 = 2 * yc + (int)((s >> 8) & 3); }
                else { ux2 = ux1 + (((s >> 6) & 1) ? 2 : -2); uy2 = uy1 + (((s >> 7) & 1) ? 2 : -2); }
                const int ulo = ux1 < ux2 ? ux1 : ux2, uhi = ux1 < ux2 ? ux2 : ux1, vlo = uy1 < uy2 ? uy1 : uy2, vhi = uy1 < uy2 ? uy2 : uy1;
                if (ulo < 0 || uhi > 88 || vlo < 0 || vhi > 88) c.ok = false;   // never produced by the tile tables
                c.tx0[k] = AABB_LO[ulo < 0 ? 0 : (ulo > 88 ? 88 : ulo)]; c.tx1[k] = 12.0 * uhi + 10.0;
                c.ty0[k] = AABB_LO[vlo < 0 ? 0 : (vlo > 88 ? 88 : vlo)]; c.ty1[k] = 12.0 * vhi + 10.0;
            }
        }
    }
    if constexpr (G > 1) {   // the "unexpected segment length" veto must be group-wide
        const int glane0 = (threadIdx.x & 63) & ~(G - 1);
        unsigned long long bad = __ballot(!c.ok) >> glane0;
        if constexpr (G < 64) bad &= (1ull << G) - 1;
        c.ok = bad == 0;
    }
}

template <int K> DEV bool cand_covers(const Cand<K> &c, const QBox &q) {
    return c.ok & (q.cx0 >= c.rx0) & (q.cx1 < c.rx1) & (q.cy0 >= c.ry0) & (q.cy1 < c.ry1);
}

// would the reference's query have returned candidate k?  (its cell inside the query's clamped cell range and the
// inclusive cell-bounds test of utils/spatial_segment_index.py:186-188)
template <int K> DEV bool cand_in_box(const Cand<K> &c, int k, const QBox &q) {
    bool in_range = (c.ox[k] <= q.cx1) & (c.ox[k] + 24.0 > q.cx0) & (c.oy[k] <= q.cy1) & (c.oy[k] + 24.0 > q.cy0);
    bool reject = (q.x1 < c.bx0[k]) | (q.x0 > c.bx1[k]) | (q.y1 < c.by0[k]) | (q.y0 > c.by1[k]);
    return (c.s[k] >> 31) & in_range & !reject;
}

template <int G> DEV bool group_any(bool v) {
    if constexpr (G == 1) return v;
    const int glane0 = (threadIdx.x & 63) & ~(G - 1);
    unsigned long long bal = __ballot(v) >> glane0;
    if constexpr (G < 64) bal &= (1ull << G) - 1;
    return bal != 0;
}

// intersect_with_ray on a decoded candidate (entities.py:82-96,180-203)
DEV double cand_toi(uint32_t s, double x1, double y1, double x2, double y2, double px, double py, double dx, double dy,
                    double vel_sq, double radius) {
    double t1, t2, t3;
    if ((s & 1u) == 0) {
        int wxu = (int)((s >> 6) & 3) - (int)((s >> 2) & 3), wyu = (int)((s >> 8) & 3) - (int)((s >> 4) & 3);
        t1 = toi_circle_point(px, py, dx, dy, vel_sq, x1, y1, radius);
        t2 = toi_circle_point(px, py, dx, dy, vel_sq, x2, y2, radius);
        t3 = toi_circle_lineseg(px, py, dx, dy, x1, y1, wxu, wyu, radius);
    } else {
        double hor = ((s >> 6) & 1) ? 1.0 : -1.0, ver = ((s >> 7) & 1) ? 1.0 : -1.0;
        t1 = toi_circle_point(px, py, dx, dy, vel_sq, x2, y1, radius);
        t2 = toi_circle_point(px, py, dx, dy, vel_sq, x1, y2, radius);
        t3 = toi_circle_arc(px, py, dx, dy, vel_sq, x1, y1, hor, ver, radius);
    }
    double t = t1;
    if (t2 < t) t = t2;
    if (t3 < t) t = t3;
    return t;
}

// GridSegmentCircular.get_closest_point on a decoded candidate (entities.py:127-157)
DEV bool cand_closest_arc(uint32_t s, double x1, double y1, double x2, double y2, double px, double py,
                                              double &a, double &b) {
    double hor = ((s >> 6) & 1) ? 1.0 : -1.0, ver = ((s >> 7) & 1) ? 1.0 : -1.0;
    bool convex = (s >> 8) & 1;
    double dx = px - x1, dy = py - y1;
    bool back = false;
    if (dx * hor > 0 && dy * ver > 0) {
        double dist = dsqrt(sq(dx) + sq(dy));
        if (dist == 0) {
            if (dx * hor > dy * ver) { a = x2; b = y1; }
            else { a = x1; b = y2; }
            return false;
        }
        a = x1 + 24.0 * dx / dist;
        b = y1 + 24.0 * dy / dist;
        back = convex ? (dist < 24.0) : (dist > 24.0);
    } else {
        if (dx * hor > dy * ver) { a = x2; b = y1; }
        else { a = x1; b = y2; }
    }
    return back;
}

// GridSegmentLinear.get_closest_point (entities.py:43-59) on decoded registers, branch-free.  max/min are the
// hardware ones: they differ from Python's max(u, 0) / min(u, 1) only in the sign of a zero u, which cannot reach
// a or b (x1 + (+-0) * w == x1).
template <int K> DEV bool cand_closest_lin(const Cand<K> &c, int k, double px, double py, double &a, double &b) {
    double dx = px - c.x1[k], dy = py - c.y1[k];
    double num = dx * c.wx[k] + dy * c.wy[k];
    double u = div_const(num, c.l2[k], c.rl2[k]);
    u = __builtin_fmax(u, 0.0);
    u = __builtin_fmin(u, 1.0);
    a = c.x1[k] + u * c.wx[k];
    b = c.y1[k] + u * c.wy[k];
    return dy * c.wx[k] - dx * c.wy[k] < 0;
}

// sweep_circle_vs_tiles (physics.py:104-128).  The early `return 0` of the reference equals the minimum.
template <int G>
__device__ __noinline__ double sweep_generic(TileRefs lv, int r, double xo, double yo, double dx, double dy, double radius) {
    double xn = xo + dx, yn = yo + dy;
    double width = radius + 1;
    double qx0 = (xo < xn ? xo : xn) - width, qy0 = (yo < yn ? yo : yn) - width;
    double qx1 = (xo > xn ? xo : xn) + width, qy1 = (yo > yn ? yo : yn) + width;
    int c0x = cell_coord(qx0, 43), c1x = cell_coord(qx1, 43), c0y = cell_coord(qy0, 24), c1y = cell_coord(qy1, 24);
    double vel_sq = sq(dx) + sq(dy);
    double shortest = 1;
    for (int xc = c0x; xc <= c1x; xc++) {
        int i0 = lv.seg_start[xc * 25 + c0y], i1 = lv.seg_start[xc * 25 + c1y + 1];   // the column's cells are contiguous
        for (int i = i0 + r; i < i1; i += G) {
            uint32_t s = lv.segs[i];
            int yc = s >> 11;
            if (!cell_passes(lv.bounds[xc * 25 + yc], xc, yc, qx0, qy0, qx1, qy1)) continue;
            double t = seg_toi(s, xc, yc, xo, yo, dx, dy, vel_sq, radius);
            if (t < shortest) shortest = t;
        }
    }
    return group_min<G>(shortest);
}

// sqrt(x) and 1/x for operands that are known to be normal and far from the exponent limits (1e-16 <= x <= 1e8):
// exactly the instruction sequences hipcc emits for IEEE f64 sqrt and division (v_rsq / v_rcp seed, Newton-Raphson
// in fma, final residual correction) WITHOUT the v_ldexp range scaling, v_div_scale and v_cmp_class / v_div_fixup
// special-case handling, which are identities in this range -- so the results are the same bits.
DEV double sqrt_inrange(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
DEV double rcp_inrange(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double rem = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(rem, y, y);
}

// The depenetration loop of collide_vs_tiles (ninja.py:303-364) -- shared by the register fast path and the LDS
// fallback.  CLOSEST(m) fills `m` with the lane's best candidate; everything else is identical.
struct DepenIO {
    double x, y, vx, vy, fnsx, fnsy, cnsx, cnsy;
    int fcount, ccount;
};
// zoo kernels also carry the crush accumulators (ninja.py:344-346); only a thwump can make them matter
struct DepenIOZ : DepenIO {
    double xcr, ycr, clen;
};
DEV void crush_add(DepenIO &, double, double, double) {}
DEV void crush_add(DepenIOZ &io, double dx, double dy, double len) { io.xcr += dx; io.ycr += dy; io.clen += len; }

#define NPP_DEPEN_STEP(io, m, BREAK)                                                                   \
    {                                                                                                  \
        /* the reference leaves the loop at three places (ninja.py:307, 326, 331); nothing is modified \
         * before the last of them, so the three tests are folded into one exit */                     \
        const bool none = (m).idx == 0x7fffffff; /* result == 0 */                                     \
        const bool back_facing = ((m).idx & 1) != 0; /* result = -1 (physics.py:179) */                \
        double ddx = (io).x - (m).a;                                                                   \
        double ddy = (io).y - (m).b;                                                                   \
        if (dabs(ddx) <= 0.0000001) { /* band-aid constants of the reference (ninja.py:313-318) */     \
            ddx = 0;                                                                                   \
            if ((io).x == 50.51197510492316 || (io).x == 49.23232124849253) ddx = -0x1p-47;            \
            if ((io).x == 49.153536108584795) ddx = 0x1p-47;                                           \
        }                                                                                              \
        double dist_sq = ddx * ddx + ddy * ddy;                                                        \
        const bool tiny = dist_sq < 1e-16;                                                             \
        double dist = sqrt_inrange(dist_sq); /* garbage when tiny: the exit below does not look at it */ \
        double depen_len = NINJA_RADIUS - (back_facing ? -dist : dist); /* dist * result, exactly */   \
        if (none | tiny | (depen_len < 0.0000001)) BREAK;                                              \
        double inv_dist = rcp_inrange(dist);                                                           \
        double norm_dx = ddx * inv_dist, norm_dy = ddy * inv_dist;                                     \
        const double depen_x = norm_dx * depen_len, depen_y = norm_dy * depen_len;                     \
        (io).x += depen_x;                                                                             \
        (io).y += depen_y;                                                                             \
        crush_add((io), depen_x, depen_y, depen_len);                                                  \
        double dot_product = (io).vx * ddx + (io).vy * ddy;                                            \
        if (dot_product < 0) {                                                                         \
            double cross_product = (io).vx * ddy - (io).vy * ddx;                                      \
            double inv_dist_sq = inv_dist * inv_dist;                                                  \
            (io).vx = cross_product * inv_dist_sq * ddy;                                               \
            (io).vy = cross_product * inv_dist_sq * (-ddx);                                            \
        }                                                                                              \
        if (ddy >= -0.0001) { (io).ccount += 1; (io).cnsx += norm_dx; (io).cnsy += norm_dy; }          \
        else { (io).fcount += 1; (io).fnsx += norm_dx; (io).fnsy += norm_dy; }                         \
    }

// LDS-table fallback of the whole loop (rare: the query left the gathered region)
// `slot`: the env's LDS spill row holds the IO block (every lane of the group carries the same values; lane 0 wrote them): a struct
// of this size would travel through scratch memory as a by-value argument / return value.
template <int G, typename IO>
__device__ __noinline__ void depen_generic(TileRefs lv, int r, double gx0, double gy0, double gx1, double gy1, IO *slot) {
    IO io = *slot;
    const int c0x = cell_coord(gx0, 43), c1x = cell_coord(gx1, 43), c0y = cell_coord(gy0, 24), c1y = cell_coord(gy1, 24);
    for (int it = 0; it < 32; it++) {
        Best m;
        m.key = __builtin_inf(); m.idx = 0x7fffffff; m.a = 0; m.b = 0;
        const double qx0 = io.x - NINJA_RADIUS, qy0 = io.y - NINJA_RADIUS, qx1 = io.x + NINJA_RADIUS, qy1 = io.y + NINJA_RADIUS;
        int base = 0;
        for (int xc = c0x; xc <= c1x; xc++) {
            int i0 = lv.seg_start[xc * 25 + c0y], i1 = lv.seg_start[xc * 25 + c1y + 1];
            for (int i = i0 + r; i < i1; i += G) {
                uint32_t s = lv.segs[i];
                int yc = s >> 11;
                if (!cell_passes(lv.bounds[xc * 25 + yc], xc, yc, gx0, gy0, gx1, gy1)) continue;
                double bx0, by0, bx1, by1;
                seg_aabb(s, xc, yc, bx0, by0, bx1, by1);
                if (bx1 < qx0 || bx0 > qx1 || by1 < qy0 || by0 > qy1) continue;
                double a, b;
                bool back = seg_closest(s, xc, yc, io.x, io.y, a, b);
                double distance_sq = sq(io.x - a) + sq(io.y - b);
                if (!back) distance_sq -= 0.1;
                if (distance_sq < m.key) { m.key = distance_sq; m.a = a; m.b = b; m.idx = ((base + i - i0) << 8) | (r << 1) | (back ? 1 : 0); }
            }
            base += i1 - i0;
        }
        group_argmin<G>(m);
        NPP_DEPEN_STEP(io, m, break)
    }
    if (r == 0) *slot = io;
}

// Ninja.collide_vs_tiles (ninja.py:269-379).  Returns the number of depenetrations applied.
struct Crush { double xcr, ycr, clen; };
DEV void crush_in(DepenIO &, const Crush &) {}
DEV void crush_in(DepenIOZ &io, const Crush &c) { io.xcr = c.xcr; io.ycr = c.ycr; io.clen = c.clen; }
DEV void crush_out(const DepenIO &, Crush &) {}
DEV void crush_out(const DepenIOZ &io, Crush &c) { c.xcr = io.xcr; c.ycr = io.ycr; c.clen = io.clen; }
template <int G, int K, bool ZOO>
DEV int collide_vs_tiles(const Lv &lv, int r, Nj &n, const Cand<K> &cd, double xold, double yold, double &fnsx, double &fnsy,
                          double &cnsx, double &cnsy, Crush &cr STAMP_ARG) {
    double dx = n.x - xold, dy = n.y - yold;
    // ---- sweep_circle_vs_tiles (physics.py:104-128); the early `return 0` of the reference equals the minimum
    double time = 1;
    {
        const double radius = NINJA_RADIUS * 0.5, width = radius + 1;
        double xn = xold + dx, yn = yold + dy;
        const QBox q = make_qbox((xold < xn ? xold : xn) - width, (yold < yn ? yold : yn) - width,
                                 (xold > xn ? xold : xn) + width, (yold > yn ? yold : yn) + width);
        if (cand_covers(cd, q)) {
            bool in[K];
            bool any = false;
#pragma unroll
            for (int k = 0; k < K; k++) { in[k] = cand_in_box(cd, k, q); any |= in[k]; }
            if (group_any<G>(any)) {   // most sweeps touch no segment at all
                double vel_sq = sq(dx) + sq(dy);
                double shortest = 1;
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (in[k]) {
                        double t = cand_toi(cd.s[k], cd.x1[k], cd.y1[k], cd.x2[k], cd.y2[k], xold, yold, dx, dy, vel_sq, radius);
                        if (t < shortest) shortest = t;
                    }
                time = group_min<G>(shortest);
            }
        } else {
            time = sweep_generic<G>(TileRefs{lv.seg_start, lv.segs, lv.bounds}, r, xold, yold, dx, dy, radius);
        }
    }
    n.x = xold + time * dx;
    n.y = yold + time * dy;
    // the segment list is gathered ONCE at the post-sweep position (ninja.py:282-285): the cell filter keeps using
    // this box for all iterations, the per-segment AABB test uses the moving position (physics.py:150-167)
    const double gx0 = n.x - NINJA_RADIUS, gy0 = n.y - NINJA_RADIUS, gx1 = n.x + NINJA_RADIUS, gy1 = n.y + NINJA_RADIUS;
    const QBox qg = make_qbox(gx0, gy0, gx1, gy1);
    const bool fast = cand_covers(cd, qg);
    uint32_t gp = 0;
    if (fast) {
#pragma unroll
        for (int k = 0; k < K; k++) gp |= cand_in_box(cd, k, qg) ? (1u << k) : 0u;
        if (!group_any<G>(gp != 0)) return 0;   // empty list: result == 0 at the first iteration
    }
    using IO = typename std::conditional<ZOO, DepenIOZ, DepenIO>::type;
    IO io;
    io.x = n.x; io.y = n.y; io.vx = n.vx; io.vy = n.vy;
    io.fnsx = fnsx; io.fnsy = fnsy; io.cnsx = cnsx; io.cnsy = cnsy;
    io.fcount = n.fcount; io.ccount = n.ccount;
    const int counts_before = n.fcount + n.ccount;   // every applied depenetration increments exactly one of the two counters
    crush_in(io, cr);
    STAMP(9);   // sweep + gather setup
    if (fast) {
        // Loop-invariant, wavefront-uniform shortcuts (scalar branches instead of exec-mask regions inside the chain):
        // does any gathered candidate of this wavefront describe an arc / sit in a slot beyond the first?
        bool arc_here = false;
#pragma unroll
        for (int k = 0; k < K; k++) arc_here |= ((gp >> k) & 1u) && (cd.s[k] & 1u);
        const bool wave_arcs = __any(arc_here);
        const bool wave_more = __any((gp >> 1) != 0);
        for (int it = 0; it < 32; it++) {
#ifdef NPP_STAMPS
            st.acc[10] += 1;   // iteration count (register fast path)
#endif
            // get_single_closest_point (physics.py:131-180) over the candidate registers
            Best m;
            m.key = __builtin_inf(); m.idx = 0x7fffffff; m.a = 0; m.b = 0;
#pragma unroll
            for (int k = 0; k < K; k++)
                if (k == 0 || (wave_more && ((gp >> k) & 1u))) {   // slot 0 is evaluated unconditionally (masked by gp below)
                    // the reference's AABB test on the rounded box x -+ 10, y -+ 10, as exact thresholds (AABB_LO)
                    bool in = ((gp >> k) & 1u) & (io.x >= cd.tx0[k]) & (io.x <= cd.tx1[k]) & (io.y >= cd.ty0[k]) & (io.y <= cd.ty1[k]);
                    double a, b;
                    bool back = cand_closest_lin(cd, k, io.x, io.y, a, b);
                    if (wave_arcs && (in & ((cd.s[k] & 1u) != 0))) back = cand_closest_arc(cd.s[k], cd.x1[k], cd.y1[k], cd.x2[k], cd.y2[k], io.x, io.y, a, b);
                    double distance_sq = sq(io.x - a) + sq(io.y - b);
                    double key = back ? distance_sq : distance_sq - 0.1;
                    const int code = ((k * G + r) << 8) | (r << 1) | (back ? 1 : 0);
                    if (k == 0) {
                        // first slot: nothing to compare with yet; a lane that is not `in` keeps key = inf / idx = none and
                        // can never be the winner, so its point need not be masked
                        m.key = in ? key : m.key; m.idx = in ? code : m.idx; m.a = a; m.b = b;
                    } else {
                        bool take = in & (key < m.key);
                        m.key = take ? key : m.key; m.a = take ? a : m.a; m.b = take ? b : m.b;
                        m.idx = take ? code : m.idx;
                    }
                }
            group_argmin<G>(m);
            NPP_DEPEN_STEP(io, m, break)
        }
    } else {
#ifdef NPP_STAMPS
        st.acc[11] += 1;   // substeps that took the LDS fallback
#endif
        IO *slot = reinterpret_cast<IO *>(lv.spill + 12);
        static_assert(sizeof(IO) + 96 <= LDS_SPILL_BYTES, "spill row too small for the depenetration block");
        if (r == 0) *slot = io;
        depen_generic<G, IO>(TileRefs{lv.seg_start, lv.segs, lv.bounds}, r, gx0, gy0, gx1, gy1, slot);
        io = *slot;
    }
    n.x = io.x; n.y = io.y; n.vx = io.vx; n.vy = io.vy;
    fnsx = io.fnsx; fnsy = io.fnsy; cnsx = io.cnsx; cnsy = io.cnsy;
    n.fcount = io.fcount; n.ccount = io.ccount;
    crush_out(io, cr);
    return io.fcount + io.ccount - counts_before;
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
// neighbourhoods is scanned.  (All lanes of a group run this redundantly and write identical LDS words.)
DEV void think_mines(const Lv &lv, Nj &n, EntBits eb) {
    if (lv.n_think == 0) return;   // pcell is only read here, so it needs no upkeep on mine-free levels
    int ccx = cell_coord(n.x, 43), ccy = cell_coord(n.y, 24);
    int pcx = n.pcell / 25, pcy = n.pcell - pcx * 25;
    n.pcell = ccx * 25 + ccy;
    bool vt = valid_target(n.state);
    if (!vt && n.state != 6) return;
    int x0 = (ccx < pcx ? ccx : pcx) - 1, x1 = (ccx > pcx ? ccx : pcx) + 1;
    int y0 = (ccy < pcy ? ccy : pcy) - 1, y1 = (ccy > pcy ? ccy : pcy) + 1;
    x0 = x0 < 0 ? 0 : x0; x1 = x1 > 43 ? 43 : x1; y0 = y0 < 0 ? 0 : y0; y1 = y1 > 24 ? 24 : y1;
    for (int xc = x0; xc <= x1; xc++) {
        int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int i = i0; i < i1; i++) {
            if ((lv.ent_meta[i] & 15u) != EK_MINE) continue;
            uint32_t st = ent_get(eb, i);
            if (vt) {
                if (st == 1) { if (overlaps(lv.ent_x[i], lv.ent_y[i], 3.5 + NINJA_RADIUS, n.x, n.y)) ent_set(eb, i, 2); }
                else if (st == 2) { if (!overlaps(lv.ent_x[i], lv.ent_y[i], 4.5 + NINJA_RADIUS, n.x, n.y)) ent_set(eb, i, 0); }
            } else if (st == 2) {
                ent_set(eb, i, 1);
            }
        }
    }
}

// logical collisions of post_collision (ninja.py:388-420) over the 3x3 neighbourhood gathered x-major
// (physics.py:79-101); an exit door added to the grid by its switch this tick is not in the snapshot.
DEV void logical_collisions(const Lv &lv, Nj &n, EntBits eb) {
    int cx = cell_coord(n.x, 43), cy = cell_coord(n.y, 24);
    int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < 43 ? cx + 1 : 43, y0 = cy > 0 ? cy - 1 : 0, y1 = cy < 24 ? cy + 1 : 24;
    int pend0 = -1, pend1 = -1;
    for (int xc = x0; xc <= x1; xc++) {
        int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int ii = i0; ii < i1; ii++) {
            const int i = lv.perm[ii];   // list order inside a cell: map order, or entity_dic order after a fast reset
            uint32_t meta = lv.ent_meta[i];
            uint32_t kind = meta & 15u;
            uint32_t st = ent_get(eb, i);
            double ex = lv.ent_x[i], ey = lv.ent_y[i];
            if (kind == EK_MINE) {   // entity_toggle_mine.py:120-128
                if (valid_target(n.state) && st == 0 && overlaps(ex, ey, 4.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(eb, i, 1);
                    ninja_kill(n, 1);
                }
            } else if (st == 0) {
                continue;   // inactive (or exit door not yet in the grid)
            } else if (kind == EK_GOLD) {   // entity_gold.py:66-74
                if (n.state != 8 && overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) { n.gold += 1; ent_set(eb, i, 0); }
            } else if (kind == EK_EXIT) {   // entity_exit.py:66-74
                if (overlaps(ex, ey, 12.0 + NINJA_RADIUS, n.x, n.y)) ninja_win(n);
            } else if (kind == EK_SWITCH) { // entity_exit_switch.py:67-129
                if (overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(eb, i, 0);
                    int door = (int)((meta >> 8) & 0xffffu);
                    if (pend0 < 0) pend0 = door; else pend1 = door;
                }
            } else if (kind == EK_LOCKED) { // entity_door_locked.py:54-67
                if (overlaps(ex, ey, 5.0 + NINJA_RADIUS, n.x, n.y)) { n.doors += 1; ent_set(eb, i, 0); }
            }
        }
    }
    if (pend0 >= 0) ent_set(eb, pend0, 1);
    if (pend1 >= 0) ent_set(eb, pend1, 1);
}

// sum the wall-probe terms of one pass in lane (= query) order (ninja.py:441)
template <int G> DEV void ordered_add(double &acc, double term) {
    if constexpr (G == 1) {
        acc += term;
    } else {
        const int glane0 = (threadIdx.x & 63) & ~(G - 1);
        unsigned long long bal = __ballot(term != 0) >> glane0;
        if constexpr (G < 64) bal &= (1ull << G) - 1;
        while (bal) {
            int k = __builtin_ctzll(bal);
            bal &= bal - 1;
            acc += __shfl(term, glane0 + k, 64);
        }
    }
}

DEV double wall_term(double px, double py, double a, double b, double rad) {
    double dx = px - a, dy = py - b;
    if (dabs(dy) < 0.00001) {
        double dist = dsqrt(sq(dx) + sq(dy));
        if (0 < dist && dist <= rad) return dx / dist;
    }
    return 0;
}

template <int G>
__device__ __noinline__ double wall_probe_generic(TileRefs lv, int r, double px, double py, double qx0, double qy0,
                                                  double qx1, double qy1, double wall_normal) {
    const double rad = NINJA_RADIUS + 0.1;
    const int c0x = cell_coord(qx0, 43), c1x = cell_coord(qx1, 43), c0y = cell_coord(qy0, 24), c1y = cell_coord(qy1, 24);
    for (int xc = c0x; xc <= c1x; xc++) {
        int i0 = lv.seg_start[xc * 25 + c0y], i1 = lv.seg_start[xc * 25 + c1y + 1];
        for (int ib = i0; ib < i1; ib += G) {   // group-uniform trip count
            int i = ib + r;
            double term = 0;
            if (i < i1) {
                uint32_t s = lv.segs[i];
                int yc = s >> 11;
                if (cell_passes(lv.bounds[xc * 25 + yc], xc, yc, qx0, qy0, qx1, qy1)) {
                    double a, b;
                    seg_closest(s, xc, yc, px, py, a, b);
                    term = wall_term(px, py, a, b, rad);
                }
            }
            ordered_add<G>(wall_normal, term);
        }
    }
    return wall_normal;
}

#include "npp_zoo.hpp"

// Ninja.post_collision (ninja.py:381-537)
template <int G, int K, bool ZOO>
DEV void post_collision(const Lv &lv, const Zoo &z, int r, Nj &n, const Cand<K> &cd, EntBits eb, double fnsx, double fnsy,
                        double cnsx, double cnsy, double xold, double yold, const ZTick &zt) {
    // wall probe (ninja.py:424-441): entity contributions first (ninja.py:397-420), then tile terms in query order
    double wall_normal = 0;
    if (ZOO && z.on) wall_normal = logical_collisions_zoo<G>(lv, z, r, n, eb, xold, yold);
    else logical_collisions(lv, n, eb);
    const double rad = NINJA_RADIUS + 0.1;
    double qx0 = n.x - rad, qy0 = n.y - rad, qx1 = n.x + rad, qy1 = n.y + rad;
    const QBox q = make_qbox(qx0, qy0, qx1, qy1);
    if (cand_covers(cd, q)) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            double term = 0;
            if (cand_in_box(cd, k, q)) {
                double a, b;
                if ((cd.s[k] & 1u) == 0) cand_closest_lin(cd, k, n.x, n.y, a, b);
                else cand_closest_arc(cd.s[k], cd.x1[k], cd.y1[k], cd.x2[k], cd.y2[k], n.x, n.y, a, b);
                term = wall_term(n.x, n.y, a, b, rad);
            }
            if (group_any<G>(term != 0)) ordered_add<G>(wall_normal, term);
        }
    } else {
        wall_normal = wall_probe_generic<G>(TileRefs{lv.seg_start, lv.segs, lv.bounds}, r, n.x, n.y, qx0, qy0, qx1, qy1, wall_normal);
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
    if (ZOO && z.on) zoo_crush_check(n, zt);
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

// Ninja.think (ninja.py:849-1059); lp = launch pad boost direction (xlp/ylp_boost_normalized) on zoo levels, else NULL
DEV void ninja_think(Nj &n, const double *lp) {
    if (n.state != n.pstate) { n.scf = 0; n.pstate = n.state; } else { n.scf += 1; }
    bool new_jump_check = n.jump ? (n.jio == 0) : false;
    n.jio = n.jump;
    n.lbuf = (-1 < n.lbuf && n.lbuf < 3) ? n.lbuf + 1 : -1;
    const bool in_lp_buffer = -1 < n.lbuf && n.lbuf < 4;
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
        if (lp && in_lp_buffer && new_jump_check) {   // lp_jump (ninja.py:610-626, 1043-1045)
            n.fbuf = -1; n.wbuf = -1; n.jbuf = -1; n.lbuf = -1;
            double boost_scalar = 2 * dabs(lp[0]) + 2;
            if (boost_scalar == 2) boost_scalar = 1.7;
            n.vx += lp[0] * boost_scalar * TWO_THIRDS;
            n.vy += lp[1] * boost_scalar * TWO_THIRDS;
            return;
        }
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
template <int G, bool ZOO, int V>
DEV void sim_tick(const Lv &lv, const Zoo &z, int r, Nj &n, EntBits eb, int hor, int jump, int n_ent STAMP_ARG) {
    constexpr int K = VariantK<G, ZOO, V>::value;
    STAMP_INIT;
    // the inputs are the same for the four ticks of a Gymnasium step: without this the compiler hoists products such as
    // GROUND_ACCEL * hor out of the tick loop and carries them (in scratch memory, as it turned out) through every tick
    asm volatile("" : "+v"(hor), "+v"(jump));
    n.frame += 1;
    n.hor = hor;
    n.jump = jump;
    const bool zoo = ZOO && z.on;
    if (zoo) zoo_entities_tick<G>(lv, z, r, n, eb, n_ent);
    else think_mines(lv, n, eb);
    STAMP(2);
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
        ZTick zt;
        zt.xcr = 0; zt.ycr = 0; zt.clen = 0; zt.crushable = 0; zt.gcx = 0; zt.gcy = 0; zt.phys_near = false;
        if (zoo) zoo_pre_collision<G>(lv, z, r, n, zt);
        // candidate segments for every query of this tick: cells touched by the path inflated by the largest
        // query radius (10.1) plus slack for depenetration drift
        // Eight ninja doubles are dead weight from here to post_collision (nothing in the substeps reads the previous speeds, the
        // normalised floor / ceiling normals or the spatial-context anchor): parked in the env's LDS spill row instead of letting
        // the register allocator push them to scratch memory (every lane of the group holds the same values; lane 0 writes, all
        // read back).  Plain kernels only: the zoo's physical collisions read some of them.
        if constexpr (!ZOO) {
            if (r == 0) {
                double *sp = lv.spill;
                sp[0] = n.vxo; sp[1] = n.vyo; sp[2] = n.fnx; sp[3] = n.fny; sp[4] = n.cnx; sp[5] = n.cny;
                int *ip = reinterpret_cast<int *>(sp + 8);   // and six counters nobody looks at before post_collision / think
                ip[0] = n.fair; ip[1] = n.scf; ip[2] = n.frame; ip[3] = n.gold; ip[4] = n.doors; ip[5] = n.fastord;
            }
            asm volatile("" ::: "memory");
        }
        Cand<K> cd;
        {
            const double pad = NINJA_RADIUS + 2.2;
            cand_gather<G, K>(lv, r, (xold < n.x ? xold : n.x) - pad, (yold < n.y ? yold : n.y) - pad,
                              (xold > n.x ? xold : n.x) + pad, (yold > n.y ? yold : n.y) + pad, cd);
        }
        STAMP(3);
        // 4 substeps (nsim.py:263-267).  Without a physically collidable entity nearby collide_vs_objects does
        // nothing, and a substep that applies no depenetration and leaves the position bit-identical has the same
        // inputs as the next one, so the remaining substeps are no-ops and are skipped.
        Crush cr{0, 0, 0};
        for (int k = 0; k < 4; k++) {
            const double xb = n.x, yb = n.y;
            if (zoo && zt.phys_near) {
                cr.xcr = zt.xcr; cr.ycr = zt.ycr; cr.clen = zt.clen;
                collide_vs_objects<G>(lv, z, r, n, zt, xold, yold, fnsx, fnsy, cnsx, cnsy);
                cr.xcr = zt.xcr; cr.ycr = zt.ycr; cr.clen = zt.clen;
            }
            const int applied = collide_vs_tiles<G, K, ZOO>(lv, r, n, cd, xold, yold, fnsx, fnsy, cnsx, cnsy, cr STAMP_PASS);
            n.work += applied;
            zt.xcr = cr.xcr; zt.ycr = cr.ycr; zt.clen = cr.clen;
            if (!(zoo && zt.phys_near) && applied == 0 && n.x == xb && n.y == yb) break;
        }
        STAMP(4);
        if constexpr (!ZOO) {
            asm volatile("" ::: "memory");
            const double *sp = lv.spill;
            n.vxo = sp[0]; n.vyo = sp[1]; n.fnx = sp[2]; n.fny = sp[3]; n.cnx = sp[4]; n.cny = sp[5];
            const int *ip = reinterpret_cast<const int *>(sp + 8);
            n.fair = ip[0]; n.scf = ip[1]; n.frame = ip[2]; n.gold = ip[3]; n.doors = ip[4]; n.fastord = ip[5];
        }
        post_collision<G, K, ZOO>(lv, z, r, n, cd, eb, fnsx, fnsy, cnsx, cnsy, xold, yold, zt);
        STAMP(5);
    }
    ninja_think(n, zoo ? z.blk : nullptr);
    STAMP(6);
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

// spatial_context (112 f32): NppEnvironment._compute_spatial_context (gym_environment/npp_environment.py:2318-2360):
//   [0, 64)   compute_local_tile_grid (gym_environment/spatial_context.py:113-176): 8 x 8 tile categories around
//             int(x // 24), int(y // 24), read from the INNER 23 x 42 tile array with WORLD tile coordinates (the
//             reference's off-by-one, base_environment.py:3346-3354); out of bounds = solid
//   [64, 112) compute_mine_overlay_from_entities (:309-367 -> :370-508): 8 nearest mines (entity_dic[1] then
//             entity_dic[21], stable sort by distance) x (dx/1056, dy/600, state code, radius code, velocity dot,
//             distance rate), recomputed only when the ninja is >= 12 px from where it was last computed.
// The lanes of a group split the copies and the nearest-8 selection.
template <int G>
DEV void write_spatial_context(const KernelArgs &a, const LevelHdr &H, Nj &n, EntBits eb, int env, int r, bool valid) {
    if (!valid) return;
    float *out = a.out.spatial_context + (size_t)env * 112;
    float *cache = a.sc_cache + (size_t)env * 48;
    const uint8_t *tiles = a.blob + H.off_tiles;
    const int col = floor12(n.x) >> 1, row = floor12(n.y) >> 1;   // int(x // 24)
    for (int k = r; k < 64; k += G) {
        int rr = row - 4 + (k >> 3), cc = col - 4 + (k & 7);
        int t = 1;
        if (rr >= 0 && rr < 23 && cc >= 0 && cc < 42) {
            t = tiles[(cc + 1) * 25 + (rr + 1)];
            t = t > 37 ? 37 : t;
        }
        // categories 0 empty, 1 solid, 2 half tiles, 3 slopes, 4 curved (spatial_context.py:42-92), / 4
        int cat = t == 0 ? 0 : (t == 1 ? 1 : (t <= 5 ? 2 : (t <= 9 ? 3 : (t <= 17 ? 4 : (t <= 33 ? 3 : 0)))));
        out[k] = (float)cat * 0.25f;
    }
    bool hit = false;
    if (n.scvalid) {
        const size_t N = (size_t)a.n;
        double dx = n.x - a.f64[F_SCX * N + env], dy = n.y - a.f64[F_SCY * N + env];
        hit = dx * dx + dy * dy < 144.0;
    }
    if (hit) {
        for (int k = r; k < 48; k += G) out[64 + k] = cache[k];
        return;
    }
    // Nearest eight, lanes of the group in parallel (round 2; the redundant serial scan of up to 130 mines with three dependent
    // loads, an fp64 square root and an 8-deep insertion each was ~50 us of a 336 us step on the door levels): lane r scans
    // draw-order entries r, r + G, ... into its own stable top-8 (distance, then entry index: `tag` = index << 15 | slot), then
    // eight rounds of "smallest (distance, tag) among the lanes' heads" merge them -- the same total order as one serial scan.
    double bd[8];
    int bt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { bd[k] = __builtin_inf(); bt[k] = 0x7fffffff; }
    const uint16_t *order = reinterpret_cast<const uint16_t *>(a.blob + H.off_raster);
    const double *ex = reinterpret_cast<const double *>(a.blob + H.off_ent_x);
    const double *ey = reinterpret_cast<const double *>(a.blob + H.off_ent_y);
    const uint32_t *meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_ent_meta);
    for (uint32_t i = r; i < H.n_ent + H.n_mov; i += G) {
        int slot = order[i];   // draw order = type 1 mines in map order, ..., type 21 mines in map order
        if (slot & 0x8000) continue;   // a mover reference (npp_level.hpp: raster_order)
        if ((meta[slot] & 15u) != EK_MINE) continue;
        double dx = ex[slot] - n.x, dy = ey[slot] - n.y;
        double d = dsqrt(dx * dx + dy * dy);
        if (!(d < bd[7])) continue;
        int tag = (int)((i << 15) | (uint32_t)slot);
        bool sw = false;
#pragma unroll
        for (int k = 0; k < 8; k++) {   // stable insertion: an equal distance stays behind earlier entries, and
            sw = sw | (d < bd[k]);      // once placed, everything after it shifts down by one unconditionally
            double td = bd[k]; int tt = bt[k];
            bd[k] = sw ? d : td; bt[k] = sw ? tag : tt;
            d = sw ? td : d; tag = sw ? tt : tag;
        }
    }
    constexpr int OWN = G >= 8 ? 1 : 8 / G;   // results per lane: result k belongs to lane k % G
    double od[OWN];
    int os[OWN];
#pragma unroll
    for (int j = 0; j < OWN; j++) { od[j] = 0; os[j] = -1; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double kmin = group_min<G>(bd[0]);
        const int cand = (bd[0] == kmin) ? bt[0] : 0x7fffffff;   // empty heads carry tag 0x7fffffff at distance inf
        const int tmin = group_min_i<G>(cand);
        if (bt[0] == tmin && tmin != 0x7fffffff) {   // this lane's head won: pop it
#pragma unroll
            for (int q = 0; q < 7; q++) { bd[q] = bd[q + 1]; bt[q] = bt[q + 1]; }
            bd[7] = __builtin_inf(); bt[7] = 0x7fffffff;
        }
        if (r == k % G) { od[k / G < OWN ? k / G : 0] = kmin; os[k / G < OWN ? k / G : 0] = tmin == 0x7fffffff ? -1 : (tmin & 0x7fff); }
    }
#pragma unroll
    for (int j = 0; j < OWN; j++) {
        const int k = j * G + r;
        if (k >= 8) continue;
        float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (os[j] >= 0) {
            double dx = ex[os[j]] - n.x, dy = ey[os[j]] - n.y, dist = od[j];
            double vd = 0.0, dr = 0.0;
            if (dist > 1e-6) {
                double dirx = dx / dist, diry = dy / dist;
                vd = (n.vx * dirx + n.vy * diry) / MAX_HOR_SPEED;
                dr = -vd;
            }
            uint32_t st = ent_get(eb, os[j]);
            f[0] = (float)clampd(dx / 1056.0, -1.0, 1.0);
            f[1] = (float)clampd(dy / 600.0, -1.0, 1.0);
            f[2] = st == 1 ? 1.0f : (st == 2 ? 0.0f : -1.0f);
            f[3] = (float)(st == 1 ? 3.5 / 5.0 : (st == 2 ? 4.5 / 5.0 : 4.0 / 5.0));
            f[4] = (float)clampd(vd, -1.0, 1.0);
            f[5] = (float)clampd(dr, -1.0, 1.0);
        }
#pragma unroll
        for (int jj = 0; jj < 6; jj++) { cache[6 * k + jj] = f[jj]; out[64 + 6 * k + jj] = f[jj]; }
    }
    n.scvalid = 1;
    if (r == 0) { a.f64[F_SCX * (size_t)a.n + env] = n.x; a.f64[F_SCY * (size_t)a.n + env] = n.y; }
}

// contiguous workgroup store of per-env rows of `width` 4-byte words staged at stage[env_in_block * width + k]
DEV void block_store_rows(const uint32_t *stage, uint32_t *dst_block, int width, int n_valid) {
    int total = n_valid * width;
    for (int j = threadIdx.x; j < total; j += blockDim.x) dst_block[j] = stage[j];
}

// Episode reset inside the step kernel (vector-env auto-reset): Simulator.reset -- every entity re-created -- or, under
// NPP_FLAG_FAST_RESET, Simulator.fast_reset (nsim.py:78-140), which is what NppEnvironment.reset does on the same level.
// Executed by all lanes of the env's group on identical data (the LDS updates are idempotent).
template <bool ZOO>
DEV void episode_reset(const KernelArgs &a, const LevelHdr &H, Lv &lv, Zoo &z, Nj &n, EntBits eb, int nw, int r, int G) {
    const bool fast = a.fast_reset && !(n.fastord & 2);   // the first reset after a level assignment is a Simulator.reset
    const int episode = next_episode(n.fastord);
    lv.spawn_x = H.spawn_x; lv.spawn_y = H.spawn_y;
    spawn_state(lv, n);
    n.fastord = episode;
    if (fast) {
        n.fastord |= 1;
        lv.perm = reinterpret_cast<const uint16_t *>(a.blob + H.off_ent_perm);
        const uint32_t *keep = reinterpret_cast<const uint32_t *>(a.blob + H.off_keep_words);
        for (int w = 0; w < nw; w++) {
            const uint32_t k = keep[w];
            eb.w[w * eb.stride] = (eb.w[w * eb.stride] & k) | (lv.init_words[w] & ~k);
        }
        if constexpr (ZOO) {
            z.ent_ord = reinterpret_cast<const uint16_t *>(a.blob + H.off_ent_rank);
            if (z.on) zoo_fast_reset_block(z, r, G);
        }
    } else {
        lv.perm = reinterpret_cast<const uint16_t *>(a.blob + H.off_ent_ident);
        for (int w = 0; w < nw; w++) eb.w[w * eb.stride] = lv.init_words[w];
        if constexpr (ZOO) {
            z.ent_ord = z.ent_seq;
            if (z.on) zoo_init_block(z, r, G, false);
        }
    }
}

// G lanes per env; EPW = 64 / G envs per wavefront; blockDim.x / 64 wavefronts per workgroup
template <int G, bool LDS_LEVEL, bool ZOO, bool MANY, int V>
DEV void run(const KernelArgs &a, unsigned char *smem) {
    constexpr int EPW = WAVE / G;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int r = lane & (G - 1);          // rank inside the env's lane group
    const int epb = EPW * (blockDim.x >> 6);   // envs per workgroup
    const int eib = wave * EPW + lane / G;     // env index inside the workgroup
    // heavy-first (npp_step): the workgroups whose envs iterated longest in an earlier launch are dispatched first, so that the
    // chain that decides the launch's duration starts at time zero instead of in the second residency round
    const int blk = a.wg_order ? (int)a.wg_order[blockIdx.x + a.wg_first] : (int)blockIdx.x + a.wg_first;
    const unsigned long long wg_t0 = a.wg_cost ? __builtin_amdgcn_s_memtime() : 0ull;
    const int env0 = blk * epb;
    const int env = env0 + eib;
    const bool valid = env < a.n;
    const int e = valid ? env : a.n - 1;
    const int n_valid = (a.n - env0) < epb ? (a.n - env0) : epb;

    uint32_t *ew = reinterpret_cast<uint32_t *>(smem + (LDS_LEVEL ? a.lds_hot_cap : 0u));
    uint32_t *stage = ew + (size_t)a.n_words_max * epb;
    EntBits eb;
    eb.w = ew + eib;
    eb.stride = epb;

    // LDS_LEVEL: the host guarantees that every env of the workgroup plays ONE level, so the level header and everything read
    // from it (table pointers, spawn / switch / door coordinates) is wavefront-uniform: scalar loads into SGPRs instead of ~36
    // VGPRs per lane that stay live for the whole kernel
    const int lvl = LDS_LEVEL ? __builtin_amdgcn_readfirstlane(a.env_level[e]) : a.env_level[e];
    const LevelHdr &H = a.hdr[lvl];
    Lv lv;
    const unsigned char *hot = a.blob + H.off_hot;
    if (LDS_LEVEL) {
        // every env of this workgroup plays level `lvl`: stage its collision table into LDS with 16-byte loads
        const uint4 *src = reinterpret_cast<const uint4 *>(hot);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const int nvec = (int)(H.hot_bytes >> 4);
        for (int i = tid; i < nvec; i += blockDim.x) dst[i] = src[i];
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
    // spawn / switch / door coordinates: the plain kernels read them from the level header where they are used (episode reset,
    // observation rows) -- loaded here they sat in 12 VGPRs through every tick; the zoo kernels may override them per env
    lv.spawn_x = 0; lv.spawn_y = 0; lv.sw_x = 0; lv.sw_y = 0; lv.door_x = 0; lv.door_y = 0;
    if constexpr (ZOO) { lv.sw_x = H.sw_x; lv.sw_y = H.sw_y; lv.door_x = H.door_x; lv.door_y = H.door_y; }
    lv.spill = reinterpret_cast<double *>(smem + lds_spill_offset(LDS_LEVEL ? a.lds_hot_cap : 0u, a.n_words_max, epb)) +
               (size_t)eib * (LDS_SPILL_BYTES / 8);

    // entity zoo: this env's block and a private copy of its level's grid edges live in LDS behind the staging rows
    Zoo z;
    z.on = false; z.blk = nullptr; z.edges = nullptr; z.n_mov = 0; z.n_door = 0; z.n_balls = 0;
    if constexpr (ZOO) {
        const size_t zoff = lds_zoo_offset(LDS_LEVEL ? a.lds_hot_cap : 0u, a.n_words_max, epb);
        z.blk = reinterpret_cast<double *>(smem + zoff) + (size_t)eib * a.zoo_words;
        z.edges = reinterpret_cast<uint32_t *>(smem + zoff + (size_t)epb * a.zoo_words * 8) + (size_t)eib * (2 * EDGE_WORDS_D);
        const uint32_t ovr = reinterpret_cast<const uint32_t *>(a.zoo + (size_t)e * a.zoo_words + 3)[0];
        z.on = H.has_zoo != 0 || (ovr & (ZOO_OVR_SWITCH | ZOO_OVR_DOOR)) != 0;
        z.obs_switch = H.obs_switch; z.obs_door = H.obs_door; z.vsw_cell = -1; z.vdoor_cell = -1;
        z.ent_seq = reinterpret_cast<const uint16_t *>(a.blob + H.off_ent_seq);
        z.ent_ord = z.ent_seq;   // chosen again once the env's state is loaded (fastord)
        z.mov_rank = reinterpret_cast<const uint16_t *>(a.blob + H.off_mov_rank);
        z.ent_cell = reinterpret_cast<const uint16_t *>(a.blob + H.off_ent_cell);
        z.mov_meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_mov_meta);
        z.mov_x0 = reinterpret_cast<const double *>(a.blob + H.off_mov_x0);
        z.mov_y0 = reinterpret_cast<const double *>(a.blob + H.off_mov_y0);
        z.door_tab = reinterpret_cast<const uint32_t *>(a.blob + H.off_door_tab);
        z.n_mov = (int)H.n_mov; z.n_door = (int)H.n_zdoor; z.door_words = (a.zoo_doors + 1) / 2;
        z.n_balls = (int)H.n_balls; z.n_created = (int)H.n_created; z.db_count = H.db_count; z.ball_first = (int)H.ball_first;
        if (z.on) {
            const int used = ZOO_HEAD + z.door_words + ZOO_MOV_WORDS * z.n_mov;
            const double *src = a.zoo + (size_t)e * a.zoo_words;
            for (int k = r; k < used; k += G) z.blk[k] = src[k];
            const uint32_t *esrc = reinterpret_cast<const uint32_t *>(a.blob + H.off_edges);
            for (int k = r; k < 2 * EDGE_WORDS_D; k += G) z.edges[k] = esrc[k];
            // npp_set_entity_pos: where the exit switch / door sit now (intermediate_goal_manager.py:698)
            if ((ovr & ZOO_OVR_SWITCH) && H.obs_switch >= 0) {
                const double *src4 = src + 4;
                lv.sw_x = src4[0]; lv.sw_y = src4[1];
                const int c = pos_cell(src4[0], src4[1]);
                if (c != (int)z.ent_cell[H.obs_switch]) z.vsw_cell = c;
            }
            if ((ovr & ZOO_OVR_DOOR) && H.obs_door >= 0) {
                lv.door_x = src[6]; lv.door_y = src[7];
                z.vdoor_cell = pos_cell(src[6], src[7]);
            }
        }
    }

#ifdef NPP_STAMPS
    Stamps st;
    for (int i = 0; i < N_STAMP; i++) st.acc[i] = 0;
#endif
    STAMP_INIT;
    Nj n;
    load_state(a, e, n);
    // list order inside a cell: map order after Simulator.reset, entity_dic order after a fast reset
    lv.perm = reinterpret_cast<const uint16_t *>(a.blob + ((n.fastord & 1) ? H.off_ent_perm : H.off_ent_ident));
    if constexpr (ZOO) {
        if (n.fastord & 1) z.ent_ord = reinterpret_cast<const uint16_t *>(a.blob + H.off_ent_rank);
    }
    const int nw = (int)lv.n_words;
    if (r == 0)
        for (int w = 0; w < nw; w++) eb.w[w * eb.stride] = a.ent_bits[(size_t)w * a.n + e];
    __syncthreads();

    STAMP(0);
    uint32_t flags = 0;
    int executed = 0;
    float reward = 0.f;
    bool done = false;
    const bool stepping = a.mode == 0 && a.n_ticks > 0;
    const bool writer = valid && r == 0;
    // npp_step_many: several Gymnasium steps in one launch (open-loop action sequences: checkpoint replay, rollouts of a
    // fixed plan).  Wavefronts walk their steps independently, so nobody waits for the slowest env of every single step.
    // (its own instantiation, MANY: the single-step kernels keep a constant trip count of one)
    const int n_steps = MANY ? (a.n_steps > 1 ? a.n_steps : 1) : 1;
#pragma nounroll
    for (int sidx = 0; sidx < n_steps; sidx++) {
        flags = 0; executed = 0; reward = 0.f;
        n.work = 0;
        const bool had_switch = lv.obs_switch >= 0 && ent_get(eb, lv.obs_switch) == 0;
        {
            // mode 0: NppEnvironment.step frame-skip loop (base_environment.py:524-609), action table :366-402; an env that
            //         is already terminal (no auto-reset) is not stepped again until the caller resets it
            // mode 1: NPlayHeadless.tick driven by replay bytes (replay/replay_executor.py:61-84), never stops early
            const bool gym = a.mode == 0;
            const int act = (gym && a.n_ticks > 0) ? a.inputs[(size_t)sidx * a.n + e] : 0;
            int hor = (act == 1 || act == 4) ? -1 : ((act == 2 || act == 5) ? 1 : 0);
            int jump = act >= 3 ? 1 : 0;
            bool live = valid && !(gym && (n.state == 8 || n.state == 6 || n.state == 7));
            for (int t = 0; t < a.n_ticks; t++) {
                if (!gym) {
                    const int b = a.inputs[(size_t)t * a.n + e];
                    const int l = (b >> 2) & 1, rr = (b >> 1) & 1;
                    hor = (l && rr) ? 0 : (l ? -1 : (rr ? 1 : 0));
                    jump = b & 1;
                }
                if (live) {
                    sim_tick<G, ZOO, V>(lv, z, r, n, eb, hor, jump, (int)H.n_ent STAMP_PASS);
                    executed++;
                    if (gym && (n.state == 8 || n.state == 6 || n.state == 7)) live = false;
                }
                if (!__any(live)) break;
            }
        }

        STAMP(1);
        const bool sw_now = lv.obs_switch >= 0 ? ent_get(eb, lv.obs_switch) == 0 : true;   // nplay_headless.py:566-576
        if (n.state == 8) flags |= 1u;
        if (n.state == 6 || n.state == 7) flags |= 2u;
        if (sw_now) flags |= 4u;
        if (n.cause == 1) flags |= 16u;
        if (n.cause == 2) flags |= 32u;
        done = (flags & 3u) != 0;
        // (the limit is read where it is used, here and in the observation rows, rather than held in a register through the ticks)
        if (stepping && !done && n.frame >= a.trunc_limit[e]) { flags |= 8u; done = true; }   // truncation_checker.py:46-77
        // sparse terminal reward: completion 200, switch 100, death -30, scaled by 0.1 (reward_constants.py:71,115,148,212)
        if (flags & 1u) reward += 20.f;
        if (flags & 2u) reward -= 3.f;
        if (sw_now && !had_switch && lv.obs_switch >= 0) reward += 10.f;

        if (writer) {
            const size_t o = (size_t)sidx * a.n + env;
            if (a.out.flags) a.out.flags[o] = (uint8_t)flags;
            if (a.out.reward) a.out.reward[o] = reward;
            if (a.out.frames) a.out.frames[o] = (uint16_t)executed;
            if (a.out.work) a.out.work[o] = (uint16_t)(n.work > 0xffff ? 0xffff : n.work);
        }
        if (MANY && sidx + 1 < n_steps && a.autoreset && stepping && done)   // intermediate steps reset on the spot
            episode_reset<ZOO>(a, H, lv, z, n, eb, nw, r, G);
    }

    const bool do_reset = a.autoreset && stepping && done;
    // game_state rows are assembled in LDS and stored as contiguous workgroup writes.  Pass 0 (only when a
    // terminal-observation buffer was given) writes the pre-reset state of the envs that are being reset; pass 1
    // resets those envs and writes the observation every env returns.
#pragma nounroll
    for (int pass = (a.out.terminal_state ? 0 : 1); pass < 2; pass++) {
        if (pass == 1 && do_reset) episode_reset<ZOO>(a, H, lv, z, n, eb, nw, r, G);
        float *gdst = pass == 0 ? a.out.terminal_state : a.out.game_state;
        if (gdst) {
            if (r == 0) write_game_state(n, a.trunc_limit[e], reinterpret_cast<float *>(stage) + eib * 41);
            __syncthreads();
            if (pass == 1) {
                block_store_rows(stage, reinterpret_cast<uint32_t *>(gdst + (size_t)env0 * 41), 41, n_valid);
            } else if (do_reset && writer) {
                const uint32_t *row = stage + eib * 41;
                uint32_t *dst = reinterpret_cast<uint32_t *>(gdst + (size_t)env * 41);
                for (int k = 0; k < 41; k++) dst[k] = row[k];
            }
            __syncthreads();
        }
    }
    if (a.out.entity_pos) {
        if (r == 0) {
            float *row = reinterpret_cast<float *>(stage) + eib * 6;
            row[0] = (float)(n.x / 1056.0); row[1] = (float)(n.y / 600.0);
            row[2] = (float)((ZOO ? lv.sw_x : H.sw_x) / 1056.0); row[3] = (float)((ZOO ? lv.sw_y : H.sw_y) / 600.0);
            row[4] = (float)((ZOO ? lv.door_x : H.door_x) / 1056.0); row[5] = (float)((ZOO ? lv.door_y : H.door_y) / 600.0);
        }
        __syncthreads();
        block_store_rows(stage, reinterpret_cast<uint32_t *>(a.out.entity_pos + (size_t)env0 * 6), 6, n_valid);
        __syncthreads();
    }
    if (a.out.positions && writer) {   // pass-through scalars player_x/y, switch_x/y, exit_door_x/y (unrounded)
        double *row = a.out.positions + (size_t)env * 6;
        row[0] = n.x; row[1] = n.y;
        row[2] = ZOO ? lv.sw_x : H.sw_x; row[3] = ZOO ? lv.sw_y : H.sw_y; row[4] = ZOO ? lv.door_x : H.door_x; row[5] = ZOO ? lv.door_y : H.door_y;
    }
    if (a.out.spatial_context) write_spatial_context<G>(a, H, n, eb, env, r, valid);
    if (a.out.action_mask && writer) {
        uint32_t m = action_mask_bits(n);
        int8_t *row = a.out.action_mask + (size_t)env * 6;
        for (int k = 0; k < 6; k++) row[k] = (int8_t)((m >> k) & 1u);
    }

    STAMP(7);
    if (writer) {
        store_state(a, env, n);
        for (int w = 0; w < nw; w++) a.ent_bits[(size_t)w * a.n + env] = eb.w[w * eb.stride];
    }
    if constexpr (ZOO) {
        if (valid && z.on) {
            const int used = ZOO_HEAD + z.door_words + ZOO_MOV_WORDS * z.n_mov;
            double *dst = a.zoo + (size_t)env * a.zoo_words;
            for (int k = r; k < used; k += G) dst[k] = z.blk[k];
        }
    }
    STAMP(8);
    if (a.wg_cost && lane == 0) atomicMax(&a.wg_cost[blk], (uint32_t)(__builtin_amdgcn_s_memtime() - wg_t0));   // slowest wavefront of the block
#ifdef NPP_STAMPS
    if (lane == 0) {
        unsigned gw = blockIdx.x * (blockDim.x >> 6) + wave;
        if (gw < 16384)
            for (int i = 0; i < N_STAMP; i++) g_stamps[gw * N_STAMP + i] += st.acc[i];
    }
#endif
}

// LDS_LEVEL is chosen by the host: true when every workgroup's envs play one level that fits the LDS budget
// (the host knows the env -> level assignment), false otherwise (tables are read through L1/L2).
// Register budget: the second launch-bound argument is the minimum number of wavefronts per SIMD.  Measured on MI355X:
// * mid-round 2 (profiles/r02_occupancy_ab.txt, tools/occupancy_ab.py), before the heavy-first workgroup order: capping the kernel at
//   256 unified registers (2 wavefronts per SIMD, all 2048 wavefronts of an 8192-env launch resident at once) changed NOTHING at 8192
//   envs on the straggler-heavy first 600 steps (160.7 vs 160.5 us) and LOST at the batch sizes where throughput matters (32 768 envs,
//   G = 8: 699 vs 404 us; 65 536 envs, G = 8 / 4: 810 vs 555 us) because the G <= 8 instantiations then spill ~360 VGPRs to scratch;
// * end of round 2, with the heavy-first order and in the bench's steady state: the G = 16 kernels capped at 2 wavefronts per SIMD
//   (196 VGPRs spilled outside the depenetration loop, 648 B of scratch per lane) take the headline from 68.2 to 75.5 M env-steps/s
//   (launch mean 120 -> 108 us, p50 111 -> 96: no second residency round any more; p95 221 -> 246: the slowest chain shares its SIMD for
//   the first ~30 us) and the mine levels from 80.0 to 97.7 M (102 -> 84 us), and cost the door levels 5 % (241 -> 253 us: every launch
//   there is one long chain).
// So: G >= 16 (up to 16 384 envs) is built for 2 wavefronts per SIMD, G <= 8 and the zoo kernels keep the allocator's free hand
// (1 wavefront per SIMD, AGPRs as spill space).  -DNPP_MIN_WAVES=1 rebuilds the uncapped G >= 16 kernels for A/B runs.
#ifndef NPP_ZOO_WAVES   // the zoo kernels (21.8 k instructions, 389 registers): 1 = uncapped; -DNPP_ZOO_WAVES=2 is the A/B build
#define NPP_ZOO_WAVES 1
#endif
#ifndef NPP_MIN_WAVES
#define NPP_MIN_WAVES 2
#endif

template <int G, bool LDS_LEVEL, bool ZOO, bool MANY, int V>
__global__ __launch_bounds__(256, (ZOO ? NPP_ZOO_WAVES : ((G < 16 || V == 2) ? 1 : NPP_MIN_WAVES))) void npp_step_kernel(KernelArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    run<G, LDS_LEVEL, ZOO, MANY, V>(a, smem);
}

// Simulator.reset / fast_reset (nsim.py:62-140) for masked envs
__global__ __launch_bounds__(64) void npp_reset_kernel(KernelArgs a) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= a.n) return;
    if (a.reset_mask && a.reset_mask[env] == 0) return;
    const LevelHdr &H = a.hdr[a.env_level[env]];
    Lv lv;
    lv.spawn_x = H.spawn_x; lv.spawn_y = H.spawn_y;
    Nj n;
    spawn_state(lv, n);
    // Simulator.fast_reset (nsim.py:78-140).  a.reset_auto (npp_reset / npp_reset_ex mode 0 under NPP_FLAG_FAST_RESET): envs that
    // have not had a Simulator.reset since their level was assigned get a full one, like the reference env's first reset()
    const uint32_t oldE = a.u32[U_E * (size_t)a.n + env];
    const bool first = ((oldE >> 18) & 1u) != 0;
    const bool fast = a.fast_reset != 0 && !(a.reset_auto && first);
    n.fastord = next_episode((int)((oldE >> 17) & 0x7fff)) | (a.reset_fresh ? 2 : (fast ? (1 | (first ? 2 : 0)) : 0));
    store_state(a, env, n);
    const uint32_t *init = reinterpret_cast<const uint32_t *>(a.blob + H.off_init_words);
    const uint32_t *keep = reinterpret_cast<const uint32_t *>(a.blob + H.off_keep_words);
    for (uint32_t w = 0; w < H.n_words; w++) {
        const size_t at = (size_t)w * a.n + env;
        a.ent_bits[at] = fast ? ((a.ent_bits[at] & keep[w]) | (init[w] & ~keep[w])) : init[w];
    }
    if (a.zoo) {
        Zoo z;
        z.blk = a.zoo + (size_t)env * a.zoo_words;
        z.edges = nullptr;
        z.mov_meta = reinterpret_cast<const uint32_t *>(a.blob + H.off_mov_meta);
        z.mov_x0 = reinterpret_cast<const double *>(a.blob + H.off_mov_x0);
        z.mov_y0 = reinterpret_cast<const double *>(a.blob + H.off_mov_y0);
        z.door_tab = reinterpret_cast<const uint32_t *>(a.blob + H.off_door_tab);
        z.n_mov = (int)H.n_mov; z.n_door = (int)H.n_zdoor; z.door_words = (a.zoo_doors + 1) / 2;
        z.n_created = (int)H.n_created;
        // never write outside this env's block, whatever the host planned (npp_load_levels checks the plan as well)
        if (z.n_door > a.zoo_doors) z.n_door = a.zoo_doors;
        if (z.n_mov > a.zoo_movers) z.n_mov = a.zoo_movers;
        z.mov_rank = reinterpret_cast<const uint16_t *>(a.blob + H.off_mov_rank);
        if (fast) zoo_fast_reset_block(z, 0, 1);
        else zoo_init_block(z, 0, 1, a.reset_fresh != 0);
    }
}

// npp_restore: copy the snapshot planes of the masked envs back into the live state
__global__ __launch_bounds__(256) void npp_restore_kernel(KernelArgs a, const double *sf, const uint32_t *su, const uint32_t *se,
                                                          const float *sc, const double *sz) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= a.n) return;
    if (a.reset_mask && a.reset_mask[env] == 0) return;
    const size_t N = (size_t)a.n;
    for (int k = 0; k < NF64; k++) a.f64[k * N + env] = sf[k * N + env];
    for (int k = 0; k < NU32; k++) a.u32[k * N + env] = su[k * N + env];
    for (int k = 0; k < a.n_words_max; k++) a.ent_bits[k * N + env] = se[k * N + env];
    for (int k = 0; k < 48; k++) a.sc_cache[(size_t)env * 48 + k] = sc[(size_t)env * 48 + k];
    if (sz && a.zoo)
        for (int k = 0; k < a.zoo_words; k++) a.zoo[(size_t)env * a.zoo_words + k] = sz[(size_t)env * a.zoo_words + k];
}

// This file is compiled four times (build_native.py: -DNPP_TU=0..3), each translation unit instantiating the step kernels
// of one (ZOO, MANY) pair, so that the 56 instantiations build in parallel.  TU 0 also holds the small kernels.  The
// diagnostic -DNPP_STAMPS build is a single translation unit with everything.
#ifndef NPP_TU
#define NPP_TU 0
#endif

template <int G, bool Z, bool M>
hipError_t launch_step_g(const KernelArgs &a, hipStream_t s) {
    const int wpb = a.waves_per_block;
    const int epb = (WAVE / G) * wpb;
    const int blocks = a.wg_count > 0 ? a.wg_count : (a.n + epb - 1) / epb;
    const dim3 grid(blocks), block(WAVE * wpb);
    // zoo levels: LDS also holds the per-env zoo blocks
    const size_t lds = lds_bytes(a.lds_level ? a.lds_hot_cap : 0, a.n_words_max, epb, Z ? a.zoo_words : 0);
    if constexpr (G == 16 && !Z) {
        if (a.variant == 1) {
            if (a.lds_level) hipLaunchKernelGGL((npp_step_kernel<G, true, Z, M, 1>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((npp_step_kernel<G, false, Z, M, 1>), grid, block, lds, s, a);
            return hipGetLastError();
        }
        if (a.variant == 2) {
            if (a.lds_level) hipLaunchKernelGGL((npp_step_kernel<G, true, Z, M, 2>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((npp_step_kernel<G, false, Z, M, 2>), grid, block, lds, s, a);
            return hipGetLastError();
        }
    }
    if (a.lds_level) hipLaunchKernelGGL((npp_step_kernel<G, true, Z, M, 0>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((npp_step_kernel<G, false, Z, M, 0>), grid, block, lds, s, a);
    return hipGetLastError();
}

template <bool Z, bool M>
hipError_t launch_step_zm(const KernelArgs &a, hipStream_t s) {
#ifdef NPP_ONLY_G   // experiments (tools/regs_experiment.sh): one instantiation compiles in seconds
    if (a.lanes_per_env == NPP_ONLY_G) return launch_step_g<NPP_ONLY_G, Z, M>(a, s);
    return hipErrorInvalidValue;
#else
    switch (a.lanes_per_env) {
        case 1: return launch_step_g<1, Z, M>(a, s);
        case 2: return launch_step_g<2, Z, M>(a, s);
        case 4: return launch_step_g<4, Z, M>(a, s);
        case 8: return launch_step_g<8, Z, M>(a, s);
        case 16: return launch_step_g<16, Z, M>(a, s);
        case 32: return launch_step_g<32, Z, M>(a, s);
        case 64: return launch_step_g<64, Z, M>(a, s);
        default: return hipErrorInvalidValue;
    }
#endif
}

}  // namespace

#if NPP_TU == 0
#ifdef NPP_STAMPS
hipError_t launch_step_tu0(const KernelArgs &a, hipStream_t s) { return launch_step_zm<false, false>(a, s); }
hipError_t launch_step_tu1(const KernelArgs &a, hipStream_t s) { return launch_step_zm<true, false>(a, s); }
hipError_t launch_step_tu2(const KernelArgs &a, hipStream_t s) { return launch_step_zm<false, true>(a, s); }
hipError_t launch_step_tu3(const KernelArgs &a, hipStream_t s) { return launch_step_zm<true, true>(a, s); }
#else
hipError_t launch_step_tu0(const KernelArgs &a, hipStream_t s) { return launch_step_zm<false, false>(a, s); }
hipError_t launch_step_tu1(const KernelArgs &a, hipStream_t s);
hipError_t launch_step_tu2(const KernelArgs &a, hipStream_t s);
hipError_t launch_step_tu3(const KernelArgs &a, hipStream_t s);
#endif

hipError_t launch_step(const KernelArgs &a, hipStream_t s) {
    switch ((a.zoo_active ? 1 : 0) | (a.n_steps > 1 ? 2 : 0)) {
        case 0: return launch_step_tu0(a, s);
        case 1: return launch_step_tu1(a, s);
        case 2: return launch_step_tu2(a, s);
        default: return launch_step_tu3(a, s);
    }
}

hipError_t launch_restore(const KernelArgs &a, const double *src_f64, const uint32_t *src_u32, const uint32_t *src_ent,
                          const float *src_sc, const double *src_zoo, hipStream_t s) {
    hipLaunchKernelGGL(npp_restore_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a, src_f64, src_u32, src_ent, src_sc, src_zoo);
    return hipGetLastError();
}

hipError_t launch_reset(const KernelArgs &a, hipStream_t s) {
    const int blocks = (a.n + 63) / 64;
    hipLaunchKernelGGL(npp_reset_kernel, dim3(blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}


#ifdef NPP_STAMPS
extern "C" int npp_debug_stamps(unsigned long long *out, int n_waves, int reset) {
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)N_STAMP * 16384);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * h.size());
    if (out) {
        for (int i = 0; i < N_STAMP; i++) out[i] = 0;
        for (int w = 0; w < n_waves && w < 16384; w++)
            for (int i = 0; i < N_STAMP; i++) out[i] += h[(size_t)w * N_STAMP + i];
        // per-wave totals (all phases) -> out[N_STAMP .. N_STAMP + n_waves)
        for (int w = 0; w < n_waves && w < 16384; w++) {
            unsigned long long t = 0;
            for (int i = 0; i < N_STAMP; i++) t += h[(size_t)w * N_STAMP + i];
            out[N_STAMP + w] = t;
        }
        // breakdown of the slowest wave after the totals
        int wmax = 0;
        for (int w = 1; w < n_waves && w < 16384; w++)
            if (out[N_STAMP + w] > out[N_STAMP + wmax]) wmax = w;
        for (int i = 0; i < N_STAMP; i++) out[N_STAMP + n_waves + i] = h[(size_t)wmax * N_STAMP + i];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), h.data(), sizeof(unsigned long long) * h.size());
    }
    return 0;
}
#endif

#elif defined(NPP_STAMPS)
// (diagnostic build: translation unit 0 holds everything, the stamp table is one device symbol)
#elif NPP_TU == 1
hipError_t launch_step_tu1(const KernelArgs &a, hipStream_t s) { return launch_step_zm<true, false>(a, s); }
#elif NPP_TU == 2
hipError_t launch_step_tu2(const KernelArgs &a, hipStream_t s) { return launch_step_zm<false, true>(a, s); }
#else
hipError_t launch_step_tu3(const KernelArgs &a, hipStream_t s) { return launch_step_zm<true, true>(a, s); }
#endif

}  // namespace npp
