// npp_reach_features.hpp -- the position-dependent part of the reachability observation: 38 floats from the per-level tables
// (npp_reach.hpp) and the ninja position.  One function, compiled for the host (CPU parity test through
// npp_reach_features_host) and for the device (npp_reach_kernel.hip), so both run the same arithmetic.
//
// Restates compute_reachability_features_from_graph (graph/reachability/feature_computation.py:197-937) and the parts of
// CachedPathDistanceCalculator.get_distance / get_geometric_distance (path_distance_calculator.py:847-1230, 1487-1830) and
// find_ninja_node (pathfinding_utils.py:1331-1497) it goes through on a level-cache hit, INCLUDING what the reference really
// does rather than what its comments say:
//   * `spatial_hash.query` does not exist (spatial_hash.py has find_closest / find_all_within_radius), every call raises and
//     is swallowed, so all node searches are the linear scans over `adjacency` in dict order (tile row, tile column, sub-node);
//   * the exit switch handed to the feature code is a dict, and `getattr(dict, "active", True)` is always True: the switch
//     counts as not collected for ever, feature 12 stays 0 and the current goal stays "switch";
//   * the level cache holds GEOMETRIC distances, so "physics cost / geometric distance" (feature 21) is 1 -> 1/3 after the log
//     normalisation whenever both exist;
//   * the directional platform ray cast snaps to multiples of 12 while nodes sit at 6 mod 12: it never finds a node and
//     features 30-37 are 1.0;
//   * features 18-19 (mine gradient) look the SDF up on the wrong object and stay 0.
// The cache-miss branch of get_distance (path_distance_calculator.py:949-960, 1218-1485) is restated for the exit door of the levels
// whose door lies within 24 px of its switch (ReachHdr::miss_exit: every exit query misses there): node selection here, the
// physics A* cost from the per-level table the host builds (npp_reach.cpp), and the calculator's per-episode
// (start cell, goal cell) -> cost dictionary as a per-env array (ReachMiss).  Elsewhere a miss (the ninja has no node with a cached
// distance) is reported through `status` bit 0 and the distance is treated as unreachable.
#pragma once
#include <cmath>

#include "npp_reach.hpp"

namespace npp {

// Per-env view of CachedPathDistanceCalculator.cache: key (start 24-px cell, goal 24-px cell) -> raw A* cost, filled on a miss,
// consulted BEFORE the level cache (path_distance_calculator.py:949-960), emptied by clear_cache() at every episode reset
// (reachability_mixin.py:67-70).  stamp[cell] == epoch marks a live entry; a new episode bumps the epoch.  One goal cell is
// tracked: the exit door's (the switch shares the entries when it lies in the same cell, ReachHdr::sw_alias).  Capacity: the
// calculator holds 5000 entries (max(max_cache_size, 5000)), a level has 1100 cells, so nothing is ever evicted.
struct ReachMiss {
    uint32_t *stamp;   // [REACH_CELLS]
    double *raw;       // [REACH_CELLS]
    uint32_t epoch;    // never 0
};

// int(v // 24) (Python float floor division is exact; the quotient of the division below can be off by one ulp)
NPP_HD inline __attribute__((always_inline)) int reach_cell24(double v) {
    int c = (int)floor(v / 24.0);
    if ((double)c * 24.0 > v) c--;
    if ((double)(c + 1) * 24.0 <= v) c++;
    return c;
}

struct ReachTabs {
    const ReachHdr *H;
    const unsigned char *blob;
    NPP_HD const unsigned char *cgoal() const { return blob + H->off_cgoal; }
    NPP_HD const double *astar(int k) const { return reinterpret_cast<const double *>(blob + H->off_astar) + (long)k * RNODES; }
    NPP_HD const unsigned char *in() const { return blob + H->off_in; }
    NPP_HD const double *dist(int g) const { return reinterpret_cast<const double *>(blob + H->off_dist) + (long)g * RNODES; }
    NPP_HD const int16_t *hop(int g) const { return reinterpret_cast<const int16_t *>(blob + H->off_hop) + (long)g * RNODES; }
    NPP_HD const double *mh(int g) const { return reinterpret_cast<const double *>(blob + H->off_mh) + (long)g * RNODES * 2; }
    NPP_HD const float *sdf() const { return reinterpret_cast<const float *>(blob + H->off_sdf); }
    NPP_HD const float *grad() const { return reinterpret_cast<const float *>(blob + H->off_grad); }
    NPP_HD const ReachRec *rec() const { return reinterpret_cast<const ReachRec *>(blob + H->off_rec); }
};

// position of a node in the iteration order of the reference's adjacency dict: tiles row-major, then (6,6) (18,6) (6,18) (18,18)
NPP_HD inline __attribute__((always_inline)) int reach_order_key(int id) {
    const int i = id / RH, j = id % RH;
    return (((j >> 1) * 42 + (i >> 1)) << 2) | ((i & 1) | ((j & 1) << 1));
}

NPP_HD inline __attribute__((always_inline)) double reach_floor(double v) { return floor(v); }

// find_ninja_node (pathfinding_utils.py:1331-1497) in two parts, because one feature vector calls it up to five times at the same
// position: reach_near() gathers the nodes the ninja overlaps ONCE (with every table read of the gather issued up front instead
// of one dependent load per lattice cell -- the device kernel is bound by exactly these chains), reach_pick() applies a call's rule.
// a node's record in registers (plain scalars, read member by member: a whole-struct copy of the 48-byte record ends up in scratch)
struct ReachNode {
    double dist0, dist1, mh0x, mh0y;
    int hop0, hop1, in, cgoal;
    NPP_HD double dist(int g) const { return g ? dist1 : dist0; }
    NPP_HD int hop(int g) const { return g ? hop1 : hop0; }
};
NPP_HD inline __attribute__((always_inline)) ReachNode reach_node_load(const ReachRec *recs, int id) {
    const ReachRec &m = recs[id];
    ReachNode q;
    q.dist0 = m.dist0; q.dist1 = m.dist1; q.mh0x = m.mh0x; q.mh0y = m.mh0y;
    q.hop0 = m.hop0; q.hop1 = m.hop1; q.in = m.in; q.cgoal = m.cgoal;
    return q;
}

struct ReachNear {
    int n;            // overlapping nodes (0 .. 4), in the iteration order of the reference's adjacency dict
    int id[4];
    double d2[4];     // squared distance to the ninja
    int fallback;     // n == 0: the first node in dict order inside the 24-px box around the ninja, -1 = none
    ReachNode rec[4]; // the table entries of id[k]
};

// the table entries of node `id`: from the gathered records when it is one of them, else one record load
NPP_HD inline __attribute__((always_inline)) ReachNode reach_rec(const ReachTabs &T, const ReachNear &N, int id) {
    ReachNode q;   // selects on scalars, member by member (a conditional copy of a whole record goes through scratch memory)
    q.dist0 = N.rec[0].dist0; q.dist1 = N.rec[0].dist1; q.mh0x = N.rec[0].mh0x; q.mh0y = N.rec[0].mh0y;
    q.hop0 = N.rec[0].hop0; q.hop1 = N.rec[0].hop1; q.in = N.rec[0].in; q.cgoal = N.rec[0].cgoal;
    bool found = N.n > 0 && N.id[0] == id;
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const bool c = k < N.n && N.id[k] == id;
        q.dist0 = c ? N.rec[k].dist0 : q.dist0; q.dist1 = c ? N.rec[k].dist1 : q.dist1;
        q.mh0x = c ? N.rec[k].mh0x : q.mh0x; q.mh0y = c ? N.rec[k].mh0y : q.mh0y;
        q.hop0 = c ? N.rec[k].hop0 : q.hop0; q.hop1 = c ? N.rec[k].hop1 : q.hop1;
        q.in = c ? N.rec[k].in : q.in; q.cgoal = c ? N.rec[k].cgoal : q.cgoal;
        found = found || c;
    }
    if (!found) {   // the fallback node of a ninja that overlaps none
        const ReachNode m = reach_node_load(T.rec(), id);
        q.dist0 = m.dist0; q.dist1 = m.dist1; q.mh0x = m.mh0x; q.mh0y = m.mh0y; q.hop0 = m.hop0; q.hop1 = m.hop1; q.in = m.in; q.cgoal = m.cgoal;
    }
    return q;
}

NPP_HD inline __attribute__((always_inline)) ReachNear reach_near(const ReachTabs &T, double px, double py) {
    const double nx = px - 24.0, ny = py - 24.0;
    const unsigned char *in = T.in();
    ReachNear N;
    N.n = 0; N.fallback = -1;
    // nodes within 10 px: at most two lattice indices per axis (the spacing is 12)
    const int i0 = (int)ceil((nx - 10.0 - 6.0) / 12.0), j0 = (int)ceil((ny - 10.0 - 6.0) / 12.0);
    const int i1 = (int)reach_floor((nx + 10.0 - 6.0) / 12.0), j1 = (int)reach_floor((ny + 10.0 - 6.0) / 12.0);
    int cid[4];
    const ReachRec *recs = T.rec();
#pragma unroll
    for (int k = 0; k < 4; k++) {   // (i0, j0) (i0, j0 + 1) (i0 + 1, j0) (i0 + 1, j0 + 1): the order of the reference's double loop
        const int i = i0 + (k >> 1), j = j0 + (k & 1);
        const bool ok = i <= i1 && j <= j1 && i >= 0 && i < RW && j >= 0 && j < RH;
        cid[k] = ok ? i * RH + j : -1;
        N.rec[k] = reach_node_load(recs, ok ? cid[k] : 0);   // unconditional, independent loads: everything a feature vector reads of the node
    }
    // the four slots stay at fixed positions (compile-time indices: registers, not scratch memory): a slot that does not qualify
    // gets the largest key, a five-exchange sorting network puts the others first in dict order
    int key[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idk = cid[k] < 0 ? 0 : cid[k];
        const double dx = reach_node_x(idk) - nx, dy = reach_node_y(idk) - ny, d2 = dx * dx + dy * dy;
        const bool ok = cid[k] >= 0 && N.rec[k].in && d2 <= 100.0;
        N.id[k] = idk; N.d2[k] = d2;
        key[k] = ok ? reach_order_key(idk) : 0x7fffffff;
        N.n += ok ? 1 : 0;
    }
#define NPP_CSEL(T, X, Y) { const T t_ = c_ ? (Y) : (X); (Y) = c_ ? (X) : (Y); (X) = t_; }
#define NPP_CSWAP(A, B)                                                                                             \
    {   /* selects on scalars, member by member: a conditional swap of whole records is compiled through scratch memory */ \
        const bool c_ = key[B] < key[A];                                                                            \
        NPP_CSEL(int, key[A], key[B]) NPP_CSEL(int, N.id[A], N.id[B]) NPP_CSEL(double, N.d2[A], N.d2[B])            \
        NPP_CSEL(double, N.rec[A].dist0, N.rec[B].dist0) NPP_CSEL(double, N.rec[A].dist1, N.rec[B].dist1)          \
        NPP_CSEL(double, N.rec[A].mh0x, N.rec[B].mh0x) NPP_CSEL(double, N.rec[A].mh0y, N.rec[B].mh0y)              \
        NPP_CSEL(int, N.rec[A].hop0, N.rec[B].hop0) NPP_CSEL(int, N.rec[A].hop1, N.rec[B].hop1)                    \
        NPP_CSEL(int, N.rec[A].in, N.rec[B].in) NPP_CSEL(int, N.rec[A].cgoal, N.rec[B].cgoal)                      \
    }
    NPP_CSWAP(0, 1) NPP_CSWAP(2, 3) NPP_CSWAP(0, 2) NPP_CSWAP(1, 3) NPP_CSWAP(1, 2)
#undef NPP_CSEL
#undef NPP_CSWAP
    if (N.n == 0) {
        // fallback: the first node in dict order with |x + 24 - px| < 24 and |y + 24 - py| < 24
        int best = -1, bk = 0x7fffffff;
        const int a0 = (int)reach_floor((nx - 24.0 - 6.0) / 12.0), a1 = (int)ceil((nx + 24.0 - 6.0) / 12.0);
        const int b0 = (int)reach_floor((ny - 24.0 - 6.0) / 12.0), b1 = (int)ceil((ny + 24.0 - 6.0) / 12.0);
        for (int i = a0; i <= a1; i++)
            for (int j = b0; j <= b1; j++) {
                if (i < 0 || i >= RW || j < 0 || j >= RH) continue;
                const int id = i * RH + j;
                if (!in[id]) continue;
                if (fabs((6 + 12 * i) + 24.0 - px) < 24.0 && fabs((6 + 12 * j) + 24.0 - py) < 24.0) {
                    const int k = reach_order_key(id);
                    if (k < bk) { bk = k; best = id; }
                }
            }
        N.fallback = best;
    }
    return N;
}

// goal_node < 0: no goal node given (closest overlapping node wins); otherwise the overlapping node with the smallest cached
// distance to goal `g` wins (Euclidean to goal_node when not cached, or always when g < 0: the call carries no goal id).
NPP_HD inline __attribute__((always_inline)) int reach_pick(const ReachTabs &T, const ReachNear &N, int goal_node, int g) {
    if (N.n == 0) return N.fallback;
    if (goal_node >= 0 && N.n > 1) {
        double dg[4];
#pragma unroll
        for (int k = 0; k < 4; k++) dg[k] = (g >= 0 && k < N.n) ? N.rec[k].dist(g) : INFINITY;
        int best = -1;
        double bd = INFINITY;
        const double gx = reach_node_x(goal_node), gy = reach_node_y(goal_node);
#pragma unroll
        for (int k = 0; k < 4; k++) {   // (no early exit: the slots keep compile-time indices, i.e. registers)
            double d = dg[k];
            if (d == INFINITY) {
                const double ex = gx - reach_node_x(N.id[k]), ey = gy - reach_node_y(N.id[k]);
                d = sqrt(ex * ex + ey * ey);   // ((gx - nx) ** 2 + (gy - ny) ** 2) ** 0.5 on integers
            }
            if (k < N.n && d < bd) { bd = d; best = N.id[k]; }
        }
        if (best >= 0) return best;
    }
    int best = N.id[0];
    double bd = N.d2[0];
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (k < N.n && N.d2[k] < bd) { bd = N.d2[k]; best = N.id[k]; }   // stable sort by distance: the first minimum in dict order
    return best;
}

NPP_HD inline __attribute__((always_inline)) int reach_find_ninja_node(const ReachTabs &T, double px, double py, int goal_node, int g) {
    return reach_pick(T, reach_near(T, px, py), goal_node, g);
}

// find_ninja_node with search_radius_override = R, or (R < 0) the "ANY closest node in the entire adjacency graph" loop that
// follows the ladder in get_distance's miss branch: a linear scan in dict order.  goal_node >= 0 with more than one node in range:
// the node closest to the goal node wins (Euclidean; first minimum in dict order), otherwise the one closest to the ninja.
NPP_HD inline __attribute__((always_inline)) int reach_scan_node(const ReachTabs &T, double px, double py, double R, int goal_node) {
    const double nx = px - 24.0, ny = py - 24.0;
    const unsigned char *in = T.in();
    int best = -1, bkey = 0, count = 0, gbest = -1, gkey = 0;
    double bd = 0.0;
    long gd = 0;
    for (int id = 0; id < RNODES; id++) {
        if (!in[id]) continue;
        const double dx = reach_node_x(id) - nx, dy = reach_node_y(id) - ny, d2 = dx * dx + dy * dy;
        if (R >= 0.0 && !(d2 <= R * R)) continue;
        const int key = reach_order_key(id);
        count++;
        if (best < 0 || d2 < bd || (d2 == bd && key < bkey)) { best = id; bd = d2; bkey = key; }
        if (goal_node >= 0) {
            const long ex = reach_node_x(goal_node) - reach_node_x(id), ey = reach_node_y(goal_node) - reach_node_y(id), e2 = ex * ex + ey * ey;
            if (gbest < 0 || e2 < gd || (e2 == gd && key < gkey)) { gbest = id; gd = e2; gkey = key; }
        }
    }
    return (goal_node >= 0 && count > 1 && R >= 0.0) ? gbest : best;
}

// The miss branch for the exit door (path_distance_calculator.py:949-960 and 1218-1485): the raw cost of the calculator's
// per-episode dictionary for the ninja's cell, computed and stored on the first query of the cell.  `peek`: look the entry up
// without creating it (the switch's query, which only ever READS the shared key); returns NaN when absent.
NPP_HD inline __attribute__((always_inline)) double reach_exit_miss_raw(const ReachTabs &T, const ReachNear &N, double px, double py, ReachMiss *M, bool peek, bool *miss) {
    int cx = reach_cell24(px), cy = reach_cell24(py);
    cx = cx < 0 ? 0 : (cx > 43 ? 43 : cx);
    cy = cy < 0 ? 0 : (cy > 24 ? 24 : cy);
    const int cell = cx * 25 + cy;
    if (M && M->stamp && M->stamp[cell] == M->epoch) return M->raw[cell];   // (stamp == NULL: no dictionary, like M == NULL)
    if (peek) return NAN;
    // temp start node: find_ninja_node, then the 48 / 150 px retries, then any closest node
    int t = reach_pick(T, N, -1, 0);
    if (t < 0) t = reach_scan_node(T, px, py, 48.0, -1);
    if (t < 0) t = reach_scan_node(T, px, py, 150.0, -1);
    if (t < 0) t = reach_scan_node(T, px, py, -1.0, -1);
    double raw = INFINITY;
    if (t >= 0) {
        const int k = T.H->off_cgoal ? (int)reach_rec(T, N, t).cgoal : 0xff;
        if (k < (int)T.H->n_cand) {
            const int c = T.H->cand[k];
            // the final start node: get_distance still carries the INFERRED goal id ("switch") at this point, so among several overlapping
            // nodes the one with the smallest cached distance to the switch wins (Euclidean to the goal node when it has none)
            int s = reach_pick(T, N, c, 0);
            if (s < 0) s = reach_scan_node(T, px, py, 48.0, c);
            if (s < 0) s = reach_scan_node(T, px, py, 150.0, c);
            if (s < 0) s = reach_scan_node(T, px, py, -1.0, c);
            if (s >= 0) {
                raw = T.astar(k)[s];
                if (raw != raw) { raw = INFINITY; *miss = true; }   // pair outside the tabulated neighbourhood
            }
        } else *miss = true;
    }
    if (M && M->stamp) { M->stamp[cell] = M->epoch; M->raw[cell] = raw; }
    return raw;
}

// get_distance / get_geometric_distance on the level-cache path (they return the same number there).  g: 0 exit switch,
// 1 exit door.  Returns +inf when unreachable; sets *miss when the reference would leave the level-cache path.
NPP_HD inline __attribute__((always_inline)) double reach_level_distance(const ReachTabs &T, const ReachNear &N, double px, double py, int g, double entity_radius, bool *miss) {
    const ReachHdr &H = *T.H;
    const double combined = 10.0 + entity_radius;
    const int gid = g == 1 ? H.exit_gid : 0;   // which goal's tables the reference reads (goal-id inference, see ReachHdr)
    const int sn = reach_pick(T, N, H.goal_node[2 + g], gid);   // goal node of get_distance: thresholds 16 / 22, then 32
    if (sn < 0) { *miss = true; return INFINITY; }
    const ReachNode q = reach_rec(T, N, sn);
    const double cached = q.dist(gid);
    if (cached == INFINITY) { *miss = true; return INFINITY; }
    const int nh = q.hop(gid);
    if (nh >= 0) {
        const double pdx = reach_node_x(nh) - reach_node_x(sn), pdy = reach_node_y(nh) - reach_node_y(sn);
        // (path_dx ** 2 + path_dy ** 2) ** 0.5: 12 for a cardinal hop, 288 ** 0.5 for a diagonal one (== sqrt(288) under glibc)
        const double plen = (pdx != 0.0 && pdy != 0.0) ? 16.970562748477139 : 12.0;
        const double dirx = pdx / plen, diry = pdy / plen;
        const double ox = px - (reach_node_x(sn) + 24), oy = py - (reach_node_y(sn) + 24);
        const double projection = ox * dirx + oy * diry;
        const double t = cached - projection - combined;
        return t > 0.0 ? t : 0.0;
    }
    const double t = cached - combined;
    return t > 0.0 ? t : 0.0;
}

// get_distance (path_distance_calculator.py:847-1485).  g: 0 exit switch, 1 exit door.  Returns +inf when unreachable; sets *miss
// when the reference would run a part of its miss branch that is not tabulated.  geometric = true: get_geometric_distance
// (path_distance_calculator.py:1487-1975), which never looks at the per-episode dictionary.
NPP_HD inline __attribute__((always_inline)) double reach_goal_distance(const ReachTabs &T, const ReachNear &N, double px, double py, int g, double entity_radius, ReachMiss *M,
                                         bool geometric, bool *miss) {
    const ReachHdr &H = *T.H;
    const int gx = H.goal_x[g], gy = H.goal_y[g];
    if (gx == 0 && gy == 0) return INFINITY;
    const double combined = 10.0 + entity_radius;
    const double dx = px - gx, dy = py - gy;
    if (dx * dx + dy * dy <= combined * combined) return 0.0;
    if (H.miss_exit && !geometric) {
        // the per-episode dictionary comes first; its key holds the goal's CELL, so a switch in the door's cell reads the door's entry
        if (g == 1 || H.sw_alias) {
            const double raw = reach_exit_miss_raw(T, N, px, py, M, g == 0, miss);
            if (raw == raw) {
                const double t = raw - combined;
                return t > 0.0 ? t : 0.0;   // max(0.0, inf - r) = inf
            }
        }
    }
    return reach_level_distance(T, N, px, py, g, entity_radius, miss);
}

NPP_HD inline __attribute__((always_inline)) float reach_clip01(double v) { return (float)(v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v)); }

// out[38]; sdf_out[3] = mine_sdf_features (value, gradient) at the ninja (npp_environment.py mine_sdf_features);
// returns status: bit 0 = the reference would have run its physics A* fallback here (not restated)
NPP_HD inline __attribute__((always_inline)) int reach_features(const ReachTabs &T, double px, double py, int total_mines, int deadly_mines, float *out, float *sdf_out,
                                 ReachMiss *M = nullptr) {
    const ReachHdr &H = *T.H;
    for (int k = 0; k < REACH_DIM; k++) out[k] = 0.f;
    bool miss = false;
    out[0] = H.f0;
    const double area = H.area_scale;
    const ReachNear N = reach_near(T, px, py);   // the nodes the ninja overlaps: shared by every node selection below
    double d_sw = INFINITY, d_ex = INFINITY;
    if (H.sw_valid) d_sw = reach_goal_distance(T, N, px, py, 0, 6.0, M, false, &miss);
    if (d_sw != INFINITY) out[1] = reach_clip01(1.0 - d_sw / area);
    else if (H.ex_valid) {   // third priority of the "next objective" ladder: the exit door (no locked-door switches reach here)
        const double d = reach_goal_distance(T, N, px, py, 1, 12.0, M, false, &miss);
        if (d != INFINITY) out[1] = reach_clip01(1.0 - d / area);
    }
    if (H.ex_valid) d_ex = reach_goal_distance(T, N, px, py, 1, 12.0, M, false, &miss);
    if (d_ex != INFINITY) out[2] = reach_clip01(1.0 - d_ex / area);
    out[3] = H.exit_reachable;
    out[4] = d_sw != INFINITY ? reach_clip01(d_sw / area) : 1.f;
    out[5] = d_ex != INFINITY ? reach_clip01(d_ex / area) : 1.f;
    const bool sw_dir = d_sw != INFINITY, ex_dir = d_ex != INFINITY;
    if (sw_dir) {
        const double dx = H.goal_x[0] - px, dy = H.goal_y[0] - py, dist = sqrt(dx * dx + dy * dy);
        if (dist > 0.001) { out[6] = (float)(dx / dist); out[7] = (float)(dy / dist); }
    }
    if (ex_dir) {
        const double dx = H.goal_x[1] - px, dy = H.goal_y[1] - py, dist = sqrt(dx * dx + dy * dy);
        if (dist > 0.001) { out[8] = (float)(dx / dist); out[9] = (float)(dy / dist); }
    }
    out[10] = reach_clip01(total_mines / 10.0);
    out[11] = total_mines > 0 ? (float)((double)deadly_mines / (double)total_mines) : 0.f;
    // out[12] = 0: the switch never reads as activated (see the header comment); current goal = "switch"
    int ninja_node = -1;
    if (sw_dir) ninja_node = reach_pick(T, N, -1, 0);
    ReachNode nrec;
    nrec.mh0x = nrec.mh0y = NAN; nrec.hop0 = -1;
    if (ninja_node >= 0) {
        nrec = reach_rec(T, N, ninja_node);
        const int nh = nrec.hop0;
        if (nh >= 0) {
            const double dx = (reach_node_x(nh) + 24) - px, dy = (reach_node_y(nh) + 24) - py, dist = sqrt(dx * dx + dy * dy);
            if (dist > 0.001) { out[13] = (float)(dx / dist); out[14] = (float)(dy / dist); }
        }
    }
    // 15-17: waypoints (none without the curriculum machinery)
    {   // 18-19 stay 0: the reference looks for the mine SDF on `path_calculator.level_cache`, which has no such attribute (it lives
        // on the calculator itself, path_distance_calculator.py:114), so its gradient branch never runs.  mine_sdf_features is the
        // separate observation key read straight from the SDF (MineSignedDistanceField.get_features_at_position).
        int col = (int)(px / 12.0), row = (int)(py / 12.0);
        col = col < 0 ? 0 : (col > SDF_W - 1 ? SDF_W - 1 : col);
        row = row < 0 ? 0 : (row > SDF_H - 1 ? SDF_H - 1 : row);
        float sv = 1.f, gx = 0.f, gy = 0.f;
        if (H.off_sdf) {
            sv = T.sdf()[row * SDF_W + col];
            gx = T.grad()[(row * SDF_W + col) * 2];
            gy = T.grad()[(row * SDF_W + col) * 2 + 1];
        }
        if (sdf_out) { sdf_out[0] = sv; sdf_out[1] = gx; sdf_out[2] = gy; }
    }
    if ((out[13] != 0.f || out[14] != 0.f) && sw_dir) {   // 20: the next hop points away from the goal
        const double gdx = H.goal_x[0] - px, gdy = H.goal_y[0] - py, gd = sqrt(gdx * gdx + gdy * gdy);
        if (gd > 0.001) {
            // float32 features times float64 directions: numpy promotes to float64
            const double al = (double)out[13] * (gdx / gd) + (double)out[14] * (gdy / gd);
            if (al < -0.3) out[20] = 1.f;
        }
    }
    if (sw_dir) {   // 21: log-normalised physics cost / geometric distance.  Both come from the level cache's geometric table (ratio 1 ->
                    // 1/3) unless the switch query was answered from the per-episode dictionary (ReachHdr::sw_alias)
        const double geo = (H.miss_exit && H.sw_alias) ? reach_goal_distance(T, N, px, py, 0, 6.0, M, true, &miss) : d_sw;
        if (geo != INFINITY && geo > 0.001) {
            double ratio = d_sw / geo;
            ratio = ratio > 0.1 ? ratio : 0.1;
            out[21] = reach_clip01((log(ratio) + 2.0) / 6.0);
        }
    }
    if (ninja_node >= 0) {   // 22-24: 4-hop look-ahead direction and its alignment with the next hop
        const double mx = nrec.mh0x, my = nrec.mh0y;
        if (mx == mx) {
            out[22] = (float)mx; out[23] = (float)my;
            if (out[13] != 0.f || out[14] != 0.f) {
                const float curv = out[13] * out[22] + out[14] * out[23];   // float32 arithmetic (numpy float32 scalars)
                out[24] = (curv + 1.0f) / 2.0f;
            }
        }
    }
    if (sw_dir && ex_dir)
        for (int k = 0; k < 4; k++) out[25 + k] = H.exit_path[k];
    if (sw_dir) {   // 29: ramps to 1 within 50 px of the switch
        const double dx = H.goal_x[0] - px, dy = H.goal_y[0] - py, d = sqrt(dx * dx + dy * dy);
        double c = d / 50.0;
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        out[29] = (float)(1.0 - c);
    }
    for (int k = 30; k < 38; k++) out[k] = 1.f;
    return miss ? 1 : 0;
}

}  // namespace npp
