// npp_reach.cpp -- host-side builder of the per-level reachability tables (npp_reach.hpp) + host-only C entry points for
// the CPU test-suite.  Restates, for the accelerated path, what the reference computes once per level:
//   graph_builder.py:969-1292   sub-node generation and the 8-connected traversability graph (tables: npp_reach_tables.inc)
//   graph_builder.py:919-966    node physics cache (grounded / walled)
//   entity_mask.py:75-118       nodes blocked by toggle mines (radius 10 + 4); locked doors block nothing (their dicts carry
//                               no "position", entity_mask.py:170-190 then lands on tile (-1, -1))
//   pathfinding_utils.py:2530   flood fill from the spawn, start node through the sub-cell lookup (subcell_node_lookup.py:260)
//   path_distance_cache.py:249  per goal: Dijkstra with geometric edge costs + the horizontal rule, next hop, 4-hop direction
//   mine_proximity_cache.py:281 mine signed-distance field
//   feature_computation.py:104  exit-path features (25-28)
#include "npp_reach.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#include "../../include/npp_amd.h"
#include "npp_level.hpp"
#include "npp_reach_build.hpp"
#include "npp_reach_features.hpp"

namespace npp {
namespace {

#include "npp_reach_tables.inc"

inline bool tbit(const uint32_t *t, int i) { return (t[i >> 5] >> (i & 31)) & 1u; }
inline bool t_valid(int type, int sx, int sy) { return type >= 0 && type < 38 && tbit(REACH_VALID, (type * 2 + sx) * 2 + sy); }
inline bool t_within(int type, int a, int b) { return type >= 0 && type < 38 && tbit(REACH_WITHIN, (type * 4 + a) * 4 + b); }
inline bool t_conn(int a, int b, int d) { return a >= 0 && a < 34 && b >= 0 && b < 34 && tbit(REACH_CONN, (a * 34 + b) * 8 + d); }
inline bool t_cross(int a, int b, int d, int u) { return a >= 0 && a < 38 && b >= 0 && b < 38 && tbit(REACH_CROSS, ((a * 38 + b) * 4 + d) * 2 + u); }

// directions in the order the reference appends neighbours (graph_builder.py:1216-1225): N E S W NE SE SW NW
const int DX[8] = {0, 1, 0, -1, 1, 1, -1, -1}, DY[8] = {-1, 0, 1, 0, -1, 1, 1, -1};
const int CONN_IDX[8] = {0, 2, 4, 6, 1, 3, 5, 7};   // index into the precomputer's N NE E SE S SW W NW

inline int nid(int i, int j) { return i * RH + j; }
inline bool inside(int i, int j) { return i >= 0 && i < RW && j >= 0 && j < RH; }

struct Graph {
    std::vector<uint8_t> in;    // node present
    std::vector<uint8_t> adj;   // bit d = edge in direction d
    Graph() : in(RNODES, 0), adj(RNODES, 0) {}
    bool edge(int id, int d) const { return (adj[id] >> d) & 1u; }
};

// inner tile (tx, ty) of the 42 x 23 grid; out of range = -1
inline int tile_at(const CompiledLevel &L, int tx, int ty) {
    if (tx < 0 || tx >= 42 || ty < 0 || ty >= 23) return -1;
    return L.tiles[(tx + 1) * GRID_H + (ty + 1)];
}

// _is_sub_node_traversable (graph_builder.py:1512-1700) for an edge between two existing sub-nodes
bool traversable(const CompiledLevel &L, int i, int j, int d) {
    const int ni = i + DX[d], nj = j + DY[d];
    const int tx = i >> 1, ty = j >> 1, sx = i & 1, sy = j & 1;
    const int ux = ni >> 1, uy = nj >> 1, rx = ni & 1, ry = nj & 1;
    const int st = tile_at(L, tx, ty), dt = tile_at(L, ux, uy);
    if (tx == ux && ty == uy) return t_within(st, sx + 2 * sy, rx + 2 * ry);
    const int tdx = ux - tx, tdy = uy - ty;
    if (d >= 4) {   // diagonal: _check_diagonal_clear
        const int side = tile_at(L, tx + tdx, ty), vert = tile_at(L, tx, ty + tdy);
        if (side < 0 || vert < 0) return false;
        if (side == 1 && vert == 1) return false;
        if (tdx != 0 && tdy != 0) {
            const int side_cx = tdx == 1 ? 0 : 1, vert_cy = tdy == -1 ? 1 : 0;
            if (!t_valid(side, side_cx, sy) || !t_valid(vert, sx, vert_cy)) return false;
        }
    } else {        // cardinal: the segment must not cross solid geometry in either tile
        const int u = (d == 0 || d == 2) ? sx : sy;
        if (!t_cross(st, dt, d, u)) return false;
    }
    return t_conn(st, dt, CONN_IDX[d]);
}

void build_base(const CompiledLevel &L, Graph &g) {
    for (int ty = 0; ty < 23; ty++)
        for (int tx = 0; tx < 42; tx++) {
            const int t = tile_at(L, tx, ty);
            if (t == 1) continue;
            for (int s = 0; s < 4; s++) {
                const int sx = s & 1, sy = s >> 1;
                if (t_valid(t, sx, sy)) g.in[nid(2 * tx + sx, 2 * ty + sy)] = 1;
            }
        }
    for (int i = 0; i < RW; i++)
        for (int j = 0; j < RH; j++) {
            if (!g.in[nid(i, j)]) continue;
            uint8_t m = 0;
            for (int d = 0; d < 8; d++) {
                const int ni = i + DX[d], nj = j + DY[d];
                if (!inside(ni, nj) || !g.in[nid(ni, nj)]) continue;
                if (traversable(L, i, j, d)) m |= (uint8_t)(1u << d);
            }
            g.adj[nid(i, j)] = m;
        }
}

// physics cache (graph_builder.py:919-966) on the BASE graph: bit 0 grounded, bit 1 walled
void build_physics(const Graph &b, std::vector<uint8_t> &ph) {
    ph.assign(RNODES, 0);
    for (int i = 0; i < RW; i++)
        for (int j = 0; j < RH; j++) {
            const int id = nid(i, j);
            if (!b.in[id]) continue;
            bool grounded = true;
            if (inside(i, j + 1) && b.in[nid(i, j + 1)] && b.edge(id, 2)) grounded = false;
            const bool walled = !(inside(i - 1, j) && b.in[nid(i - 1, j)]) || !(inside(i + 1, j) && b.in[nid(i + 1, j)]);
            ph[id] = (uint8_t)((grounded ? 1 : 0) | (walled ? 2 : 0));
        }
}

// SubcellNodeLookupLoader.find_closest_node_position (subcell_node_lookup.py:260-390) + the tail of
// find_closest_node_to_position (pathfinding_utils.py:885-1008).  (wx, wy) world position; returns node id or -1.
int find_closest_node(const Graph &g, const std::vector<uint8_t> &ph, double wx, double wy, double threshold, bool prefer_grounded) {
    const double qx = wx - 24.0, qy = wy - 24.0;
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int sx0 = clampi((int)std::floor((qx + 48.0 - 6.0) / 12.0), 0, 91), sy0 = clampi((int)std::floor((qy + 48.0 - 6.0) / 12.0), 0, 53);
    std::vector<int> cands;
    auto cell_node = [&](int sx, int sy) -> int {   // lookup[sx][sy] = (-42 + 12 sx, -42 + 12 sy); negative = invalid
        const int cx = -42 + 12 * sx, cy = -42 + 12 * sy;
        if (cx < 0 || cy < 0) return -1;
        const int id = reach_node_id(cx, cy);
        return (id >= 0 && g.in[id]) ? id : -1;
    };
    {
        const int id = cell_node(sx0, sy0);
        if (id >= 0) {
            if (!prefer_grounded) return id;   // no distance test on the primary candidate
            cands.push_back(id);
        }
    }
    const int R = (int)std::ceil(threshold / 12.0);
    for (int r = 1; r <= R; r++)
        for (int dx = -r; dx <= r; dx++)
            for (int dy = -r; dy <= r; dy++) {
                if (dx == 0 && dy == 0) continue;
                const int id = cell_node(clampi(sx0 + dx, 0, 91), clampi(sy0 + dy, 0, 53));
                if (id < 0) continue;
                const double ex = reach_node_x(id) - qx, ey = reach_node_y(id) - qy;
                if (ex * ex + ey * ey <= threshold * threshold) {
                    if (!prefer_grounded) return id;
                    cands.push_back(id);
                }
            }
    if (prefer_grounded && !cands.empty()) {
        int best = -1;
        double bd = 0;
        for (int pass = 0; pass < 2 && best < 0; pass++)   // grounded candidates first, then the others; min() keeps the first minimum
            for (int id : cands) {
                if (((ph[id] & 1) != 0) != (pass == 0)) continue;
                const double ex = reach_node_x(id) - qx, ey = reach_node_y(id) - qy, d = ex * ex + ey * ey;
                if (best < 0 || d < bd) { best = id; bd = d; }
            }
        return best;
    }
    // the linear scan that follows in the reference can only find nodes beyond the threshold here (every lattice node within
    // `threshold` was visited above), so it returns None
    return -1;
}

int flood_fill(const Graph &g, const std::vector<uint8_t> &ph, double wx, double wy, std::vector<uint8_t> &reach) {
    reach.assign(RNODES, 0);
    int start = find_closest_node(g, ph, wx, wy, 10.0, true);
    if (start < 0) start = find_closest_node(g, ph, wx, wy, 50.0, true);
    if (start < 0) return 0;
    std::vector<int> q{start};
    reach[start] = 1;
    int n = 0;
    for (size_t h = 0; h < q.size(); h++) {
        const int id = q[h];
        n++;
        const int i = id / RH, j = id % RH;
        for (int d = 0; d < 8; d++)
            if (g.edge(id, d)) {
                const int nb = nid(i + DX[d], j + DY[d]);
                if (!reach[nb]) { reach[nb] = 1; q.push_back(nb); }
            }
    }
    return n;
}

struct PQ {
    double d;
    int x, y, id;
    bool operator<(const PQ &o) const {   // std::priority_queue is a max-heap: invert (dist, (x, y)) tuple order of heapq
        if (d != o.d) return d > o.d;
        if (x != o.x) return x > o.x;
        return y > o.y;
    }
};

// bfs_distance_from_start(..., use_geometric_costs=True, return_parents=True) from the goal node (pathfinding_utils.py:1499)
void dijkstra(const Graph &g, const std::vector<uint8_t> &ph, int start, std::vector<double> &dist, std::vector<int16_t> &parent) {
    dist.assign(RNODES, INFINITY);
    parent.assign(RNODES, -1);
    std::vector<uint8_t> visited(RNODES, 0);
    std::priority_queue<PQ> pq;
    dist[start] = 0.0;
    pq.push({0.0, reach_node_x(start), reach_node_y(start), start});
    while (!pq.empty()) {
        const PQ cur = pq.top();
        pq.pop();
        if (visited[cur.id]) continue;
        visited[cur.id] = 1;
        const int i = cur.id / RH, j = cur.id % RH;
        for (int d = 0; d < 8; d++) {
            if (!g.edge(cur.id, d)) continue;
            const int nb = nid(i + DX[d], j + DY[d]);
            if (visited[nb]) continue;
            // _violates_horizontal_rule: two consecutive horizontal edges unless both ends of this one are grounded
            if (DY[d] == 0 && DX[d] != 0 && !((ph[cur.id] & 1) && (ph[nb] & 1))) {
                const int p = parent[cur.id];
                if (p >= 0 && reach_node_y(p) == reach_node_y(cur.id) && reach_node_x(p) != reach_node_x(cur.id)) continue;
            }
            const double cost = d < 4 ? 12.0 : std::pow(288.0, 0.5);
            const double nd = cur.d + cost;
            if (dist[nb] == INFINITY || nd < dist[nb]) {
                dist[nb] = nd;
                parent[nb] = (int16_t)cur.id;
                pq.push({nd, reach_node_x(nb), reach_node_y(nb), nb});
            }
        }
    }
}

// _compute_multi_hop_direction(node, parents, max_hops=4) (path_distance_cache.py:187-247)
void multi_hop(const std::vector<int16_t> &parent, int node, double &ox, double &oy) {
    static const double W[4] = {0.45, 0.25, 0.15, 0.08};
    double tx = 0.0, ty = 0.0;
    int cur = node;
    for (int k = 0; k < 4; k++) {
        const int nx = parent[cur];
        if (nx < 0) break;
        tx += W[k] * (double)(reach_node_x(nx) - reach_node_x(cur));
        ty += W[k] * (double)(reach_node_y(nx) - reach_node_y(cur));
        cur = nx;
    }
    const double mag = std::pow(tx * tx + ty * ty, 0.5);
    if (mag < 0.001) { ox = NAN; oy = NAN; return; }
    ox = tx / mag; oy = ty / mag;
}

}  // namespace

bool build_reach(const double *map, int64_t n, ReachBuilt &R, std::string &err) {
    CompiledLevel L;
    if (!compile_level(map, n, L, err)) return false;
    build_reach(L, R);
    return true;
}

void build_reach(const CompiledLevel &L, ReachBuilt &R) {
    R = ReachBuilt();
    ReachHdr &H = R.hdr;
    std::memset(&H, 0, sizeof(H));
    H.supported = 1;
    // ---- entities the feature code sees
    std::vector<int> switches, doors, mines1, mines21;
    for (size_t k = 0; k < L.ent_map_order.size(); k++) {
        const int s = L.ent_map_order[k];
        const uint32_t kind = L.ent_meta[s] & 15u, type = (L.ent_meta[s] >> 24) & 63u;
        if (kind == EK_SWITCH) switches.push_back(s);
        else if (kind == EK_EXIT) doors.push_back(s);
        else if (kind == EK_MINE) (type == 1 ? mines1 : mines21).push_back(s);
    }
    if (!((switches.size() == 1 && doors.size() == 1) || (switches.empty() && doors.empty()))) {
        // several exits: the feature code takes the LAST switch (nplay_headless.py _sim_exit_switch) while the level cache keys
        // its goals on the FIRST (level_data_helpers.py:34-51); the mismatch sends the reference into its physics A* branch.
        // (No exit at all is fine: every goal-dependent feature keeps its "unreachable" value.)
        H.supported = 0;
        R.note = "needs one exit switch / door pair (or none)";
    }
    const int sw = switches.empty() ? -1 : switches.back(), dr = doors.empty() ? -1 : doors.back();
    if (sw >= 0) { H.goal_x[0] = (int)L.ent_x[sw]; H.goal_y[0] = (int)L.ent_y[sw]; }
    if (dr >= 0) { H.goal_x[1] = (int)L.ent_x[dr]; H.goal_y[1] = (int)L.ent_y[dr]; }
    H.sw_valid = sw >= 0 && !(L.ent_x[sw] == 0.0 && L.ent_y[sw] == 0.0);
    H.ex_valid = dr >= 0 && !(L.ent_x[dr] == 0.0 && L.ent_y[dr] == 0.0);
    H.n_mines = (int)(mines1.size() + mines21.size());
    R.mine_mask.assign((L.ent_meta.size() + 15) / 16, 0u);
    for (int s : mines1) R.mine_mask[s >> 4] |= 1u << ((s & 15) * 2);
    for (int s : mines21) R.mine_mask[s >> 4] |= 1u << ((s & 15) * 2);
    if (R.mine_mask.empty()) R.mine_mask.push_back(0u);
    // ---- base graph, physics, entity mask, flood fill from the spawn
    Graph base;
    build_base(L, base);
    build_physics(base, R.phys);
    R.base_in = base.in; R.base_adj = base.adj;
    Graph masked = base;
    {
        std::vector<uint8_t> blocked(RNODES, 0);
        auto block = [&](const std::vector<int> &ms) {
            for (int s : ms) {
                const int mx = (int)L.ent_x[s] - 24, my = (int)L.ent_y[s] - 24;   // _get_entity_pixel_position: int() then the offset
                for (int id = 0; id < RNODES; id++) {
                    if (!base.in[id]) continue;
                    const long ex = reach_node_x(id) - mx, ey = reach_node_y(id) - my;
                    if ((double)(ex * ex + ey * ey) < 14.0 * 14.0) blocked[id] = 1;   // NINJA_RADIUS + RADII[0]
                }
            }
        };
        block(mines1);
        block(mines21);
        R.blocked = blocked;
        for (int id = 0; id < RNODES; id++) {
            if (!masked.in[id]) continue;
            if (blocked[id]) { masked.in[id] = 0; masked.adj[id] = 0; continue; }
            const int i = id / RH, j = id % RH;
            uint8_t m = masked.adj[id];
            for (int d = 0; d < 8; d++)
                if ((m >> d) & 1u) {
                    const int nb = nid(i + DX[d], j + DY[d]);
                    if (blocked[nb]) m &= (uint8_t)~(1u << d);
                }
            masked.adj[id] = m;
        }
    }
    Graph fin = masked;
    {
        std::vector<uint8_t> reach;
        // graph_builder.py:844-866: physics_cache is None there, grounding comes from base_adjacency -- the same bits
        int cnt = flood_fill(masked, R.phys, L.spawn_x, L.spawn_y, reach);
        if (cnt == 0) {   // "Using ALL adjacency nodes as fallback"
            for (int id = 0; id < RNODES; id++) reach[id] = masked.in[id];
        }
        for (int id = 0; id < RNODES; id++) {
            if (!fin.in[id]) continue;
            if (!reach[id]) { fin.in[id] = 0; fin.adj[id] = 0; continue; }
            const int i = id / RH, j = id % RH;
            uint8_t m = fin.adj[id];
            for (int d = 0; d < 8; d++)
                if (((m >> d) & 1u) && !reach[nid(i + DX[d], j + DY[d])]) m &= (uint8_t)~(1u << d);
            fin.adj[id] = m;
        }
    }
    R.in = fin.in; R.adj = fin.adj;
    int n_adj = 0;
    for (int id = 0; id < RNODES; id++) n_adj += fin.in[id];
    H.n_adj = (uint32_t)n_adj;
    if (n_adj == 0) { H.supported = 0; R.note = "empty adjacency"; }
    {
        double v = (double)n_adj / 966.0;
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        H.f0 = (float)v;
    }
    // ---- area scale (feature_computation.py:300-332): flood fill from int(start_position) + 24 (sic: the spawn is already in
    //      world space, the offset is added once more); failure -> LEVEL_DIAGONAL
    {
        std::vector<uint8_t> reach;
        const double sx = (double)((long)L.spawn_x + 24), sy = (double)((long)L.spawn_y + 24);
        const int cnt = n_adj ? flood_fill(fin, R.phys, sx, sy, reach) : 0;
        R.surface_area = cnt;
        H.area_scale = cnt > 0 ? std::sqrt((double)cnt) * 12.0 : std::sqrt(1056.0 * 1056.0 + 600.0 * 600.0);
    }
    // ---- level cache: goals exit_switch_0 and exit_door_0
    for (int gi = 0; gi < 2; gi++) {
        R.dist[gi].assign(RNODES, INFINITY);
        R.hop[gi].assign(RNODES, -1);
        R.mh[gi].assign(2 * RNODES, NAN);
        H.goal_node[gi] = -1;
        H.goal_node[2 + gi] = -1;
        const bool ok = gi == 0 ? H.sw_valid : H.ex_valid;
        if (!ok || !n_adj) continue;
        const double gx = H.goal_x[gi], gy = H.goal_y[gi];
        const int gn = find_closest_node(fin, R.phys, gx, gy, 50.0, false);
        H.goal_node[gi] = gn;
        // get_distance's own goal node: threshold ninja radius + entity radius, then 32 (path_distance_calculator.py:970-1000)
        const double thr = 10.0 + (gi == 0 ? 6.0 : 12.0);
        int g2 = find_closest_node(fin, R.phys, gx, gy, thr, false);
        if (g2 < 0) g2 = find_closest_node(fin, R.phys, gx, gy, 32.0, false);
        H.goal_node[2 + gi] = g2;
        if (g2 < 0) { H.supported = 0; R.note = "goal node not found (the reference raises RuntimeError)"; }
        if (gn < 0) continue;
        std::vector<int16_t> parent;
        dijkstra(fin, R.phys, gn, R.dist[gi], parent);
        for (int id = 0; id < RNODES; id++) {
            if (R.dist[gi][id] == INFINITY) continue;
            R.hop[gi][id] = parent[id];
            multi_hop(parent, id, R.mh[gi][2 * id], R.mh[gi][2 * id + 1]);
        }
    }
    // goal-id inference for the exit door position (path_distance_calculator.py:1004-1022): the switch positions are tried first
    H.exit_gid = 1;
    if (H.sw_valid && H.ex_valid && sw >= 0 && dr >= 0 && std::fabs(L.ent_x[sw] - H.goal_x[1]) < 24.0 && std::fabs(L.ent_y[sw] - H.goal_y[1]) < 24.0) {
        H.exit_gid = 0;
        if (std::abs(H.goal_x[0] - H.goal_x[1]) > 12 || std::abs(H.goal_y[0] - H.goal_y[1]) > 12) {
            H.supported = 0;
            R.note = "exit door within 24 px of the switch: the reference validates against the wrong goal and leaves the level cache";
        }
    }
    // ---- feature 3 and features 25-28 (static)
    H.exit_reachable = (H.ex_valid && H.goal_node[1] >= 0) ? 1.f : 0.f;
    for (int k = 0; k < 4; k++) H.exit_path[k] = 0.f;
    if (H.sw_valid && H.goal_node[0] >= 0) {
        const int sn = H.goal_node[0];   // find_closest_node_to_position(switch_pos, threshold=50)
        const int nh = R.hop[1][sn];
        if (R.dist[1][sn] != INFINITY && nh >= 0) {
            const double dx = (double)(reach_node_x(nh) + 24) - (double)H.goal_x[0], dy = (double)(reach_node_y(nh) + 24) - (double)H.goal_y[0];
            const double dist = std::sqrt(dx * dx + dy * dy);
            if (dist > 0.001) { H.exit_path[0] = (float)(dx / dist); H.exit_path[1] = (float)(dy / dist); }
        }
        if (R.dist[1][sn] != INFINITY && R.mh[1][2 * sn] == R.mh[1][2 * sn]) {
            H.exit_path[2] = (float)R.mh[1][2 * sn];
            H.exit_path[3] = (float)R.mh[1][2 * sn + 1];
        }
    }
    // ---- mine SDF (mine_proximity_cache.py:315-380): every toggle mine counts as deadly at level load (type 1 starts in state
    //      0, type 21 is forced to state 0 by the entity extractor)
    R.has_sdf = H.n_mines > 0;
    if (R.has_sdf) {
        R.sdf.assign(SDF_W * SDF_H, 1.f);
        R.grad.assign(2 * SDF_W * SDF_H, 0.f);
        std::vector<int> all = mines1;
        all.insert(all.end(), mines21.begin(), mines21.end());
        for (int row = 0; row < SDF_H; row++)
            for (int col = 0; col < SDF_W; col++) {
                const double cx = (col + 0.5) * 12.0, cy = (row + 0.5) * 12.0;
                double best = INFINITY, bx = 0.0, by = 0.0;
                for (int s : all) {
                    const double dx = cx - L.ent_x[s], dy = cy - L.ent_y[s], d = std::sqrt(dx * dx + dy * dy);
                    if (d < best) { best = d; bx = dx; by = dy; }
                }
                double v;
                if (best <= 20.0) v = best / 20.0 - 1.0;
                else { v = (best - 20.0) / 40.0; v = v < 1.0 ? v : 1.0; }
                R.sdf[row * SDF_W + col] = (float)v;
                if (best > 1e-6) {
                    R.grad[(row * SDF_W + col) * 2] = (float)(bx / best);
                    R.grad[(row * SDF_W + col) * 2 + 1] = (float)(by / best);
                }
            }
    }
}

// pack one level's tables behind `hdr` into `blob` (16-byte aligned sections); offsets are relative to the blob start
void pack_reach(const ReachBuilt &R, ReachHdr &hdr, std::vector<unsigned char> &blob) {
    hdr = R.hdr;
    const size_t base = (blob.size() + 15) / 16 * 16;   // this level's offsets are relative to `base` (stored in hdr.base)
    blob.resize(base + 16, 0);                           // offset 0 means "absent"
    hdr.base = base;
    auto append = [&](const void *p, size_t bytes) -> uint32_t {
        const size_t off = (blob.size() + 15) / 16 * 16;
        blob.resize(off + bytes);
        std::memcpy(blob.data() + off, p, bytes);
        return (uint32_t)(off - base);
    };
    hdr.off_in = append(R.in.data(), RNODES);
    std::vector<double> d2(2 * RNODES);
    std::memcpy(d2.data(), R.dist[0].data(), 8 * RNODES);
    std::memcpy(d2.data() + RNODES, R.dist[1].data(), 8 * RNODES);
    hdr.off_dist = append(d2.data(), 16 * RNODES);
    std::vector<int16_t> h2(2 * RNODES);
    std::memcpy(h2.data(), R.hop[0].data(), 2 * RNODES);
    std::memcpy(h2.data() + RNODES, R.hop[1].data(), 2 * RNODES);
    hdr.off_hop = append(h2.data(), 4 * RNODES);
    std::vector<double> m2(4 * RNODES);
    std::memcpy(m2.data(), R.mh[0].data(), 16 * RNODES);
    std::memcpy(m2.data() + 2 * RNODES, R.mh[1].data(), 16 * RNODES);
    hdr.off_mh = append(m2.data(), 32 * RNODES);
    hdr.off_mine_mask = append(R.mine_mask.data(), 4 * R.mine_mask.size());
    hdr.n_words = (uint32_t)R.mine_mask.size();
    hdr.off_sdf = hdr.off_grad = 0;
    if (R.has_sdf) {
        hdr.off_sdf = append(R.sdf.data(), 4 * R.sdf.size());
        hdr.off_grad = append(R.grad.data(), 4 * R.grad.size());
    }
}

}  // namespace npp

using namespace npp;

extern "C" {

int npp_reach_compile(const double *map, int64_t n, int32_t *info, uint8_t *base_in, uint8_t *base_adj, uint8_t *phys, uint8_t *in,
                      uint8_t *adj, double *dist, int16_t *hop, double *mh, float *sdf, float *grad, double *scalars) {
    if (!map) return NPP_ERR_INVALID;
    ReachBuilt R;
    std::string err;
    if (!build_reach(map, n, R, err)) return NPP_ERR_INVALID;
    const ReachHdr &H = R.hdr;
    if (info) {
        info[0] = (int32_t)H.supported; info[1] = (int32_t)H.n_adj;
        for (int k = 0; k < 4; k++) info[2 + k] = H.goal_node[k];
        info[6] = H.goal_x[0]; info[7] = H.goal_y[0]; info[8] = H.goal_x[1]; info[9] = H.goal_y[1];
        info[10] = H.exit_gid; info[11] = H.n_mines; info[12] = R.has_sdf ? 1 : 0; info[13] = (int32_t)R.surface_area;
    }
    if (base_in) std::memcpy(base_in, R.base_in.data(), RNODES);
    if (base_adj) std::memcpy(base_adj, R.base_adj.data(), RNODES);
    if (phys) std::memcpy(phys, R.phys.data(), RNODES);
    if (in) std::memcpy(in, R.in.data(), RNODES);
    if (adj) std::memcpy(adj, R.adj.data(), RNODES);
    for (int g = 0; g < 2; g++) {
        if (dist) std::memcpy(dist + (size_t)g * RNODES, R.dist[g].data(), 8 * RNODES);
        if (hop) std::memcpy(hop + (size_t)g * RNODES, R.hop[g].data(), 2 * RNODES);
        if (mh) std::memcpy(mh + (size_t)g * 2 * RNODES, R.mh[g].data(), 16 * RNODES);
    }
    if (sdf && R.has_sdf) std::memcpy(sdf, R.sdf.data(), 4 * R.sdf.size());
    if (grad && R.has_sdf) std::memcpy(grad, R.grad.data(), 4 * R.grad.size());
    if (scalars) {
        scalars[0] = H.area_scale; scalars[1] = H.f0; scalars[2] = H.exit_reachable;
        for (int k = 0; k < 4; k++) scalars[3 + k] = H.exit_path[k];
    }
    return NPP_OK;
}

int npp_reach_features_host(const double *map, int64_t n, const double *pos, const int32_t *mines, int count, float *out, float *sdf_out,
                            int32_t *status) {
    if (!map || !pos || !out || count < 0) return NPP_ERR_INVALID;
    ReachBuilt R;
    std::string err;
    if (!build_reach(map, n, R, err)) return NPP_ERR_INVALID;
    ReachHdr H;
    std::vector<unsigned char> blob;
    pack_reach(R, H, blob);
    ReachTabs T{&H, blob.data() + H.base};
    for (int k = 0; k < count; k++) {
        float sd[3];
        const int st = reach_features(T, pos[2 * k], pos[2 * k + 1], mines ? mines[2 * k] : H.n_mines, mines ? mines[2 * k + 1] : 0,
                                      out + (size_t)k * REACH_DIM, sd);
        if (sdf_out) { sdf_out[3 * k] = sd[0]; sdf_out[3 * k + 1] = sd[1]; sdf_out[3 * k + 2] = sd[2]; }
        if (status) status[k] = st | (H.supported ? 0 : 2);
    }
    return NPP_OK;
}

}  // extern "C"
