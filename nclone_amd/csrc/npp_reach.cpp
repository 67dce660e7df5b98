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


// ---- the cache-miss branch of CachedPathDistanceCalculator.get_distance (path_distance_calculator.py:1218-1485) ------------------
// CPython's float ** float is libm pow(); glibc's pow(x, 2.0) / pow(x, 0.5) differ from x * x / sqrt(x) in the last bit for ~0.1 % of
// the inputs, and the compiler would fold a direct call: go through a volatile pointer.
double (*volatile py_pow)(double, double) = static_cast<double (*)(double, double)>(std::pow);

// MineProximityCostCache._precompute_mine_proximity_costs (mine_proximity_cache.py:70-196): node positions are in tile-data space,
// mine positions in world space, and the reference subtracts them as they are
void mine_multipliers(const Graph &fin, const CompiledLevel &L, const std::vector<int> &mines, std::vector<double> &mult) {
    mult.assign(RNODES, 1.0);
    if (mines.empty()) return;
    for (int id = 0; id < RNODES; id++) {
        if (!fin.in[id]) continue;
        double best = INFINITY;
        for (int s : mines) {
            const double dx = (double)reach_node_x(id) - L.ent_x[s], dy = (double)reach_node_y(id) - L.ent_y[s];
            const double d = py_pow(dx * dx + dy * dy, 0.5);
            best = d < best ? d : best;   // min(min_distance, distance)
        }
        if (best < 48.0) {   // MINE_HAZARD_RADIUS
            const double pf = 1.0 - (best / 48.0);
            mult[id] = 1.0 + py_pow(pf, 2.0) * (1.1 - 1.0);   // MINE_HAZARD_COST_MULTIPLIER - 1.0
        }
    }
}

struct AStarCtx {
    const Graph *adj;                  // masked + flood-filled adjacency (neighbour lists)
    const std::vector<uint8_t> *ph;    // physics cache (base graph): bit 0 grounded, bit 1 walled
    const std::vector<double> *mine;   // mine proximity multiplier per node
    const std::vector<float> *grad;    // mine SDF gradient grid [SDF_H][SDF_W][2], empty = zeros
};

// _get_aerial_chain_multiplier (pathfinding_utils.py:80-111)
double aerial_chain_multiplier(int chain) {
    if (chain <= 2) return py_pow(3.0, (double)(chain + 1));
    return 500.0 * py_pow(3.0, (double)(chain - 2));
}

// _calculate_physics_aware_cost (pathfinding_utils.py:174-539) with ninja_velocity = None and hazard_cost_multiplier = None, which is
// how compute_reachability_features_from_graph reaches it.  parent / grandparent: node ids, -1 = None.
double physics_cost(const AStarCtx &C, int src, int dst, int parent, int grandparent, int chain) {
    const int sx = reach_node_x(src), sy = reach_node_y(src), tx = reach_node_x(dst), ty = reach_node_y(dst);
    const int dx = tx - sx, dy = ty - sy;
    const bool src_g = ((*C.ph)[src] & 1) != 0, dst_g = ((*C.ph)[dst] & 1) != 0, src_w = ((*C.ph)[src] & 2) != 0;
    bool x_change = false, y_to_rising = false, from_hair = false;
    int pdx = 0, pdy = 0, px = 0;
    if (parent >= 0) {
        px = reach_node_x(parent);
        pdx = sx - px; pdy = sy - reach_node_y(parent);
        if (pdx != 0 && dx != 0) x_change = (pdx > 0) != (dx > 0);
        if (pdy >= 0 && dy < 0) y_to_rising = true;
        if (pdy == 0 && !src_g) from_hair = true;
    }
    const double base_cost = (dx != 0 && dy != 0) ? 1.414 : 1.0;
    double mult;
    if (dx != 0 && dy > 0 && !src_g && !src_w) mult = 0.15;
    else if (dx != 0 && dy < 0 && !src_g && src_w) {
        if (from_hair || y_to_rising) mult = INFINITY;
        else if (x_change) mult = 1.5;
        else mult = 1.0;
    } else if (dx != 0 && dy < 0 && src_g) mult = 0.2;
    else if (dx != 0 && dy < 0 && !src_g && !src_w) {
        if (from_hair) mult = INFINITY;
        else if (chain > 0 && parent >= 0 && pdy >= 0) mult = INFINITY;
        else if (y_to_rising) mult = INFINITY;
        else mult = 1.0;   // both remaining branches of the reference
    } else if (dy < 0) {
        if (src_g) mult = 0.7;
        else if (src_w) mult = (from_hair || y_to_rising) ? INFINITY : 1.2;
        else if (from_hair) mult = INFINITY;
        else if (chain > 0 && parent >= 0 && pdy >= 0) mult = INFINITY;
        else if (y_to_rising) mult = INFINITY;
        else mult = aerial_chain_multiplier(chain);
    } else if (dy > 0) mult = 0.5;
    else {
        if (src_g && dst_g) mult = 0.15;
        else if (x_change && !src_g && !src_w) mult = 1.5;
        else if (!src_g && !src_w) mult = 0.5;
        else mult = 1.0;
    }
    double momentum = 1.0;
    if (dy == 0 && src_g && dst_g && grandparent >= 0 && parent >= 0) {
        const int recent = sx - px, prev = px - reach_node_x(grandparent);
        if (recent * prev > 0 && std::abs(recent) >= 12 && dx != 0) {
            const int dir = recent < 0 ? -1 : 1;
            momentum = dir * dx > 0 ? 0.7 : 2.5;
        }
    }
    double mine = (*C.mine)[dst];
    if (parent >= 0 && !C.grad->empty()) {   // velocity-aware mine cost from the SDF gradient at the destination (tile-data coordinates)
        int col = (int)((double)tx / 12.0), row = (int)((double)ty / 12.0);
        col = col < 0 ? 0 : (col > SDF_W - 1 ? SDF_W - 1 : col);
        row = row < 0 ? 0 : (row > SDF_H - 1 ? SDF_H - 1 : row);
        const double gx = (double)(*C.grad)[(row * SDF_W + col) * 2], gy = (double)(*C.grad)[(row * SDF_W + col) * 2 + 1];
        if (gx * gx + gy * gy > 0.0001) {
            const double vmag = py_pow((double)(dx * dx + dy * dy), 0.5);
            const double vx = (double)dx / vmag, vy = (double)dy / vmag;
            const double toward = -(vx * gx + vy * gy);
            if (toward > 0.3) mine *= 1.0 + toward * vmag * 0.16667;
        }
    }
    return base_cost * mult * momentum * mine * 1.0;
}

struct AQ {
    double f, g;
    int x, y, id;
    bool operator<(const AQ &o) const {   // max-heap of std::priority_queue inverted: heapq order of (f, g, (x, y))
        if (f != o.f) return f > o.f;
        if (g != o.g) return g > o.g;
        if (x != o.x) return x > o.x;
        return y > o.y;
    }
};

// CachedPathDistanceCalculator._calculate_distance -> _astar_distance (path_distance_calculator.py:581-657, 744-845)
struct AStar {
    std::vector<double> g;
    std::vector<uint8_t> has, visited;
    std::vector<int16_t> parent, grand, chain;
    AStar() : g(RNODES), has(RNODES), visited(RNODES), parent(RNODES), grand(RNODES), chain(RNODES) {}
    double run(const AStarCtx &C, int start, int goal) {
        const Graph &A = *C.adj;
        if (!A.in[start] || !A.in[goal]) return INFINITY;
        if (start == goal) return 0.0;
        std::fill(has.begin(), has.end(), 0);
        std::fill(visited.begin(), visited.end(), 0);
        const int gx = reach_node_x(goal), gy = reach_node_y(goal);
        auto h = [&](int id) { return (double)(std::abs(reach_node_x(id) - gx) + std::abs(reach_node_y(id) - gy)); };
        std::priority_queue<AQ> open;
        open.push({h(start), 0.0, reach_node_x(start), reach_node_y(start), start});
        g[start] = 0.0; has[start] = 1; parent[start] = -1; grand[start] = -1; chain[start] = 0;
        while (!open.empty()) {
            const AQ cur = open.top();
            open.pop();
            if (visited[cur.id]) continue;
            visited[cur.id] = 1;
            if (cur.id == goal) return cur.g;
            const bool grounded = ((*C.ph)[cur.id] & 1) != 0;
            const int cchain = chain[cur.id], cpar = parent[cur.id], cgrand = grand[cur.id];
            const int i = cur.id / RH, j = cur.id % RH;
            for (int d = 0; d < 8; d++) {
                if (!A.edge(cur.id, d)) continue;
                const int nb = nid(i + DX[d], j + DY[d]);
                if (visited[nb]) continue;
                // _violates_horizontal_rule (pathfinding_utils.py:582-630)
                if (DY[d] == 0 && !(grounded && ((*C.ph)[nb] & 1)) && cpar >= 0 && reach_node_y(cpar) == reach_node_y(cur.id) &&
                    reach_node_x(cpar) != reach_node_x(cur.id))
                    continue;
                const int nchain = (!grounded && DY[d] < 0) ? cchain + 1 : 0;
                const double cost = physics_cost(C, cur.id, nb, cpar, cgrand, cchain);
                const double tg = cur.g + cost;
                if (!has[nb] || tg < g[nb]) {
                    g[nb] = tg; has[nb] = 1;
                    parent[nb] = (int16_t)cur.id; grand[nb] = (int16_t)cpar; chain[nb] = (int16_t)nchain;
                    open.push({tg + h(nb), tg, reach_node_x(nb), reach_node_y(nb), nb});
                }
            }
        }
        return INFINITY;
    }
};

// find_goal_node_closest_to_start with get_distance's ladder of search radii (path_distance_calculator.py:1290-1372): the nodes
// within the radius of the goal (dict order), those within the entity radius preferred, the one closest to the start node wins
int goal_node_for_start(const std::vector<int> &order, int qx, int qy, int entity_r, int start) {
    static const double RADII[3] = {-1.0, 48.0, 150.0};
    const long sx = reach_node_x(start), sy = reach_node_y(start);
    for (int step = 0; step < 3; step++) {
        const double R = RADII[step] < 0.0 ? 10.0 + (double)entity_r : RADII[step];
        int best = -1, best_o = -1;
        long bd = 0, bdo = 0;
        for (int id : order) {
            const long ex = reach_node_x(id) - qx, ey = reach_node_y(id) - qy, d2 = ex * ex + ey * ey;
            if ((double)d2 > R * R) continue;
            const long tx = reach_node_x(id) - sx, ty = reach_node_y(id) - sy, ds = tx * tx + ty * ty;
            if (d2 <= (long)entity_r * entity_r) {   // ((..) ** 0.5 <= entity_radius on integer sums)
                if (best_o < 0 || ds < bdo) { best_o = id; bdo = ds; }
            } else if (best < 0 || ds < bd) { best = id; bd = ds; }
        }
        if (best_o >= 0) return best_o;
        if (best >= 0) return best;
    }
    int best = -1;
    long bd = 0;
    for (int id : order) {   // "ANY closest node in the entire adjacency graph"
        const long ex = reach_node_x(id) - qx, ey = reach_node_y(id) - qy, d2 = ex * ex + ey * ey;
        if (best < 0 || d2 < bd) { best = id; bd = d2; }
    }
    return best;
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
        R.spawn_area = cnt;
        if (cnt == 0 && n_adj) {   // the PBRS calculator's own flood fill starts at int(spawn) (pbrs_potentials.py:876-886); it only
            std::vector<uint8_t> r2;   // runs when the feature code's call above failed (both share one cache entry per level)
            R.spawn_area = flood_fill(fin, R.phys, (double)(long)L.spawn_x, (double)(long)L.spawn_y, r2);
        }
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
        // validation against the switch's cached position (path_distance_calculator.py:1046-1072) fails beyond 12 px: every exit-door
        // query then falls through to the cache-miss branch -- tabulated at the end of this function
        if (std::abs(H.goal_x[0] - H.goal_x[1]) > 12 || std::abs(H.goal_y[0] - H.goal_y[1]) > 12) H.miss_exit = 1;
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
    // ---- the cache-miss branch for the exit door (see ReachHdr::miss_exit).  Everything it computes is a function of the level and
    //      of the ninja position alone: goal node = f(temp start node), cost = physics A* (start node -> goal node).  The per-episode
    //      (start cell, goal cell) cache in front of it is the only history, and lives per env on the device.
    {
        std::vector<int> all = mines1;
        all.insert(all.end(), mines21.begin(), mines21.end());
        mine_multipliers(fin, L, all, R.mine_mult);
    }
    H.sw_alias = (H.sw_valid && H.ex_valid && H.goal_x[0] / 24 == H.goal_x[1] / 24 && H.goal_y[0] / 24 == H.goal_y[1] / 24) ? 1u : 0u;
    if (H.miss_exit && H.supported) {
        std::vector<int> order;
        for (int id = 0; id < RNODES; id++)
            if (fin.in[id]) order.push_back(id);
        std::sort(order.begin(), order.end(), [](int a, int b) { return reach_order_key(a) < reach_order_key(b); });
        R.cgoal.assign(RNODES, 0xff);
        const int qx = H.goal_x[1] - 24, qy = H.goal_y[1] - 24;
        std::vector<int> cand;
        for (int t : order) {
            const int c = goal_node_for_start(order, qx, qy, 12, t);
            size_t k = std::find(cand.begin(), cand.end(), c) - cand.begin();
            if (k == cand.size()) cand.push_back(c);
            R.cgoal[t] = (uint8_t)(k < 0xff ? k : 0xfe);
        }
        if (cand.size() > (size_t)REACH_MAX_CAND) {
            H.supported = 0;
            R.note = "more than 16 goal nodes around the exit door";
        } else {
            H.n_cand = (uint32_t)cand.size();
            for (size_t k = 0; k < cand.size(); k++) H.cand[k] = cand[k];
            R.astar.assign(cand.size() * (size_t)RNODES, NAN);
            const AStarCtx C{&fin, &R.phys, &R.mine_mult, &R.grad};
            AStar A;
            // start node and temp start node of one query both lie within a tile of the ninja: tabulate every start node within
            // four lattice steps of a temp start node that selects the goal node (anything else reads NaN -> status bit 0)
            for (size_t k = 0; k < cand.size(); k++) {
                std::vector<uint8_t> want(RNODES, 0);
                for (int t : order) {
                    if (R.cgoal[t] != k) continue;
                    const int ti = t / RH, tj = t % RH;
                    for (int i = std::max(0, ti - 4); i <= std::min(RW - 1, ti + 4); i++)
                        for (int j = std::max(0, tj - 4); j <= std::min(RH - 1, tj + 4); j++)
                            if (fin.in[nid(i, j)]) want[nid(i, j)] = 1;
                }
                for (int s2 : order)
                    if (want[s2]) R.astar[k * RNODES + s2] = A.run(C, s2, cand[k]);
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
    hdr.off_cgoal = hdr.off_astar = 0;
    if (R.hdr.miss_exit && !R.astar.empty()) {
        hdr.off_cgoal = append(R.cgoal.data(), RNODES);
        hdr.off_astar = append(R.astar.data(), 8 * R.astar.size());
    }
    hdr.off_sdf = hdr.off_grad = 0;
    if (R.has_sdf) {
        hdr.off_sdf = append(R.sdf.data(), 4 * R.sdf.size());
        hdr.off_grad = append(R.grad.data(), 4 * R.grad.size());
    }
    std::vector<ReachRec> rec(RNODES);
    std::memset(rec.data(), 0, sizeof(ReachRec) * RNODES);
    for (int id = 0; id < RNODES; id++) {
        ReachRec &q = rec[id];
        q.dist0 = R.dist[0][id]; q.dist1 = R.dist[1][id];
        q.mh0x = R.mh[0][2 * id]; q.mh0y = R.mh[0][2 * id + 1];
        q.hop0 = R.hop[0][id]; q.hop1 = R.hop[1][id];
        q.in = R.in[id];
        q.cgoal = hdr.off_cgoal ? R.cgoal[id] : 0xff;
    }
    hdr.off_rec = append(rec.data(), sizeof(ReachRec) * RNODES);
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

int npp_reach_compile_miss(const double *map, int64_t n, int32_t *info, uint8_t *cgoal, double *astar, double *mine_mult) {
    if (!map) return NPP_ERR_INVALID;
    ReachBuilt R;
    std::string err;
    if (!build_reach(map, n, R, err)) return NPP_ERR_INVALID;
    const ReachHdr &H = R.hdr;
    if (info) {
        info[0] = (int32_t)H.miss_exit; info[1] = (int32_t)H.n_cand; info[2] = (int32_t)H.sw_alias; info[3] = (int32_t)H.supported;
        for (int k = 0; k < REACH_MAX_CAND; k++) info[4 + k] = k < (int)H.n_cand ? H.cand[k] : -1;
    }
    if (cgoal) {
        if (R.cgoal.empty()) std::memset(cgoal, 0xff, RNODES);
        else std::memcpy(cgoal, R.cgoal.data(), RNODES);
    }
    if (astar && !R.astar.empty()) std::memcpy(astar, R.astar.data(), 8 * R.astar.size());
    if (mine_mult) std::memcpy(mine_mult, R.mine_mult.data(), 8 * RNODES);
    return NPP_OK;
}

int npp_reach_rollout_host(const double *map, int64_t n, const double *pos, const int32_t *mines, const uint8_t *new_episode, int count,
                           float *out, int32_t *status, double *raw_out) {
    if (!map || !pos || !out || count < 0) return NPP_ERR_INVALID;
    ReachBuilt R;
    std::string err;
    if (!build_reach(map, n, R, err)) return NPP_ERR_INVALID;
    ReachHdr H;
    std::vector<unsigned char> blob;
    pack_reach(R, H, blob);
    ReachTabs T{&H, blob.data() + H.base};
    std::vector<uint32_t> stamp(REACH_CELLS, 0u);
    std::vector<double> raw(REACH_CELLS, 0.0);
    ReachMiss M{stamp.data(), raw.data(), 1u};
    for (int k = 0; k < count; k++) {
        if (new_episode && new_episode[k]) M.epoch++;   // clear_cache() of the path calculator (reachability_mixin.py:67-70)
        const int st = reach_features(T, pos[2 * k], pos[2 * k + 1], mines ? mines[2 * k] : H.n_mines, mines ? mines[2 * k + 1] : 0,
                                      out + (size_t)k * REACH_DIM, nullptr, &M);
        if (status) status[k] = st | (H.supported ? 0 : 2);
        if (raw_out) {   // the dictionary entry of the ninja's cell after this query (NaN = none)
            int cx = reach_cell24(pos[2 * k]), cy = reach_cell24(pos[2 * k + 1]);
            cx = cx < 0 ? 0 : (cx > 43 ? 43 : cx);
            cy = cy < 0 ? 0 : (cy > 24 ? 24 : cy);
            raw_out[k] = stamp[cx * 25 + cy] == M.epoch ? raw[cx * 25 + cy] : NAN;
        }
    }
    return NPP_OK;
}

}  // extern "C"
