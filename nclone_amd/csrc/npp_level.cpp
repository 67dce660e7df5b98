// npp_level.cpp -- see npp_level.hpp.  Host C++ only (no HIP).
#include "npp_level.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

namespace npp {
namespace {

// Orthogonal half-edge contributions of tile types 0..33 as two 12-bit masks (bit k = k-th entry of the
// reference's per-tile list, nclone/tile_definitions.py:137-185: 6 horizontal half-edges left->right,
// top->bottom, then 6 vertical ones top->bottom, left->right).  NEG = normal up/left (-1), POS = down/right (+1).
const uint16_t ORTHO_NEG[34] = {0x000, 0x0c3, 0x043, 0x302, 0x08c, 0x0c1, 0x0c3, 0x003, 0x000, 0x0c0, 0x0c3, 0x003,
                                0x000, 0x0c0, 0x0c3, 0x003, 0x000, 0x0c0, 0x043, 0x003, 0x000, 0x080, 0x0c3, 0x043,
                                0x080, 0x0c0, 0x0c1, 0x002, 0x000, 0x0c0, 0x0c3, 0x003, 0x002, 0x0c1};
const uint16_t ORTHO_POS[34] = {0x000, 0xc30, 0x40c, 0xc20, 0x830, 0x310, 0x000, 0xc00, 0xc30, 0x030, 0x000, 0xc00,
                                0xc30, 0x030, 0x000, 0xc00, 0xc30, 0x030, 0x000, 0x400, 0x830, 0x030, 0x400, 0xc00,
                                0xc30, 0x830, 0x000, 0xc00, 0xc20, 0x010, 0x010, 0xc20, 0xc30, 0x030};

// Grid-edge toggles of tile types 0..33 (nclone/tile_definitions.py TILE_GRID_EDGE_MAP, same 6 + 6 layout, bit k =
// k-th entry): what drones, thwumps and shove thwumps test with is_empty_row / is_empty_column (physics.py:210-235).
const uint16_t GRID_EDGE[34] = {0x000, 0xcf3, 0x44f, 0xf22, 0x8bc, 0x3d1, 0x6db, 0xe67, 0xdb6, 0x9f9, 0xcf3, 0xcf3,
                                0xcf3, 0xcf3, 0x6db, 0xe67, 0xdb6, 0x9f9, 0x44f, 0x44f, 0x8bc, 0x8bc, 0xcf3, 0xcf3,
                                0xcf3, 0xcf3, 0x3d1, 0xf22, 0xf22, 0x3d1, 0xcf3, 0xcf3, 0xcf3, 0xcf3};

// Non-orthogonal piece of a tile, already in packed-segment units (12 px): 0 = none.
// Diagonals (tile_definitions.py:189-210) and quarter circles (:214-223).
uint16_t pack_linear(int x1, int y1, int x2, int y2) {
    return (uint16_t)(0u | (x1 << 2) | (y1 << 4) | (x2 << 6) | (y2 << 8));
}
uint16_t pack_arc(int cx, int cy, int hor, int ver, int convex) {
    return (uint16_t)(1u | (cx << 2) | (cy << 4) | ((hor > 0) << 6) | ((ver > 0) << 7) | ((convex != 0) << 8));
}

bool tile_special(int t, uint16_t &seg) {
    switch (t) {
        case 6: seg = pack_linear(0, 2, 2, 0); return true;
        case 7: seg = pack_linear(0, 0, 2, 2); return true;
        case 8: seg = pack_linear(2, 0, 0, 2); return true;
        case 9: seg = pack_linear(2, 2, 0, 0); return true;
        case 10: seg = pack_arc(0, 0, 1, 1, 1); return true;
        case 11: seg = pack_arc(2, 0, -1, 1, 1); return true;
        case 12: seg = pack_arc(2, 2, -1, -1, 1); return true;
        case 13: seg = pack_arc(0, 2, 1, -1, 1); return true;
        case 14: seg = pack_arc(2, 2, -1, -1, 0); return true;
        case 15: seg = pack_arc(0, 2, 1, -1, 0); return true;
        case 16: seg = pack_arc(0, 0, 1, 1, 0); return true;
        case 17: seg = pack_arc(2, 0, -1, 1, 0); return true;
        case 18: seg = pack_linear(0, 1, 2, 0); return true;
        case 19: seg = pack_linear(0, 0, 2, 1); return true;
        case 20: seg = pack_linear(2, 1, 0, 2); return true;
        case 21: seg = pack_linear(2, 2, 0, 1); return true;
        case 22: seg = pack_linear(0, 2, 2, 1); return true;
        case 23: seg = pack_linear(0, 1, 2, 2); return true;
        case 24: seg = pack_linear(2, 0, 0, 1); return true;
        case 25: seg = pack_linear(2, 1, 0, 0); return true;
        case 26: seg = pack_linear(0, 2, 1, 0); return true;
        case 27: seg = pack_linear(1, 0, 2, 2); return true;
        case 28: seg = pack_linear(2, 0, 1, 2); return true;
        case 29: seg = pack_linear(1, 2, 0, 0); return true;
        case 30: seg = pack_linear(1, 2, 2, 0); return true;
        case 31: seg = pack_linear(0, 0, 1, 2); return true;
        case 32: seg = pack_linear(1, 0, 0, 2); return true;
        case 33: seg = pack_linear(2, 2, 1, 0); return true;
        default: return false;
    }
}

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct RawEnt {
    uint32_t kind;
    double x, y;
    int cell;
    uint32_t init;
    int link_raw;  // index into raw list (door of a switch), -1 otherwise
    uint32_t type = 0;  // Entity.type (1/21 mine, 2 gold, 3 exit door, 4 exit switch, 6 locked-door switch)
    int seq = 0;        // creation order
    int extra = -1;     // link field for kinds without a linked entity: door table index / orientation
};

struct RawMover {
    uint32_t kind, orientation, mode, type;
    double x, y;
    int seq;
};

// physics.py:317-332
void orientation_vector(int o, double &vx, double &vy) {
    const double diag = std::sqrt(2.0) / 2;
    static const int sx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, sy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    o &= 7;
    vx = (o & 1) ? sx[o] * diag : sx[o];
    vy = (o & 1) ? sy[o] * diag : sy[o];
}

void seg_bounds_units(uint16_t s, int &x0, int &y0, int &x1, int &y1) {
    if ((s & 1u) == 0) {
        int ax = (s >> 2) & 3, ay = (s >> 4) & 3, bx = (s >> 6) & 3, by = (s >> 8) & 3;
        x0 = std::min(ax, bx); x1 = std::max(ax, bx);
        y0 = std::min(ay, by); y1 = std::max(ay, by);
    } else {
        // entities.py:119-125: centre, centre + 24*hor, centre + 24*ver
        int cx = (s >> 2) & 3, cy = (s >> 4) & 3;
        int hx = cx + (((s >> 6) & 1) ? 2 : -2), vy = cy + (((s >> 7) & 1) ? 2 : -2);
        x0 = std::min(cx, hx); x1 = std::max(cx, hx);
        y0 = std::min(cy, vy); y1 = std::max(cy, vy);
    }
}

}  // namespace

bool compile_level(const double *map, int64_t n, CompiledLevel &L, std::string &err) {
    if (n < 1233) {
        err = "map_data shorter than 1233 values";
        return false;
    }
    // map_data holds bytes (and, for generated maps, small fractional coordinates): anything non-finite or beyond 16 bits is a
    // corrupted blob -- refused here so that no later stage has to convert an out-of-range double to an integer
    for (int64_t i = 0; i < n; i++)
        if (!(std::fabs(map[i]) <= 65535.0)) {
            err = "map_data[" + std::to_string(i) + "] is not a finite value within +-65535";
            return false;
        }
    L = CompiledLevel();
    // ---- tiles: map_data[184:1150] is the 42x23 interior, index x + 42*y; border cells are solid (map_loader.py:22-37)
    L.tiles.assign(N_CELLS, 1);
    for (int x = 0; x < 42; x++)
        for (int y = 0; y < 23; y++) {
            double v = map[184 + x + 42 * y];
            int t = (v == std::floor(v) && v >= 0 && v <= 254) ? (int)v : 255;
            L.tiles[(x + 1) * GRID_H + (y + 1)] = (uint8_t)t;
        }
    // ---- accumulate signed half-edges on the 89x51 half-cell lattice (tile_segment_factory.py:190-208)
    static thread_local int8_t hsum[89][51], vsum[89][51];
    std::memset(hsum, 0, sizeof(hsum));
    std::memset(vsum, 0, sizeof(vsum));
    std::vector<std::vector<uint16_t>> cell(N_CELLS);
    for (int x = 0; x < GRID_W; x++)
        for (int y = 0; y < GRID_H; y++) {
            int t = L.tiles[x * GRID_H + y];
            if (t == 0 || t >= 34) continue;  // 34..37 glitched tiles are empty (tile_segment_factory.py:184-186)
            uint16_t ng = ORTHO_NEG[t], ps = ORTHO_POS[t];
            for (int k = 0; k < 6; k++) {
                int s = ((ps >> k) & 1) - ((ng >> k) & 1);
                hsum[2 * x + (k & 1)][2 * y + (k >> 1)] += (int8_t)s;
            }
            for (int k = 0; k < 6; k++) {
                int s = ((ps >> (k + 6)) & 1) - ((ng >> (k + 6)) & 1);
                vsum[2 * x + (k >> 1)][2 * y + (k & 1)] += (int8_t)s;
            }
            uint16_t sp;
            if (tile_special(t, sp)) cell[x * GRID_H + y].push_back(sp);  // always first in its cell
        }
    // ---- surviving half-edges become segments owned by a cell (tile_segment_factory.py:231-262); the lattice is
    //      walked x-major because the reference pre-seeds its dicts that way (nsim.py:218-219)
    for (int xc = 0; xc < 89; xc++)
        for (int yc = 0; yc < 51; yc++) {
            int st = hsum[xc][yc];
            if (st == 0) continue;
            int cx = (int)std::floor(xc / 2.0);
            int cy = (int)std::floor((yc - 0.1 * st) / 2);
            if (cx < 0 || cx >= GRID_W || cy < 0 || cy >= GRID_H) continue;
            int ux = xc - 2 * cx, uy = yc - 2 * cy;  // units of 12 px inside the owning cell
            cell[cx * GRID_H + cy].push_back(st == -1 ? pack_linear(ux + 1, uy, ux, uy) : pack_linear(ux, uy, ux + 1, uy));
        }
    for (int xc = 0; xc < 89; xc++)
        for (int yc = 0; yc < 51; yc++) {
            int st = vsum[xc][yc];
            if (st == 0) continue;
            int cx = (int)std::floor((xc - 0.1 * st) / 2);
            int cy = (int)std::floor(yc / 2.0);
            if (cx < 0 || cx >= GRID_W || cy < 0 || cy >= GRID_H) continue;
            int ux = xc - 2 * cx, uy = yc - 2 * cy;
            cell[cx * GRID_H + cy].push_back(st == -1 ? pack_linear(ux, uy, ux, uy + 1) : pack_linear(ux, uy + 1, ux, uy));
        }
    L.seg_start.assign(N_CELLS + 1, 0);
    L.cell_bounds.assign(N_CELLS, 0);
    for (int c = 0; c < N_CELLS; c++) {
        L.seg_start[c] = (uint16_t)L.segs.size();
        int bx0 = 3, by0 = 3, bx1 = 0, by1 = 0;
        for (uint16_t s : cell[c]) {
            int x0, y0, x1, y1;
            seg_bounds_units(s, x0, y0, x1, y1);
            if (x0 < 0 || y0 < 0 || x1 > 2 || y1 > 2) {
                err = "internal: segment leaves its cell";
                return false;
            }
            bx0 = std::min(bx0, x0); by0 = std::min(by0, y0);
            bx1 = std::max(bx1, x1); by1 = std::max(by1, y1);
            L.segs.push_back((uint16_t)(s | (uint16_t)((c % GRID_H) << 11)));   // bits 11-15: the owning cell's y
        }
        if (!cell[c].empty()) L.cell_bounds[c] = (uint8_t)(bx0 | (by0 << 2) | (bx1 << 4) | (by1 << 6));
    }
    if (L.segs.size() > 65000) {
        err = "too many segments";
        return false;
    }
    L.seg_start[N_CELLS] = (uint16_t)L.segs.size();

    // ---- entities (map_loader.py:84-141, entity_factory.py:145-233)
    L.spawn_x = map[1231] * 6;
    L.spawn_y = map[1232] * 6;
    std::vector<RawEnt> raw;
    std::vector<RawMover> movers;
    int next_seq = 0;       // creation order over static entities and movers alike
    size_t seq_done = 0;    // raw entries that already have their creation number
    std::vector<std::array<double, 5>> raw_doors;
    auto cell_of = [](double px, double py) {
        int cx = clampi((int)std::fmax(std::fmin(std::floor(px / 24), 1e6), -1e6), 0, 43);
        int cy = clampi((int)std::fmax(std::fmin(std::floor(py / 24), 1e6), -1e6), 0, 24);
        return cx * GRID_H + cy;
    };
    int64_t index = 1230;
    const double exit_count = n > 1156 ? map[1156] : 0;
    int last_switch_raw = -1;
    while (index < n) {
        if (index + 4 >= n) break;
        double tv = map[index];
        int type = (tv == std::floor(tv) && tv >= 0 && tv < 64) ? (int)tv : -1;
        double xc = map[index + 1], yc = map[index + 2];
        if (type == 1 || type == 21) {
            // type 1 starts toggled/deadly (state 0), type 21 starts untoggled (state 1): entity_factory.py:181-182,230-231
            raw.push_back({EK_MINE, xc * 6, yc * 6, cell_of(xc * 6, yc * 6), type == 1 ? 0u : 1u, -1, (uint32_t)type});
            L.n_thinkable++;
        } else if (type == 2) {
            raw.push_back({EK_GOLD, xc * 6, yc * 6, cell_of(xc * 6, yc * 6), 1u, -1, 2u});
        } else if (type == 3) {
            int64_t ci = index + 5 * (int64_t)exit_count;
            if (ci + 2 >= n || ci < 0) {
                err = "exit switch coordinates out of range";
                return false;
            }
            double sx = map[ci + 1] * 6, sy = map[ci + 2] * 6;
            raw.push_back({EK_EXIT, xc * 6, yc * 6, cell_of(xc * 6, yc * 6), 0u, -1, 3u});
            raw.push_back({EK_SWITCH, sx, sy, cell_of(sx, sy), 1u, (int)raw.size() - 1, 4u});
            last_switch_raw = (int)raw.size() - 1;
        } else if (type == 5 || type == 6 || type == 8) {
            // the entity lives at its switch (entity_door_base.py:94-97; a regular door is its own switch,
            // entity_factory.py:198-201); its door segment never reaches the ninja's region queries because the
            // spatial index is snapshotted before entities load.  What the door does change are the grid edges that
            // drones and thwumps test (entity_door_base.py:62-89).
            if (type != 5 && index + 7 >= n) {
                err = "door record truncated";
                return false;
            }
            double dx = xc * 6, dy = yc * 6, orient = map[index + 3];
            double sx = type == 5 ? dx : map[index + 6] * 6, sy = type == 5 ? dy : map[index + 7] * 6;
            bool vertical = (orient == 0 || orient == 4);
            double vx, vy;
            orientation_vector((int)orient, vx, vy);
            int dcx = clampi((int)std::floor((dx - 12 * vx) / 24), 0, 43), dcy = clampi((int)std::floor((dy - 12 * vy) / 24), 0, 24);
            int hx = 2 * (dcx + 1), hy = 2 * (dcy + 1);
            uint32_t k0, k1;
            if (vertical) { k0 = (uint32_t)(hx * EDGE_H + hy - 2) | 0x8000u; k1 = (uint32_t)(hx * EDGE_H + hy - 1) | 0x8000u; }
            else { k0 = (uint32_t)((hx - 2) * EDGE_H + hy); k1 = (uint32_t)((hx - 1) * EDGE_H + hy); }
            L.door_tab.push_back(k0 | (k1 << 16));
            // initial edge counter (trap doors start open, entity_door_trap.py:53) | door class << 8 (0 locked, 1 regular, 2 trap)
            L.door_tab.push_back((type == 8 ? 0u : 1u) | ((type == 5 ? 1u : (type == 8 ? 2u : 0u)) << 8));
            uint32_t kind = type == 5 ? EK_DOOR_REG : (type == 6 ? EK_LOCKED : EK_DOOR_TRAP);
            // 2-bit state: bit 0 active, bit 1 "closed" for regular doors (locked: closed == active, trap: closed == !active)
            raw.push_back({kind, sx, sy, cell_of(sx, sy), type == 5 ? 3u : 1u, -1, (uint32_t)type});
            raw.back().extra = (int)(L.door_tab.size() / 2) - 1;
            // the door's stroke (entity_door_base.py:52-85): 24 px long, vertical for orientations 0 and 4
            raw_doors.push_back({vertical ? dx : dx - 12, vertical ? dy - 12 : dy, vertical ? dx : dx + 12,
                                 vertical ? dy + 12 : dy, (double)(raw.size() - 1)});
            if (type != 6) L.has_zoo = true;
        } else if (type == 10 || type == 11) {
            // launch pad (entity_launch_pad.py) / one-way platform (entity_one_way_platform.py): static, oriented
            raw.push_back({type == 10 ? EK_LAUNCH : EK_ONEWAY, xc * 6, yc * 6, cell_of(xc * 6, yc * 6), 1u, -1, (uint32_t)type});
            raw.back().extra = ((int)map[index + 3]) & 7;
            L.has_zoo = true;
        } else if (type == 24) {
            // boost pad (entity_boost_pad.py): state 1 = not touching, 3 = touching the ninja
            raw.push_back({EK_BOOST, xc * 6, yc * 6, cell_of(xc * 6, yc * 6), 1u, -1, 24u});
            L.has_zoo = true;
        } else if (type == 14 || type == 17 || type == 20 || type == 25 || type == 26 || type == 28) {
            uint32_t mk = type == 14 ? MK_DRONE : type == 17 ? MK_BOUNCE : type == 20 ? MK_THWUMP : type == 25 ? MK_BALL
                          : type == 26 ? MK_MINI : MK_SHOVE;
            movers.push_back({mk, ((uint32_t)(int)map[index + 3]) & 7u, ((uint32_t)(int)map[index + 4]) & 3u, (uint32_t)type,
                              xc * 6, yc * 6, next_seq++});
            L.has_zoo = true;
        }
        for (; seq_done < raw.size(); seq_done++) raw[seq_done].seq = next_seq++;
        if (type == 6 || type == 8) {
            if (index + 9 < n && map[index + 7] != 0 && map[index + 8] == 0 && map[index + 9] == 0)
                index += 10;
            else
                index += 9;
        } else {
            index += 5;
        }
    }
    if (raw.size() + movers.size() > 4096) {
        err = "too many entities (max 4096 per level)";
        return false;
    }
    // CSR by cell: map order inside a cell, exit doors after everything else of that cell (they are appended to
    // the cell's list only when their switch is hit, entity_exit_switch.py:120)
    std::vector<int> order(raw.size());
    for (size_t i = 0; i < raw.size(); i++) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        int ka = raw[a].cell * 2 + (raw[a].kind == EK_EXIT), kb = raw[b].cell * 2 + (raw[b].kind == EK_EXIT);
        return ka < kb;
    });
    std::vector<int> slot_of(raw.size());
    for (size_t s = 0; s < order.size(); s++) slot_of[order[s]] = (int)s;
    size_t ne = raw.size();
    L.ent_x.resize(ne); L.ent_y.resize(ne); L.ent_meta.resize(ne); L.ent_map_order.resize(ne);
    L.ent_seq.resize(ne); L.ent_cell.resize(ne);
    L.ent_start.assign(N_CELLS + 1, 0);
    L.ent_init_words.assign((ne + 15) / 16, 0);
    std::vector<int> count(N_CELLS, 0);
    for (size_t s = 0; s < ne; s++) {
        const RawEnt &r = raw[order[s]];
        L.ent_x[s] = r.x;
        L.ent_y[s] = r.y;
        uint32_t link = r.link_raw >= 0 ? (uint32_t)slot_of[r.link_raw] : (r.extra >= 0 ? (uint32_t)r.extra : 0xffffu);
        L.ent_meta[s] = r.kind | (r.init << 4) | (link << 8) | (r.type << 24);
        L.ent_seq[s] = (uint16_t)r.seq;
        L.ent_cell[s] = (uint16_t)r.cell;
        L.ent_init_words[s >> 4] |= r.init << ((s & 15) * 2);
        count[r.cell]++;
    }
    for (size_t i = 0; i < ne; i++) L.ent_map_order[i] = (uint16_t)slot_of[i];
    int acc = 0;
    for (int c = 0; c < N_CELLS; c++) {
        L.ent_start[c] = (uint16_t)acc;
        acc += count[c];
    }
    L.ent_start[N_CELLS] = (uint16_t)acc;
    // movers in entity_dic order (type ascending, map order inside a type): nsim.py:235-251 walks them like that
    std::stable_sort(movers.begin(), movers.end(), [](const RawMover &a, const RawMover &b) { return a.type < b.type; });
    // draw order of the reference's entity layer
    {
        // entity_renderer.py:100-150: groups by Entity.type in order of first appearance while walking entity_dic
        // (keys ascending; under key 3 the door precedes its switch, type 4), creation order inside a group; movers are
        // referenced as 0x8000 | index
        struct DrawRef { double key; int seq; uint16_t ref; };
        std::vector<DrawRef> dr;
        for (size_t i = 0; i < raw.size(); i++)
            dr.push_back({raw[i].type == 4 ? 3.5 : (double)raw[i].type, raw[i].seq, (uint16_t)slot_of[i]});
        for (size_t i = 0; i < movers.size(); i++) dr.push_back({(double)movers[i].type, movers[i].seq, (uint16_t)(0x8000u | i)});
        std::stable_sort(dr.begin(), dr.end(), [](const DrawRef &a, const DrawRef &b) {
            return a.key != b.key ? a.key < b.key : a.seq < b.seq;
        });
        L.raster_order.clear();
        for (const DrawRef &d : dr) L.raster_order.push_back(d.ref);
        L.draw_recs.clear();
        for (uint16_t ref : L.raster_order) {
            float fx = 0.f, fy = 0.f;
            uint32_t info;
            if (ref & 0x8000u) {
                const RawMover &m = movers[ref & 0x7fffu];
                info = (m.kind & 15u) | ((m.type & 63u) << 4) | ((m.orientation & 7u) << 10) | 0x8000u | ((uint32_t)(ref & 0x7fffu) << 16);
            } else {
                const uint32_t mm = L.ent_meta[ref];
                fx = (float)L.ent_x[ref]; fy = (float)L.ent_y[ref];
                info = (mm & 15u) | (((mm >> 24) & 63u) << 4) | (((mm >> 8) & 7u) << 10) | ((uint32_t)ref << 16);
            }
            uint32_t bx, by;
            std::memcpy(&bx, &fx, 4); std::memcpy(&by, &fy, 4);
            L.draw_recs.push_back(bx); L.draw_recs.push_back(by); L.draw_recs.push_back(info); L.draw_recs.push_back(0u);
        }
        for (auto &d : raw_doors) {
            for (int k = 0; k < 4; k++) L.door_segs.push_back(d[k]);
            L.door_segs.push_back((double)slot_of[(int)d[4]]);
        }
    }
    // ---- entity zoo tables
    {
        for (const RawMover &m : movers) {
            L.mov_meta.push_back(m.kind | (m.orientation << 3) | (m.mode << 6) | ((uint32_t)m.seq << 8));
            L.mov_x0.push_back(m.x);
            L.mov_y0.push_back(m.y);
            if (m.kind == MK_BALL) L.n_balls++;
        }
        L.n_created = next_seq;
        L.db_count = n > 1200 ? map[1200] : 0;
        // grid edges: nsim.py:208-215 (frame preset) + tile_segment_factory.py:283-302 (mod-2 toggles per tile)
        std::vector<uint8_t> hor(EDGE_W * EDGE_H, 0), ver(EDGE_W * EDGE_H, 0);
        for (int x = 0; x < EDGE_W; x++)
            for (int y = 0; y < EDGE_H; y++) {
                hor[x * EDGE_H + y] = (y == 0 || y == 50) ? 1 : 0;
                ver[x * EDGE_H + y] = (x == 0 || x == 88) ? 1 : 0;
            }
        for (int x = 0; x < GRID_W; x++)
            for (int y = 0; y < GRID_H; y++) {
                int t = L.tiles[x * GRID_H + y];
                if (t == 0 || t >= 34) continue;
                uint16_t g = GRID_EDGE[t];
                for (int j = 0; j < 3; j++)
                    for (int i = 0; i < 2; i++) hor[(2 * x + i) * EDGE_H + 2 * y + j] ^= (g >> (2 * j + i)) & 1;
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 2; j++) ver[(2 * x + i) * EDGE_H + 2 * y + j] ^= (g >> (2 * i + j + 6)) & 1;
            }
        L.edges.assign(2 * EDGE_WORDS, 0);
        for (int k = 0; k < EDGE_W * EDGE_H; k++) {
            if (hor[k]) L.edges[k >> 5] |= 1u << (k & 31);
            if (ver[k]) L.edges[EDGE_WORDS + (k >> 5)] |= 1u << (k & 31);
        }
        // entity_dic walk (keys ascending, creation order inside a key) for dumps / checksums
        struct DicRef { uint32_t type; int seq; uint32_t ref; };
        std::vector<DicRef> dic;
        for (size_t i = 0; i < raw.size(); i++)
            dic.push_back({raw[i].type == 4 ? 3u : raw[i].type, raw[i].seq, (uint32_t)slot_of[i]});
        for (size_t i = 0; i < movers.size(); i++) dic.push_back({movers[i].type, movers[i].seq, 0x80000000u | (uint32_t)i});
        std::stable_sort(dic.begin(), dic.end(), [](const DicRef &a, const DicRef &b) {
            return a.type != b.type ? a.type < b.type : a.seq < b.seq;
        });
        for (const DicRef &d : dic) L.dic_order.push_back(d.ref);
        // Simulator.fast_reset rebuilds every cell list while walking entity_dic: an entity's position in that walk is
        // its list-order number afterwards
        L.ent_rank.assign(ne, 0);
        L.mov_rank.assign(movers.size(), 0);
        for (size_t k = 0; k < L.dic_order.size(); k++) {
            const uint32_t ref = L.dic_order[k];
            if (ref & 0x80000000u) L.mov_rank[ref & 0x7fffffffu] = (uint16_t)k;
            else L.ent_rank[ref] = (uint16_t)k;
        }
        // walk order of the CSR under that rule: inside a cell by rank, exit doors still last (they join their cell's
        // list only when their switch is hit)
        L.ent_perm.resize(ne);
        L.ent_ident.resize(ne);
        for (size_t s2 = 0; s2 < ne; s2++) { L.ent_perm[s2] = (uint16_t)s2; L.ent_ident[s2] = (uint16_t)s2; }
        for (int c = 0; c < N_CELLS; c++)
            std::stable_sort(L.ent_perm.begin() + L.ent_start[c], L.ent_perm.begin() + L.ent_start[c + 1], [&](uint16_t a, uint16_t b) {
                const int ka = ((L.ent_meta[a] & 15u) == EK_EXIT) ? 0x10000 : L.ent_rank[a];
                const int kb = ((L.ent_meta[b] & 15u) == EK_EXIT) ? 0x10000 : L.ent_rank[b];
                return ka < kb;
            });
        // what a fast reset leaves alone: classes without reset_state() only get active = True -- a boost pad keeps
        // "touching" (bit 1), a regular door keeps "closed" (bit 1); everything else returns to its initial bits
        L.ent_keep_words.assign((ne + 15) / 16, 0);
        for (size_t s2 = 0; s2 < ne; s2++) {
            const uint32_t kind = L.ent_meta[s2] & 15u;
            if (kind == EK_BOOST || kind == EK_DOOR_REG) L.ent_keep_words[s2 >> 4] |= 2u << ((s2 & 15) * 2);
        }
    }
    {
        int k = 0;
        for (size_t i = 0; i < raw.size() && k < 5; i++)
            if (raw[i].kind == EK_LOCKED) L.locked_slots[k++] = slot_of[i];
    }
    if (last_switch_raw >= 0) {
        L.obs_switch = slot_of[last_switch_raw];
        L.obs_door = slot_of[raw[last_switch_raw].link_raw];
    }
    return true;
}

int dump_segments(const CompiledLevel &lv, int16_t *out, int max_rows) {
    int r = 0;
    for (int c = 0; c < N_CELLS; c++) {
        int cx = c / GRID_H, cy = c % GRID_H;
        for (int i = lv.seg_start[c]; i < lv.seg_start[c + 1]; i++) {
            if (r >= max_rows) return -1;
            uint16_t s = lv.segs[i];
            int16_t *o = out + 8 * r++;
            o[0] = (int16_t)cx; o[1] = (int16_t)cy; o[2] = (int16_t)(s & 1);
            if ((s & 1) == 0) {
                o[3] = (int16_t)(24 * cx + 12 * ((s >> 2) & 3)); o[4] = (int16_t)(24 * cy + 12 * ((s >> 4) & 3));
                o[5] = (int16_t)(24 * cx + 12 * ((s >> 6) & 3)); o[6] = (int16_t)(24 * cy + 12 * ((s >> 8) & 3));
                o[7] = 1;
            } else {
                o[3] = (int16_t)(24 * cx + 12 * ((s >> 2) & 3)); o[4] = (int16_t)(24 * cy + 12 * ((s >> 4) & 3));
                o[5] = (int16_t)(((s >> 6) & 1) ? 1 : -1); o[6] = (int16_t)(((s >> 7) & 1) ? 1 : -1);
                o[7] = (int16_t)((s >> 8) & 1);
            }
        }
    }
    return r;
}

}  // namespace npp
