// npp_level.hpp -- host-side level compiler: raw map_data -> packed per-cell tables for the HIP kernels.
//
// Replaces, for the accelerated path, the reference's level-load chain
//   nclone/map_loader.py:18-145 (tiles, entities), nclone/utils/tile_segment_factory.py:170-262
//   (tile -> segments with ortho cancellation), nclone/utils/spatial_segment_index.py:69-110
//   (per-cell snapshot + cell AABB), nclone/utils/entity_factory.py:145-233, nclone/entities.py:209-236.
// Output layout is documented in DESIGN.md ("level tables").
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace npp {

constexpr int GRID_W = 44;
constexpr int GRID_H = 25;
constexpr int N_CELLS = GRID_W * GRID_H;  // cell index = cx * 25 + cy  (x-major, like the reference's queries)

// entity kinds understood by the kernels (low nibble of ent_meta)
enum EntKind : uint32_t { EK_NONE = 0, EK_MINE = 1, EK_GOLD = 2, EK_EXIT = 3, EK_SWITCH = 4, EK_DOOR_REG = 5, EK_LOCKED = 6,
                          EK_DOOR_TRAP = 7, EK_LAUNCH = 8, EK_ONEWAY = 9, EK_BOOST = 10 };
// entities that move between grid cells ("movers"); they live in their own table, in entity_dic order
enum MoverKind : uint32_t { MK_DRONE = 1, MK_BOUNCE = 2, MK_THWUMP = 3, MK_BALL = 4, MK_MINI = 5, MK_SHOVE = 6 };
constexpr int EDGE_W = 89, EDGE_H = 51;                      // half-cell grid edges, key = x * 51 + y
constexpr int EDGE_WORDS = (EDGE_W * EDGE_H + 31) / 32;      // 142 u32 per orientation

// Packed collision segment (uint16), coordinates in units of 12 px relative to the owning cell's origin:
//   linear : bit0 = 0, bits 2-3 x1, 4-5 y1, 6-7 x2, 8-9 y2
//   arc    : bit0 = 1, bits 2-3 cx, 4-5 cy, bit6 hor>0, bit7 ver>0, bit8 convex
//   both   : bits 11-15 = y of the owning cell (a query walks whole cell columns, which are contiguous in the CSR)
// Packed cell bounds (uint8): bits 0-1 min x, 2-3 min y, 4-5 max x, 6-7 max y (same units).
struct CompiledLevel {
    std::vector<uint16_t> seg_start;   // [N_CELLS+1] CSR over cells
    std::vector<uint16_t> segs;        // packed, query order
    std::vector<uint8_t> cell_bounds;  // [N_CELLS]
    std::vector<uint16_t> ent_start;   // [N_CELLS+1] CSR over cells, map order inside a cell, exit doors last
    std::vector<double> ent_x, ent_y;  // [n_ent] pixel positions (coord * 6)
    std::vector<uint32_t> ent_meta;    // [n_ent] kind | init2bit << 4 | link << 8 (switch -> its door's index) | raw type << 24
    std::vector<uint16_t> ent_map_order;  // [n_ent] CSR slot of the i-th entity in map order (for dumps)
    std::vector<uint32_t> ent_init_words;  // 2 bits per entity, 16 per word
    std::vector<uint8_t> tiles;        // [N_CELLS] tile id per cell (border = 1), for the rasteriser
    std::vector<uint16_t> raster_order;  // CSR slots in draw order (entity_renderer.py:100-150: by type, then map order)
    // the same walk as compact records for the rasteriser, 4 u32 per drawable: x, y (float bits; unused for movers),
    // kind | raw type << 4 | orientation << 10 | mover << 15 | (CSR slot or mover index) << 16, 0
    std::vector<uint32_t> draw_recs;
    std::vector<double> door_segs;     // closed-door strokes: x1, y1, x2, y2, slot of the owning entity (5 per door)
    double spawn_x = 0, spawn_y = 0;
    int obs_switch = -1, obs_door = -1;  // CSR slots of the exit switch / door reported in observations
    int n_thinkable = 0;               // mines
    uint32_t unsupported_mask = 0;     // bit t set if entity type t present but not simulated
    // ---- entity zoo ----
    std::vector<uint16_t> ent_seq;     // [n_ent] creation order (= list order inside a grid cell) of a CSR entity
    std::vector<uint16_t> ent_cell;    // [n_ent] cell index of a CSR entity
    std::vector<uint32_t> mov_meta;    // [n_mov] MoverKind | orientation << 3 | mode << 6 | creation order << 8
    std::vector<double> mov_x0, mov_y0;  // [n_mov] position at creation (= spring / thwump origin)
    std::vector<uint32_t> edges;       // hor[EDGE_WORDS] then ver[EDGE_WORDS]: tile grid edges (bit = edge present)
    std::vector<uint32_t> door_tab;    // [n_zdoor][2]: key0 | key1 << 16 (bit 15 of a key = vertical); initial counter | door class << 8
    std::vector<uint32_t> dic_order;   // entity_dic walk: CSR slot, or 0x80000000 | mover index
    // ---- Simulator.fast_reset (nsim.py:78-140): cell lists rebuilt while walking entity_dic ----
    std::vector<uint16_t> ent_rank;    // [n_ent] position of a CSR entity in the entity_dic walk (its list-order number then)
    std::vector<uint16_t> mov_rank;    // [n_mov] the same for movers
    std::vector<uint16_t> ent_perm;    // [n_ent] CSR walk position -> slot when the cell lists are in entity_dic order
    std::vector<uint16_t> ent_ident;   // [n_ent] identity (the walk order after Simulator.reset: map order)
    std::vector<uint32_t> ent_keep_words;  // 2 bits per entity: state bits a fast reset keeps (classes without reset_state)
    int locked_slots[5] = {-1, -1, -1, -1, -1};   // CSR slots of the first five locked doors in creation order
    int n_created = 0;                 // entities created at load (first free list-order number)
    int n_balls = 0;
    double db_count = 0;               // map_data[1200]
    bool has_zoo = false;              // any kind beyond mines / gold / exit / locked doors
};

// Returns false (and fills err) on malformed input.
bool compile_level(const double *map, int64_t n, CompiledLevel &out, std::string &err);

// rows of 8 int16, see npp_dump_level_segments in npp_amd.h
int dump_segments(const CompiledLevel &lv, int16_t *out, int max_rows);

}  // namespace npp
