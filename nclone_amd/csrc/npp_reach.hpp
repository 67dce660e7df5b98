// npp_reach.hpp -- reachability observation (BASELINE.json config 5): data model shared by the host
// builder (npp_reach.cpp), the feature function (host + device) and the kernel (npp_reach_kernel.hip).
//
// What the reference does (citations: /root/reference/nclone/graph/reachability/...):
//   graph_builder.py:735 build_graph          12-px sub-node graph of the level, entity mask (mines), flood fill from the spawn
//   path_distance_cache.py:249                per goal (exit switch, exit door) a Dijkstra with GEOMETRIC edge costs from the goal
//                                             over that graph -> distance, next hop, 4-hop look-ahead direction per node
//   mine_proximity_cache.py:281               mine signed-distance field on a 12-px grid
//   feature_computation.py:197                38 floats from those tables + the ninja position, recomputed by the env only when
//                                             (ninja cell, exit_switch_activated) changes (mixins/reachability_mixin.py:150-222)
// All of it is static per level except the position-dependent arithmetic, so: the host builds the per-level tables once at
// npp_load_levels (milliseconds per level), they live in HBM, and the kernel does table look-ups + the feature arithmetic for
// the envs whose cache key changed.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define NPP_HD __host__ __device__
#else
#define NPP_HD
#endif

namespace npp {

// sub-node lattice in tile-data space (the 42 x 23 inner tiles): node (i, j) sits at x = 6 + 12 i, y = 6 + 12 j
constexpr int RW = 84, RH = 46, RNODES = RW * RH;   // 3864
constexpr int SDF_W = 88, SDF_H = 50;                // mine SDF grid, 12-px cells over the 1056 x 600 world
constexpr int REACH_DIM = 38;
constexpr int REACH_MAX_CAND = 16;                   // goal nodes the miss branch can select for one goal
constexpr int REACH_CELLS = 44 * 25;                 // 24-px cells: keys of the calculator's per-episode (start cell, goal cell) cache

// Per-level tables as laid out in HBM (one ReachHdr per level + a blob; offsets in bytes from `base` in the blob).
struct ReachHdr {
    uint32_t supported;       // 0: the level needs a code path of the reference that is not restated (see npp_reach.cpp)
    uint32_t n_adj;           // len(adjacency)
    uint32_t off_in;          // u8 [RNODES]   1 = node in the final adjacency (entity mask + flood fill)
    uint32_t off_dist;        // f64[2][RNODES] geometric distance to goal 0 (exit switch) / 1 (exit door); +inf = not reached
    uint32_t off_hop;         // i16[2][RNODES] next hop node id toward the goal, -1 = none
    uint32_t off_mh;          // f64[2][RNODES][2] multi-hop direction, NaN = none
    uint32_t off_sdf;         // f32[SDF_H][SDF_W] (0 = no mines: value 1, gradient 0)
    uint32_t off_grad;        // f32[SDF_H][SDF_W][2]
    int32_t goal_x[2], goal_y[2];      // int(entity position) of exit switch / exit door, world space
    int32_t goal_node[4];              // [0], [1]: node nearest to the goal in the adjacency (threshold 50), -1 = none;
                                       // [2], [3]: the goal node get_distance looks up (threshold 10 + radius, then 32)
    int32_t exit_gid;                  // goal id the reference infers for the exit door position: 1 ("exit"), or 0 when the
                                       // door lies within 24 px of the switch on both axes (it then reads the SWITCH tables)
    int32_t sw_valid, ex_valid;        // the positions are not (0, 0)
    float exit_path[4];                // features 25-28 (static per level)
    float f0;                          // feature 0: clip(len(adjacency) / 966, 0, 1)
    float exit_reachable;              // feature 3
    double area_scale;                 // sqrt(surface area) * 12, or LEVEL_DIAGONAL when the reference's flood fill fails
    int32_t n_mines;                   // toggle mines of both types
    uint32_t n_words;                  // entity-state words of the level
    uint32_t off_mine_mask;            // u32[n_words]: bit 2k of word w set when entity 16 w + k is a toggle mine
    // ---- the cache-miss branch of get_distance for the exit door (path_distance_calculator.py:1218-1485), taken on EVERY query when
    //      the door lies within 24 px of its switch but not within 12: goal-id inference says "switch", validation against the
    //      switch's cached position fails.  Static per level: the goal node as a function of the temp start node, and the physics
    //      A* cost (path_distance_calculator.py:744-845) from every start node to that goal node.
    uint32_t miss_exit;                // 1: exit-door queries take the miss branch
    uint32_t n_cand;                   // goal nodes find_goal_node_closest_to_start can return for the exit door
    int32_t cand[REACH_MAX_CAND];      // their node ids
    uint32_t off_cgoal;                // u8 [RNODES] index into cand[] for a temp start node, 0xff = none
    uint32_t off_astar;                // f64[n_cand][RNODES] raw A* cost start node -> cand[k]; NaN = pair not tabulated
    uint32_t sw_alias;                 // 1: switch and door share a 24-px cell, i.e. one key of the per-episode cache
    uint32_t off_rec;                  // ReachRec[RNODES]: the per-node entries of the tables above side by side (what one feature vector
                                       // reads of a node in ONE 48-byte record: the device kernel is a chain of dependent loads)
    uint64_t base;                     // byte offset of this level's tables in the blob of all levels
};

struct alignas(16) ReachRec {   // scalar members only: indexing a member array by the goal id would put the record into scratch memory
    double dist0, dist1;   // off_dist
    double mh0x, mh0y;     // off_mh, goal 0
    int16_t hop0, hop1;    // off_hop
    uint8_t in;            // off_in
    uint8_t cgoal;         // off_cgoal (0xff without the table)
    uint16_t pad0_;
    uint64_t pad1_;
    NPP_HD double dist(int g) const { return g ? dist1 : dist0; }
    NPP_HD int hop(int g) const { return g ? hop1 : hop0; }
};
static_assert(sizeof(ReachRec) == 48, "three 16-byte loads per node");

NPP_HD inline int reach_node_id(int x, int y) {   // tile-data pixel position (6 mod 12) -> node id, -1 outside the lattice
    if (x < 6 || y < 6) return -1;
    const int i = (x - 6) / 12, j = (y - 6) / 12;
    if (i >= RW || j >= RH || (x - 6) % 12 != 0 || (y - 6) % 12 != 0) return -1;
    return i * RH + j;
}
NPP_HD inline int reach_node_x(int id) { return 6 + 12 * (id / RH); }
NPP_HD inline int reach_node_y(int id) { return 6 + 12 * (id % RH); }

}  // namespace npp
