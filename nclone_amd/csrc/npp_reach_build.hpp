// npp_reach_build.hpp -- host-side result of the per-level reachability builder (npp_reach.cpp): the tables that travel to
// HBM (pack_reach) plus the intermediate stages the CPU tests compare with tests/golden/reach.npz.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "npp_reach.hpp"

namespace npp {

struct ReachBuilt {
    ReachHdr hdr;
    std::vector<uint8_t> base_in, base_adj;   // base graph (tiles only): node present, edge bits N E S W NE SE SW NW
    std::vector<uint8_t> phys;                // bit 0 grounded, bit 1 walled (on the base graph)
    std::vector<uint8_t> blocked;             // nodes within 14 px of a toggle mine
    std::vector<uint8_t> in, adj;             // final adjacency (mask + flood fill from the spawn)
    std::vector<double> dist[2];              // [RNODES] per goal
    std::vector<int16_t> hop[2];              // [RNODES]
    std::vector<double> mh[2];                // [RNODES][2]
    std::vector<float> sdf, grad;             // empty when the level has no mines
    std::vector<uint32_t> mine_mask;          // per entity-state word: bit 2k set when entity 16 w + k is a toggle mine
    std::vector<double> mine_mult;            // [RNODES] MineProximityCostCache multiplier of the adjacency nodes (1.0 = none)
    std::vector<uint8_t> cgoal;               // [RNODES] miss branch: index into hdr.cand for a temp start node (0xff = none)
    std::vector<double> astar;                // [n_cand][RNODES] miss branch: physics A* cost to hdr.cand[k] (NaN = not tabulated)
    bool has_sdf = false;
    int surface_area = 0;                     // node count of the area-scale flood fill (0 = it failed)
    int spawn_area = 0;                       // what the reference's truncation limit sees: the same, or the fill from the true spawn
    std::string note;                         // why hdr.supported == 0
};

struct CompiledLevel;
bool build_reach(const double *map, int64_t n, ReachBuilt &out, std::string &err);
void build_reach(const CompiledLevel &level, ReachBuilt &out);
// appends the level's tables to `blob` (offsets in `hdr` relative to hdr.base)
void pack_reach(const ReachBuilt &R, ReachHdr &hdr, std::vector<unsigned char> &blob);

}  // namespace npp
