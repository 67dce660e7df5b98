// npp_host.hpp -- what the GPU-free part of the library (npp_host.cpp: the host-only C entry points; npp_level.cpp, npp_reach.cpp)
// shares with the C ABI proper (npp_capi.cpp).  Nothing here needs HIP: tools/build_sanitized.sh compiles these files with
// g++ -fsanitize=address,undefined into a test-only library.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "npp_level.hpp"
#include "npp_zoo_layout.hpp"

namespace npp {

// message of the last failing call that had no handle (npp_create, the host-only entries); per thread
std::string &host_error();

// The per-env zoo block must hold the doors and movers of EVERY level of the set: the reset kernel writes n_zdoor door
// words and n_mov mover records for whatever level an env plays, zoo level or not (a locked door has an edge counter too).
inline void zoo_block_plan(const std::vector<CompiledLevel> &lv, int &doors, int &movers) {
    doors = 0; movers = 0;
    for (const CompiledLevel &L : lv) {
        doors = std::max(doors, (int)(L.door_tab.size() / 2));
        movers = std::max(movers, (int)L.mov_meta.size());
    }
}


// calculate_truncation_limit(surface_area, 0) (gym_environment/truncation_calculator.py:19-57; the env passes 0 mines,
// npp_environment.py:1238-1256): int(clip((sqrt(area) * 20.0 + 0 * 75.0) * 25, 1200, 10000))
inline int32_t truncation_limit_for_area(int surface_area) {
    // the PBRS calculator's fallback when the flood fill finds nothing (reward_calculation/pbrs_potentials.py:885-895)
    const double area = surface_area > 0 ? (double)surface_area : 1000.0;
    double v = (std::sqrt(area) * 20.0 + 0.0 * 75.0) * 25.0;
    v = v < 1200.0 ? 1200.0 : (v > 10000.0 ? 10000.0 : v);
    return (int32_t)v;
}


}  // namespace npp
