// npp_zoo_layout.hpp -- layout constants of the per-env zoo block, shared by the kernels (npp_internal.hpp) and the GPU-free host
// code (npp_host.hpp).
#pragma once
#include <cstdint>

namespace npp {

// per-env zoo block, in 8-byte words: [0] xlp_boost_normalized, [1] ylp_boost_normalized, [2] lo32 = next list-order
// number (grid_move appends), hi32 = 1 while the entities are in their first creation since the level was assigned
// (Entity.index bug-compat, see entity_death_ball.py:153-166), [3] lo32 = override flags (ZOO_OVR_*), hi32 = list-order
// number the exit door got when its switch appended it to the grid (0 = not yet), [4..7] = exit switch x, y, exit door x, y
// set by npp_set_entity_pos (they survive resets), then ceil(zoo_doors / 2) words of i32 door state (low 16 bits signed
// edge counter, bits 16-23 open_timer), then 5 words per mover: x, y, a, b, lo32 = cell | bits << 11, hi32 = list-order number.
constexpr int ZOO_HEAD = 8;
constexpr uint32_t ZOO_OVR_SWITCH = 1u;   // the exit switch sits at [4], [5]
constexpr uint32_t ZOO_OVR_DOOR = 2u;     // the exit door sits at [6], [7]
constexpr int ZOO_MOV_WORDS = 5;
inline int zoo_words_for(int doors, int movers) { return ZOO_HEAD + (doors + 1) / 2 + ZOO_MOV_WORDS * movers; }
}  // namespace npp
