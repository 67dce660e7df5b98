// npp_internal.hpp -- structures shared between the C ABI (npp_capi.cpp) and the HIP kernels (npp_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "npp_zoo_layout.hpp"

namespace npp {

// ---- SoA ninja state: f64 plane k of env e lives at d_f64[k * n_envs + e] (coalesced per wave) -------------------
enum F64Plane { F_X = 0, F_Y, F_VX, F_VY, F_VXO, F_VYO, F_FNX, F_FNY, F_CNX, F_CNY, F_SCX, F_SCY, NF64 };  // F_SC*: position of the cached mine overlay
// u32 planes, bit layout in npp_kernels.hip (pack_state / unpack_state)
enum U32Plane { U_A = 0, U_B, U_C, U_D, U_E, NU32 };

// Per-level header (device copy); offsets are bytes into the level blob.
// The "hot" region [off_hot, off_hot + hot_bytes) = seg_start | ent_start | cell_bounds | segs is what a
// workgroup stages into LDS when all of its envs play this level.
struct LevelHdr {
    uint32_t off_hot;
    uint32_t hot_bytes;   // multiple of 16
    uint32_t off_ent_x;   // f64[n_ent]
    uint32_t off_ent_y;   // f64[n_ent]
    uint32_t off_ent_meta;  // u32[n_ent]
    uint32_t off_init_words;  // u32[n_words]
    uint32_t off_tiles;   // u8[1100]
    uint32_t off_raster;  // u16[n_ent] draw order
    uint32_t off_doors;   // f64[5 * n_door]
    uint32_t n_door;
    uint32_t n_seg;
    uint32_t n_ent;
    uint32_t n_words;
    uint32_t n_think;
    int32_t obs_switch;
    int32_t obs_door;
    uint32_t fits_lds;
    // entity zoo (all zero / unused when has_zoo == 0)
    uint32_t has_zoo;
    uint32_t off_ent_seq;   // u16[n_ent]
    uint32_t off_ent_cell;  // u16[n_ent]
    uint32_t off_mov_meta;  // u32[n_mov]
    uint32_t off_mov_x0;    // f64[n_mov]
    uint32_t off_mov_y0;    // f64[n_mov]
    uint32_t off_edges;     // u32[2 * EDGE_WORDS]
    uint32_t off_door_tab;  // u32[2 * n_zdoor]
    uint32_t n_mov;
    uint32_t n_zdoor;
    uint32_t n_created;
    uint32_t n_balls;
    uint32_t ball_first;    // mover slot of the first death ball
    uint32_t off_ent_rank;  // u16[n_ent] list-order number after a fast reset (entity_dic rank)
    uint32_t off_mov_rank;  // u16[n_mov]
    uint32_t off_ent_perm;  // u16[n_ent] CSR walk order after a fast reset
    uint32_t off_ent_ident; // u16[n_ent] identity walk order
    uint32_t off_keep_words;  // u32[n_words] state bits a fast reset keeps
    uint32_t off_draw_recs;   // uint4[n_ent + n_mov] drawables in draw order (npp_level.hpp: draw_recs); 16-byte aligned
    double db_count;
    int32_t locked_slots[5];   // CSR slots of the first five locked doors (entity_dic[6] order), -1 = none
    int32_t pad_;
    double spawn_x, spawn_y;
    double sw_x, sw_y, door_x, door_y;
};

// offsets inside the hot region (bytes)
constexpr uint32_t HOT_SEG_START = 0;      // u16[1101] (2202 -> padded 2208)
constexpr uint32_t HOT_ENT_START = 2208;   // u16[1101]
constexpr uint32_t HOT_BOUNDS = 4416;      // u8[1100] (-> padded 1104)
constexpr uint32_t HOT_SEGS = 5520;        // u16[n_seg]

struct StepOut {
    float *game_state;
    int8_t *action_mask;
    float *entity_pos;
    uint8_t *flags;
    float *reward;
    uint16_t *frames;
    float *terminal_state;
    float *spatial_context;
    double *positions;
    uint16_t *work;
};

struct KernelArgs {
    double *f64;          // [NF64][n]
    uint32_t *u32;        // [NU32][n]
    uint32_t *ent_bits;   // [n_words_max][n]
    float *sc_cache;      // [n][48] cached mine overlay of spatial_context
    const int32_t *env_level;   // [n]
    const int32_t *trunc_limit; // [n]
    const LevelHdr *hdr;  // [n_levels]
    const unsigned char *blob;
    const uint8_t *tile_canvas;   // u8 [n_levels][600][1056]: tile-layer coverage counts (render kernels only; built lazily)
    const uint8_t *inputs;  // actions [n] (mode 0) or replay bytes [n_ticks][n] (mode 1)
    const uint8_t *reset_mask;  // reset kernel only; NULL = all
    int n;
    int n_ticks;          // frame_skip (mode 0) or tick count (mode 1); 0 = observe only
    int n_steps;          // mode 0: Gymnasium steps per launch (npp_step_many); 0 / 1 = one
    int mode;             // 0 gym step, 1 raw ticks
    int autoreset;
    int n_words_max;
    int lanes_per_env;    // G
    int waves_per_block;  // WPB
    uint32_t lds_hot_cap; // bytes reserved for a staged level
    int lds_level;        // 1: every workgroup is level-uniform and its level fits lds_hot_cap -> stage it in LDS
    // entity zoo: per-env block of zoo_words 8-byte words at zoo[env * zoo_words] (layout: ZOO_* below)
    double *zoo;
    int zoo_words;
    int zoo_doors;        // door slots per env (max over levels)
    int zoo_movers;       // mover slots per env (max over levels)
    int zoo_active;       // 1: some env currently plays a level with zoo entities -> the zoo step kernel runs
    int reset_fresh;      // reset kernel: 1 = this reset is the first creation after a (re)assignment of levels
    int fast_reset;       // reset kernel: this reset is a Simulator.fast_reset; step kernels: auto-resets are fast resets
    int reset_auto;       // reset kernel, with fast_reset: envs without a Simulator.reset since their level assignment (state word E
                          // bit 18) get a full reset instead -- the reference env's first reset() (npp_environment.py:518-557)
    // npp_step only (null elsewhere): heavy-first launch order of the workgroups -- wg_order[blockIdx.x] is the block of envs this
    // workgroup steps, wg_cost[block] the shader clocks its last launch took (launch_cost_order rebuilds the order from the costs)
    const uint32_t *wg_order;
    uint32_t *wg_cost;
    int wg_first, wg_count;   // split launch: this launch covers entries [wg_first, wg_first + wg_count) of the order; wg_count == 0 =
                              // the whole grid
    // observation overlap (npp_set_obs_overlap): the step is launched in parts -- the workgroups expected to run long on streams of
    // their own -- and `phase` (u8[n], written by npp_phase_kernel before the parts are launched) says which part steps an env; the
    // observation kernels are then launched once per part, each instance on its part's stream, skipping the envs of the other
    // parts (phase == NULL: no filter).  The step kernel itself does not look at it.
    uint8_t *phase;
    int phase_id;
    int variant;          // build variant of the G = 16 plain step kernels (npp_kernels.hip: VariantK); 0 everywhere else
    StepOut out;
};

constexpr int WAVE = 64;

constexpr int EDGE_WORDS_D = 142;   // == npp::EDGE_WORDS (npp_level.hpp)

// Launch geometry: G lanes cooperate on one environment (G in {1,2,4,8,16,32,64}); a wavefront holds 64/G envs; a
// workgroup holds WPB wavefronts that share one LDS copy of a level when all their envs play the same level.
// dynamic LDS layout: [hot_cap][ent words: n_words_max * envs_per_block * 4][obs staging: envs_per_block * 41 * 4][pad to 8]
//                     [spill rows: envs_per_block * LDS_SPILL_BYTES][zoo blocks + grid edges]
// spill row of an env (round 3): what the step kernel parks in LDS instead of scratch memory -- the eight ninja doubles that are
// dead during the collision loops (64 B) and the DepenIO / DepenIOZ block handed to the out-of-line depenetration fallback
constexpr int LDS_SPILL_BYTES = 208;   // 8 doubles | 8 ints | DepenIOZ (104 B, from byte 96)
__host__ __device__ inline size_t lds_spill_offset(uint32_t hot_cap, int n_words_max, int envs_per_block) {
    const size_t b = (size_t)hot_cap + (size_t)n_words_max * envs_per_block * 4 + (size_t)envs_per_block * 41 * 4;
    return (b + 7) & ~(size_t)7;
}
__host__ __device__ inline size_t lds_zoo_offset(uint32_t hot_cap, int n_words_max, int envs_per_block) {
    return lds_spill_offset(hot_cap, n_words_max, envs_per_block) + (size_t)envs_per_block * LDS_SPILL_BYTES;
}
inline size_t lds_bytes(uint32_t hot_cap, int n_words_max, int envs_per_block, int zoo_words = 0) {
    size_t b = lds_zoo_offset(hot_cap, n_words_max, envs_per_block);
    // zoo kernels add, per env: the zoo block and a private copy of the level's grid-edge bitmaps
    if (zoo_words) b += (size_t)envs_per_block * ((size_t)zoo_words * 8 + 2 * EDGE_WORDS_D * 4);
    return b;
}

hipError_t launch_step(const KernelArgs &a, hipStream_t s);
hipError_t launch_reset(const KernelArgs &a, hipStream_t s);
// per-env copy of every state plane from `src` into the live state (a.reset_mask selects envs; NULL = all)
hipError_t launch_restore(const KernelArgs &a, const double *src_f64, const uint32_t *src_u32, const uint32_t *src_ent,
                          const float *src_sc, const double *src_zoo, hipStream_t s);
hipError_t launch_render(const KernelArgs &a, uint8_t *d_out, int centered, hipStream_t s);
// max_records: the largest number of draw records (closed-door strokes + entities + movers) of a loaded level; sizes the LDS
// xscr: the per-env scratch of the split cell pass (GV_XSTRIDE bytes per env); null keeps the whole cell pass inside the first kernel
constexpr size_t GV_XSTRIDE = 14336;
// order / cost: u32[n] each -- the launch order of the envs (heaviest first) and the clocks every env's wavefront took; `reorder`
// rebuilds the order from the costs before the launch (null order = env order)
hipError_t launch_global_view(const KernelArgs &a, int max_records, const uint8_t *gv_p, const float *gv_h, const uint8_t *gv_v,
                              uint8_t *d_out, unsigned char *xscr, uint32_t *order, uint32_t *cost, int reorder, hipStream_t s);
// per-level static tables of global_view (the level right after a reset; needs the tile canvas): gv_p u8[n_levels][600][1056]
// (+ 16 bytes) picture, gv_h f32[n_levels][600][100] horizontal sums, gv_v u8[n_levels][176][100] view
hipError_t launch_gv_static(const KernelArgs &a, int n_levels, uint8_t *gv_p, float *gv_h, uint8_t *gv_v, hipStream_t s);
hipError_t launch_full_frame(const KernelArgs &a, int env0, int count, uint8_t *d_out, hipStream_t s);
hipError_t launch_switch_states(const KernelArgs &a, float *d_out, hipStream_t s);
// npp_reach_kernel.hip (tables: npp_reach.hpp)
struct ReachHdr;
// per-env arrays behind ReachMiss (npp_reach_features.hpp): allocated only when a loaded level takes the reference's miss branch
struct ReachMissDev {
    uint32_t *stamp;   // [n][REACH_CELLS]
    double *raw;       // [n][REACH_CELLS]
    uint32_t *epoch;   // [n] current epoch of the env's entries (stamp == epoch: live); 0 = never called
    uint32_t *last_episode;   // [n] episode counter (state word E bits 19-31) seen at the last call
};
hipError_t launch_reach(const KernelArgs &a, const ReachHdr *rh, const unsigned char *rblob, uint32_t *key, float *cache,
                        const ReachMissDev &md, float *out, float *sdf_out, int32_t *status, float *sw_out, hipStream_t s);
// envs selected by a.reset_mask (NULL = all): key / cache <- the snapshot's, or "no cached vector" when src_key == NULL
hipError_t launch_reach_restore(const KernelArgs &a, const uint32_t *src_key, const float *src_cache, uint32_t *key, float *cache,
                                const ReachMissDev &md, hipStream_t s);
// order <- the indices 0 .. n - 1 sorted by cost, heaviest first (128 logarithmic bins; any costs give a permutation)
hipError_t launch_phase_assign(const uint32_t *order, int blocks, int epb, int n, const int *edge, int parts, uint8_t *phase, hipStream_t s);
hipError_t launch_spin(long long ticks, hipStream_t s);   // a bounded idle wavefront (stream calibration)
hipError_t launch_cost_order(const uint32_t *cost, uint32_t *order, int n, int fold, hipStream_t s);
hipError_t launch_tile_tables(hipStream_t s);   // per-device tile gray tables of the player_frame kernel
hipError_t launch_tile_canvas(const LevelHdr *d_hdr, const unsigned char *d_blob, uint8_t *d_canvas, int n_levels, hipStream_t s);

}  // namespace npp
