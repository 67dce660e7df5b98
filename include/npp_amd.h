/*
 * npp_amd.h -- C ABI of the MI355X batched N++ environment stepper.
 *
 * The reference (Tetramputechture/nclone) is pure Python and has no FFI seam; the seam it
 * does have is the object protocol of NPlayHeadless (nclone/nplay_headless.py:28) under the
 * Gymnasium surface of NppEnvironment (nclone/gym_environment/base_environment.py:483 step,
 * npp_environment.py:504 reset).  Each entry point below names the reference interface it
 * replaces for N environments at once.  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 = NPP_OK);
 *     npp_last_error(h) returns a human-readable message for the last failure on h
 *     (h may be NULL for errors raised by npp_create).
 *   - "d_" parameters are DEVICE-ACCESSIBLE pointers (hipMalloc'ed or hipHostMalloc'ed pinned
 *     host memory); everything else is ordinary host memory.
 *   - calls are asynchronous and ordered on the handle's HIP stream; npp_sync blocks.
 *   - one host thread per handle.  One handle per GPU (one process per GPU for multi-GPU).
 *   - all arithmetic of the physics path is IEEE fp64 without fused contraction, as in the
 *     reference (CPython floats).
 */
#ifndef NPP_AMD_H
#define NPP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct npp_handle_s *npp_handle;

enum {
    NPP_OK = 0,
    NPP_ERR_INVALID = 1,     /* bad argument */
    NPP_ERR_HIP = 2,         /* HIP runtime failure (message has the hipError string) */
    NPP_ERR_UNSUPPORTED = 3, /* level uses entity types outside the accelerated path */
    NPP_ERR_STATE = 4        /* call order (e.g. step before levels were loaded) */
};

/* npp_create flags */
enum {
    NPP_FLAG_AUTORESET = 1u << 0,         /* reset an env in-kernel when it terminates/truncates
                                             (vector-env semantics; obs returned is the reset obs) */
    NPP_FLAG_ALLOW_UNSUPPORTED = 1u << 1, /* load levels with unsupported entity types, ignoring
                                             those entities (they are skipped, never simulated) */
    NPP_FLAG_FRAME_CENTERED = 1u << 2,    /* player_frame cropped around (x, y) as intended; default (0) reproduces
                                             the reference's axis-swapped crop (observation_processor.py:219-231) */
    NPP_FLAG_FAST_RESET = 1u << 3         /* every reset after the first one of a level assignment has the semantics of
                                             Simulator.fast_reset (nsim.py:78-140), which NppEnvironment.reset uses for
                                             same-level resets (npp_environment.py:541-557): entities are reset in place,
                                             so (a) the per-cell entity lists are rebuilt in entity_dic order (type key,
                                             then creation order) instead of map order, and (b) entities whose class has
                                             no reset_state() (drones, thwumps, bounce blocks, death balls, shove thwumps,
                                             regular doors, boost pads) keep position and state, only `active` is set.
                                             Default (0): every reset is Simulator.reset (nsim.py:62-76). */
};

/* out_flags bits written by npp_step */
enum {
    NPP_F_WON = 1u << 0,        /* ninja.has_won()  (ninja.py:1396)  */
    NPP_F_DEAD = 1u << 1,       /* ninja.has_died() (ninja.py:1399)  */
    NPP_F_SWITCH = 1u << 2,     /* exit_switch_activated() (nplay_headless.py:566) */
    NPP_F_TRUNCATED = 1u << 3,  /* sim.frame >= limit (truncation_checker.py:46-77) */
    NPP_F_CAUSE_MINE = 1u << 4, /* ninja.death_cause == "mine" */
    NPP_F_CAUSE_IMPACT = 1u << 5/* ninja.death_cause == "terminal_impact" */
};

#define NPP_GAME_STATE_DIM 41   /* gym_environment/constants.py:25 */
#define NPP_ACTION_DIM 6        /* base_environment.py:150 Discrete(6) */
#define NPP_ENTITY_POS_DIM 6    /* observation_processor.py:340-361 */
#define NPP_SPATIAL_CONTEXT_DIM 112 /* gym_environment/constants.py SPATIAL_CONTEXT_DIM */
#define NPP_FRAME_W 84          /* gym_environment/constants.py:12-13 */
#define NPP_FRAME_H 84
#define NPP_DUMP_F64 12
#define NPP_DUMP_I32 32

/* Output block of npp_step.  Any pointer may be NULL (that output is skipped). */
typedef struct {
    float *d_game_state;      /* [N,41] f32: get_ninja_state() (nplay_headless.py:735-924) + time_remaining
                                 (base_environment.py:2811-2829), cast like observation_processor.py:284-302 */
    int8_t *d_action_mask;    /* [N,6]  i8 : ninja.get_valid_action_mask() (ninja.py:628-839) */
    float *d_entity_pos;      /* [N,6]  f32: [ninja, exit switch, exit door] / (1056, 600) */
    uint8_t *d_flags;         /* [N]    u8 : NPP_F_* of the state the step ended in (before auto-reset) */
    float *d_reward;          /* [N]    f32: sparse terminal reward only */
    uint16_t *d_frames;       /* [N]    u16: ticks executed this step (info["frame_skip_stats"]) */
    float *d_terminal_state;  /* [N,41] f32: game_state of the terminal state for envs that were
                                 auto-reset this step (rows of other envs are left untouched) */
    float *d_spatial_context; /* [N,112] f32: 8x8 local tile categories + 8 nearest mines x 6 features
                                 (gym_environment/spatial_context.py:113-176,309-508 as called from
                                 npp_environment.py:2318-2360, incl. its >= 12 px position cache) */
    double *d_positions;      /* [N,6]  f64: player_x, player_y, switch_x, switch_y, exit_door_x, exit_door_y in pixels:
                                 the pass-through scalars of the raw observation (base_environment.py _get_observation ->
                                 observation_processor.py:374-399), unrounded */
    uint16_t *d_work;         /* [N]    u16: depenetration iterations this env ran in this step (collide_vs_tiles,
                                 ninja.py:303-364) -- a profiling aid: the launch lasts as long as its busiest env */
} npp_step_out;

/* Simulator()+NPlayHeadless() for n_envs environments on GPU device_id. */
int npp_create(int n_envs, int device_id, unsigned flags, npp_handle *out);
int npp_destroy(npp_handle h);
const char *npp_last_error(npp_handle h);

/* Use an existing hipStream_t (e.g. torch's current stream) for all launches of h. NULL = default stream. */
int npp_set_stream(npp_handle h, void *hip_stream);
int npp_sync(npp_handle h);

/* NPlayHeadless.load_map_from_map_data (nplay_headless.py:195) -> Simulator.load (nsim.py:51) for a SET of
 * levels.  blob holds the raw map_data values of all levels back to back as doubles (generated levels carry
 * fractional entity coordinates); level i is blob[offsets[i] .. offsets[i+1]).
 * The host compiles every level (tiles -> ordered per-cell segment lists, entities -> per-cell tables) and
 * uploads the tables.  Replaces any previously loaded set and resets every env to level 0. */
int npp_load_levels(npp_handle h, const double *blob, const int64_t *offsets, int n_levels);

/* Which level each env plays (EnvMapLoader.load_map choice, env_map_loader.py:111-208).  env_ids == NULL
 * means envs 0..n-1.  The listed envs are reset (Simulator.reset, nsim.py:62). */
int npp_assign_levels(npp_handle h, const int32_t *env_ids, const int32_t *level_ids, int n);

/* Reset the envs whose mask byte is non-zero (NULL = all).  Without NPP_FLAG_FAST_RESET: Simulator.reset (nsim.py:62-76).
 * With it: Simulator.reset for an env's first reset after npp_load_levels / npp_assign_levels, Simulator.fast_reset
 * (nsim.py:78-140) afterwards -- the choice NppEnvironment.reset makes (npp_environment.py:541-557). */
int npp_reset(npp_handle h, const uint8_t *env_mask);
/* The same with an explicit choice: mode 0 = as npp_reset, 1 = Simulator.reset, 2 = Simulator.fast_reset
 * (NPlayHeadless.reset / fast_reset, nplay_headless.py). */
int npp_reset_ex(npp_handle h, const uint8_t *env_mask, int mode);

/* Truncation limit in frames (truncation_checker.py:21-29); limits == NULL sets `all` for every env. */
int npp_set_truncation_limit(npp_handle h, const int32_t *limits, int32_t all);
/* The reference env's DYNAMIC limit (npp_environment.py:1238-1256, base_environment.py:1078-1100 ->
 * truncation_calculator.py:19-57): per level int(clip(sqrt(reachable surface area) * 20 * 25, 1200, 10000)), the surface area
 * being the node count the reachability graph's flood fill finds from the spawn (npp_reach.cpp, pinned by the reference's own
 * values in tests/golden/reach.npz).  enable != 0: every env takes its level's limit now and again at every npp_load_levels /
 * npp_assign_levels (for the envs assigned); it drives both the truncation flag and game_state[40] (time_remaining).  A later
 * npp_set_truncation_limit overrides it until the next assignment.  Known deviation: the reference holds the 10 000-frame
 * fallback until its first reward calculation after a level load, i.e. for the first step's observation. */
int npp_set_dynamic_truncation(npp_handle h, int enable);
/* Host-only: that limit and the surface area for one level. */
int npp_level_truncation_limit(const double *map, int64_t n, int32_t *limit, int32_t *surface_area);

/* NppEnvironment.step for all envs (base_environment.py:483-755): d_actions[N] in 0..5
 * (_actions_to_execute, :366-402), up to frame_skip ticks with early stop on win/death (:535-609),
 * truncation check (:613), observation (:627,680).
 * d_reward: the sparse terminal constants only (+20 win, -3 death, +10 exit switch) -- REWARD PARITY: NONE; the reference's
 * reward is its PBRS calculator (reward_calculation/main_reward_calculator.py:225), which is out of scope. */
int npp_step(npp_handle h, const uint8_t *d_actions, int frame_skip, const npp_step_out *out);

/* Several Gymnasium steps in ONE launch, for action sequences that do not depend on the observations in between: batched
 * checkpoint replay (the reference's ActionReplayer.replay_to_checkpoint, action_replayer.py, replays one env at a time),
 * rollouts of a fixed plan.  d_actions: u8[n_steps][n_envs].  d_flags / d_reward / d_frames of `out` receive
 * [n_steps][n_envs] values (one row per step); the observations are those after the last step; with auto-reset an env that
 * terminates mid-sequence restarts on the spot (no terminal observation in this mode).  Wavefronts advance through their
 * steps independently, so the launch costs the sum of average step times instead of the sum of worst-case step times. */
int npp_step_many(npp_handle h, const uint8_t *d_actions, int n_steps, int frame_skip, const npp_step_out *out);

/* NPlayHeadless.tick(h, j) (nplay_headless.py:322) for all envs, n_ticks times, driven by replay input
 * bytes d_inputs[n_ticks][N] (bit0 jump, bit1 right, bit2 left: replay/replay_executor.py:61-84).
 * No early stop, no truncation, no auto-reset: the caller polls state like tools/test_replay_playback.py. */
int npp_tick(npp_handle h, const uint8_t *d_inputs, int n_ticks);

/* Observation of the current state without stepping (NppEnvironment._get_observation, used by reset()). */
int npp_observe(npp_handle h, const npp_step_out *out);

/* player_frame (84x84 u8) around each ninja, rasterised on device (nsim_renderer.py:71-134 +
 * observation_processor.py:207-282 crop incl. its axis swap).  d_out is [N,84,84]. */
int npp_render_player_frame(npp_handle h, uint8_t *d_out);
/* Goal-curriculum repositioning (gym_environment/reward_calculation/intermediate_goal_manager.py:698 apply_to_simulator): move
 * the exit switch (kind 0) or the exit door (kind 1) of ONE env to pixel position (x, y).  Like there, the entity's grid
 * cell follows (a switch that changes cell is appended to the new cell's list; the door joins its new cell's list when its
 * switch is hit) and the position survives resets until it is set again; NaN clears it.  Call between episodes.  Envs with
 * a moved entity run in the zoo kernels (merged neighbourhood walk). */
int npp_set_entity_pos(npp_handle h, int env, int kind, double x, double y);

/* switch_states observation (gym_environment/npp_environment.py:1782-1847): f32[n_envs][25] = up to 5 locked doors x
 * [switch x / 1056, switch y / 600, door x / 1056, door y / 600, collected]; the reference's door position falls back to the
 * switch position (its segment has no `p1`), reproduced.  Pinned by the reference's own two methods run on live entities
 * (tests/golden/obs.npz, make_golden_obs.py). */
int npp_switch_states(npp_handle h, float *d_out);

/* reachability_features (f32[n_envs][38]) and mine_sdf_features (f32[n_envs][3]) of the current state of every env: what
 * ReachabilityMixin._get_reachability_features (gym_environment/mixins/reachability_mixin.py:72-222) and
 * MineSignedDistanceField.get_features_at_position (graph/reachability/mine_proximity_cache.py:476) return, including the
 * reference's cache rule -- the 38 floats are recomputed only when (ninja cell, exit_switch_activated) differs from the key of
 * the env's previous call, so call it once per observation (after npp_step / npp_reset), as the reference does.  The
 * per-level tables (sub-node graph, entity mask, flood fill, geometric Dijkstra from both goals, mine SDF) are built on the
 * host at the first call.  Levels whose exit door lies 12-24 px from its switch send every exit-door query of the reference through
 * its cache-miss branch (physics A*, path_distance_calculator.py:744-845, 1218-1485): restated since round 3 -- per-level A* table
 * on the host, the calculator's per-episode (start cell, goal cell) dictionary per env on the device (13 KB per env, allocated only
 * when such a level is loaded; emptied when the env's episode counter moves, on npp_restore and on npp_assign_levels).
 * Either output may be NULL.  d_status (i32[n_envs], may be NULL): bit 0 set where the reference would have taken that branch
 * for a query the tables do not cover (no node with a cached distance under the ninja on a level without the A* table) -- those
 * rows carry the "unreachable" values.  NPP_ERR_UNSUPPORTED: a loaded level has several exit switches (the reference then also
 * leaves the level cache for the switch and runs a second search that is not restated), or an entity was moved with
 * npp_set_entity_pos.  Not meaningful between the steps of npp_step_many (the cache rule needs every observation).
 * mine_sdf_features: the VALUES are the reference's (MineSignedDistanceField); WHEN the reference's observation holds them (its
 * reward calculator builds / clears that SDF) is not modelled -- parity of that life cycle is unpinned. */
int npp_reachability(npp_handle h, float *d_features, float *d_mine_sdf, int32_t *d_status);
/* ... and, when d_switch_states (f32[n_envs][25]) is not NULL, npp_switch_states's output from the same launch (an idle lane of
 * every env's lane group writes it: the full observation Dict needs one kernel less). */
int npp_reachability_ex(npp_handle h, float *d_features, float *d_mine_sdf, int32_t *d_status, float *d_switch_states);

/* The whole gray frame of envs [env0, env0 + count): what NPlayHeadless.render() returns in grayscale mode
 * (nplay_headless.py:144-156, nsim_renderer.py:71-134).  d_out: u8[count][600][1056]. */
int npp_render_frame(npp_handle h, int env0, int count, uint8_t *d_out);

/* global_view (observation_processor.py:304-328): the (600 x 1056) gray frame through cv2.resize(frame, (100, 176), INTER_AREA).
 * The reference's RENDERED_VIEW_WIDTH / HEIGHT are swapped (gym_environment/constants.py:18-19), so the frame is squashed to
 * 176 rows x 100 columns; reproduced as is (area-weighted mean of the source pixels).  d_out: u8[n_envs][176][100]. */
int npp_render_global_view(npp_handle h, uint8_t *d_out);

/* Parity hook: copies the simulator state of envs [env0, env0+count) to HOST buffers.
 * f64 [count][NPP_DUMP_F64]: xpos ypos xspeed yspeed floor_nx floor_ny ceil_nx ceil_ny xspeed_old yspeed_old
 *                            applied_gravity applied_drag
 * i32 [count][NPP_DUMP_I32]: see DESIGN.md ("state dump layout"). */
int npp_dump_state(npp_handle h, int env0, int count, double *f64_out, int32_t *i32_out);

/* Parity hook: per-entity dynamic state of one env in level (map) order:
 * mines -> state 0/1/2, exit door -> switch_hit, others -> active.  Returns count via *n_out. */
int npp_dump_entities(npp_handle h, int env, int32_t *out, int max, int *n_out);

/* Parity hook: the compiled collision table of a level as rows of 8 int16
 * (cell x, cell y, kind, then x1,y1,x2,y2,oriented | cx,cy,hor,ver,convex), in query order. */
int npp_dump_level_segments(npp_handle h, int level, int16_t *out, int max_rows, int *n_out);

/* Host-only utilities (no GPU, no handle): run the level compiler on one level.  Used by the CPU test-suite to
 * check the compiler against the reference's ordered segment dump and entity tables.
 * segments: rows of 8 int16 as in npp_dump_level_segments.  entities: rows of 6 doubles in map order:
 * kind (1 mine, 2 gold, 3 exit door, 4 exit switch, 6 locked-door switch), x, y, cell x, cell y, initial 2-bit state. */
int npp_compile_level_segments(const double *map, int64_t n, int16_t *out, int max_rows, int *n_out,
                               uint32_t *unsupported_mask);
int npp_compile_level_entities(const double *map, int64_t n, double *out, int max_rows, int *n_out);

/* Build variant of the step kernel (a speed knob like npp_set_launch_geometry; results are bit-identical for every variant; no
 * counterpart in the reference).  The 16-lanes-per-env kernels of levels without moving entities exist in three builds -- 0: two
 * wavefronts per SIMD, two candidate slots per lane; 1: two wavefronts per SIMD, one slot (fastest on sparse geometry, slow where
 * creases gather more than 16 segments); 2: one wavefront per SIMD, two slots (fastest single chain) -- and npp_step times them
 * against each other on the handle's own workload (HIP events, no synchronisation) after every npp_load_levels /
 * npp_assign_levels and keeps the fastest.  variant = -1 (default) selects that autotuner, 0..2 pin a build.
 * npp_get_step_variant: *variant = the build npp_step currently launches, *tuned = 1 once the autotuner has decided (or a build is
 * pinned). */
int npp_set_step_variant(npp_handle h, int variant);
int npp_get_step_variant(npp_handle h, int *variant, int *tuned);

/* Observation overlap (a speed knob for the full observation Dict; results are bit-identical with it on or off; no counterpart in the
 * reference).  A step launch ends with a tail of a few long environments that leaves most of the chip idle.  With cuts c1 < c2 < ...
 * (percentages, at most three) npp_step cuts its heavy-first workgroup order at those points and launches every piece ("part") as a
 * kernel of its own, the most expensive first: the last piece (the cheap end) on the handle's stream, the others on HIP streams the
 * handle owns (chosen at set-up so that their kernels are seen to run beside the handle's stream: that call synchronises).  Until the next join npp_render_player_frame, npp_render_global_view, npp_reachability and npp_switch_states launch
 * one kernel per part, each behind the part it reads: the observations of the cheap environments are produced while the expensive
 * ones are still stepping.  npp_join makes the handle's stream wait for the others -- call it before anything else consumes the
 * step's or the observation kernels' outputs on the handle's stream; every other entry point (npp_sync, npp_step, npp_reset,
 * npp_snapshot ...) joins by itself.  npp_set_obs_overlap(h, percent) = one cut; percent = 0 / n_cuts = 0 (default) switch it off. */
int npp_set_obs_overlap(npp_handle h, int percent);
int npp_set_obs_overlap_parts(npp_handle h, const int *cuts, int n_cuts);
int npp_join(npp_handle h);

/* Host-only: the per-level reachability tables of one level, stage by stage (CPU tests compare them with the reference's,
 * tests/golden/reach.npz).  Node id = i * 46 + j for the sub-node at tile-data pixel (6 + 12 i, 6 + 12 j), 84 x 46 = 3864 ids.
 * info i32[16]: supported, len(adjacency), goal node of exit_switch_0 / exit_door_0 in the level cache, the two goal nodes
 * get_distance resolves, switch x, y, door x, y (int), goal id inferred for the door position (0 = "switch"), mines, has
 * SDF, surface area (nodes).  base_in / base_adj / phys: tile-only graph (node present, edge bits N E S W NE SE SW NW,
 * grounded | walled << 1).  in / adj: final adjacency.  dist f64[2][3864], hop i16[2][3864], mh f64[2][3864][2]: level
 * cache per goal.  sdf f32[50][88], grad f32[50][88][2].  scalars f64[8]: area scale, feature 0, feature 3, features 25-28.
 * Any pointer may be NULL. */
int npp_reach_compile(const double *map, int64_t n, int32_t *info, uint8_t *base_in, uint8_t *base_adj, uint8_t *phys, uint8_t *in,
                      uint8_t *adj, double *dist, int16_t *hop, double *mh, float *sdf, float *grad, double *scalars);
/* Host-only: the feature function the device kernel runs (npp_reach_features.hpp compiles for both), on `count` positions
 * (f64[count][2]) of one level; mines i32[count][2] = (total, deadly) toggle mines (NULL: all safe).  out f32[count][38],
 * sdf_out f32[count][3] (may be NULL), status i32[count] (may be NULL; bit 0 as in npp_reachability, bit 1 = level unsupported). */
int npp_reach_features_host(const double *map, int64_t n, const double *pos, const int32_t *mines, int count, float *out, float *sdf_out,
                            int32_t *status);
/* Host-only: the tables behind the cache-miss branch of CachedPathDistanceCalculator.get_distance
 * (graph/reachability/path_distance_calculator.py:1218-1485, physics A* :744-845), built for the exit door of levels whose door lies
 * within 24 px (but not 12) of its switch -- there the reference's goal-id inference sends EVERY exit-door query down that branch.
 * info i32[20]: miss branch active, number of goal nodes, switch and door share a 24-px cell, supported, then the goal node ids
 * (-1 padded).  cgoal u8[3864]: index of the goal node find_goal_node_closest_to_start picks for a temp start node (255 = node
 * not in the adjacency).  astar f64[16][3864]: _calculate_distance(start node, goal node) (NaN = pair not tabulated).
 * mine_mult f64[3864]: MineProximityCostCache multiplier per node (1 = none).  Any pointer may be NULL. */
int npp_reach_compile_miss(const double *map, int64_t n, int32_t *info, uint8_t *cgoal, double *astar, double *mine_mult);
/* Host-only: npp_reach_features_host along a ROLLOUT -- the `count` positions are consecutive feature recomputations of one env,
 * new_episode u8[count] marks the first recomputation after an episode reset (the path calculator's per-episode
 * (start cell, goal cell) dictionary is emptied there, reachability_mixin.py:67-70); what the device keeps per env.  raw_out f64[count]
 * (may be NULL): the dictionary's entry for the ninja's cell after each query (the raw A* cost; NaN = no entry). */
int npp_reach_rollout_host(const double *map, int64_t n, const double *pos, const int32_t *mines, const uint8_t *new_episode, int count,
                           float *out, int32_t *status, double *raw_out);

/* Host-only: the zoo tables the level compiler derives from map_data.  edges_out: int32[2][89*51] grid-edge counters at
 * load (horizontal then vertical, key = x * 51 + y; tile edges of tile_segment_factory.py:283-302 plus closed doors,
 * entity_door_base.py:78-89).  movers_out: rows of 4 doubles (Entity.type, x, y, creation order) in entity_dic order. */
int npp_compile_level_zoo(const double *map, int64_t n, int32_t *edges_out, double *movers_out, int max_movers, int *n_movers);

/* Test / debug aid for levels with moving entities: per env 6 doubles summed over the entities in the reference's
 * entity_dic order (keys ascending, creation order inside a key): sum x, sum y, sum xspeed, sum yspeed (bounce blocks and
 * death balls), sum of state codes (3*closed + 5*(state mod 7) + 11*dir + 13*touching + 17*activated), number of active
 * entities -- the row tests/golden/make_golden_zoo.py records from the reference after every tick. */
int npp_entity_checksum(npp_handle h, int env0, int count, double *out);

/* Go-Explore style checkpoints (state_checkpoint.py / action_replayer.py in the reference restore a state by
 * reset + replaying the action sequence and validating |dpos| < 0.01 px).  Here a checkpoint is a raw copy of the
 * SoA state of ALL envs kept on the device (one slot per handle): npp_snapshot stores it, npp_restore puts it back for
 * the envs whose mask byte is non-zero (NULL = all).  The env -> level assignment must not have changed in between. */
int npp_snapshot(npp_handle h);
int npp_restore(npp_handle h, const uint8_t *env_mask);

/* Launch geometry: lanes_per_env wavefront lanes cooperate on one environment (power of two, 1..64; 0 = choose from
 * n_envs so that the grid fills the chip), waves_per_block wavefronts share one LDS copy of a level (1..4, 0 = auto).
 * Results are bit-identical for every geometry; only speed changes. */
int npp_set_launch_geometry(npp_handle h, int lanes_per_env, int waves_per_block);
int npp_get_launch_geometry(npp_handle h, int *lanes_per_env, int *waves_per_block);

/* Host-only (no GPU, no handle): the per-env "zoo block" (doors' edge counters + moving entities) a level SET needs.
 * The block is sized over ALL levels of the set -- the reset kernel initialises the doors / movers of every level, not
 * only of those with moving entities -- and npp_load_levels refuses a plan that does not cover one of its levels. */
int npp_plan_zoo_block(const double *blob, const int64_t *offsets, int n_levels, int *doors, int *movers, int *words);

int npp_num_envs(npp_handle h);
int npp_num_levels(npp_handle h);

#ifdef __cplusplus
}
#endif
#endif /* NPP_AMD_H */
