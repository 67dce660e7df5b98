// npp_reach_kernel.hip -- reachability_features (f32[N, 38]) and mine_sdf_features (f32[N, 3]) of every env from the
// per-level tables in HBM (npp_reach.hpp) and the env state.  The work per env is a handful of dependent table look-ups
// (3 node searches of at most 4 lattice cells, a few f64 distances) and it only runs for the envs whose (ninja cell,
// exit_switch_activated) key changed since their last observation -- the reference's own cache rule
// (gym_environment/mixins/reachability_mixin.py:150-222), kept because the cached vector is what the reference returns.
// Roofline: latency/issue bound; bytes per env = 152 (features out) + 12 (sdf out) + 16 (x, y) + 8 (key, level) + 156 cache
// row read (+ written back when recomputed) ~ 350 B -> 2.9 MB per launch at 8192 envs.
#include <hip/hip_runtime.h>

#include "npp_internal.hpp"
#include "npp_reach_features.hpp"

namespace npp {
namespace {

// 16 envs per 64-lane workgroup (512 workgroups at 8192 envs: the look-ups of a recomputing env are a chain of dependent loads,
// so the launch wants many small wavefronts rather than 128 full ones).  Lane 4 e of the workgroup owns env e's key test and, on
// a miss, its recomputation; then ALL 64 lanes copy the 16 rows -- cache rows, output rows and the LDS staging between them are
// contiguous for consecutive envs, so every global access of the copy is a full coalesced line (the AoS row [38 floats + status]
// is the right layout for that; round 2 had one lane walk its own 156-byte row).
constexpr int REACH_EPB = 16;

__global__ __launch_bounds__(64) void npp_reach_kernel(KernelArgs a, const ReachHdr *rh, const unsigned char *rblob, uint32_t *key,
                                                       float *cache, ReachMissDev md, float *out, float *sdf_out, int32_t *status,
                                                       float *sw_out) {
    __shared__ float rows[REACH_EPB * (REACH_DIM + 1)];
    __shared__ float sdfs[REACH_EPB * 3];
    __shared__ int fresh[REACH_EPB];      // 1: the env's row in `rows` was recomputed (write it back to the cache)
    const int env0 = blockIdx.x * REACH_EPB;
    // observation overlap: the host splits a step only when its workgroups hold a multiple of 16 envs, so these 16 share a phase
    if (a.phase && a.phase_id >= 0 && a.phase[env0] != (uint8_t)a.phase_id) return;
    const int n_here = min(REACH_EPB, a.n - env0);
    const int l = threadIdx.x;
    const int el = l >> 2, env = env0 + el;
    constexpr int ROW = REACH_DIM + 1;
    // ---- stage the cached rows (contiguous for the workgroup's envs)
    for (int i = l; i < n_here * ROW; i += 64) rows[i] = cache[(size_t)env0 * ROW + i];
    if (l < REACH_EPB) fresh[l] = 0;
    __syncthreads();
    // switch_states (npp_switch_states_kernel's arithmetic, npp_render.hip) by the env's second lane, which has nothing else to do:
    // npp_reachability_ex saves the launch of that 8-us kernel
    if (sw_out && (l & 3) == 1 && el < n_here) {
        const LevelHdr &L = a.hdr[a.env_level[env]];
        const double *ex = reinterpret_cast<const double *>(a.blob + L.off_ent_x);
        const double *ey = reinterpret_cast<const double *>(a.blob + L.off_ent_y);
        float *o = sw_out + (size_t)env * 25;
        for (int k = 0; k < 5; k++) {
            const int slot = L.locked_slots[k];
            float v0 = 0.f, v1 = 0.f, v4 = 0.f;
            if (slot >= 0) {
                const double sx = ex[slot] / 1056.0, sy = ey[slot] / 600.0;
                v0 = (float)(sx < 0.0 ? 0.0 : (sx > 1.0 ? 1.0 : sx)); v1 = (float)(sy < 0.0 ? 0.0 : (sy > 1.0 ? 1.0 : sy));
                const uint32_t st = (a.ent_bits[(size_t)(slot >> 4) * a.n + env] >> ((slot & 15) * 2)) & 3u;
                v4 = (st & 1u) ? 0.f : 1.f;
            }
            o[5 * k] = v0; o[5 * k + 1] = v1; o[5 * k + 2] = v0; o[5 * k + 3] = v1; o[5 * k + 4] = v4;
        }
    }
    if ((l & 3) == 0 && el < n_here) {
        const int lvl = a.env_level[env];
        const LevelHdr &L = a.hdr[lvl];
        const ReachHdr &H = rh[lvl];
        const ReachTabs T{&H, rblob + H.base};
        const double px = a.f64[(size_t)F_X * a.n + env], py = a.f64[(size_t)F_Y * a.n + env];
        // exit_switch_activated (nplay_headless.py:566-576): not switch.active
        bool sw = true;
        if (L.obs_switch >= 0) sw = ((a.ent_bits[(size_t)(L.obs_switch >> 4) * a.n + env] >> ((L.obs_switch & 15) * 2)) & 3u) == 0;
        const int cx = reach_cell24(px), cy = reach_cell24(py);
        const uint32_t k = 0x80000000u | ((uint32_t)(cx & 0x3fff)) | ((uint32_t)(cy & 0x3fff) << 14) | ((uint32_t)sw << 28);
        // per-episode dictionary of the path calculator (ReachMiss): a reset of the env since the last call empties it.  The episode
        // counter lives in state word E (npp_kernels.hip: Nj::fastord bits 2-14)
        ReachMiss M{nullptr, nullptr, 1u};
        if (md.stamp) {
            const uint32_t ep = (a.u32[(size_t)U_E * a.n + env] >> 19) & 0x1fffu;
            uint32_t epoch = md.epoch[env];   // 0 = never called (stamps are 0 too: nothing may match)
            if (epoch == 0u || md.last_episode[env] != ep) {
                epoch++;
                if (epoch == 0u) epoch = 1u;
                md.epoch[env] = epoch;
                md.last_episode[env] = ep;
            }
            M.stamp = md.stamp + (size_t)env * REACH_CELLS;
            M.raw = md.raw + (size_t)env * REACH_CELLS;
            M.epoch = epoch;
        }
        float sd[3];
        if (key[env] == k) {
            // mine_sdf_features is read fresh at every observation (npp_environment.py: get_features_at_position)
            int col = (int)(px / 12.0), row = (int)(py / 12.0);
            col = col < 0 ? 0 : (col > SDF_W - 1 ? SDF_W - 1 : col);
            row = row < 0 ? 0 : (row > SDF_H - 1 ? SDF_H - 1 : row);
            sd[0] = 1.f; sd[1] = 0.f; sd[2] = 0.f;
            if (H.off_sdf) {
                sd[0] = T.sdf()[row * SDF_W + col];
                sd[1] = T.grad()[(row * SDF_W + col) * 2];
                sd[2] = T.grad()[(row * SDF_W + col) * 2 + 1];
            }
        } else {
            // live toggle-mine counts (feature_computation.py:541-565): deadly = state 0
            const uint32_t *mm = reinterpret_cast<const uint32_t *>(rblob + H.base + H.off_mine_mask);
            int deadly = 0;
            if (H.n_mines > 0)
                for (uint32_t w = 0; w < H.n_words; w++) {
                    const uint32_t m = mm[w];
                    if (!m) continue;
                    const uint32_t b = a.ent_bits[(size_t)w * a.n + env];
                    deadly += __popc(~(b | (b >> 1)) & m);
                }
            float f[REACH_DIM];
            const int st = reach_features(T, px, py, H.n_mines, deadly, f, sd, &M);   // (M.stamp == NULL: no dictionary; a selected pointer would put M into scratch)
#pragma unroll
            for (int i = 0; i < REACH_DIM; i++) rows[el * ROW + i] = f[i];
            rows[el * ROW + REACH_DIM] = (float)st;
            fresh[el] = 1;
            key[env] = k;
        }
        sdfs[el * 3] = sd[0]; sdfs[el * 3 + 1] = sd[1]; sdfs[el * 3 + 2] = sd[2];
    }
    __syncthreads();
    // ---- write back: recomputed rows to the cache, every row to the outputs
    for (int i = l; i < n_here * ROW; i += 64) {
        const int e = i / ROW, c = i - e * ROW;
        const float v = rows[i];
        if (fresh[e]) cache[(size_t)env0 * ROW + i] = v;
        if (c < REACH_DIM) {
            if (out) out[(size_t)(env0 + e) * REACH_DIM + c] = v;
        } else if (status) status[env0 + e] = (int)v;
    }
    if (sdf_out && l < n_here * 3) sdf_out[(size_t)env0 * 3 + l] = sdfs[l];
}

__global__ __launch_bounds__(256) void npp_reach_restore_kernel(KernelArgs a, const uint32_t *src_key, const float *src_cache,
                                                                uint32_t *key, float *cache, ReachMissDev md) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= a.n) return;
    if (a.reset_mask && !a.reset_mask[env]) return;
    // a restored checkpoint is "reset + replay of the action sequence" in the reference (base_environment.py:1769-1789): the path
    // calculator's dictionary is empty afterwards; a newly assigned level starts empty as well
    if (md.stamp) md.last_episode[env] = 0xffffffffu;
    if (!src_key) { key[env] = 0u; return; }
    key[env] = src_key[env];
    for (int i = 0; i <= REACH_DIM; i++) cache[(size_t)env * (REACH_DIM + 1) + i] = src_cache[(size_t)env * (REACH_DIM + 1) + i];
}

}  // namespace

hipError_t launch_reach_restore(const KernelArgs &a, const uint32_t *src_key, const float *src_cache, uint32_t *key, float *cache,
                                const ReachMissDev &md, hipStream_t s) {
    hipLaunchKernelGGL(npp_reach_restore_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a, src_key, src_cache, key, cache, md);
    return hipGetLastError();
}

hipError_t launch_reach(const KernelArgs &a, const ReachHdr *rh, const unsigned char *rblob, uint32_t *key, float *cache,
                        const ReachMissDev &md, float *out, float *sdf_out, int32_t *status, float *sw_out, hipStream_t s) {
    hipLaunchKernelGGL(npp_reach_kernel, dim3((a.n + REACH_EPB - 1) / REACH_EPB), dim3(64), 0, s, a, rh, rblob, key, cache, md, out, sdf_out, status,
                       sw_out);
    return hipGetLastError();
}

}  // namespace npp
