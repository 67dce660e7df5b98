// npp_reach_kernel.hip -- reachability_features (f32[N, 38]) and mine_sdf_features (f32[N, 3]) of every env from the
// per-level tables in HBM (npp_reach.hpp) and the env state.  One lane per env: the work per env is a handful of dependent
// table look-ups (3 node searches of at most 4 lattice cells, a few f64 distances) and it only runs for the envs whose
// (ninja cell, exit_switch_activated) key changed since their last observation -- the reference's own cache rule
// (gym_environment/mixins/reachability_mixin.py:150-222), kept because the cached vector is what the reference returns.
// Roofline: latency/issue bound; bytes per env = 152 (features out) + 12 (sdf out) + 16 (x, y) + 8 (key, level) + 152 cache
// read or write ~ 340 B -> 2.8 MB per launch at 8192 envs.
#include <hip/hip_runtime.h>

#include "npp_internal.hpp"
#include "npp_reach_features.hpp"

namespace npp {
namespace {

__global__ __launch_bounds__(64) void npp_reach_kernel(KernelArgs a, const ReachHdr *rh, const unsigned char *rblob, uint32_t *key,
                                                       float *cache, ReachMissDev md, float *out, float *sdf_out, int32_t *status) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= a.n) return;
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
    float *c = cache + (size_t)env * (REACH_DIM + 1);
    float f[REACH_DIM], sd[3];
    int st = 0;
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
    if (key[env] == k) {
#pragma unroll
        for (int i = 0; i < REACH_DIM; i++) f[i] = c[i];
        st = (int)c[REACH_DIM];
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
        st = reach_features(T, px, py, H.n_mines, deadly, f, sd, M.stamp ? &M : nullptr);
#pragma unroll
        for (int i = 0; i < REACH_DIM; i++) c[i] = f[i];
        c[REACH_DIM] = (float)st;
        key[env] = k;
    }
    if (out) {
        float *o = out + (size_t)env * REACH_DIM;
#pragma unroll
        for (int i = 0; i < REACH_DIM; i++) o[i] = f[i];
    }
    if (sdf_out) { sdf_out[3 * env] = sd[0]; sdf_out[3 * env + 1] = sd[1]; sdf_out[3 * env + 2] = sd[2]; }
    if (status) status[env] = st;
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
                        const ReachMissDev &md, float *out, float *sdf_out, int32_t *status, hipStream_t s) {
    hipLaunchKernelGGL(npp_reach_kernel, dim3((a.n + 63) / 64), dim3(64), 0, s, a, rh, rblob, key, cache, md, out, sdf_out, status);
    return hipGetLastError();
}

}  // namespace npp
