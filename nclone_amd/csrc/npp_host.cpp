// npp_host.cpp -- the host-only entry points of include/npp_amd.h (no GPU, no handle): the level compiler's views for the CPU
// test-suite, the zoo block plan, the dynamic truncation limit.  No HIP header is included, so this file, npp_level.cpp and
// npp_reach.cpp also build with plain g++ (tools/build_sanitized.sh: -fsanitize=address,undefined).
#include "npp_host.hpp"

#include <cstring>

#include "../../include/npp_amd.h"
#include "npp_reach_build.hpp"

using namespace npp;

namespace npp {
std::string &host_error() {
    static thread_local std::string e;
    return e;
}
}  // namespace npp

namespace {
int fail(std::nullptr_t, int code, const std::string &msg) {
    host_error() = msg;
    return code;
}
}  // namespace

extern "C" {

int npp_compile_level_segments(const double *map, int64_t n, int16_t *out, int max_rows, int *n_out, uint32_t *unsupported_mask) {
    if (!map || !out || !n_out) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_segments: bad arguments");
    CompiledLevel L;
    std::string err;
    if (!compile_level(map, n, L, err)) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_segments: " + err);
    int r = dump_segments(L, out, max_rows);
    if (r < 0) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_segments: buffer too small");
    *n_out = r;
    if (unsupported_mask) *unsupported_mask = L.unsupported_mask;
    return NPP_OK;
}

int npp_compile_level_zoo(const double *map, int64_t n, int32_t *edges_out, double *movers_out, int max_movers, int *n_movers) {
    if (!map || !edges_out || !n_movers) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_zoo: bad arguments");
    CompiledLevel L;
    std::string err;
    if (!compile_level(map, n, L, err)) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_zoo: " + err);
    const int NK = EDGE_W * EDGE_H;
    for (int k = 0; k < NK; k++) {
        edges_out[k] = (int32_t)((L.edges[k >> 5] >> (k & 31)) & 1u);
        edges_out[NK + k] = (int32_t)((L.edges[EDGE_WORDS + (k >> 5)] >> (k & 31)) & 1u);
    }
    for (size_t d = 0; d + 1 < L.door_tab.size(); d += 2) {
        uint32_t keys[2] = {L.door_tab[d] & 0xffffu, L.door_tab[d] >> 16};
        for (uint32_t k : keys) edges_out[((k & 0x8000u) ? NK : 0) + (int)(k & 0x7fffu)] += (int32_t)(L.door_tab[d + 1] & 0xffu);
    }
    int nm = (int)L.mov_meta.size();
    if (movers_out) {
        if (nm > max_movers) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_zoo: buffer too small");
        static const double type_of[7] = {0, 14, 17, 20, 25, 26, 28};
        for (int m = 0; m < nm; m++) {
            movers_out[4 * m] = type_of[L.mov_meta[m] & 7u];
            movers_out[4 * m + 1] = L.mov_x0[m];
            movers_out[4 * m + 2] = L.mov_y0[m];
            movers_out[4 * m + 3] = (double)(L.mov_meta[m] >> 8);
        }
    }
    *n_movers = nm;
    return NPP_OK;
}

int npp_plan_zoo_block(const double *blob, const int64_t *offsets, int n_levels, int *doors, int *movers, int *words) {
    if (!blob || !offsets || n_levels <= 0) return fail(nullptr, NPP_ERR_INVALID, "npp_plan_zoo_block: bad arguments");
    std::vector<CompiledLevel> lv(n_levels);
    for (int i = 0; i < n_levels; i++) {
        std::string err;
        if (offsets[i + 1] < offsets[i] || !compile_level(blob + offsets[i], offsets[i + 1] - offsets[i], lv[i], err))
            return fail(nullptr, NPP_ERR_INVALID, "npp_plan_zoo_block: level " + std::to_string(i) + ": " + err);
    }
    int d = 0, m = 0;
    zoo_block_plan(lv, d, m);
    if (doors) *doors = d;
    if (movers) *movers = m;
    if (words) *words = zoo_words_for(d, m);
    return NPP_OK;
}

int npp_compile_level_entities(const double *map, int64_t n, double *out, int max_rows, int *n_out) {
    if (!map || !out || !n_out) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_entities: bad arguments");
    CompiledLevel L;
    std::string err;
    if (!compile_level(map, n, L, err)) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_entities: " + err);
    int ne = (int)L.ent_map_order.size();
    if (ne > max_rows) return fail(nullptr, NPP_ERR_INVALID, "npp_compile_level_entities: buffer too small");
    // cell of a slot from the CSR
    std::vector<int> cell_of_slot(ne, 0);
    for (int c = 0; c < N_CELLS; c++)
        for (int s = L.ent_start[c]; s < L.ent_start[c + 1]; s++) cell_of_slot[s] = c;
    for (int i = 0; i < ne; i++) {
        int s = L.ent_map_order[i];
        double *o = out + (size_t)i * 6;
        o[0] = L.ent_meta[s] & 15u;
        o[1] = L.ent_x[s];
        o[2] = L.ent_y[s];
        o[3] = cell_of_slot[s] / GRID_H;
        o[4] = cell_of_slot[s] % GRID_H;
        o[5] = (L.ent_meta[s] >> 4) & 3u;
    }
    *n_out = ne;
    return NPP_OK;
}


int npp_level_truncation_limit(const double *map, int64_t n, int32_t *limit, int32_t *surface_area) {
    if (!map) return NPP_ERR_INVALID;
    ReachBuilt R;
    std::string err;
    if (!build_reach(map, n, R, err)) return NPP_ERR_INVALID;
    if (limit) *limit = truncation_limit_for_area(R.spawn_area);
    if (surface_area) *surface_area = R.spawn_area;
    return NPP_OK;
}

}  // extern "C"
