// npp_zoo.hpp -- the "entity zoo" of SURVEY.md 8(f) row 2, device side.  Included by npp_kernels.hip (it uses that
// file's Nj / Lv / EntBits, the lane-group butterflies and the tile primitives) and only instantiated by the ZOO step
// kernel, which runs when some environment plays a level holding any of:
//   regular / trap doors (entity_door_regular.py, entity_door_trap.py), launch pads (entity_launch_pad.py), one-way
//   platforms (entity_one_way_platform.py), zap / mini drones (entity_drone_base.py), bounce blocks
//   (entity_bounce_block.py), thwumps (entity_thwump.py), boost pads (entity_boost_pad.py), death balls
//   (entity_death_ball.py), shove thwumps (entity_shove_thwump.py).
//
// Data model.  Static kinds stay in the level's per-cell CSR (2-bit state word in LDS).  Entities that change grid
// cell ("movers") live in a per-environment block in LDS (copied from / to HBM at kernel entry / exit) together with
// the door edge counters; a mover carries its cell and a list-order number, which reproduces the reference's per-cell
// Python lists exactly: an entity that enters a cell is appended to that cell's list (entities.py Entity.grid_move), so
// inside a cell the iteration order is "creation order for entities that never moved, then order of arrival".  Every
// neighbourhood walk of the ninja (physics.py:79-101) is a merge of the CSR stream with the movers, by (cell, number).
//
// Work split.  Movers are owned by lanes (slot m belongs to lane m mod G of the env's group) for the move / think passes;
// everything that touches the ninja is executed redundantly by all lanes of the group on identical data, like the
// plain path's entity code, so no result has to be broadcast.  Death balls use the group-cooperative tile queries.
#pragma once

struct Zoo {
    double *blk;              // LDS: this env's zoo block (layout: ZOO_HEAD in npp_internal.hpp)
    uint32_t *edges;          // LDS: hor[EDGE_WORDS_D] | ver[EDGE_WORDS_D] tile grid edges of this env's level
    const uint16_t *ent_seq;  // creation order of a CSR entity
    const uint16_t *ent_ord;  // its list-order number inside a cell: creation order, or the entity_dic rank after a fast reset
    const uint16_t *mov_rank; // entity_dic rank of a mover (its list-order number after a fast reset)
    const uint16_t *ent_cell;
    const uint32_t *mov_meta;
    const double *mov_x0;
    const double *mov_y0;
    const uint32_t *door_tab;
    int n_mov, n_door, door_words, n_balls, n_created, ball_first;
    double db_count;
    bool on;                  // this env's level has zoo entities, or its exit switch / door was repositioned
    // npp_set_entity_pos: cells of the repositioned exit switch (only when it left its original cell) / exit door, else -1
    int obs_switch, obs_door, vsw_cell, vdoor_cell;
};

// per-tick accumulators of pre_collision (ninja.py:217-219)
struct ZTick {
    double xcr, ycr, clen;
    int crushable;
    int gcx, gcy;             // cell of the _cached_entities gather (ninja.py:220-222)
    bool phys_near;
};

DEV double *zoo_mov(const Zoo &z, int m) { return z.blk + ZOO_HEAD + z.door_words + ZOO_MOV_WORDS * m; }
DEV uint32_t *zoo_mov_w(const Zoo &z, int m) { return reinterpret_cast<uint32_t *>(zoo_mov(z, m) + 4); }
DEV int *zoo_door(const Zoo &z, int d) { return reinterpret_cast<int *>(z.blk + ZOO_HEAD) + d; }
DEV uint32_t *zoo_head_w(const Zoo &z) { return reinterpret_cast<uint32_t *>(z.blk + 2); }   // [0] counter, [1] fresh
DEV uint32_t *zoo_ovr_w(const Zoo &z) { return reinterpret_cast<uint32_t *>(z.blk + 3); }    // [0] ZOO_OVR_* flags, [1] exit door's list-order number
// static mover attributes are mirrored into the LDS record (bits 17-19 kind, 20-22 orientation, 23-24 mode of word 0) so that
// the per-tick passes never go to the level tables in global memory for them
DEV uint32_t mov_kind(const Zoo &z, int m) { return (zoo_mov_w(z, m)[0] >> 17) & 7u; }
DEV uint32_t mov_orient(const Zoo &z, int m) { return (zoo_mov_w(z, m)[0] >> 20) & 7u; }
DEV uint32_t mov_mode(const Zoo &z, int m) { return (zoo_mov_w(z, m)[0] >> 23) & 3u; }
DEV int door_counter(int v) { return (int)(short)(v & 0xffff); }
DEV int door_pack(int counter, int timer) { return (counter & 0xffff) | (timer << 16); }
DEV int pos_cell(double x, double y) { return cell_coord(x, 43) * 25 + cell_coord(y, 24); }

constexpr uint32_t MKD_DRONE = 1, MKD_BOUNCE = 2, MKD_THWUMP = 3, MKD_BALL = 4, MKD_MINI = 5, MKD_SHOVE = 6;   // == npp::MoverKind

// Level -> initial zoo block (Simulator.reset semantics: every entity re-created).  Called by all lanes of the group.
DEV void zoo_init_block(const Zoo &z, int r, int G, bool fresh) {
    if (r == 0) {
        z.blk[0] = 0; z.blk[1] = 0;
        zoo_head_w(z)[0] = (uint32_t)z.n_created + 1u;   // n_created itself is the number of a repositioned exit switch
        zoo_head_w(z)[1] = fresh ? 1u : 0u;
        if (fresh) { zoo_ovr_w(z)[0] = 0; z.blk[4] = 0; z.blk[5] = 0; z.blk[6] = 0; z.blk[7] = 0; }   // new level: nothing moved
        zoo_ovr_w(z)[1] = 0;
    }
    for (int d = r; d < z.n_door; d += G) *zoo_door(z, d) = door_pack((int)(z.door_tab[2 * d + 1] & 0xffu), 0);
    for (int m = r; m < z.n_mov; m += G) {
        const uint32_t meta = z.mov_meta[m];
        const uint32_t kind = meta & 7u, orient = (meta >> 3) & 7u;
        double *p = zoo_mov(z, m);
        const double x = z.mov_x0[m], y = z.mov_y0[m];
        p[0] = x; p[1] = y;
        uint32_t bits = 0;
        if (kind == MKD_DRONE || kind == MKD_MINI) { p[2] = x; p[3] = y; bits = orient >> 1; }   // dir = orientation // 2
        else { p[2] = 0; p[3] = 0; }
        if (kind == MKD_THWUMP) bits = 1;   // state + 1
        uint32_t *w = zoo_mov_w(z, m);
        w[0] = (uint32_t)pos_cell(x, y) | (bits << 11) | (kind << 17) | (orient << 20) | (((meta >> 6) & 3u) << 23);
        w[1] = meta >> 8;
    }
}

// Simulator.fast_reset (nsim.py:78-140) on the zoo block: nothing is re-created.  Locked and trap doors return to their
// initial edge counter (reset_state -> change_state), regular doors keep counter and timer; movers keep position, speed,
// state and cell and get their entity_dic rank as list-order number (the cell lists are refilled while walking entity_dic);
// the ninja's launch-pad boost direction is not touched by Ninja.reset_state; the "first creation" flag stays.
DEV void zoo_fast_reset_block(const Zoo &z, int r, int G) {
    if (r == 0) {
        zoo_head_w(z)[0] = (uint32_t)z.n_created + 1u;
        zoo_ovr_w(z)[1] = 0;   // the exit door leaves the grid until its switch is hit again
    }
    for (int d = r; d < z.n_door; d += G) {
        const uint32_t t = z.door_tab[2 * d + 1];
        if (((t >> 8) & 3u) != 1u) *zoo_door(z, d) = door_pack((int)(t & 0xffu), 0);   // 1 = regular door: untouched
    }
    for (int m = r; m < z.n_mov; m += G) zoo_mov_w(z, m)[1] = z.mov_rank[m];
}

// ---- grid edges (physics.py:16-18, 210-235; entity_door_base.py:78-89, 99-108) ------------------------------------
DEV bool zoo_edge(const Zoo &z, bool vertical, int x, int y) {
    x = clampi(x, 0, 87);
    y = clampi(y, 0, 49);
    const int key = x * 51 + y;
    int v = (int)((z.edges[(vertical ? EDGE_WORDS_D : 0) + (key >> 5)] >> (key & 31)) & 1u);
    if (z.n_door) {
        const uint32_t want = (uint32_t)key | (vertical ? 0x8000u : 0u);
        for (int d = 0; d < z.n_door; d++) {
            const uint32_t k = z.door_tab[2 * d];
            if ((k & 0xffffu) == want || (k >> 16) == want) v += door_counter(*zoo_door(z, d));
        }
    }
    return v != 0;   // plain integer counters: a negative one is truthy, like in the reference
}
DEV bool is_empty_row(const Zoo &z, int x1, int x2, int y, int dir) {
    if (dir != 1 && dir != -1) return false;
    const int yy = dir == 1 ? y + 1 : y;
    for (int x = x1; x <= x2; x++)
        if (zoo_edge(z, false, x, yy)) return false;
    return true;
}
DEV bool is_empty_column(const Zoo &z, int x, int y1, int y2, int dir) {
    if (dir != 1 && dir != -1) return false;
    const int xx = dir == 1 ? x + 1 : x;
    for (int y = y1; y <= y2; y++)
        if (zoo_edge(z, true, xx, y)) return false;
    return true;
}
DEV void door_add(const Zoo &z, int d, int delta) {
    int v = *zoo_door(z, d);
    *zoo_door(z, d) = door_pack(door_counter(v) + delta, (v >> 16) & 0xff);
}

// physics.py:183-201
DEV bool pen_square(double sx, double sy, double px, double py, double semi, double &nx, double &ny, double &len, double &len2) {
    const double dx = px - sx, dy = py - sy;
    const double penx = semi - dabs(dx), peny = semi - dabs(dy);
    if (penx > 0 && peny > 0) {
        if (peny <= penx) { nx = 0; ny = dy < 0 ? -1 : 1; len = peny; len2 = penx; }
        else { nx = dx < 0 ? -1 : 1; ny = 0; len = penx; len2 = peny; }
        return true;
    }
    return false;
}
// physics.py:457-471
DEV bool overlap_circle_segment(double xpos, double ypos, double radius, double px1, double py1, double px2, double py2) {
    const double px = px2 - px1, py = py2 - py1;
    const double dx = xpos - px1, dy = ypos - py1;
    const double seg_lensq = sq(px) + sq(py);
    double u = (dx * px + dy * py) / seg_lensq;
    u = pymax(u, 0);
    u = pymin(u, 1);
    const double a = px1 + u * px, b = py1 + u * py;
    return sq(xpos - a) + sq(ypos - b) < sq(radius);
}
// physics.py:317-332
DEV void orientation_vec(uint32_t o, double &vx, double &vy) {
    const double diag = 0.70710678118654757;   // math.sqrt(2) / 2
    const int sx = (o == 0 || o == 1 || o == 7) ? 1 : ((o >= 3 && o <= 5) ? -1 : 0);
    const int sy = (o >= 1 && o <= 3) ? 1 : ((o >= 5) ? -1 : 0);
    vx = (o & 1) ? sx * diag : (double)sx;
    vy = (o & 1) ? sy * diag : (double)sy;
}

// ---- movers: move() -------------------------------------------------------------------------------------------------
DEV int dir_vx(int d) { return d == 0 ? 1 : (d == 2 ? -1 : 0); }   // entity_drone_base.py:68
DEV int dir_vy(int d) { return d == 1 ? 1 : (d == 3 ? -1 : 0); }
DEV int dir_list(int mode, int i) {                                // entity_drone_base.py:72
    const uint32_t tab = mode == 0 ? 0x2301u : (mode == 1 ? 0x2103u : (mode == 2 ? 0x2310u : 0x2130u));
    return (int)((tab >> (4 * i)) & 15u);
}

// entity_drone_base.py:138-164
DEV bool drone_test(const Zoo &z, double x, double y, int dir, double R, double grid, double &xt, double &yt) {
    const int xdir = dir_vx(dir), ydir = dir_vy(dir);
    const double xtarget = x + grid * xdir, ytarget = y + grid * ydir;
    if (!ydir) {
        int cell_x = floor12(x + xdir * R);
        const int cell_xtarget = floor12(xtarget + xdir * R);
        const int cell_y1 = floor12(y - R), cell_y2 = floor12(y + R);
        int guard = 0;
        while (cell_x != cell_xtarget && guard++ < 64) {
            if (!is_empty_column(z, cell_x, cell_y1, cell_y2, xdir)) return false;
            cell_x += xdir;
        }
    } else {
        int cell_y = floor12(y + ydir * R);
        const int cell_ytarget = floor12(ytarget + ydir * R);
        const int cell_x1 = floor12(x - R), cell_x2 = floor12(x + R);
        int guard = 0;
        while (cell_y != cell_ytarget && guard++ < 64) {
            if (!is_empty_row(z, cell_x1, cell_x2, cell_y, ydir)) return false;
            cell_y += ydir;
        }
    }
    xt = xtarget; yt = ytarget;
    return true;
}

// One mover's move() (nsim.py:246-247).  Returns true when Entity.grid_move put it into another cell (the caller
// assigns the list-order number); `newcell` is that cell.
DEV bool mover_move(const Zoo &z, int m, int &newcell) {
    const uint32_t kind = mov_kind(z, m);
    double *p = zoo_mov(z, m);
    uint32_t *w = zoo_mov_w(z, m);
    const int cell = (int)(w[0] & 0x7ffu);
    bool gm = false;   // reached Entity.grid_move
    if (kind == MKD_DRONE || kind == MKD_MINI) {   // entity_drone_base.py:93-136
        const double speed = kind == MKD_DRONE ? 8.0 / 7 : 1.3, R = kind == MKD_DRONE ? 7.5 : 4.0, grid = kind == MKD_DRONE ? 24.0 : 12.0;
        const int mode = (int)mov_mode(z, m);
        int dir = (int)((w[0] >> 11) & 3u);
        double x = p[0], y = p[1], xt = p[2], yt = p[3];
        const double xspeed = speed * dir_vx(dir), yspeed = speed * dir_vy(dir);
        const double dx = xt - x, dy = yt - y;
        const double dist = dsqrt(sq(dx) + sq(dy));
        if (dist < 0.000001 || (dx * (xt - (x + xspeed)) + dy * (yt - (y + yspeed))) < 0) {
            x = xt; y = yt;
            bool can_move = false;
            for (int i = 0; i < 4; i++) {
                const int nd = (dir + dir_list(mode, i)) % 4;
                if (drone_test(z, x, y, nd, R, grid, xt, yt)) { dir = nd; can_move = true; break; }
            }
            if (can_move) {
                const double disp = speed - dist;
                x += disp * dir_vx(dir);
                y += disp * dir_vy(dir);
            }
        } else {
            x += xspeed;
            y += yspeed;
            gm = true;
        }
        p[0] = x; p[1] = y; p[2] = xt; p[3] = yt;
        w[0] = (w[0] & ~(3u << 11)) | ((uint32_t)dir << 11);
    } else if (kind == MKD_BOUNCE) {   // entity_bounce_block.py:107-122
        double x = p[0], y = p[1], vx = p[2], vy = p[3];
        vx *= 0.98; vy *= 0.98;
        x += vx; y += vy;
        const double xforce = 0.02222222222222222 * (z.mov_x0[m] - x);
        const double yforce = 0.02222222222222222 * (z.mov_y0[m] - y);
        x += xforce; y += yforce;
        vx += xforce; vy += yforce;
        p[0] = x; p[1] = y; p[2] = vx; p[3] = vy;
        gm = true;
    } else if (kind == MKD_THWUMP) {   // entity_thwump.py:109-156
        int state = (int)((w[0] >> 11) & 3u) - 1;
        if (state) {
            const uint32_t orient = mov_orient(z, m);
            const bool horizontal = (orient == 0 || orient == 4);
            const int direction = (orient == 0 || orient == 2) ? 1 : -1;
            const double speed = state == 1 ? 20.0 / 7 : 8.0 / 7;
            const int speed_dir = direction * state;
            double x = p[0], y = p[1];
            const double origin = horizontal ? z.mov_x0[m] : z.mov_y0[m];
            const double cur = horizontal ? x : y, other = horizontal ? y : x;
            const double pos_new = cur + speed * speed_dir;
            if (state == -1 && (pos_new - origin) * (cur - origin) < 0) {
                if (horizontal) p[0] = origin; else p[1] = origin;
                state = 0;
            } else {
                const int c = floor12(cur + speed_dir * 11), c_new = floor12(pos_new + speed_dir * 11);
                bool blocked = false;
                if (c != c_new) {
                    const int o1 = floor12(other - 11), o2 = floor12(other + 11);
                    blocked = horizontal ? !is_empty_column(z, c, o1, o2, speed_dir) : !is_empty_row(z, o1, o2, c, speed_dir);
                }
                if (blocked) state = -1;
                else {
                    if (horizontal) p[0] = pos_new; else p[1] = pos_new;
                    gm = true;
                }
            }
            w[0] = (w[0] & ~(3u << 11)) | ((uint32_t)(state + 1) << 11);
        }
    }
    if (gm) {
        newcell = pos_cell(p[0], p[1]);
        return newcell != cell;
    }
    return false;
}

// move / think passes over lane-owned movers with list-order numbers handed out in entity_dic order
template <int G, typename F>
DEV void zoo_pass(const Zoo &z, int r, F body) {
    uint32_t ctr = zoo_head_w(z)[0];
    const int glane0 = (threadIdx.x & 63) & ~(G - 1);
    for (int base = 0; base < z.n_mov; base += G) {
        const int m = base + r;
        int newcell = 0;
        const bool moved = (m < z.n_mov) && body(m, newcell);
        unsigned long long bal = __ballot(moved) >> glane0;
        if constexpr (G < 64) bal &= (1ull << G) - 1;
        if (moved) {
            uint32_t *w = zoo_mov_w(z, m);
            w[0] = (w[0] & ~0x7ffu) | (uint32_t)newcell;
            w[1] = ctr + (uint32_t)__builtin_popcountll(bal & ((1ull << r) - 1));
        }
        ctr += (uint32_t)__builtin_popcountll(bal);
    }
    zoo_head_w(z)[0] = ctr;
}

// ---- thinkers ----------------------------------------------------------------------------------------------------------
// entity_thwump.py:158-206
DEV void thwump_think(const Zoo &z, int m, const Nj &n) {
    uint32_t *w = zoo_mov_w(z, m);
    const int state = (int)((w[0] >> 11) & 3u) - 1;
    if (state || !valid_target(n.state)) return;
    const uint32_t orient = mov_orient(z, m);
    const bool horizontal = (orient == 0 || orient == 4);
    const int direction = (orient == 0 || orient == 2) ? 1 : -1;
    const double *p = zoo_mov(z, m);
    const double activation_range = 2 * (9 + NINJA_RADIUS);
    const double mine = horizontal ? p[1] : p[0], theirs = horizontal ? n.y : n.x;   // the axis across the travel
    if (!(dabs(mine - theirs) < activation_range)) return;
    const double along = horizontal ? p[0] : p[1], nalong = horizontal ? n.x : n.y;
    const int ninja_cell = floor12(nalong);
    int tc = floor12(along - direction * 11);
    const int c1 = floor12(mine - 11), c2 = floor12(mine + 11);
    int d = ninja_cell - tc;
    if (d * direction >= 0) {
        int i = 0;
        for (; i < 100; i++) {
            const bool empty = horizontal ? is_empty_column(z, tc, c1, c2, direction) : is_empty_row(z, c1, c2, tc, direction);
            if (!empty) { d = ninja_cell - tc; break; }
            tc += direction;
        }
        if (i == 100) i = 99;   // Python's loop variable after an exhausted range(100)
        if (i > 0 && d * direction <= 0) w[0] = (w[0] & ~(3u << 11)) | (2u << 11);   // set_state(1)
    }
}

// shove thwump bits: state 11-12, activated 13, direction code 14-16 (0 none, 1 +x, 2 -x, 3 +y, 4 -y)
DEV void shove_dir(uint32_t w0, double &xd, double &yd) {
    const uint32_t c = (w0 >> 14) & 7u;
    xd = c == 1 ? 1 : (c == 2 ? -1 : 0);
    yd = c == 3 ? 1 : (c == 4 ? -1 : 0);
}
// entity_shove_thwump.py:109-153.  Returns true when grid_move changed the cell.
DEV bool shove_think(const Zoo &z, int m, int &newcell) {
    double *p = zoo_mov(z, m);
    uint32_t *w = zoo_mov_w(z, m);
    uint32_t w0 = w[0];
    int state = (int)((w0 >> 11) & 3u);
    double xdir, ydir;
    shove_dir(w0, xdir, ydir);
    bool gm = false;
    bool go = true;
    if (state == 1) {
        if (w0 & (1u << 13)) { w0 &= ~(1u << 13); go = false; }   // activated -> False, return
        else state = 2;
    }
    if (go) {
        double mx = 0, my = 0, speed = 0;
        bool moving = false;
        if (state == 3) {
            const double origin_dist = dabs(p[0] - z.mov_x0[m]) + dabs(p[1] - z.mov_y0[m]);
            if (origin_dist >= 1) { mx = xdir; my = ydir; speed = 1; moving = true; }
            else { p[0] = z.mov_x0[m]; p[1] = z.mov_y0[m]; state = 0; }
        } else if (state == 2) {
            mx = -xdir; my = -ydir; speed = 4; moving = true;
        }
        if (moving) {   // move_if_possible
            if (ydir == 0) {
                const double xpos_new = p[0] + mx * speed;
                const int cell_x = floor12(p[0]), cell_x_new = floor12(xpos_new);
                bool blocked = false;
                if (cell_x != cell_x_new) {
                    const int cell_y1 = floor12(p[1] - 8), cell_y2 = floor12(p[1] + 8);
                    blocked = !is_empty_column(z, cell_x, cell_y1, cell_y2, (int)mx);
                }
                if (blocked) state = 3; else { p[0] = xpos_new; gm = true; }
            } else {
                const double ypos_new = p[1] + my * speed;
                const int cell_y = floor12(p[1]), cell_y_new = floor12(ypos_new);
                bool blocked = false;
                if (cell_y != cell_y_new) {
                    const int cell_x1 = floor12(p[0] - 8), cell_x2 = floor12(p[0] + 8);
                    blocked = !is_empty_row(z, cell_x1, cell_x2, cell_y, (int)my);
                }
                if (blocked) state = 3; else { p[1] = ypos_new; gm = true; }
            }
        }
    }
    w[0] = (w0 & ~(3u << 11)) | ((uint32_t)state << 11);
    if (gm) {
        newcell = pos_cell(p[0], p[1]);
        return newcell != (int)(w0 & 0x7ffu);
    }
    return false;
}

// get_single_closest_point with its own region query (physics.py:131-180, segments=None), group-cooperative
template <int G>
__device__ __noinline__ Best closest_generic(TileRefs lv, int r, double px, double py, double radius) {
    const double qx0 = px - radius, qy0 = py - radius, qx1 = px + radius, qy1 = py + radius;
    const int c0x = cell_coord(qx0, 43), c1x = cell_coord(qx1, 43), c0y = cell_coord(qy0, 24), c1y = cell_coord(qy1, 24);
    Best m;
    m.key = __builtin_inf(); m.idx = 0x7fffffff; m.a = 0; m.b = 0;
    int base = 0;
    for (int xc = c0x; xc <= c1x; xc++) {
        const int i0 = lv.seg_start[xc * 25 + c0y], i1 = lv.seg_start[xc * 25 + c1y + 1];
        for (int i = i0 + r; i < i1; i += G) {
            const uint32_t s = lv.segs[i];
            const int yc = s >> 11;
            if (!cell_passes(lv.bounds[xc * 25 + yc], xc, yc, qx0, qy0, qx1, qy1)) continue;
            double bx0, by0, bx1, by1;
            seg_aabb(s, xc, yc, bx0, by0, bx1, by1);
            if (bx1 < qx0 || bx0 > qx1 || by1 < qy0 || by0 > qy1) continue;
            double a, b;
            const bool back = seg_closest(s, xc, yc, px, py, a, b);
            double distance_sq = sq(px - a) + sq(py - b);
            if (!back) distance_sq -= 0.1;
            if (distance_sq < m.key) { m.key = distance_sq; m.a = a; m.b = b; m.idx = ((base + i - i0) << 8) | (r << 1) | (back ? 1 : 0); }
        }
        base += i1 - i0;
    }
    group_argmin<G>(m);
    return m;
}

// get_single_closest_point (physics.py:131-180) answered from candidate registers: the query's own region (cell filter with
// the inclusive cell-bounds test) and per-segment AABB both use the box q = position +- radius; first-wins argmin in the
// reference's iteration order (the gather order).  Same per-candidate arithmetic as the ninja's depenetration loop.
template <int G, int K>
DEV Best cand_closest_query(const Cand<K> &cd, int r, double px, double py, const QBox &q) {
    Best m;
    m.key = __builtin_inf(); m.idx = 0x7fffffff; m.a = 0; m.b = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!cand_in_box(cd, k, q)) continue;
        const double bx0 = __builtin_fmin(cd.x1[k], cd.x2[k]), bx1 = __builtin_fmax(cd.x1[k], cd.x2[k]);
        const double by0 = __builtin_fmin(cd.y1[k], cd.y2[k]), by1 = __builtin_fmax(cd.y1[k], cd.y2[k]);
        if ((bx1 < q.x0) | (bx0 > q.x1) | (by1 < q.y0) | (by0 > q.y1)) continue;
        double a, b;
        bool back;
        if ((cd.s[k] & 1u) == 0) back = cand_closest_lin(cd, k, px, py, a, b);
        else back = cand_closest_arc(cd.s[k], cd.x1[k], cd.y1[k], cd.x2[k], cd.y2[k], px, py, a, b);
        const double distance_sq = sq(px - a) + sq(py - b);
        const double key = back ? distance_sq : distance_sq - 0.1;
        if (key < m.key) { m.key = key; m.a = a; m.b = b; m.idx = ((k * G + r) << 8) | (r << 1) | (back ? 1 : 0); }
    }
    group_argmin<G>(m);
    return m;
}

// entity_death_ball.py:74-167 for ball slot m; all lanes of the group cooperate on the tile queries.  Returns true when
// grid_move changed the cell.
template <int G>
DEV bool ball_think(const Lv &lv, const Zoo &z, int r, const Nj &n, int m, int ball_first, int &newcell) {
    double *p = zoo_mov(z, m);
    double x = p[0], y = p[1], vx = p[2], vy = p[3];
    if (!valid_target(n.state)) {
        vx *= 0.95; vy *= 0.95;
    } else {
        double dx = n.x - x, dy = n.y - y;
        const double dist = dsqrt(sq(dx) + sq(dy));
        if (dist > 0) { dx /= dist; dy /= dist; }
        vx += dx * 0.04;
        vy += dy * 0.04;
        const double speed = dsqrt(sq(vx) + sq(vy));
        if (speed > 0.85) {
            double new_speed = (speed - 0.85) * 0.9;
            if (new_speed <= 0.01) new_speed = 0;
            new_speed += 0.85;
            vx = vx / speed * new_speed;
            vy = vy / speed * new_speed;
        }
    }
    const double xold = x, yold = y;
    const TileRefs tr{lv.seg_start, lv.segs, lv.bounds};
    // cells that the sweep (radius 4 + 1) and the first closest-point query (radius 8) can look at: when they hold no
    // segment at all, the sweep returns 1 and the query returns "nothing" (physics.py:116-128, 141-180)
    const double xn = xold + vx, yn = yold + vy;
    const double ux0 = (xold < xn ? xold : xn), ux1 = (xold > xn ? xold : xn), uy0 = (yold < yn ? yold : yn), uy1 = (yold > yn ? yold : yn);
    bool open_space;
    {
        const int c0x = cell_coord(ux0 - 8, 43), c1x = cell_coord(ux1 + 8, 43), c0y = cell_coord(uy0 - 8, 24), c1y = cell_coord(uy1 + 8, 24);
        int nseg = 0;
        for (int xc = c0x; xc <= c1x; xc++) nseg += (int)lv.seg_start[xc * 25 + c1y + 1] - (int)lv.seg_start[xc * 25 + c0y];
        open_space = nseg == 0;
    }
    double xnormal = 0, ynormal = 0;
    bool bail = false;
    if (open_space) {
        x = xn; y = yn;
    } else {
        // like the ninja's tick: gather the segments around the ball's path once into registers and answer the sweep and
        // every closest-point query from them; a query that leaves the gathered cells takes the table walk
        constexpr int K = KSlots<G>::value;
        Cand<K> cd;
        cand_gather<G, K>(lv, r, ux0 - 10.0, uy0 - 10.0, ux1 + 10.0, uy1 + 10.0, cd);
        double time = 1;
        {
            const double radius = 8 * 0.5, width = radius + 1;
            const QBox q = make_qbox(ux0 - width, uy0 - width, ux1 + width, uy1 + width);
            if (cand_covers(cd, q)) {
                const double vel_sq = sq(vx) + sq(vy);
                double shortest = 1;
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (cand_in_box(cd, k, q)) {
                        const double t = cand_toi(cd.s[k], cd.x1[k], cd.y1[k], cd.x2[k], cd.y2[k], xold, yold, vx, vy, vel_sq, radius);
                        if (t < shortest) shortest = t;
                    }
                time = group_min<G>(shortest);
            } else {
                time = sweep_generic<G>(tr, r, xold, yold, vx, vy, radius);
            }
        }
        x = xold + time * vx;
        y = yold + time * vy;
        for (int it = 0; it < 16; it++) {
            const QBox q = make_qbox(x - 8.0, y - 8.0, x + 8.0, y + 8.0);
            const Best c = cand_covers(cd, q) ? cand_closest_query<G, K>(cd, r, x, y, q) : closest_generic<G>(tr, r, x, y, 8.0);
            if (c.idx == 0x7fffffff) break;
            const int result = (c.idx & 1) ? -1 : 1;
            const double dx = x - c.a, dy = y - c.b;
            const double dist = dsqrt(sq(dx) + sq(dy));
            const double depen_len = 8 - dist * result;
            if (depen_len < 0.0000001) break;
            if (dist == 0) { bail = true; break; }   // `return` in the reference: nothing below runs
            const double xnorm = dx / dist, ynorm = dy / dist;
            x += xnorm * depen_len;
            y += ynorm * depen_len;
            xnormal += xnorm;
            ynormal += ynorm;
        }
    }
    if (!bail) {
        const double normal_len = dsqrt(sq(xnormal) + sq(ynormal));
        if (normal_len > 0) {
            const double dx = xnormal / normal_len, dy = ynormal / normal_len;
            const double dot_product = vx * dx + vy * dy;
            if (dot_product < 0) {
                const double speed = dsqrt(sq(vx) + sq(vy));
                const int bounce_strength = speed <= 1.35 ? 1 : 2;
                vx -= dx * dot_product * bounce_strength;
                vy -= dy * dot_product * bounce_strength;
            }
        }
    }
    p[0] = x; p[1] = y; p[2] = vx; p[3] = vy;
    if (bail) return false;
    // ball-ball repulsion (:153-166): Entity.index is a per-Simulator creation counter that Simulator.reset() never
    // clears, so `self.index + 1 < db_count` only holds while the entities are in their first creation
    const int index = m - ball_first;
    if (zoo_head_w(z)[1] != 0 && (double)(index + 1) < z.db_count) {
        for (int k = index + 1; k < z.n_balls; k++) {
            double *t = zoo_mov(z, ball_first + k);
            double dx = x - t[0], dy = y - t[1];
            const double dist = dsqrt(sq(dx) + sq(dy));
            if (dist < 16) {
                dx = dx / dist * 4;
                dy = dy / dist * 4;
                vx += dx; vy += dy;
                t[2] -= dx; t[3] -= dy;
            }
        }
        p[2] = vx; p[3] = vy;
    }
    newcell = pos_cell(x, y);
    return newcell != (int)(zoo_mov_w(z, m)[0] & 0x7ffu);
}

// ---- the ninja's neighbourhood: movers merged into the CSR walk by (cell, list-order number) ---------------------------
DEV int zoo_key(int cell, uint32_t seq) { return (cell << 20) | (int)(seq & 0xfffffu); }

// next mover of the 3x3 block [x0..x1] x [y0..y1] with lo < key < hi; `phys` restricts to physically collidable kinds
template <int G>
DEV int zoo_next(const Zoo &z, int r, int lo, int hi, int x0, int x1, int y0, int y1, bool phys, int &key_out) {
    int best = 0x7fffffff, slot = 0x7fffffff;
    for (int m = r; m < z.n_mov; m += G) {
        const uint32_t kind = mov_kind(z, m);
        if (phys && !(kind == MKD_BOUNCE || kind == MKD_THWUMP || kind == MKD_SHOVE)) continue;
        const uint32_t *w = zoo_mov_w(z, m);
        const int cell = (int)(w[0] & 0x7ffu);
        const int cx = cell / 25, cy = cell - cx * 25;
        if (cx < x0 || cx > x1 || cy < y0 || cy > y1) continue;
        const int key = zoo_key(cell, w[1]);
        if (key > lo && key < hi && key < best) { best = key; slot = m; }
    }
    if (!phys) {
        // a repositioned exit switch that left its cell was appended to the new cell's list right after level load (number
        // n_created); the exit door joins its cell's list when the switch is hit (number taken then).  Slots n_mov, n_mov + 1.
        if (z.vsw_cell >= 0) {
            const int cx = z.vsw_cell / 25, cy = z.vsw_cell - cx * 25;
            const int key = zoo_key(z.vsw_cell, (uint32_t)z.n_created);
            if (cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1 && key > lo && key < hi && key < best) { best = key; slot = z.n_mov; }
        }
        if (z.vdoor_cell >= 0 && zoo_ovr_w(z)[1] != 0) {
            const int cx = z.vdoor_cell / 25, cy = z.vdoor_cell - cx * 25;
            const int key = zoo_key(z.vdoor_cell, zoo_ovr_w(z)[1]);
            if (cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1 && key > lo && key < hi && key < best) { best = key; slot = z.n_mov + 1; }
        }
    }
    const int kmin = group_min_i<G>(best);
    if (kmin == 0x7fffffff) return -1;
    const int s = group_min_i<G>(best == kmin ? slot : 0x7fffffff);
    key_out = kmin;
    return s;
}

// entity_one_way_platform.py:79-105
DEV bool oneway_depen(double ex, double ey, double nx, double ny, const Nj &n, double xold, double yold, double &len) {
    const double dx = n.x - ex, dy = n.y - ey;
    const double lateral_dist = dy * nx - dx * ny;
    const double direction = (n.vy * nx - n.vx * ny) * lateral_dist;
    const double radius_scalar = direction < 0 ? 0.91 : 0.51;
    if (dabs(lateral_dist) < radius_scalar * NINJA_RADIUS + 12) {
        const double normal_dist = dx * nx + dy * ny;
        if (0 < normal_dist && normal_dist <= NINJA_RADIUS) {
            const double normal_proj = n.vx * nx + n.vy * ny;
            if (normal_proj <= 0) {
                const double dx_old = xold - ex, dy_old = yold - ey;
                const double normal_dist_old = dx_old * nx + dy_old * ny;
                if (NINJA_RADIUS - normal_dist_old <= 1.1) { len = NINJA_RADIUS - normal_dist; return true; }
            }
        }
    }
    return false;
}

// the body of Ninja.collide_vs_objects for one depenetration (ninja.py:231-267); type = Entity.type
DEV void apply_physical(Nj &n, ZTick &zt, int type, double depen_x, double depen_y, double depen_len, double &fnsx, double &fnsy,
                        double &cnsx, double &cnsy) {
    const double pop_x = depen_x * depen_len, pop_y = depen_y * depen_len;
    n.x += pop_x;
    n.y += pop_y;
    if (type != 17) { zt.xcr += pop_x; zt.ycr += pop_y; zt.clen += depen_len; }
    if (type == 20) zt.crushable = 1;
    if (type == 17 || type == 20 || type == 28) { n.vx += pop_x; n.vy += pop_y; }
    if (type == 11) {
        const double xspeed_new = (n.vx * depen_y - n.vy * depen_x) * depen_y;
        const double yspeed_new = (n.vx * depen_y - n.vy * depen_x) * (-depen_x);
        n.vx = xspeed_new;
        n.vy = yspeed_new;
    }
    if (depen_y >= -0.0001) { n.ccount += 1; cnsx += depen_x; cnsy += depen_y; }
    else { n.fcount += 1; fnsx += depen_x; fnsy += depen_y; }
}

// physical_collision() of a mover (entity_bounce_block.py:127-145, entity_thwump.py:208-213, entity_shove_thwump.py:155-171)
DEV void mover_physical(const Zoo &z, int m, Nj &n, ZTick &zt, double &fnsx, double &fnsy, double &cnsx, double &cnsy) {
    const uint32_t kind = mov_kind(z, m);
    double *p = zoo_mov(z, m);
    double nx, ny, len, len2;
    if (kind == MKD_BOUNCE) {
        if (!pen_square(p[0], p[1], n.x, n.y, 9 + NINJA_RADIUS, nx, ny, len, len2)) return;
        p[0] -= nx * len * (1 - 0.2);
        p[1] -= ny * len * (1 - 0.2);
        p[2] -= nx * len * (1 - 0.2);
        p[3] -= ny * len * (1 - 0.2);
        apply_physical(n, zt, 17, nx, ny, len * 0.2, fnsx, fnsy, cnsx, cnsy);
    } else if (kind == MKD_THWUMP) {
        if (!pen_square(p[0], p[1], n.x, n.y, 9 + NINJA_RADIUS, nx, ny, len, len2)) return;
        apply_physical(n, zt, 20, nx, ny, len, fnsx, fnsy, cnsx, cnsy);
    } else if (kind == MKD_SHOVE) {
        const uint32_t w0 = zoo_mov_w(z, m)[0];
        const int state = (int)((w0 >> 11) & 3u);
        if (state > 1) return;
        if (!pen_square(p[0], p[1], n.x, n.y, 12 + NINJA_RADIUS, nx, ny, len, len2)) return;
        double xdir, ydir;
        shove_dir(w0, xdir, ydir);
        if (state == 0 || xdir * nx + ydir * ny >= 0.01) apply_physical(n, zt, 28, nx, ny, len, fnsx, fnsy, cnsx, cnsy);
    }
}

// block of the cached gather + whether anything physical is in it (ninja.py:220-222)
template <int G>
DEV void zoo_pre_collision(const Lv &lv, const Zoo &z, int r, const Nj &n, ZTick &zt) {
    zt.xcr = 0; zt.ycr = 0; zt.clen = 0; zt.crushable = 0;
    zt.gcx = cell_coord(n.x, 43);
    zt.gcy = cell_coord(n.y, 24);
    const int cx = zt.gcx, cy = zt.gcy;
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < 43 ? cx + 1 : 43, y0 = cy > 0 ? cy - 1 : 0, y1 = cy < 24 ? cy + 1 : 24;
    bool any = false;
    for (int xc = x0; xc <= x1; xc++) {
        const int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int i = i0; i < i1; i++) any = any || ((lv.ent_meta[i] & 15u) == EK_ONEWAY);
    }
    int key;
    any = any || zoo_next<G>(z, r, -1, 0x7fffffff, x0, x1, y0, y1, true, key) >= 0;
    zt.phys_near = any;
}

// Ninja.collide_vs_objects (ninja.py:224-267) over the cached neighbourhood
template <int G>
DEV void collide_vs_objects(const Lv &lv, const Zoo &z, int r, Nj &n, ZTick &zt, double xold, double yold, double &fnsx,
                            double &fnsy, double &cnsx, double &cnsy) {
    const int cx = zt.gcx, cy = zt.gcy;
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < 43 ? cx + 1 : 43, y0 = cy > 0 ? cy - 1 : 0, y1 = cy < 24 ? cy + 1 : 24;
    // the next mover in list order is looked up once and again only after it has been consumed
    int nk = 0;
    int nm = zoo_next<G>(z, r, -1, 0x7fffffff, x0, x1, y0, y1, true, nk);
    for (int xc = x0; xc <= x1; xc++) {
        const int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int i = i0; i < i1; i++) {
            const uint32_t meta = lv.ent_meta[i];
            if ((meta & 15u) != EK_ONEWAY) continue;
            const int key = zoo_key(z.ent_cell[i], z.ent_ord[i]);
            while (nm >= 0 && nk < key) {
                mover_physical(z, nm, n, zt, fnsx, fnsy, cnsx, cnsy);
                const int last = nk;
                nm = zoo_next<G>(z, r, last, 0x7fffffff, x0, x1, y0, y1, true, nk);
            }
            double nx, ny, len;
            orientation_vec((meta >> 8) & 7u, nx, ny);
            if (oneway_depen(lv.ent_x[i], lv.ent_y[i], nx, ny, n, xold, yold, len))
                apply_physical(n, zt, 11, nx, ny, len, fnsx, fnsy, cnsx, cnsy);
        }
    }
    while (nm >= 0) {
        mover_physical(z, nm, n, zt, fnsx, fnsy, cnsx, cnsy);
        const int last = nk;
        nm = zoo_next<G>(z, r, last, 0x7fffffff, x0, x1, y0, y1, true, nk);
    }
}

// logical_collision() of a mover; adds to wall_normal what the reference's post_collision would (ninja.py:419-420)
DEV void mover_logical(const Zoo &z, int m, Nj &n, double &wall_normal) {
    const uint32_t kind = mov_kind(z, m);
    double *p = zoo_mov(z, m);
    double nx, ny, len, len2;
    if (kind == MKD_DRONE || kind == MKD_MINI) {   // entity_drone_zap.py:57-64, entity_mini_drone.py:60-67
        if (valid_target(n.state) && overlaps(p[0], p[1], (kind == MKD_DRONE ? 7.5 : 4.0) + NINJA_RADIUS, n.x, n.y)) ninja_kill(n, 3);
    } else if (kind == MKD_BOUNCE) {   // entity_bounce_block.py:147-158
        if (pen_square(p[0], p[1], n.x, n.y, 9 + NINJA_RADIUS + 0.1, nx, ny, len, len2)) wall_normal += nx;
    } else if (kind == MKD_THWUMP) {   // entity_thwump.py:215-243
        if (valid_target(n.state) && pen_square(p[0], p[1], n.x, n.y, 9 + NINJA_RADIUS + 0.1, nx, ny, len, len2)) {
            const uint32_t orient = mov_orient(z, m);
            const bool horizontal = (orient == 0 || orient == 4);
            const int direction = (orient == 0 || orient == 2) ? 1 : -1;
            double px1, py1, px2, py2;
            if (horizontal) {
                const double dx = (9 + 2) * direction, dy = 9 - 2;
                px1 = p[0] + dx; py1 = p[1] - dy; px2 = p[0] + dx; py2 = p[1] + dy;
            } else {
                const double dx = 9 - 2, dy = (9 + 2) * direction;
                px1 = p[0] - dx; py1 = p[1] + dy; px2 = p[0] + dx; py2 = p[1] + dy;
            }
            if (overlap_circle_segment(n.x, n.y, NINJA_RADIUS + 2, px1, py1, px2, py2)) ninja_kill(n, 3);
            wall_normal += nx;
        }
    } else if (kind == MKD_BALL) {   // entity_death_ball.py:169-181
        if (valid_target(n.state) && overlaps(p[0], p[1], 5.0 + NINJA_RADIUS, n.x, n.y)) {
            const double dx = p[0] - n.x, dy = p[1] - n.y;
            const double dist = dsqrt(sq(dx) + sq(dy));
            p[2] += dx / dist * 10;
            p[3] += dy / dist * 10;
            ninja_kill(n, 3);
        }
    } else if (kind == MKD_SHOVE) {   // entity_shove_thwump.py:173-202
        uint32_t *w = zoo_mov_w(z, m);
        uint32_t w0 = w[0];
        const int state = (int)((w0 >> 11) & 3u);
        const bool depen = pen_square(p[0], p[1], n.x, n.y, 12 + NINJA_RADIUS + 0.1, nx, ny, len, len2);
        if (depen && state <= 1) {
            if (state == 0) {
                w0 |= 1u << 13;
                if (len2 > 0.2) {
                    const uint32_t code = nx > 0 ? 1u : (nx < 0 ? 2u : (ny > 0 ? 3u : 4u));
                    w0 = (w0 & ~(7u << 14) & ~(3u << 11)) | (code << 14) | (1u << 11);
                }
                w[0] = w0;
                wall_normal += nx;
            } else {
                double xdir, ydir;
                shove_dir(w0, xdir, ydir);
                if (xdir * nx + ydir * ny >= 0.01) { w[0] = w0 | (1u << 13); wall_normal += nx; }
            }
            return;
        }
        if (overlaps(n.x, n.y, NINJA_RADIUS + 8.0, p[0], p[1])) ninja_kill(n, 3);
    }
}

// logical collisions of post_collision over the merged neighbourhood (ninja.py:388-420).  Returns the entities' part
// of wall_normal.
template <int G>
DEV double logical_collisions_zoo(const Lv &lv, const Zoo &z, int r, Nj &n, EntBits eb, double xold, double yold) {
    const int cx = cell_coord(n.x, 43), cy = cell_coord(n.y, 24);
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < 43 ? cx + 1 : 43, y0 = cy > 0 ? cy - 1 : 0, y1 = cy < 24 ? cy + 1 : 24;
    int pend0 = -1, pend1 = -1;
    double wall_normal = 0;
    int nk = 0;
    int nm = (z.n_mov || z.vsw_cell >= 0 || z.vdoor_cell >= 0) ? zoo_next<G>(z, r, -1, 0x7fffffff, x0, x1, y0, y1, false, nk) : -1;
    // a mover, or one of the two repositioned entities (slots n_mov: exit switch, n_mov + 1: exit door)
    auto visit = [&](int m) {
        if (m < z.n_mov) { mover_logical(z, m, n, wall_normal); return; }
        if (m == z.n_mov) {   // entity_exit_switch.py:67-129 at its new place
            if (ent_get(eb, z.obs_switch) == 1 && overlaps(z.blk[4], z.blk[5], 6.0 + NINJA_RADIUS, n.x, n.y)) {
                ent_set(eb, z.obs_switch, 0);
                zoo_ovr_w(z)[1] = zoo_head_w(z)[0];
                zoo_head_w(z)[0] += 1;
                if (pend0 < 0) pend0 = z.obs_door; else pend1 = z.obs_door;
            }
        } else if (ent_get(eb, z.obs_door) == 1) {   // entity_exit.py:66-74 at its new place
            if (overlaps(z.blk[6], z.blk[7], 12.0 + NINJA_RADIUS, n.x, n.y)) ninja_win(n);
        }
    };
    for (int xc = x0; xc <= x1; xc++) {
        const int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
        for (int i = i0; i < i1; i++) {
            const uint32_t meta = lv.ent_meta[i];
            const uint32_t kind = meta & 15u;
            const uint32_t st = ent_get(eb, i);
            if (kind == EK_EXIT && st == 0) continue;   // not in the grid: it does not even take a place in the order
            if ((i == z.obs_switch && z.vsw_cell >= 0) || (i == z.obs_door && z.vdoor_cell >= 0)) continue;   // lives elsewhere now
            if (nm >= 0) {
                // the exit door was appended to its cell's list when its switch was hit: it carries that number
                const int key = zoo_key(z.ent_cell[i], (i == z.obs_door && zoo_ovr_w(z)[1] != 0) ? zoo_ovr_w(z)[1] : (uint32_t)z.ent_ord[i]);
                while (nm >= 0 && nk < key) {
                    visit(nm);
                    const int last = nk;
                    nm = zoo_next<G>(z, r, last, 0x7fffffff, x0, x1, y0, y1, false, nk);
                }
            }
            double ex = lv.ent_x[i], ey = lv.ent_y[i];
            if (i == z.obs_switch && (zoo_ovr_w(z)[0] & ZOO_OVR_SWITCH)) { ex = z.blk[4]; ey = z.blk[5]; }   // moved inside its cell
            if (kind == EK_MINE) {   // entity_toggle_mine.py:120-128
                if (valid_target(n.state) && st == 0 && overlaps(ex, ey, 4.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(eb, i, 1);
                    ninja_kill(n, 1);
                }
            } else if ((st & 1u) == 0) {
                continue;   // inactive
            } else if (kind == EK_GOLD) {
                if (n.state != 8 && overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) { n.gold += 1; ent_set(eb, i, 0); }
            } else if (kind == EK_EXIT) {
                if (overlaps(ex, ey, 12.0 + NINJA_RADIUS, n.x, n.y)) ninja_win(n);
            } else if (kind == EK_SWITCH) {
                if (overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(eb, i, 0);
                    const int door = (int)((meta >> 8) & 0xffffu);
                    if (door == z.obs_door) {   // list-order number of the append (entity_exit_switch.py:120)
                        zoo_ovr_w(z)[1] = zoo_head_w(z)[0];
                        zoo_head_w(z)[0] += 1;
                    }
                    if (pend0 < 0) pend0 = door; else pend1 = door;
                }
            } else if (kind == EK_LOCKED) {   // entity_door_locked.py:54-67
                if (overlaps(ex, ey, 5.0 + NINJA_RADIUS, n.x, n.y)) {
                    n.doors += 1;
                    ent_set(eb, i, 0);
                    door_add(z, (int)((meta >> 8) & 0xffffu), -1);
                }
            } else if (kind == EK_DOOR_REG) {   // entity_door_regular.py:55-63: EVERY overlapping frame decrements again
                if (overlaps(ex, ey, 10.0 + NINJA_RADIUS, n.x, n.y)) {
                    const int d = (int)((meta >> 8) & 0xffffu);
                    ent_set(eb, i, 1);   // active, open
                    *zoo_door(z, d) = door_pack(door_counter(*zoo_door(z, d)) - 1, 0);
                }
            } else if (kind == EK_DOOR_TRAP) {   // entity_door_trap.py:60-67
                if (overlaps(ex, ey, 5.0 + NINJA_RADIUS, n.x, n.y)) {
                    ent_set(eb, i, 0);
                    door_add(z, (int)((meta >> 8) & 0xffffu), 1);
                }
            } else if (kind == EK_LAUNCH) {   // entity_launch_pad.py:82-100 + ninja.py:401-418
                if (valid_target(n.state) && overlaps(ex, ey, 6.0 + NINJA_RADIUS, n.x, n.y)) {
                    double nx, ny;
                    orientation_vec((meta >> 8) & 7u, nx, ny);
                    if (((ex - (n.x - NINJA_RADIUS * nx)) * nx + (ey - (n.y - NINJA_RADIUS * ny)) * ny) >= -0.1) {
                        double yboost_scale = 1;
                        if (ny < 0) yboost_scale = 1 - ny;
                        const double xb = nx * (36.0 / 7), yb = ny * (36.0 / 7) * yboost_scale;
                        const double xboost = xb * 2 / 3, yboost = yb * 2 / 3;
                        n.x += xboost;
                        n.y += yboost;
                        n.vx = xboost;
                        n.vy = yboost;
                        n.fcount = 0;
                        n.fbuf = -1;
                        const double boost_scalar = dsqrt(sq(xboost) + sq(yboost));
                        z.blk[0] = xboost / boost_scalar;
                        z.blk[1] = yboost / boost_scalar;
                        n.lbuf = 0;
                        if (n.state == 3) n.gjump = 0;
                        n.state = 4;
                    }
                }
            } else if (kind == EK_ONEWAY) {   // entity_one_way_platform.py:111-116
                double nx, ny, len;
                orientation_vec((meta >> 8) & 7u, nx, ny);
                if (oneway_depen(ex, ey, nx, ny, n, xold, yold, len) && dabs(nx) == 1) wall_normal += nx;
            }
        }
    }
    while (nm >= 0) {
        visit(nm);
        const int last = nk;
        nm = zoo_next<G>(z, r, last, 0x7fffffff, x0, x1, y0, y1, false, nk);
    }
    if (pend0 >= 0) ent_set(eb, pend0, 1);
    if (pend1 >= 0) ent_set(eb, pend1, 1);
    return wall_normal;
}

// boost pads' move() (entity_boost_pad.py:46-64) + mines' think() + regular doors' think() for everything that can change
// this tick: only entities within one cell of the ninja now or at the previous tick can (radii <= 16 px < 24 px).
DEV void zoo_think_static(const Lv &lv, const Zoo &z, Nj &n, EntBits eb) {
    const int ccx = cell_coord(n.x, 43), ccy = cell_coord(n.y, 24);
    const int pcx = n.pcell / 25, pcy = n.pcell - pcx * 25;
    n.pcell = ccx * 25 + ccy;
    const bool vt = valid_target(n.state);
    int x0 = (ccx < pcx ? ccx : pcx) - 1, x1 = (ccx > pcx ? ccx : pcx) + 1;
    int y0 = (ccy < pcy ? ccy : pcy) - 1, y1 = (ccy > pcy ? ccy : pcy) + 1;
    x0 = x0 < 0 ? 0 : x0; x1 = x1 > 43 ? 43 : x1; y0 = y0 < 0 ? 0 : y0; y1 = y1 > 24 ? 24 : y1;
    // boost pads first (they are movable, nsim.py:246-247); first touches are applied in creation order
    int done_seq = -1;
    for (;;) {
        int best = 0x7fffffff, best_i = -1, pending = 0;
        for (int xc = x0; xc <= x1; xc++) {
            const int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
            for (int i = i0; i < i1; i++) {
                if ((lv.ent_meta[i] & 15u) != EK_BOOST) continue;
                const uint32_t st = ent_get(eb, i);
                const bool ov = vt && overlaps(lv.ent_x[i], lv.ent_y[i], 6.0 + NINJA_RADIUS, n.x, n.y);
                if (!ov) { if (st != 1) ent_set(eb, i, 1); continue; }
                if (st == 3) continue;   // already touching
                const int sq_ = z.ent_seq[i];
                if (sq_ <= done_seq) continue;
                pending++;
                if (sq_ < best) { best = sq_; best_i = i; }
            }
        }
        if (best_i < 0) break;
        const double vel_norm = dsqrt(sq(n.vx) + sq(n.vy));
        if (vel_norm > 0) {
            const double x_boost = 2 * n.vx / vel_norm, y_boost = 2 * n.vy / vel_norm;
            n.vx += x_boost;
            n.vy += y_boost;
        }
        ent_set(eb, best_i, 3);
        done_seq = best;
        if (pending <= 1) break;
    }
    // mines (entity_toggle_mine.py:90-118)
    if (lv.n_think && (vt || n.state == 6)) {
        for (int xc = x0; xc <= x1; xc++) {
            const int i0 = lv.ent_start[xc * 25 + y0], i1 = lv.ent_start[xc * 25 + y1 + 1];
            for (int i = i0; i < i1; i++) {
                if ((lv.ent_meta[i] & 15u) != EK_MINE) continue;
                const uint32_t st = ent_get(eb, i);
                if (vt) {
                    if (st == 1) { if (overlaps(lv.ent_x[i], lv.ent_y[i], 3.5 + NINJA_RADIUS, n.x, n.y)) ent_set(eb, i, 2); }
                    else if (st == 2) { if (!overlaps(lv.ent_x[i], lv.ent_y[i], 4.5 + NINJA_RADIUS, n.x, n.y)) ent_set(eb, i, 0); }
                } else if (st == 2) {
                    ent_set(eb, i, 1);
                }
            }
        }
    }
}

// regular doors' think (entity_door_regular.py:46-53); doors are few, so every door of the level is visited
DEV void zoo_think_doors(const Lv &lv, const Zoo &z, EntBits eb, int n_ent) {
    if (!z.n_door) return;
    for (int i = 0; i < n_ent; i++) {
        if ((lv.ent_meta[i] & 15u) != EK_DOOR_REG) continue;
        if (ent_get(eb, i) != 1) continue;   // closed
        const int d = (int)((lv.ent_meta[i] >> 8) & 0xffffu);
        const int v = *zoo_door(z, d);
        const int timer = ((v >> 16) & 0xff) + 1;
        if (timer > 5) {
            ent_set(eb, i, 3);
            *zoo_door(z, d) = door_pack(door_counter(v) + 1, timer);
        } else {
            *zoo_door(z, d) = door_pack(door_counter(v), timer);
        }
    }
}

// everything of Simulator.tick that precedes the ninja's integrate (nsim.py:235-251)
template <int G>
DEV void zoo_entities_tick(const Lv &lv, const Zoo &z, int r, Nj &n, EntBits eb, int n_ent) {
    // move(): drones, bounce blocks, thwumps (lane-owned), then boost pads
    if (z.n_mov) zoo_pass<G>(z, r, [&](int m, int &newcell) { return mover_move(z, m, newcell); });
    zoo_think_static(lv, z, n, eb);
    // think(): regular doors (key 5), thwumps (20), death balls (25), shove thwumps (28)
    zoo_think_doors(lv, z, eb, n_ent);
    if (z.n_mov) {
        for (int m = r; m < z.n_mov; m += G)
            if (mov_kind(z, m) == MKD_THWUMP) thwump_think(z, m, n);
        if (z.n_balls) {
            const int ball_first = z.ball_first;
            uint32_t ctr = zoo_head_w(z)[0];
            for (int k = 0; k < z.n_balls; k++) {
                const int m = ball_first + k;
                int newcell = 0;
                if (ball_think<G>(lv, z, r, n, m, ball_first, newcell)) {
                    uint32_t *w = zoo_mov_w(z, m);
                    w[0] = (w[0] & ~0x7ffu) | (uint32_t)newcell;
                    w[1] = ctr++;
                }
            }
            zoo_head_w(z)[0] = ctr;
        }
        zoo_pass<G>(z, r, [&](int m, int &newcell) {
            return mov_kind(z, m) == MKD_SHOVE ? shove_think(z, m, newcell) : false;
        });
    }
}

// ninja.py:531-537
DEV void zoo_crush_check(Nj &n, const ZTick &zt) {
    if (zt.crushable && zt.clen > 0) {
        if (dsqrt(sq(zt.xcr) + sq(zt.ycr)) / zt.clen < 0.05) ninja_kill(n, 3);
    }
}
