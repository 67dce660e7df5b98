/*
 * nsim_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64) of the
 * reference's per-tick physics/collision path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (nclone_amd/)
 * never does.
 *
 * Parity status: PINNED.  In the build container this restatement is checked
 * bit-for-bit (fp64 positions/velocities, every discrete field) against fixtures
 * produced by running the reference itself (tests/golden/make_golden.py): all 104
 * in-scope bc_replays (25 575 ticks) plus 31 random-action rollouts (49 396 ticks), and
 * -- with the entity zoo of SURVEY.md 8(f) row 2 (regular/trap doors, launch pads, one-way
 * platforms, zap and mini drones, bounce blocks, thwumps, boost pads, death balls, shove
 * thwumps) -- the remaining 26 bc_replays (5 886 ticks, per-tick entity checksums) and
 * 9 zoo rollouts (10 776 ticks) from tests/golden/make_golden_zoo.py: 130 of 130 replays.
 * plus 28 random "entity soup" levels run through the reference (make_golden_fuzz.py, 12 537 ticks:
 * regular / trap doors, shove thwumps, every orientation and drone mode) and repositioned exit
 * switches / doors (make_golden_moved.py).
 *
 * Where the reference writes `x**2` CPython calls libm pow(|x|, 2.0), which is NOT
 * always equal to x*x on glibc.  The default build uses
 * pow() so that it matches the reference's bits; -DOSIM_SQ_MUL builds the variant
 * that squares by multiplication, which is what the HIP kernel computes (the GPU is
 * compared bit-for-bit with that variant and within 1e-5 with the pow variant).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/nclone/).  The structure is deliberately close to the reference's
 * object graph (per-cell segment lists, per-cell entity lists) and deliberately
 * unlike the product's (packed CSR + SoA lanes), so the two are independent checks.
 */
#define _GNU_SOURCE
#include <math.h>
#include <alloca.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef OSIM_SQ_MUL
static inline double SQ(double x) { return x * x; }
static inline double POWHALF(double x) { return sqrt(x); }
#else
/* CPython float_pow: negative base with integral exponent -> pow(-x, w). */
static inline double SQ(double x) { return pow(fabs(x), 2.0); }
static inline double POWHALF(double x) { return pow(x, 0.5); }
#endif

/* constants/physics_constants.py:11-60,282-344 */
#define NINJA_RADIUS 10.0
#define GRAVITY_FALL 0.06666666666666665
#define GRAVITY_JUMP 0.01111111111111111
#define GROUND_ACCEL 0.06666666666666665
#define AIR_ACCEL 0.04444444444444444
#define DRAG_REGULAR 0.9933221725495059
#define DRAG_SLOW 0.8617738760127536
#define FRICTION_GROUND 0.9459290248857720
#define FRICTION_GROUND_SLOW 0.8617738760127536
#define FRICTION_WALL 0.9113380468927672
#define MAX_HOR_SPEED 3.333
#define MAX_JUMP_DURATION 45
#define MAX_SURVIVABLE_IMPACT 6.0
#define MIN_SURVIVABLE_CRUSHING 0.05

#define GW 44
#define GH 25
#define MAXSEG_CELL 16
#define MAXQ 160

/* ---- tile tables (tile_definitions.py:137-223), re-encoded -------------------------
 * ORTHO[t]: 12 chars, first 6 = horizontal half-edges (left->right, top->bottom),
 * last 6 = vertical half-edges (top->bottom, left->right); '-' normal up/left,
 * '+' normal down/right, '.' none. */
static const char *ORTHO[34] = {
    "............", "--..++--..++", "--++..-...+.", ".-...+..--++", "..--++.-...+", "-...+.--++..",
    "--....--....", "--........++", "....++....++", "....++--....",
    "--....--....", "--........++", "....++....++", "....++--....",
    "--....--....", "--........++", "....++....++", "....++--....",
    "--....-.....", "--........+.", "....++.....+", "....++.-....",
    "--....--..+.", "--....-...++", "....++.-..++", "....++--...+",
    "-.....--....", ".-........++", ".....+....++", "....+.--....",
    "--..+.--....", "--...+....++", ".-..++....++", "-...++--....",
};
/* GEDGE[t]: tile_definitions.py TILE_GRID_EDGE_MAP, same 6 + 6 half-edge layout as ORTHO, '1' = edge present */
static const char *GEDGE[34] = {
    "000000000000", "110011110011", "111100100010", "010001001111", "001111010001", "100010111100",
    "110110110110", "111001100111", "011011011011", "100111111001",
    "110011110011", "110011110011", "110011110011", "110011110011",
    "110110110110", "111001100111", "011011011011", "100111111001",
    "111100100010", "111100100010", "001111010001", "001111010001",
    "110011110011", "110011110011", "110011110011", "110011110011",
    "100010111100", "010001001111", "010001001111", "100010111100",
    "110011110011", "110011110011", "110011110011", "110011110011",
};
/* DIAG[t] = x1,y1,x2,y2 for t in 6..9 and 18..33 (tile_definitions.py:189-210) */
static const signed char DIAG[34][4] = {
    {0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},
    {0,24,24,0},{0,0,24,24},{24,0,0,24},{24,24,0,0},
    {0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0},
    {0,12,24,0},{0,0,24,12},{24,12,0,24},{24,24,0,12},
    {0,24,24,12},{0,12,24,24},{24,0,0,12},{24,12,0,0},
    {0,24,12,0},{12,0,24,24},{24,0,12,24},{12,24,0,0},
    {12,24,24,0},{0,0,12,24},{12,0,0,24},{24,24,12,0},
};
/* ARC[t-10] = cx,cy,hor,ver,convex for t in 10..17 (tile_definitions.py:214-223) */
static const signed char ARC[8][5] = {
    {0,0,1,1,1},{24,0,-1,1,1},{24,24,-1,-1,1},{0,24,1,-1,1},
    {24,24,-1,-1,0},{0,24,1,-1,0},{0,0,1,1,0},{24,0,-1,1,0},
};

typedef struct {
    int kind;                 /* 0 linear (entities.py:9), 1 circular (entities.py:99) */
    double x1, y1, x2, y2;    /* linear end points */
    int oriented;
    double px, py, seg_lensq;
    double cx, cy;            /* circular */
    double hor, ver, radius;
    double phx, phy, pvx, pvy;
    int convex;
    double b[4];              /* cached bounds */
} Seg;

typedef struct {
    int present;
    int n;
    Seg *s[MAXSEG_CELL];
    double b[4];
} IndexCell;                  /* utils/spatial_segment_index.py:69-110 */

enum { K_MINE = 1, K_GOLD = 2, K_EXIT = 3, K_SWITCH = 4, K_DOOR_REG = 5, K_LOCKED = 6, K_DOOR_TRAP = 8,
       K_LAUNCH = 10, K_ONEWAY = 11, K_DRONE = 14, K_BOUNCE = 17, K_THWUMP = 20, K_BOOST = 24, K_BALL = 25,
       K_SHOVE = 28 };   /* K_DRONE covers zap drones (14) and mini drones (26) */

typedef struct Entity {
    int dic_key;              /* entity_dic key the object lives under */
    int type;                 /* Entity.type */
    int kind;
    double x, y;
    int cx, cy;
    int active;
    int state, init_state;    /* mines */
    double radius;
    int switch_hit;           /* exit door */
    struct Entity *parent;    /* exit switch -> door */
    int closed;               /* doors */
    int thinkable, logical, movable, physical;
    /* ---- entity zoo ---- */
    double x0, y0;            /* position at creation */
    int cx0, cy0;
    int orientation, mode;
    double nx, ny;            /* launch pad / one-way normal (physics.py:317-332) */
    double xspeed, yspeed;    /* bounce block, death ball */
    double xorigin, yorigin;  /* bounce block, thwump, shove thwump */
    double speed, xtarget, ytarget, drone_radius, grid_width;   /* drones */
    int dir;
    int is_horizontal, direction;   /* thwump */
    double xdir, ydir;        /* shove thwump */
    int activated;
    int touching;             /* boost pad */
    double door_x, door_y;    /* the door's own position (the entity itself sits at its switch) */
    int is_vertical, edge[2][2], open_timer;   /* doors: grid edges they own (entity_door_base.py:69-89) */
    int index;                /* Entity.index (entities.py: per-type creation counter) */
} Entity;

typedef struct {
    int n, cap;
    Entity **e;
} EList;

typedef struct {
    double xpos, ypos, xspeed, yspeed;
    double applied_gravity, applied_drag, applied_friction;
    int state, airborn, airborn_old, walled;
    double wall_normal;
    int jump_input_old, jump_duration;
    int jump_buffer, floor_buffer, wall_buffer, launch_pad_buffer;
    double floor_normalized_x, floor_normalized_y, ceiling_normalized_x, ceiling_normalized_y;
    int hor_input, jump_input;
    double xpos_old, ypos_old, xspeed_old, yspeed_old;
    int floor_count, wall_count, ceiling_count;
    double floor_normal_x, floor_normal_y, ceiling_normal_x, ceiling_normal_y;
    int is_crushable;
    double x_crush, y_crush, crush_len;
    int gold_collected, doors_opened;
    int death_cause;          /* 0 none, 1 mine, 2 terminal_impact */
    int terminal_impact;
    int frames_airborne, state_change_frame, previous_state;
    int consecutive_floor_frames, consecutive_wall_frames;
    double xlp_boost_normalized, ylp_boost_normalized;
} Ninja;

typedef struct OSim {
    double *map;
    int nmap;
    int frame;
    int tiles[GW][GH];
    Seg *segs;
    int nsegs;
    IndexCell index[GW][GH];
    Entity *ents;
    int nents;
    /* entity_dic order: keys ascending, list order inside a key */
    Entity **dic_order;
    int ndic;
    EList grid[GW][GH];
    Ninja nj;
    int unsupported;          /* bitmask of entity types present but not restated */
    /* grid edges used by drones / thwumps (nsim.py:208-215, tile_segment_factory.py:283-302, doors) */
    int hor_base[89][51], ver_base[89][51];
    int hor_edge[89][51], ver_edge[89][51];
    int has_zoo;              /* any movable / physically collidable / extra thinkable entity */
    int creations;            /* number of times the entities were (re)created since load: sim.entity_counts persists
                                 across Simulator.reset(), so Entity.index keeps growing (entities.py:129-133) */
    /* curriculum repositioning of the exit switch [0] / exit door [1] (intermediate_goal_manager.py:698), re-applied after
     * every reset like the reference does after every map load */
    int ovr_set[2];
    double ovr_x[2], ovr_y[2];
    Entity *cached[256];      /* ninja._cached_entities (ninja.py:220-222) */
    int ncached;
    /* mine-overlay cache of gym_environment/spatial_context.py:103-110,309-367 (per env = per process there) */
    int sc_valid;
    double sc_lx, sc_ly;
    float sc_overlay[48];
} OSim;

static int iclamp(int n, int a, int b) { return n < a ? a : (n > b ? b : n); }
static double pymax(double a, double b) { return b > a ? b : a; } /* max(a, b): first maximal */
static double pymin(double a, double b) { return b < a ? b : a; } /* min(a, b): first minimal */

/* ------------------------------------------------------------------------------------
 * Level geometry: map_loader.py:18-41, utils/tile_segment_factory.py:170-262,
 * utils/tile_segment_cache.py:36-60, entities.py:12-41,102-125
 * ---------------------------------------------------------------------------------- */
static void seg_linear(Seg *s, double x1, double y1, double x2, double y2)
{
    memset(s, 0, sizeof(*s));
    s->kind = 0;
    s->x1 = x1; s->y1 = y1; s->x2 = x2; s->y2 = y2;
    s->oriented = 1;
    s->px = x2 - x1;
    s->py = y2 - y1;
    s->seg_lensq = SQ(s->px) + SQ(s->py);
    if (s->seg_lensq == 0) s->seg_lensq = 1e-9;
    s->b[0] = x1 < x2 ? x1 : x2; s->b[1] = y1 < y2 ? y1 : y2;
    s->b[2] = x1 > x2 ? x1 : x2; s->b[3] = y1 > y2 ? y1 : y2;
}

static void seg_circular(Seg *s, double cx, double cy, int hor, int ver, int convex)
{
    memset(s, 0, sizeof(*s));
    s->kind = 1;
    s->cx = cx; s->cy = cy; s->hor = hor; s->ver = ver; s->radius = 24; s->convex = convex;
    s->phx = cx + 24 * hor; s->phy = cy;
    s->pvx = cx; s->pvy = cy + 24 * ver;
    s->b[0] = cx < s->phx ? cx : s->phx; s->b[2] = cx > s->phx ? cx : s->phx;
    s->b[1] = cy < s->pvy ? cy : s->pvy; s->b[3] = cy > s->pvy ? cy : s->pvy;
}

static void build_geometry(OSim *S)
{
    static int hor[89][51], ver[89][51];
    /* per-cell temporary lists in append order */
    int (*cnt)[GH] = calloc(GW, sizeof(*cnt));
    int (*lst)[GH][MAXSEG_CELL] = calloc(GW, sizeof(*lst));
    memset(hor, 0, sizeof(hor));
    memset(ver, 0, sizeof(ver));
    S->segs = malloc(sizeof(Seg) * GW * GH * 13);
    S->nsegs = 0;
    /* map_loader.py:22-37: inner tiles from map_data[184:1150], border forced to 1 */
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++)
            S->tiles[x][y] = 1;
    for (int x = 0; x < 42; x++)
        for (int y = 0; y < 23; y++) {
            double v = S->map[184 + x + y * 42];
            int t = (v == floor(v) && v >= 0 && v < 256) ? (int)v : 255;
            S->tiles[x + 1][y + 1] = t;
        }
    /* grid edges: nsim.py:208-215 (outer frame preset to 1) then tile_segment_factory.py:283-302 (mod-2 toggles) */
    for (int xc = 0; xc < 89; xc++)
        for (int yc = 0; yc < 51; yc++) {
            S->hor_base[xc][yc] = (yc == 0 || yc == 50) ? 1 : 0;
            S->ver_base[xc][yc] = (xc == 0 || xc == 88) ? 1 : 0;
        }
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) {
            int t = S->tiles[x][y];
            if (t == 0 || t >= 34) continue;
            const char *g = GEDGE[t];
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 2; i++)
                    S->hor_base[2 * x + i][2 * y + j] = (S->hor_base[2 * x + i][2 * y + j] + (g[2 * j + i] - '0')) % 2;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 2; j++)
                    S->ver_base[2 * x + i][2 * y + j] = (S->ver_base[2 * x + i][2 * y + j] + (g[2 * i + j + 6] - '0')) % 2;
        }
    /* tile_segment_factory.py:170-215; the iteration order over tiles only matters for the
     * (at most one) diagonal/arc of a cell, which is always first in its cell's list. */
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) {
            int t = S->tiles[x][y];
            if (t == 0 || t >= 34) continue;
            const char *o = ORTHO[t];
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 2; i++) {
                    char c = o[2 * j + i];
                    hor[2 * x + i][2 * y + j] += (c == '-') ? -1 : (c == '+') ? 1 : 0;
                }
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 2; j++) {
                    char c = o[2 * i + j + 6];
                    ver[2 * x + i][2 * y + j] += (c == '-') ? -1 : (c == '+') ? 1 : 0;
                }
            if ((t >= 6 && t <= 9) || t >= 18) {
                Seg *s = &S->segs[S->nsegs];
                seg_linear(s, 24 * x + DIAG[t][0], 24 * y + DIAG[t][1], 24 * x + DIAG[t][2], 24 * y + DIAG[t][3]);
                lst[x][y][cnt[x][y]++] = S->nsegs++;
            } else if (t >= 10 && t <= 17) {
                Seg *s = &S->segs[S->nsegs];
                const signed char *a = ARC[t - 10];
                seg_circular(s, 24 * x + a[0], 24 * y + a[1], a[2], a[3], a[4]);
                lst[x][y][cnt[x][y]++] = S->nsegs++;
            }
        }
    /* tile_segment_factory.py:231-262: dict pre-seeded x-major (nsim.py:218-219) */
    for (int xc = 0; xc < 89; xc++)
        for (int yc = 0; yc < 51; yc++) {
            int st = hor[xc][yc];
            if (st == 0) continue;
            int cx = (int)floor(xc / 2.0);
            int cy = (int)floor((yc - 0.1 * st) / 2);
            double p1x = 12 * xc, p1y = 12 * yc, p2x = 12 * xc + 12, p2y = 12 * yc;
            if (st == -1) { double tx = p1x, ty = p1y; p1x = p2x; p1y = p2y; p2x = tx; p2y = ty; }
            if (cx < 0 || cx >= GW || cy < 0 || cy >= GH) continue; /* KeyError in the reference */
            seg_linear(&S->segs[S->nsegs], p1x, p1y, p2x, p2y);
            lst[cx][cy][cnt[cx][cy]++] = S->nsegs++;
        }
    for (int xc = 0; xc < 89; xc++)
        for (int yc = 0; yc < 51; yc++) {
            int st = ver[xc][yc];
            if (st == 0) continue;
            int cx = (int)floor((xc - 0.1 * st) / 2);
            int cy = (int)floor(yc / 2.0);
            double p1x = 12 * xc, p1y = 12 * yc + 12, p2x = 12 * xc, p2y = 12 * yc;
            if (st == -1) { double tx = p1x, ty = p1y; p1x = p2x; p1y = p2y; p2x = tx; p2y = ty; }
            if (cx < 0 || cx >= GW || cy < 0 || cy >= GH) continue;
            seg_linear(&S->segs[S->nsegs], p1x, p1y, p2x, p2y);
            lst[cx][cy][cnt[cx][cy]++] = S->nsegs++;
        }
    /* spatial_segment_index.py:69-110: snapshot per cell + cell bounds */
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) {
            IndexCell *c = &S->index[x][y];
            c->present = cnt[x][y] > 0;
            c->n = cnt[x][y];
            c->b[0] = c->b[1] = INFINITY;
            c->b[2] = c->b[3] = -INFINITY;
            for (int k = 0; k < c->n; k++) {
                Seg *s = &S->segs[lst[x][y][k]];
                c->s[k] = s;
                c->b[0] = pymin(c->b[0], s->b[0]); c->b[1] = pymin(c->b[1], s->b[1]);
                c->b[2] = pymax(c->b[2], s->b[2]); c->b[3] = pymax(c->b[3], s->b[3]);
            }
        }
    free(cnt);
    free(lst);
}

/* ------------------------------------------------------------------------------------
 * Entities: map_loader.py:69-145, utils/entity_factory.py:145-233, entities.py:209-236
 * ---------------------------------------------------------------------------------- */
static void elist_push(EList *l, Entity *e)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 4;
        l->e = realloc(l->e, sizeof(Entity *) * l->cap);
    }
    l->e[l->n++] = e;
}

static void ent_base(Entity *e, int type, double xc, double yc)
{
    memset(e, 0, sizeof(*e));
    e->type = type;
    e->x = xc * 6;
    e->y = yc * 6;
    e->active = 1;
    e->cx = iclamp((int)floor(e->x / 24), 0, 43);
    e->cy = iclamp((int)floor(e->y / 24), 0, 24);
}

static void mine_set_state(Entity *e, int st)
{
    /* entity_toggle_mine.py:130-135, radii constants/physics_constants.py:48 */
    e->state = st;
    e->radius = st == 0 ? 4.0 : (st == 1 ? 3.5 : 4.5);
}

/* physics.py:317-332 */
static void orientation_vector(int o, double *vx, double *vy)
{
    double diag = sqrt(2.0) / 2;
    static const int sx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, sy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    o &= 7;
    if (o & 1) { *vx = sx[o] * diag; *vy = sy[o] * diag; }
    else { *vx = sx[o]; *vy = sy[o]; }
}

/* entity_door_base.py:52-97: the door's own position decides the grid edges it blocks, then the entity moves to
 * its switch.  The door segment itself never reaches the ninja's queries. */
static void door_init(Entity *e, int orientation, double sw_xc, double sw_yc)
{
    double vx, vy;
    orientation_vector(orientation, &vx, &vy);
    e->orientation = orientation;
    e->door_x = e->x;
    e->door_y = e->y;
    e->is_vertical = (orientation == 0 || orientation == 4);
    int dcx = iclamp((int)floor((e->x - 12 * vx) / 24), 0, 43);
    int dcy = iclamp((int)floor((e->y - 12 * vy) / 24), 0, 24);
    int hx = 2 * (dcx + 1), hy = 2 * (dcy + 1);
    if (e->is_vertical) {
        e->edge[0][0] = hx; e->edge[0][1] = hy - 2;
        e->edge[1][0] = hx; e->edge[1][1] = hy - 1;
    } else {
        e->edge[0][0] = hx - 2; e->edge[0][1] = hy;
        e->edge[1][0] = hx - 1; e->edge[1][1] = hy;
    }
    e->x = 6 * sw_xc;
    e->y = 6 * sw_yc;
    e->cx = iclamp((int)floor(e->x / 24), 0, 43);
    e->cy = iclamp((int)floor(e->y / 24), 0, 24);
    e->logical = 1;
}

static void load_entities(OSim *S)
{
    int n = S->nmap;
    const double *m = S->map;
    int cap = (n - 1230) / 5 * 2 + 8;
    S->ents = calloc(cap, sizeof(Entity));
    S->nents = 0;
    S->unsupported = 0;
    S->has_zoo = 0;
    int counts[40];
    memset(counts, 0, sizeof(counts));
    int index = 1230;
    double exit_door_count = n > 1156 ? m[1156] : 0;
    while (index < n) {
        if (index + 4 >= n) break;
        double tv = m[index];
        int type = (tv == floor(tv) && tv >= 0 && tv < 64) ? (int)tv : -1;
        double xc = m[index + 1], yc = m[index + 2];
        int orientation = (int)m[index + 3], mode = (int)m[index + 4];
        Entity *e = NULL;
        if (type == 1 || type == 21) {
            e = &S->ents[S->nents++];
            ent_base(e, type, xc, yc);
            e->kind = K_MINE; e->dic_key = type; e->thinkable = 1; e->logical = 1;
            e->init_state = (type == 1) ? 0 : 1;   /* entity_factory.py:181-182,230-231 */
            mine_set_state(e, e->init_state);
        } else if (type == 2) {
            e = &S->ents[S->nents++];
            ent_base(e, type, xc, yc);
            e->kind = K_GOLD; e->dic_key = 2; e->logical = 1; e->radius = 6;
        } else if (type == 3) {
            /* entity_factory.py:185-197: door goes to entity_dic[3] only; the returned switch
             * goes to entity_dic[3] and to the grid. */
            Entity *door = &S->ents[S->nents++];
            ent_base(door, 3, xc, yc);
            door->kind = K_EXIT; door->dic_key = 3; door->logical = 1; door->radius = 12;
            door->x0 = door->x; door->y0 = door->y; door->cx0 = door->cx; door->cy0 = door->cy;
            int ci = index + 5 * (int)exit_door_count;
            double sx = (ci + 2 < n) ? m[ci + 1] : 0, sy = (ci + 2 < n) ? m[ci + 2] : 0;
            e = &S->ents[S->nents++];
            ent_base(e, 4, sx, sy);
            e->kind = K_SWITCH; e->dic_key = 3; e->logical = 1; e->radius = 6; e->parent = door;
        } else if (type == 5) {
            /* entity_factory.py:198-201, entity_door_regular.py:41-44: the switch is the door itself */
            e = &S->ents[S->nents++];
            ent_base(e, 5, xc, yc);
            e->kind = K_DOOR_REG; e->dic_key = 5; e->radius = 10; e->thinkable = 1;
            door_init(e, orientation, xc, yc);
        } else if (type == 6 || type == 8) {
            /* entity_factory.py:202-216, entity_door_locked.py:46-52, entity_door_trap.py:51-58 */
            double sx = (index + 7 < n) ? m[index + 6] : 0, sy = (index + 7 < n) ? m[index + 7] : 0;
            e = &S->ents[S->nents++];
            ent_base(e, type, xc, yc);
            e->kind = type == 6 ? K_LOCKED : K_DOOR_TRAP; e->dic_key = type; e->radius = 5;
            door_init(e, orientation, sx, sy);
        } else if (type == 10) {
            e = &S->ents[S->nents++];
            ent_base(e, 10, xc, yc);
            e->kind = K_LAUNCH; e->dic_key = 10; e->logical = 1; e->radius = 6; e->orientation = orientation;
            orientation_vector(orientation, &e->nx, &e->ny);
        } else if (type == 11) {
            e = &S->ents[S->nents++];
            ent_base(e, 11, xc, yc);
            e->kind = K_ONEWAY; e->dic_key = 11; e->logical = 1; e->physical = 1; e->orientation = orientation;
            orientation_vector(orientation, &e->nx, &e->ny);
        } else if (type == 14 || type == 26) {
            /* entity_drone_zap.py:53-55 (speed 8/7, radius 7.5, grid 24), entity_mini_drone.py:56-58 (1.3, 4, 12) */
            e = &S->ents[S->nents++];
            ent_base(e, type, xc, yc);
            e->kind = K_DRONE; e->dic_key = type; e->logical = 1; e->movable = 1;
            e->speed = type == 14 ? 8.0 / 7 : 1.3;
            e->drone_radius = type == 14 ? 7.5 : 4.0;
            e->grid_width = type == 14 ? 24 : 12;
            e->orientation = orientation; e->mode = mode & 3;
        } else if (type == 17) {
            e = &S->ents[S->nents++];
            ent_base(e, 17, xc, yc);
            e->kind = K_BOUNCE; e->dic_key = 17; e->logical = 1; e->physical = 1; e->movable = 1;
        } else if (type == 20) {
            e = &S->ents[S->nents++];
            ent_base(e, 20, xc, yc);
            e->kind = K_THWUMP; e->dic_key = 20; e->logical = 1; e->physical = 1; e->movable = 1; e->thinkable = 1;
            e->orientation = orientation;
            e->is_horizontal = (orientation == 0 || orientation == 4);
            e->direction = (orientation == 0 || orientation == 2) ? 1 : -1;
        } else if (type == 24) {
            e = &S->ents[S->nents++];
            ent_base(e, 24, xc, yc);
            e->kind = K_BOOST; e->dic_key = 24; e->movable = 1; e->radius = 6;
        } else if (type == 25) {
            e = &S->ents[S->nents++];
            ent_base(e, 25, xc, yc);
            e->kind = K_BALL; e->dic_key = 25; e->thinkable = 1; e->logical = 1;
        } else if (type == 28) {
            e = &S->ents[S->nents++];
            ent_base(e, 28, xc, yc);
            e->kind = K_SHOVE; e->dic_key = 28; e->thinkable = 1; e->logical = 1; e->physical = 1;
        }
        if (e) {
            e->x0 = e->x; e->y0 = e->y; e->cx0 = e->cx; e->cy0 = e->cy;
            e->index = counts[e->type < 40 ? e->type : 39]++;
            if (e->kind == K_DOOR_REG || e->kind >= K_DOOR_TRAP) S->has_zoo = 1;
        }
        if (type == 6 || type == 8) {
            if (index + 9 < n) {
                if (m[index + 7] != 0 && m[index + 8] == 0 && m[index + 9] == 0) index += 10;
                else index += 9;
            } else index += 9;
        } else index += 5;
    }
    /* entity_dic iteration order (nsim.py:190-192,237-243): keys 1..28, list order */
    S->dic_order = malloc(sizeof(Entity *) * (S->nents + 1));
    S->ndic = 0;
    for (int key = 1; key <= 28; key++)
        for (int i = 0; i < S->nents; i++)
            if (S->ents[i].dic_key == key) S->dic_order[S->ndic++] = &S->ents[i];
}

static void door_edges_add(OSim *S, const Entity *e, int d)
{   /* entity_door_base.py:78-89,104-108: plain integer counters (they can go negative, and negative is truthy) */
    for (int k = 0; k < 2; k++) {
        if (e->is_vertical) S->ver_edge[e->edge[k][0]][e->edge[k][1]] += d;
        else S->hor_edge[e->edge[k][0]][e->edge[k][1]] += d;
    }
}

static void grid_rebuild(OSim *S)
{
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) S->grid[x][y].n = 0;
    /* map_loader.py:113-120: appended in map order; the exit door itself is not in the grid */
    for (int i = 0; i < S->nents; i++) {
        Entity *e = &S->ents[i];
        if (e->kind == K_EXIT) continue;
        elist_push(&S->grid[e->cx][e->cy], e);
    }
}

/* entities.py Entity.grid_move: leave the old cell's list, join the END of the new cell's list */
static void grid_move(OSim *S, Entity *e)
{
    int ncx = iclamp((int)floor(e->x / 24), 0, 43), ncy = iclamp((int)floor(e->y / 24), 0, 24);
    if (ncx == e->cx && ncy == e->cy) return;
    EList *l = &S->grid[e->cx][e->cy];
    for (int k = 0; k < l->n; k++)
        if (l->e[k] == e) {
            memmove(&l->e[k], &l->e[k + 1], sizeof(Entity *) * (l->n - k - 1));
            l->n--;
            break;
        }
    e->cx = ncx; e->cy = ncy;
    elist_push(&S->grid[ncx][ncy], e);
}

static void ninja_init(OSim *S)
{
    /* ninja.py:80-196 */
    Ninja *n = &S->nj;
    memset(n, 0, sizeof(*n));
    n->xpos = S->map[1231] * 6;
    n->ypos = S->map[1232] * 6;
    n->applied_gravity = GRAVITY_FALL;
    n->applied_drag = DRAG_REGULAR;
    n->applied_friction = FRICTION_GROUND;
    n->jump_buffer = n->floor_buffer = n->wall_buffer = n->launch_pad_buffer = -1;
    n->floor_normalized_y = -1;
    n->ceiling_normalized_y = 1;
}

/* intermediate_goal_manager.py:698-1000 apply_to_simulator: the LAST EntityExitSwitch of entity_dic[3] and its parent door get
 * new positions; a switch that changes cell leaves its list and is appended to the new cell's list; the door only gets its
 * new cell (it joins that cell's list when the switch is hit, entity_exit_switch.py:120) */
static void apply_overrides(OSim *S)
{
    Entity *sw = NULL;
    for (int i = S->ndic - 1; i >= 0 && !sw; i--)
        if (S->dic_order[i]->kind == K_SWITCH) sw = S->dic_order[i];
    if (!sw) return;
    if (S->ovr_set[0]) {
        sw->x = S->ovr_x[0]; sw->y = S->ovr_y[0];
        int ncx = iclamp((int)floor(sw->x / 24), 0, 43), ncy = iclamp((int)floor(sw->y / 24), 0, 24);
        if (ncx != sw->cx || ncy != sw->cy) {
            EList *l = &S->grid[sw->cx][sw->cy];
            for (int k = 0; k < l->n; k++)
                if (l->e[k] == sw) { memmove(&l->e[k], &l->e[k + 1], sizeof(Entity *) * (l->n - k - 1)); l->n--; break; }
            sw->cx = ncx; sw->cy = ncy;
            elist_push(&S->grid[ncx][ncy], sw);
        }
    }
    if (S->ovr_set[1] && sw->parent) {
        Entity *d = sw->parent;
        d->x = S->ovr_x[1]; d->y = S->ovr_y[1];
        d->cx = iclamp((int)floor(d->x / 24), 0, 43); d->cy = iclamp((int)floor(d->y / 24), 0, 24);
    }
}

void osim_set_entity_pos(OSim *S, int kind, double x, double y)
{
    if (kind < 0 || kind > 1) return;
    S->ovr_set[kind] = 1; S->ovr_x[kind] = x; S->ovr_y[kind] = y;
    apply_overrides(S);
}

void osim_reset(OSim *S)
{
    /* nsim.py:62-76: Simulator.reset() re-creates every entity from map_data (the semantics restated here;
     * fast_reset, nsim.py:78-140, is equivalent for the entity kinds that implement reset_state()) */
    S->frame = 0;
    S->sc_valid = 0;   /* reset_mine_overlay_cache(): npp_environment.py:569-571 */
    S->creations += 1;
    S->ncached = 0;
    ninja_init(S);
    memcpy(S->hor_edge, S->hor_base, sizeof(S->hor_edge));
    memcpy(S->ver_edge, S->ver_base, sizeof(S->ver_edge));
    for (int i = 0; i < S->nents; i++) {
        Entity *e = &S->ents[i];
        e->active = 1;
        e->x = e->x0; e->y = e->y0; e->cx = e->cx0; e->cy = e->cy0;
        switch (e->kind) {
        case K_MINE: mine_set_state(e, e->init_state); break;
        case K_EXIT: e->switch_hit = 0; break;
        case K_LOCKED: e->closed = 1; door_edges_add(S, e, 1); break;
        case K_DOOR_REG: e->closed = 1; e->open_timer = 0; door_edges_add(S, e, 1); break;
        case K_DOOR_TRAP: e->closed = 0; door_edges_add(S, e, 1); door_edges_add(S, e, -1); break;
        case K_DRONE:     /* entity_drone_base.py:74-84 */
            e->dir = e->orientation / 2;
            e->xtarget = e->x; e->ytarget = e->y;
            break;
        case K_BOUNCE: e->xspeed = e->yspeed = 0; e->xorigin = e->x; e->yorigin = e->y; break;
        case K_THWUMP: e->xorigin = e->x; e->yorigin = e->y; e->state = 0; break;
        case K_BOOST: e->touching = 0; break;
        case K_BALL: e->xspeed = e->yspeed = 0; break;
        case K_SHOVE: e->xorigin = e->x; e->yorigin = e->y; e->xdir = e->ydir = 0; e->state = 0; e->activated = 0; break;
        default: break;
        }
    }
    grid_rebuild(S);
    apply_overrides(S);
}

/* Simulator.fast_reset (nsim.py:78-140): the reset NppEnvironment.reset uses for same-level resets
 * (npp_environment.py:541-557).  Nothing is re-created: the ninja is reset in place (ninja.py:1288-1394), the per-cell lists
 * are cleared and refilled while walking entity_dic (keys ascending, creation order inside a key: nsim.py:124-140), every
 * entity whose class defines reset_state() goes back to its initial position / state (toggle mine, gold, exit door, exit
 * switch, locked door, trap door), every other entity only gets active = True and keeps position, cell, speed, state,
 * door counters and timers (drones, thwumps, bounce blocks, death balls, shove thwumps, boost pads, regular doors, launch
 * pads, one-ways).  The grid edges are NOT restored from the tiles: doors fix theirs through change_state. */
void osim_fast_reset(OSim *S)
{
    S->frame = 0;
    S->sc_valid = 0;   /* reset_mine_overlay_cache(): npp_environment.py:569-571 */
    S->ncached = 0;
    {   /* Ninja.reset_state keeps xlp / ylp_boost_normalized (not listed in ninja.py:1288-1394) */
        double lx = S->nj.xlp_boost_normalized, ly = S->nj.ylp_boost_normalized;
        ninja_init(S);
        S->nj.xlp_boost_normalized = lx; S->nj.ylp_boost_normalized = ly;
    }
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) S->grid[x][y].n = 0;
    for (int i = 0; i < S->ndic; i++) {
        Entity *e = S->dic_order[i];
        switch (e->kind) {
        case K_MINE: case K_GOLD: case K_EXIT: case K_SWITCH: case K_LOCKED: case K_DOOR_TRAP:
            e->x = e->x0; e->y = e->y0; e->cx = e->cx0; e->cy = e->cy0;
            e->active = 1;
            if (e->kind == K_MINE) mine_set_state(e, e->init_state);
            if (e->kind == K_EXIT) e->switch_hit = 0;
            if (e->kind == K_LOCKED && !e->closed) { e->closed = 1; door_edges_add(S, e, 1); }     /* entity_door_locked.py:69-77 */
            if (e->kind == K_DOOR_TRAP && e->closed) { e->closed = 0; door_edges_add(S, e, -1); }  /* entity_door_trap.py:69-77 */
            break;
        default:
            e->active = 1;
            break;
        }
        if (e->kind != K_EXIT) elist_push(&S->grid[e->cx][e->cy], e);
    }
    apply_overrides(S);
}

OSim *osim_create(void) { return calloc(1, sizeof(OSim)); }

static void free_level(OSim *S)
{
    free(S->map); S->map = NULL;
    free(S->segs); S->segs = NULL;
    free(S->ents); S->ents = NULL;
    free(S->dic_order); S->dic_order = NULL;
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) {
            free(S->grid[x][y].e);
            S->grid[x][y].e = NULL;
            S->grid[x][y].n = S->grid[x][y].cap = 0;
        }
}

void osim_destroy(OSim *S)
{
    if (!S) return;
    free_level(S);
    free(S);
}

/* nsim.py:51-56 */
int osim_load(OSim *S, const double *map, int n)
{
    if (n < 1233) return -1;
    free_level(S);
    S->map = malloc(sizeof(double) * n);
    memcpy(S->map, map, sizeof(double) * n);
    S->nmap = n;
    build_geometry(S);
    load_entities(S);
    S->creations = 0;   /* a load restated here is a fresh Simulator */
    S->ovr_set[0] = S->ovr_set[1] = 0;
    osim_reset(S);
    return S->unsupported;
}

/* ------------------------------------------------------------------------------------
 * physics.py
 * ---------------------------------------------------------------------------------- */
/* physics.py:39-52 -> utils/spatial_segment_index.py:112-158 */
static int query_region(OSim *S, double x1, double y1, double x2, double y2, Seg **out)
{
    double min_x = pymin(x1, x2), min_y = pymin(y1, y2), max_x = pymax(x1, x2), max_y = pymax(y1, y2);
    int c0x = iclamp((int)floor(min_x / 24), 0, 43), c1x = iclamp((int)floor(max_x / 24), 0, 43);
    int c0y = iclamp((int)floor(min_y / 24), 0, 24), c1y = iclamp((int)floor(max_y / 24), 0, 24);
    int n = 0;
    for (int xc = c0x; xc <= c1x; xc++)
        for (int yc = c0y; yc <= c1y; yc++) {
            IndexCell *c = &S->index[xc][yc];
            if (!c->present) continue;
            if (max_x < c->b[0] || min_x > c->b[2] || max_y < c->b[1] || min_y > c->b[3]) continue;
            for (int k = 0; k < c->n && n < MAXQ; k++) out[n++] = c->s[k];
        }
    return n;
}

/* physics.py:247-261 */
static double toi_circle_circle(double xpos, double ypos, double vx, double vy, double a, double b, double radius)
{
    double dx = xpos - a, dy = ypos - b;
    double dist_sq = SQ(dx) + SQ(dy);
    double vel_sq = SQ(vx) + SQ(vy);
    double dot_prod = dx * vx + dy * vy;
    if (dist_sq - SQ(radius) > 0) {
        double radicand = SQ(dot_prod) - vel_sq * (dist_sq - SQ(radius));
        if (vel_sq > 0.0001 && dot_prod < 0 && radicand >= 0)
            return (-dot_prod - sqrt(radicand)) / vel_sq;
        return 1;
    }
    return 0;
}

/* physics.py:264-285 */
static double toi_circle_lineseg(double xpos, double ypos, double dx, double dy,
                                 double a1, double b1, double a2, double b2, double radius)
{
    double wx = a2 - a1, wy = b2 - b1;
    double seg_len = sqrt(SQ(wx) + SQ(wy));
    double nx = wx / seg_len, ny = wy / seg_len;
    double normal_proj = (xpos - a1) * ny - (ypos - b1) * nx;
    double hor_proj = (xpos - a1) * nx + (ypos - b1) * ny;
    if (fabs(normal_proj) >= radius) {
        double dir = dx * ny - dy * nx;
        if (dir * normal_proj < 0) {
            double t = pymin((fabs(normal_proj) - radius) / fabs(dir), 1);
            double hor_proj2 = hor_proj + t * (dx * nx + dy * ny);
            if (0 <= hor_proj2 && hor_proj2 <= seg_len) return t;
        }
    } else {
        if (0 <= hor_proj && hor_proj <= seg_len) return 0;
    }
    return 1;
}

/* physics.py:288-314 */
static double toi_circle_arc(double xpos, double ypos, double vx, double vy, double a, double b,
                             double hor, double ver, double radius_arc, double radius_circle)
{
    double dx = xpos - a, dy = ypos - b;
    double dist_sq = SQ(dx) + SQ(dy);
    double vel_sq = SQ(vx) + SQ(vy);
    double dot_prod = dx * vx + dy * vy;
    double radius1 = radius_arc + radius_circle;
    double radius2 = radius_arc - radius_circle;
    double t = 1;
    if (dist_sq > SQ(radius1)) {
        double radicand = SQ(dot_prod) - vel_sq * (dist_sq - SQ(radius1));
        if (vel_sq > 0.0001 && dot_prod < 0 && radicand >= 0)
            t = (-dot_prod - sqrt(radicand)) / vel_sq;
    } else if (dist_sq < SQ(radius2)) {
        double radicand = SQ(dot_prod) - vel_sq * (dist_sq - SQ(radius2));
        if (vel_sq > 0.0001)
            t = pymin((-dot_prod + sqrt(radicand)) / vel_sq, 1);
    } else {
        t = 0;
    }
    if ((dx + t * vx) * hor > 0 && (dy + t * vy) * ver > 0) return t;
    return 1;
}

/* entities.py:82-96, 180-203 */
static double seg_intersect_with_ray(const Seg *s, double xpos, double ypos, double dx, double dy, double radius)
{
    double t1, t2, t3;
    if (s->kind == 0) {
        t1 = toi_circle_circle(xpos, ypos, dx, dy, s->x1, s->y1, radius);
        t2 = toi_circle_circle(xpos, ypos, dx, dy, s->x2, s->y2, radius);
        t3 = toi_circle_lineseg(xpos, ypos, dx, dy, s->x1, s->y1, s->x2, s->y2, radius);
    } else {
        t1 = toi_circle_circle(xpos, ypos, dx, dy, s->phx, s->phy, radius);
        t2 = toi_circle_circle(xpos, ypos, dx, dy, s->pvx, s->pvy, radius);
        t3 = toi_circle_arc(xpos, ypos, dx, dy, s->cx, s->cy, s->hor, s->ver, s->radius, radius);
    }
    double r = t1;
    if (t2 < r) r = t2;
    if (t3 < r) r = t3;
    return r;
}

/* physics.py:104-128 */
static double sweep_circle_vs_tiles(OSim *S, double xpos_old, double ypos_old, double dx, double dy, double radius)
{
    double xpos_new = xpos_old + dx, ypos_new = ypos_old + dy;
    double width = radius + 1;
    double x1 = (xpos_old < xpos_new ? xpos_old : xpos_new) - width;
    double y1 = (ypos_old < ypos_new ? ypos_old : ypos_new) - width;
    double x2 = (xpos_old > xpos_new ? xpos_old : xpos_new) + width;
    double y2 = (ypos_old > ypos_new ? ypos_old : ypos_new) + width;
    Seg *q[MAXQ];
    int n = query_region(S, x1, y1, x2, y2, q);
    double shortest = 1;
    for (int i = 0; i < n; i++) {
        double t = seg_intersect_with_ray(q[i], xpos_old, ypos_old, dx, dy, radius);
        if (t == 0) return 0;
        if (t < shortest) shortest = t;
    }
    return shortest;
}

/* entities.py:43-59, 127-157 */
static int seg_closest_point(const Seg *s, double xpos, double ypos, double *a, double *b)
{
    if (s->kind == 0) {
        double dx = xpos - s->x1, dy = ypos - s->y1;
        double u = (dx * s->px + dy * s->py) / s->seg_lensq;
        u = pymax(u, 0);
        u = pymin(u, 1);
        *a = s->x1 + u * s->px;
        *b = s->y1 + u * s->py;
        return (dy * s->px - dx * s->py < 0) && s->oriented;
    }
    double dx = xpos - s->cx, dy = ypos - s->cy;
    int back = 0;
    if (dx * s->hor > 0 && dy * s->ver > 0) {
        double dist_sq = SQ(dx) + SQ(dy);
        double dist = sqrt(dist_sq);
        if (dist == 0) {
            if (dx * s->hor > dy * s->ver) { *a = s->phx; *b = s->phy; }
            else { *a = s->pvx; *b = s->pvy; }
            return 0;
        }
        *a = s->cx + s->radius * dx / dist;
        *b = s->cy + s->radius * dy / dist;
        back = s->convex ? (dist < s->radius) : (dist > s->radius);
    } else {
        if (dx * s->hor > dy * s->ver) { *a = s->phx; *b = s->phy; }
        else { *a = s->pvx; *b = s->pvy; }
    }
    return back;
}

/* physics.py:131-180 */
static int get_single_closest_point(double xpos, double ypos, double radius, Seg **segs, int n, double *oa, double *ob)
{
    double shortest = INFINITY;
    int result = 0;
    double qx0 = xpos - radius, qy0 = ypos - radius, qx1 = xpos + radius, qy1 = ypos + radius;
    for (int i = 0; i < n; i++) {
        const Seg *s = segs[i];
        if (s->b[2] < qx0 || s->b[0] > qx1 || s->b[3] < qy0 || s->b[1] > qy1) continue;
        double a, b;
        int back = seg_closest_point(s, xpos, ypos, &a, &b);
        double distance_sq = SQ(xpos - a) + SQ(ypos - b);
        if (!back) distance_sq -= 0.1;
        if (distance_sq < shortest) {
            shortest = distance_sq;
            *oa = a; *ob = b;
            result = back ? -1 : 1;
        }
    }
    return result;
}

/* physics.py:204-207 */
static int overlap_circle_vs_circle(double x1, double y1, double r1, double x2, double y2, double r2)
{
    double dist = sqrt(SQ(x1 - x2) + SQ(y1 - y2));
    return dist < r1 + r2;
}

/* physics.py:79-101 */
static int gather_entities(OSim *S, double xpos, double ypos, Entity **out, int max)
{
    int cx = iclamp((int)floor(xpos / 24), 0, 43), cy = iclamp((int)floor(ypos / 24), 0, 24);
    int x0 = cx - 1 > 0 ? cx - 1 : 0, x1 = cx + 1 < 43 ? cx + 1 : 43;
    int y0 = cy - 1 > 0 ? cy - 1 : 0, y1 = cy + 1 < 24 ? cy + 1 : 24;
    int n = 0;
    for (int x = x0; x <= x1; x++)
        for (int y = y0; y <= y1; y++) {
            EList *l = &S->grid[x][y];
            for (int k = 0; k < l->n && n < max; k++)
                if (l->e[k]->active) out[n++] = l->e[k];
        }
    return n;
}

/* ------------------------------------------------------------------------------------
 * ninja.py
 * ---------------------------------------------------------------------------------- */
static int ninja_valid_target(const Ninja *n) { return !(n->state == 6 || n->state == 8 || n->state == 9); } /* :1272 */

static void ninja_kill(Ninja *n, int cause)
{   /* ninja.py:1253-1270 */
    if (n->state < 6) {
        n->death_cause = cause;
        if (n->state == 3) n->applied_gravity = GRAVITY_FALL;
        n->state = 7;
    }
}

static void ninja_win(Ninja *n)
{   /* ninja.py:1246-1251 */
    if (n->state < 6) {
        if (n->state == 3) n->applied_gravity = GRAVITY_FALL;
        n->state = 8;
    }
}

/* physics.py:16-18, 210-235 */
static int half_x(int x) { return iclamp(x, 0, 87); }
static int half_y(int y) { return iclamp(y, 0, 49); }

static int is_empty_row(const OSim *S, int x1, int x2, int y, int dir)
{
    if (dir != 1 && dir != -1) return 0;   /* the reference returns None there, which its callers read as "blocked" */
    int yy = half_y(dir == 1 ? y + 1 : y);
    for (int x = x1; x <= x2; x++)
        if (S->hor_edge[half_x(x)][yy]) return 0;
    return 1;
}

static int is_empty_column(const OSim *S, int x, int y1, int y2, int dir)
{
    if (dir != 1 && dir != -1) return 0;
    int xx = half_x(dir == 1 ? x + 1 : x);
    for (int y = y1; y <= y2; y++)
        if (S->ver_edge[xx][half_y(y)]) return 0;
    return 1;
}

/* physics.py:183-201; returns 0 when there is no penetration */
static int penetration_square_vs_point(double sx, double sy, double px, double py, double semi_side,
                                       double *nx, double *ny, double *len, double *len2)
{
    double dx = px - sx, dy = py - sy;
    double penx = semi_side - fabs(dx), peny = semi_side - fabs(dy);
    if (penx > 0 && peny > 0) {
        if (peny <= penx) { *nx = 0; *ny = dy < 0 ? -1 : 1; *len = peny; *len2 = penx; }
        else { *nx = dx < 0 ? -1 : 1; *ny = 0; *len = penx; *len2 = peny; }
        return 1;
    }
    return 0;
}

/* physics.py:457-471 */
static int overlap_circle_vs_segment(double xpos, double ypos, double radius, double px1, double py1, double px2, double py2)
{
    double px = px2 - px1, py = py2 - py1;
    double dx = xpos - px1, dy = ypos - py1;
    double seg_lensq = SQ(px) + SQ(py);
    double u = (dx * px + dy * py) / seg_lensq;
    u = pymax(u, 0);
    u = pymin(u, 1);
    double a = px1 + u * px, b = py1 + u * py;
    return SQ(xpos - a) + SQ(ypos - b) < SQ(radius);
}

static const int DIR_VX[4] = {1, 0, -1, 0}, DIR_VY[4] = {0, 1, 0, -1};       /* entity_drone_base.py:68 */
static const int DIR_LIST[4][4] = {{1, 0, 3, 2}, {3, 0, 1, 2}, {0, 1, 3, 2}, {0, 3, 1, 2}};   /* :72 */

/* entity_drone_base.py:138-164 */
static int drone_test_direction(OSim *S, Entity *e, int dir)
{
    int xdir = DIR_VX[dir], ydir = DIR_VY[dir];
    double xtarget = e->x + e->grid_width * xdir;
    double ytarget = e->y + e->grid_width * ydir;
    double R = e->drone_radius;
    if (!ydir) {
        int cell_x = (int)floor((e->x + xdir * R) / 12);
        int cell_xtarget = (int)floor((xtarget + xdir * R) / 12);
        int cell_y1 = (int)floor((e->y - R) / 12);
        int cell_y2 = (int)floor((e->y + R) / 12);
        while (cell_x != cell_xtarget) {
            if (!is_empty_column(S, cell_x, cell_y1, cell_y2, xdir)) return 0;
            cell_x += xdir;
        }
    } else {
        int cell_y = (int)floor((e->y + ydir * R) / 12);
        int cell_ytarget = (int)floor((ytarget + ydir * R) / 12);
        int cell_x1 = (int)floor((e->x - R) / 12);
        int cell_x2 = (int)floor((e->x + R) / 12);
        while (cell_y != cell_ytarget) {
            if (!is_empty_row(S, cell_x1, cell_x2, cell_y, ydir)) return 0;
            cell_y += ydir;
        }
    }
    e->xtarget = xtarget;
    e->ytarget = ytarget;
    return 1;
}

/* entity_drone_base.py:93-136 */
static void drone_move(OSim *S, Entity *e)
{
    double xspeed = e->speed * DIR_VX[e->dir], yspeed = e->speed * DIR_VY[e->dir];
    double dx = e->xtarget - e->x, dy = e->ytarget - e->y;
    double dist = sqrt(SQ(dx) + SQ(dy));
    if (dist < 0.000001 || (dx * (e->xtarget - (e->x + xspeed)) + dy * (e->ytarget - (e->y + yspeed))) < 0) {
        e->x = e->xtarget;
        e->y = e->ytarget;
        int can_move = 0;
        for (int i = 0; i < 4; i++) {
            int new_dir = (e->dir + DIR_LIST[e->mode][i]) % 4;
            if (drone_test_direction(S, e, new_dir)) { e->dir = new_dir; can_move = 1; break; }
        }
        if (can_move) {
            double disp = e->speed - dist;
            e->x += disp * DIR_VX[e->dir];
            e->y += disp * DIR_VY[e->dir];
        }
    } else {
        e->x += xspeed;
        e->y += yspeed;
        grid_move(S, e);
    }
}

/* entity_bounce_block.py:107-125 (the data-tracking tail does not feed back into the physics) */
static void bounce_move(OSim *S, Entity *e)
{
    e->xspeed *= 0.98;
    e->yspeed *= 0.98;
    e->x += e->xspeed;
    e->y += e->yspeed;
    double xforce = 0.02222222222222222 * (e->xorigin - e->x);
    double yforce = 0.02222222222222222 * (e->yorigin - e->y);
    e->x += xforce;
    e->y += yforce;
    e->xspeed += xforce;
    e->yspeed += yforce;
    grid_move(S, e);
}

/* entity_thwump.py:109-156 */
static void thwump_move(OSim *S, Entity *e)
{
    if (!e->state) return;
    double speed = e->state == 1 ? 20.0 / 7 : 8.0 / 7;
    int speed_dir = e->direction * e->state;
    if (!e->is_horizontal) {
        double ypos_new = e->y + speed * speed_dir;
        if (e->state == -1 && (ypos_new - e->yorigin) * (e->y - e->yorigin) < 0) {
            e->y = e->yorigin;
            e->state = 0;
            return;
        }
        int cell_y = (int)floor((e->y + speed_dir * 11) / 12);
        int cell_y_new = (int)floor((ypos_new + speed_dir * 11) / 12);
        if (cell_y != cell_y_new) {
            int cell_x1 = (int)floor((e->x - 11) / 12), cell_x2 = (int)floor((e->x + 11) / 12);
            if (!is_empty_row(S, cell_x1, cell_x2, cell_y, speed_dir)) { e->state = -1; return; }
        }
        e->y = ypos_new;
    } else {
        double xpos_new = e->x + speed * speed_dir;
        if (e->state == -1 && (xpos_new - e->xorigin) * (e->x - e->xorigin) < 0) {
            e->x = e->xorigin;
            e->state = 0;
            return;
        }
        int cell_x = (int)floor((e->x + speed_dir * 11) / 12);
        int cell_x_new = (int)floor((xpos_new + speed_dir * 11) / 12);
        if (cell_x != cell_x_new) {
            int cell_y1 = (int)floor((e->y - 11) / 12), cell_y2 = (int)floor((e->y + 11) / 12);
            if (!is_empty_column(S, cell_x, cell_y1, cell_y2, speed_dir)) { e->state = -1; return; }
        }
        e->x = xpos_new;
    }
    grid_move(S, e);
}

/* entity_boost_pad.py:46-64 */
static void boost_move(OSim *S, Entity *e)
{
    Ninja *n = &S->nj;
    if (!(n->state != 6 && n->state != 8 && n->state != 9)) { e->touching = 0; return; }
    if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
        if (!e->touching) {
            double vel_norm = sqrt(SQ(n->xspeed) + SQ(n->yspeed));
            if (vel_norm > 0) {
                double x_boost = 2 * n->xspeed / vel_norm;
                double y_boost = 2 * n->yspeed / vel_norm;
                n->xspeed += x_boost;
                n->yspeed += y_boost;
            }
            e->touching = 1;
        }
    } else {
        e->touching = 0;
    }
}

/* entity_thwump.py:158-206 */
static void thwump_think(OSim *S, Entity *e)
{
    Ninja *n = &S->nj;
    if (e->state || !(n->state != 6 && n->state != 8 && n->state != 9)) return;
    double activation_range = 2 * (9 + NINJA_RADIUS);
    int i = 0;
    if (!e->is_horizontal) {
        if (fabs(e->x - n->xpos) < activation_range) {
            int ninja_ycell = (int)floor(n->ypos / 12);
            int thwump_ycell = (int)floor((e->y - e->direction * 11) / 12);
            int thwump_xcell1 = (int)floor((e->x - 11) / 12), thwump_xcell2 = (int)floor((e->x + 11) / 12);
            int dy = ninja_ycell - thwump_ycell;
            if (dy * e->direction >= 0) {
                for (i = 0; i < 100; i++) {
                    if (!is_empty_row(S, thwump_xcell1, thwump_xcell2, thwump_ycell, e->direction)) {
                        dy = ninja_ycell - thwump_ycell;
                        break;
                    }
                    thwump_ycell += e->direction;
                }
                if (i == 100) i = 99;   /* Python leaves the loop variable at 99 when range(100) is exhausted */
                if (i > 0 && dy * e->direction <= 0) e->state = 1;
            }
        }
    } else {
        if (fabs(e->y - n->ypos) < activation_range) {
            int ninja_xcell = (int)floor(n->xpos / 12);
            int thwump_xcell = (int)floor((e->x - e->direction * 11) / 12);
            int thwump_ycell1 = (int)floor((e->y - 11) / 12), thwump_ycell2 = (int)floor((e->y + 11) / 12);
            int dx = ninja_xcell - thwump_xcell;
            if (dx * e->direction >= 0) {
                for (i = 0; i < 100; i++) {
                    if (!is_empty_column(S, thwump_xcell, thwump_ycell1, thwump_ycell2, e->direction)) {
                        dx = ninja_xcell - thwump_xcell;
                        break;
                    }
                    thwump_xcell += e->direction;
                }
                if (i == 100) i = 99;
                if (i > 0 && dx * e->direction <= 0) e->state = 1;
            }
        }
    }
}

/* entity_shove_thwump.py:127-153 */
static void shove_move_if_possible(OSim *S, Entity *e, double xdir, double ydir, double speed)
{
    if (e->ydir == 0) {
        double xpos_new = e->x + xdir * speed;
        int cell_x = (int)floor(e->x / 12), cell_x_new = (int)floor(xpos_new / 12);
        if (cell_x != cell_x_new) {
            int cell_y1 = (int)floor((e->y - 8) / 12), cell_y2 = (int)floor((e->y + 8) / 12);
            if (!is_empty_column(S, cell_x, cell_y1, cell_y2, (int)xdir)) { e->state = 3; return; }
        }
        e->x = xpos_new;
    } else {
        double ypos_new = e->y + ydir * speed;
        int cell_y = (int)floor(e->y / 12), cell_y_new = (int)floor(ypos_new / 12);
        if (cell_y != cell_y_new) {
            int cell_x1 = (int)floor((e->x - 8) / 12), cell_x2 = (int)floor((e->x + 8) / 12);
            if (!is_empty_row(S, cell_x1, cell_x2, cell_y, (int)ydir)) { e->state = 3; return; }
        }
        e->y = ypos_new;
    }
    grid_move(S, e);
}

/* entity_shove_thwump.py:109-125 */
static void shove_think(OSim *S, Entity *e)
{
    if (e->state == 1) {
        if (e->activated) { e->activated = 0; return; }
        e->state = 2;
    }
    if (e->state == 3) {
        double origin_dist = fabs(e->x - e->xorigin) + fabs(e->y - e->yorigin);
        if (origin_dist >= 1) shove_move_if_possible(S, e, e->xdir, e->ydir, 1);
        else { e->x = e->xorigin; e->y = e->yorigin; e->state = 0; }
    } else if (e->state == 2) {
        shove_move_if_possible(S, e, -e->xdir, -e->ydir, 4);
    }
}

static double sweep_circle_vs_tiles(OSim *S, double xpos_old, double ypos_old, double dx, double dy, double radius);
static int query_region(OSim *S, double x1, double y1, double x2, double y2, Seg **out);
static int get_single_closest_point(double xpos, double ypos, double radius, Seg **segs, int n, double *oa, double *ob);

/* entity_death_ball.py:74-167 */
static void ball_think(OSim *S, Entity *e)
{
    Ninja *n = &S->nj;
    if (!(n->state != 6 && n->state != 8 && n->state != 9)) {
        e->xspeed *= 0.95;
        e->yspeed *= 0.95;
    } else {
        double dx = n->xpos - e->x, dy = n->ypos - e->y;
        double dist = sqrt(SQ(dx) + SQ(dy));
        if (dist > 0) { dx /= dist; dy /= dist; }
        e->xspeed += dx * 0.04;
        e->yspeed += dy * 0.04;
        double speed = sqrt(SQ(e->xspeed) + SQ(e->yspeed));
        if (speed > 0.85) {
            double new_speed = (speed - 0.85) * 0.9;
            if (new_speed <= 0.01) new_speed = 0;
            new_speed += 0.85;
            e->xspeed = e->xspeed / speed * new_speed;
            e->yspeed = e->yspeed / speed * new_speed;
        }
    }
    double xpos_old = e->x, ypos_old = e->y;
    e->x += e->xspeed;
    e->y += e->yspeed;
    double time = sweep_circle_vs_tiles(S, xpos_old, ypos_old, e->xspeed, e->yspeed, 8 * 0.5);
    e->x = xpos_old + time * e->xspeed;
    e->y = ypos_old + time * e->yspeed;
    double xnormal = 0, ynormal = 0;
    for (int it = 0; it < 16; it++) {
        Seg *segs[MAXQ];
        int nseg = query_region(S, e->x - 8, e->y - 8, e->x + 8, e->y + 8, segs);
        double a = 0, b = 0;
        int result = get_single_closest_point(e->x, e->y, 8, segs, nseg, &a, &b);
        if (result == 0) break;
        double dx = e->x - a, dy = e->y - b;
        double dist = sqrt(SQ(dx) + SQ(dy));
        double depen_len = 8 - dist * result;
        if (depen_len < 0.0000001) break;
        if (dist == 0) return;
        double xnorm = dx / dist, ynorm = dy / dist;
        e->x += xnorm * depen_len;
        e->y += ynorm * depen_len;
        xnormal += xnorm;
        ynormal += ynorm;
    }
    double normal_len = sqrt(SQ(xnormal) + SQ(ynormal));
    if (normal_len > 0) {
        double dx = xnormal / normal_len, dy = ynormal / normal_len;
        double dot_product = e->xspeed * dx + e->yspeed * dy;
        if (dot_product < 0) {
            double speed = sqrt(SQ(e->xspeed) + SQ(e->yspeed));
            int bounce_strength = speed <= 1.35 ? 1 : 2;
            e->xspeed -= dx * dot_product * bounce_strength;
            e->yspeed -= dy * dot_product * bounce_strength;
        }
    }
    /* :153-166.  Entity.index counts creations over the Simulator's lifetime (sim.entity_counts is never
     * cleared), so the ball-ball repulsion only ever runs before the first Simulator.reset(). */
    double db_count = S->nmap > 1200 ? S->map[1200] : 0;
    int nballs = 0;
    Entity *balls[64];
    for (int i = 0; i < S->ndic; i++)
        if (S->dic_order[i]->kind == K_BALL && nballs < 64) balls[nballs++] = S->dic_order[i];
    int index = e->index + (S->creations - 1) * nballs;
    if (index + 1 < db_count) {
        for (int k = index + 1; k < nballs; k++) {
            Entity *t = balls[k];
            double dx = e->x - t->x, dy = e->y - t->y;
            double dist = sqrt(SQ(dx) + SQ(dy));
            if (dist < 16) {
                dx = dx / dist * 4;
                dy = dy / dist * 4;
                e->xspeed += dx;
                e->yspeed += dy;
                t->xspeed -= dx;
                t->yspeed -= dy;
            }
        }
    }
    grid_move(S, e);
}

/* entity_door_regular.py:46-53 */
static void door_regular_think(OSim *S, Entity *e)
{
    if (!e->closed) {
        e->open_timer += 1;
        if (e->open_timer > 5) { e->closed = 1; door_edges_add(S, e, 1); }
    }
}

/* entity_one_way_platform.py:79-105 */
static int oneway_depen(const OSim *S, const Entity *e, double *len)
{
    const Ninja *n = &S->nj;
    double dx = n->xpos - e->x, dy = n->ypos - e->y;
    double lateral_dist = dy * e->nx - dx * e->ny;
    double direction = (n->yspeed * e->nx - n->xspeed * e->ny) * lateral_dist;
    double radius_scalar = direction < 0 ? 0.91 : 0.51;
    if (fabs(lateral_dist) < radius_scalar * NINJA_RADIUS + 12) {
        double normal_dist = dx * e->nx + dy * e->ny;
        if (0 < normal_dist && normal_dist <= NINJA_RADIUS) {
            double normal_proj = n->xspeed * e->nx + n->yspeed * e->ny;
            if (normal_proj <= 0) {
                double dx_old = n->xpos_old - e->x, dy_old = n->ypos_old - e->y;
                double normal_dist_old = dx_old * e->nx + dy_old * e->ny;
                if (NINJA_RADIUS - normal_dist_old <= 1.1) {
                    *len = NINJA_RADIUS - normal_dist;
                    return 1;
                }
            }
        }
    }
    return 0;
}

/* physical_collision() of the four physically collidable kinds; 1 = (nx, ny), (len, len2) returned */
static int entity_physical_collision(OSim *S, Entity *e, double *nx, double *ny, double *len)
{
    Ninja *n = &S->nj;
    double len2;
    switch (e->kind) {
    case K_ONEWAY:
        if (oneway_depen(S, e, len)) { *nx = e->nx; *ny = e->ny; return 1; }
        return 0;
    case K_BOUNCE: { /* entity_bounce_block.py:127-145 */
        double dl;
        if (!penetration_square_vs_point(e->x, e->y, n->xpos, n->ypos, 9 + NINJA_RADIUS, nx, ny, &dl, &len2)) return 0;
        e->x -= *nx * dl * (1 - 0.2);
        e->y -= *ny * dl * (1 - 0.2);
        e->xspeed -= *nx * dl * (1 - 0.2);
        e->yspeed -= *ny * dl * (1 - 0.2);
        *len = dl * 0.2;
        return 1;
    }
    case K_THWUMP: /* entity_thwump.py:208-213 */
        return penetration_square_vs_point(e->x, e->y, n->xpos, n->ypos, 9 + NINJA_RADIUS, nx, ny, len, &len2);
    case K_SHOVE: /* entity_shove_thwump.py:155-171 */
        if (e->state <= 1) {
            if (penetration_square_vs_point(e->x, e->y, n->xpos, n->ypos, 12 + NINJA_RADIUS, nx, ny, len, &len2)) {
                if (e->state == 0 || e->xdir * *nx + e->ydir * *ny >= 0.01) return 1;
            }
        }
        return 0;
    default:
        return 0;
    }
}

/* entity_toggle_mine.py:90-118 */
static void mine_think(OSim *S, Entity *e)
{
    Ninja *n = &S->nj;
    if (ninja_valid_target(n)) {
        if (e->state == 1) {
            if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) mine_set_state(e, 2);
        } else if (e->state == 2) {
            if (!overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) mine_set_state(e, 0);
        }
    } else {
        if (e->state == 2 && n->state == 6) mine_set_state(e, 1);
    }
}

/* logical_collision() of every kind.  Returns 0 = None, 1 = a wall-normal contribution in *r0 (may be 0, which the
 * caller's truth test drops like the reference's), 2 = launch pad boost (r0, r1). */
static int entity_logical_collision(OSim *S, Entity *e, double *r0, double *r1)
{
    Ninja *n = &S->nj;
    double nx, ny, len, len2;
    switch (e->kind) {
    case K_MINE: /* entity_toggle_mine.py:120-128 */
        if (ninja_valid_target(n) && e->state == 0) {
            if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
                mine_set_state(e, 1);
                ninja_kill(n, 1);
            }
        }
        break;
    case K_GOLD: /* entity_gold.py:66-74 */
        if (n->state != 8) {
            if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
                n->gold_collected += 1;
                e->active = 0;
            }
        }
        break;
    case K_EXIT: /* entity_exit.py:66-74 */
        if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) ninja_win(n);
        break;
    case K_SWITCH: /* entity_exit_switch.py:67-129 */
        if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
            e->active = 0;
            elist_push(&S->grid[e->parent->cx][e->parent->cy], e->parent);
            e->parent->switch_hit = 1;
        }
        break;
    case K_LOCKED: /* entity_door_locked.py:54-67 */
        if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
            n->doors_opened += 1;
            e->closed = 0;
            door_edges_add(S, e, -1);
            e->active = 0;
        }
        break;
    case K_DOOR_REG: /* entity_door_regular.py:55-63: every overlapping frame decrements the edge counters again */
        if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
            e->closed = 0;
            door_edges_add(S, e, -1);
            e->open_timer = 0;
        }
        break;
    case K_DOOR_TRAP: /* entity_door_trap.py:60-67 */
        if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
            e->closed = 1;
            door_edges_add(S, e, 1);
            e->active = 0;
        }
        break;
    case K_LAUNCH: /* entity_launch_pad.py:82-100 */
        if (ninja_valid_target(n)) {
            if (overlap_circle_vs_circle(e->x, e->y, e->radius, n->xpos, n->ypos, NINJA_RADIUS)) {
                if (((e->x - (n->xpos - NINJA_RADIUS * e->nx)) * e->nx + (e->y - (n->ypos - NINJA_RADIUS * e->ny)) * e->ny) >= -0.1) {
                    double yboost_scale = 1;
                    if (e->ny < 0) yboost_scale = 1 - e->ny;
                    *r0 = e->nx * (36.0 / 7);
                    *r1 = e->ny * (36.0 / 7) * yboost_scale;
                    return 2;
                }
            }
        }
        break;
    case K_ONEWAY: /* entity_one_way_platform.py:111-116 */
        if (oneway_depen(S, e, &len)) {
            if (fabs(e->nx) == 1) { *r0 = e->nx; return 1; }
        }
        break;
    case K_DRONE: /* entity_drone_zap.py:57-64, entity_mini_drone.py:60-67 */
        if (ninja_valid_target(n)) {
            if (overlap_circle_vs_circle(e->x, e->y, e->drone_radius, n->xpos, n->ypos, NINJA_RADIUS)) ninja_kill(n, 3);
        }
        break;
    case K_BOUNCE: /* entity_bounce_block.py:147-158 */
        if (penetration_square_vs_point(e->x, e->y, n->xpos, n->ypos, 9 + NINJA_RADIUS + 0.1, &nx, &ny, &len, &len2)) {
            *r0 = nx;
            return 1;
        }
        break;
    case K_THWUMP: /* entity_thwump.py:215-243 */
        if (ninja_valid_target(n)) {
            if (penetration_square_vs_point(e->x, e->y, n->xpos, n->ypos, 9 + NINJA_RADIUS + 0.1, &nx, &ny, &len, &len2)) {
                double px1, py1, px2, py2;
                if (e->is_horizontal) {
                    double dx = (9 + 2) * e->direction, dy = 9 - 2;
                    px1 = e->x + dx; py1 = e->y - dy; px2 = e->x + dx; py2 = e->y + dy;
                } else {
                    double dx = 9 - 2, dy = (9 + 2) * e->direction;
                    px1 = e->x - dx; py1 = e->y + dy; px2 = e->x + dx; py2 = e->y + dy;
                }
                if (overlap_circle_vs_segment(n->xpos, n->ypos, NINJA_RADIUS + 2, px1, py1, px2, py2)) ninja_kill(n, 3);
                *r0 = nx;
                return 1;
            }
        }
        break;
    case K_BALL: /* entity_death_ball.py:169-181 */
        if (ninja_valid_target(n)) {
            if (overlap_circle_vs_circle(e->x, e->y, 5, n->xpos, n->ypos, NINJA_RADIUS)) {
                double dx = e->x - n->xpos, dy = e->y - n->ypos;
                double dist = sqrt(SQ(dx) + SQ(dy));
                e->xspeed += dx / dist * 10;
                e->yspeed += dy / dist * 10;
                ninja_kill(n, 3);
            }
        }
        break;
    case K_SHOVE: { /* entity_shove_thwump.py:173-202 */
        int depen = penetration_square_vs_point(e->x, e->y, n->xpos, n->ypos, 12 + NINJA_RADIUS + 0.1, &nx, &ny, &len, &len2);
        if (depen && e->state <= 1) {
            if (e->state == 0) {
                e->activated = 1;
                if (len2 > 0.2) { e->xdir = nx; e->ydir = ny; e->state = 1; }
            } else {
                if (e->xdir * nx + e->ydir * ny >= 0.01) e->activated = 1;
                else return 0;
            }
            *r0 = nx;
            return 1;
        }
        if (overlap_circle_vs_circle(n->xpos, n->ypos, NINJA_RADIUS, e->x, e->y, 8)) ninja_kill(n, 3);
        break;
    }
    }
    return 0;
}

/* ninja.py:198-206 */
static void ninja_integrate(Ninja *n)
{
    n->xspeed *= n->applied_drag;
    n->yspeed *= n->applied_drag;
    n->yspeed += n->applied_gravity;
    n->xpos_old = n->xpos;
    n->ypos_old = n->ypos;
    n->xpos += n->xspeed;
    n->ypos += n->yspeed;
}

/* ninja.py:208-222 */
static void ninja_pre_collision(OSim *S)
{
    Ninja *n = &S->nj;
    n->xspeed_old = n->xspeed;
    n->yspeed_old = n->yspeed;
    n->floor_count = n->wall_count = n->ceiling_count = 0;
    n->floor_normal_x = n->floor_normal_y = 0;
    n->ceiling_normal_x = n->ceiling_normal_y = 0;
    n->is_crushable = 0;
    n->x_crush = n->y_crush = 0;
    n->crush_len = 0;
    /* _cached_entities only feeds collide_vs_objects, which only looks at physically collidable entities */
    S->ncached = S->has_zoo ? gather_entities(S, n->xpos, n->ypos, S->cached, 256) : 0;
}

/* ninja.py:224-267 */
static void ninja_collide_vs_objects(OSim *S)
{
    Ninja *n = &S->nj;
    for (int i = 0; i < S->ncached; i++) {
        Entity *e = S->cached[i];
        if (!e->physical) continue;
        double depen_x, depen_y, depen_len;
        if (!entity_physical_collision(S, e, &depen_x, &depen_y, &depen_len)) continue;
        double pop_x = depen_x * depen_len, pop_y = depen_y * depen_len;
        n->xpos += pop_x;
        n->ypos += pop_y;
        if (e->type != 17) {
            n->x_crush += pop_x;
            n->y_crush += pop_y;
            n->crush_len += depen_len;
        }
        if (e->type == 20) n->is_crushable = 1;
        if (e->type == 17 || e->type == 20 || e->type == 28) {
            n->xspeed += pop_x;
            n->yspeed += pop_y;
        }
        if (e->type == 11) {
            double xspeed_new = (n->xspeed * depen_y - n->yspeed * depen_x) * depen_y;
            double yspeed_new = (n->xspeed * depen_y - n->yspeed * depen_x) * (-depen_x);
            n->xspeed = xspeed_new;
            n->yspeed = yspeed_new;
        }
        if (depen_y >= -0.0001) {
            n->ceiling_count += 1;
            n->ceiling_normal_x += depen_x;
            n->ceiling_normal_y += depen_y;
        } else {
            n->floor_count += 1;
            n->floor_normal_x += depen_x;
            n->floor_normal_y += depen_y;
        }
    }
}

/* ninja.py:269-379 */
static void ninja_collide_vs_tiles(OSim *S)
{
    Ninja *n = &S->nj;
    double dx = n->xpos - n->xpos_old;
    double dy = n->ypos - n->ypos_old;
    double time = sweep_circle_vs_tiles(S, n->xpos_old, n->ypos_old, dx, dy, NINJA_RADIUS * 0.5);
    n->xpos = n->xpos_old + time * dx;
    n->ypos = n->ypos_old + time * dy;

    double rad = NINJA_RADIUS;
    Seg *segs[MAXQ];
    int nseg = query_region(S, n->xpos - rad, n->ypos - rad, n->xpos + rad, n->ypos + rad, segs);

    double xpos = n->xpos, ypos = n->ypos, xspeed = n->xspeed, yspeed = n->yspeed;
    for (int it = 0; it < 32; it++) {
        double a = 0, b = 0;
        int result = get_single_closest_point(xpos, ypos, NINJA_RADIUS, segs, nseg, &a, &b);
        if (result == 0) break;
        dx = xpos - a;
        dy = ypos - b;
        if (fabs(dx) <= 0.0000001) {
            dx = 0;
            if (xpos == 50.51197510492316 || xpos == 49.23232124849253) dx = -ldexp(1.0, -47);
            if (xpos == 49.153536108584795) dx = ldexp(1.0, -47);
        }
        double dist_sq = dx * dx + dy * dy;
        if (dist_sq < 1e-16) break;
        double dist = sqrt(dist_sq);
        double depen_len = NINJA_RADIUS - dist * result;
        if (depen_len < 0.0000001) break;
        double inv_dist = 1.0 / dist;
        double norm_dx = dx * inv_dist, norm_dy = dy * inv_dist;
        double depen_x = norm_dx * depen_len, depen_y = norm_dy * depen_len;
        xpos += depen_x;
        ypos += depen_y;
        n->x_crush += depen_x;
        n->y_crush += depen_y;
        n->crush_len += depen_len;
        double dot_product = xspeed * dx + yspeed * dy;
        if (dot_product < 0) {
            double cross_product = xspeed * dy - yspeed * dx;
            double inv_dist_sq = inv_dist * inv_dist;
            xspeed = cross_product * inv_dist_sq * dy;
            yspeed = cross_product * inv_dist_sq * (-dx);
        }
        if (dy >= -0.0001) {
            n->ceiling_count += 1;
            n->ceiling_normal_x += norm_dx;
            n->ceiling_normal_y += norm_dy;
        } else {
            n->floor_count += 1;
            n->floor_normal_x += norm_dx;
            n->floor_normal_y += norm_dy;
        }
    }
    n->xpos = xpos; n->ypos = ypos; n->xspeed = xspeed; n->yspeed = yspeed;
}

/* ninja.py:381-537 */
static void ninja_post_collision(OSim *S)
{
    Ninja *n = &S->nj;
    double wall_normal = 0;
    Entity *near[256];
    int ne = gather_entities(S, n->xpos, n->ypos, near, 256);
    for (int i = 0; i < ne; i++) {
        if (!near[i]->logical) continue;
        double r0 = 0, r1 = 0;
        int kind = entity_logical_collision(S, near[i], &r0, &r1);
        if (kind == 2) {   /* ninja.py:401-418: launch pad */
            double xboost = r0 * 2 / 3, yboost = r1 * 2 / 3;
            n->xpos += xboost;
            n->ypos += yboost;
            n->xspeed = xboost;
            n->yspeed = yboost;
            n->floor_count = 0;
            n->floor_buffer = -1;
            double boost_scalar = sqrt(SQ(xboost) + SQ(yboost));
            n->xlp_boost_normalized = xboost / boost_scalar;
            n->ylp_boost_normalized = yboost / boost_scalar;
            n->launch_pad_buffer = 0;
            if (n->state == 3) n->applied_gravity = GRAVITY_FALL;
            n->state = 4;
        } else if (kind == 1 && r0 != 0) {
            wall_normal += r0;
        }
    }

    double rad = NINJA_RADIUS + 0.1;
    Seg *segs[MAXQ];
    int nseg = query_region(S, n->xpos - rad, n->ypos - rad, n->xpos + rad, n->ypos + rad, segs);
    for (int i = 0; i < nseg; i++) {
        double a, b;
        seg_closest_point(segs[i], n->xpos, n->ypos, &a, &b);
        double dx = n->xpos - a, dy = n->ypos - b;
        double dist = sqrt(SQ(dx) + SQ(dy));
        if (fabs(dy) < 0.00001 && 0 < dist && dist <= rad) wall_normal += dx / dist;
    }

    n->airborn_old = n->airborn;
    n->airborn = 1;
    n->walled = 0;
    if (wall_normal != 0) {
        n->walled = 1;
        n->wall_normal = wall_normal / fabs(wall_normal);
    }

    if (n->floor_count > 0) {
        n->airborn = 0;
        double floor_scalar = sqrt(SQ(n->floor_normal_x) + SQ(n->floor_normal_y));
        if (floor_scalar == 0) {
            n->floor_normalized_x = 0;
            n->floor_normalized_y = -1;
        } else {
            n->floor_normalized_x = n->floor_normal_x / floor_scalar;
            n->floor_normalized_y = n->floor_normal_y / floor_scalar;
        }
        if (n->state != 8 && n->airborn_old) {
            double impact_vel = -(n->floor_normalized_x * n->xspeed_old + n->floor_normalized_y * n->yspeed_old);
            if (impact_vel > MAX_SURVIVABLE_IMPACT - 4.0 / 3 * fabs(n->floor_normalized_y)) {
                n->xspeed = n->xspeed_old;
                n->yspeed = n->yspeed_old;
                ninja_kill(n, 2);
                n->terminal_impact = 1;
            }
        }
    }

    if (n->airborn) n->frames_airborne += 1; else n->frames_airborne = 0;
    if (!n->airborn) n->consecutive_floor_frames += 1; else n->consecutive_floor_frames = 0;
    if (n->walled) n->consecutive_wall_frames += 1; else n->consecutive_wall_frames = 0;

    if (n->ceiling_count > 0) {
        double ceiling_scalar = sqrt(SQ(n->ceiling_normal_x) + SQ(n->ceiling_normal_y));
        if (ceiling_scalar == 0) {
            n->ceiling_normalized_x = 0;
            n->ceiling_normalized_y = 1;
        } else {
            n->ceiling_normalized_x = n->ceiling_normal_x / ceiling_scalar;
            n->ceiling_normalized_y = n->ceiling_normal_y / ceiling_scalar;
        }
        if (n->state != 8) {
            double impact_vel = -(n->ceiling_normalized_x * n->xspeed_old + n->ceiling_normalized_y * n->yspeed_old);
            if (impact_vel > MAX_SURVIVABLE_IMPACT - 4.0 / 3 * fabs(n->ceiling_normalized_y)) {
                n->xspeed = n->xspeed_old;
                n->yspeed = n->yspeed_old;
                ninja_kill(n, 2);
                n->terminal_impact = 1;
            }
        }
    }
    /* ninja.py:531-537 */
    if (n->is_crushable && n->crush_len > 0) {
        if (sqrt(SQ(n->x_crush) + SQ(n->y_crush)) / n->crush_len < MIN_SURVIVABLE_CRUSHING) ninja_kill(n, 3);
    }
}

/* ninja.py:539-579 */
static void ninja_floor_jump(Ninja *n)
{
    n->jump_buffer = -1;
    n->floor_buffer = -1;
    n->launch_pad_buffer = -1;
    n->state = 3;
    n->applied_gravity = GRAVITY_JUMP;
    double jx, jy;
    if (n->floor_normalized_x == 0) {
        jx = 0; jy = -2;
    } else {
        double dx = n->floor_normalized_x, dy = n->floor_normalized_y;
        if (n->xspeed * dx >= 0) {
            if (n->xspeed * n->hor_input >= 0) { jx = 2.0 / 3 * dx; jy = 2 * dy; }
            else { jx = 0; jy = -1.4; }
        } else {
            if (n->xspeed * n->hor_input > 0) { jx = 0; jy = -1.4; }
            else { n->xspeed = 0; jx = 2.0 / 3 * dx; jy = 2 * dy; }
        }
    }
    if (n->yspeed > 0) n->yspeed = 0;
    n->xspeed += jx; n->yspeed += jy;
    n->xpos += jx; n->ypos += jy;
    n->jump_duration = 0;
}

/* ninja.py:581-608 */
static void ninja_wall_jump(Ninja *n)
{
    double jx, jy;
    if (n->hor_input * n->wall_normal < 0 && n->state == 5) { jx = 2.0 / 3; jy = -1; }
    else { jx = 1; jy = -1.4; }
    n->state = 3;
    n->applied_gravity = GRAVITY_JUMP;
    if (n->xspeed * n->wall_normal < 0) n->xspeed = 0;
    if (n->yspeed > 0) n->yspeed = 0;
    n->xspeed += jx * n->wall_normal;
    n->yspeed += jy;
    n->xpos += jx * n->wall_normal;
    n->ypos += jy;
    n->jump_buffer = -1;
    n->wall_buffer = -1;
    n->launch_pad_buffer = -1;
    n->jump_duration = 0;
}

/* ninja.py:610-626 */
static void ninja_lp_jump(Ninja *n)
{
    n->floor_buffer = n->wall_buffer = n->jump_buffer = n->launch_pad_buffer = -1;
    double boost_scalar = 2 * fabs(n->xlp_boost_normalized) + 2;
    if (boost_scalar == 2) boost_scalar = 1.7;
    n->xspeed += n->xlp_boost_normalized * boost_scalar * (2.0 / 3);
    n->yspeed += n->ylp_boost_normalized * boost_scalar * (2.0 / 3);
}

/* ninja.py:849-1059 */
static void ninja_think(Ninja *n)
{
    if (n->state != n->previous_state) {
        n->state_change_frame = 0;
        n->previous_state = n->state;
    } else {
        n->state_change_frame += 1;
    }
    int new_jump_check = n->jump_input ? (n->jump_input_old == 0) : 0;
    n->jump_input_old = n->jump_input;

    if (-1 < n->launch_pad_buffer && n->launch_pad_buffer < 3) n->launch_pad_buffer += 1; else n->launch_pad_buffer = -1;
    int in_lp_buffer = -1 < n->launch_pad_buffer && n->launch_pad_buffer < 4;
    if (-1 < n->jump_buffer && n->jump_buffer < 5) n->jump_buffer += 1; else n->jump_buffer = -1;
    int in_jump_buffer = -1 < n->jump_buffer && n->jump_buffer < 5;
    if (-1 < n->wall_buffer && n->wall_buffer < 5) n->wall_buffer += 1; else n->wall_buffer = -1;
    int in_wall_buffer = -1 < n->wall_buffer && n->wall_buffer < 5;
    if (-1 < n->floor_buffer && n->floor_buffer < 5) n->floor_buffer += 1; else n->floor_buffer = -1;
    int in_floor_buffer = -1 < n->floor_buffer && n->floor_buffer < 5;

    if (new_jump_check && n->airborn) n->jump_buffer = 0;
    if (n->walled) n->wall_buffer = 0;
    if (!n->airborn) n->floor_buffer = 0;

    if (n->state == 6 || n->state == 9) return;
    if (n->state == 7) { n->state = 6; return; }   /* ninja.py:1061-1063 */
    if (n->state == 8) { n->applied_drag = n->airborn ? DRAG_REGULAR : DRAG_SLOW; return; }

    if (!n->airborn) {
        double xspeed_new = n->xspeed + GROUND_ACCEL * n->hor_input;
        if (fabs(xspeed_new) < MAX_HOR_SPEED) n->xspeed = xspeed_new;
        if (n->state > 2) {
            if (n->xspeed * n->hor_input <= 0) {
                if (n->state == 3) n->applied_gravity = GRAVITY_FALL;
                n->state = 2;
            } else {
                if (n->state == 3) n->applied_gravity = GRAVITY_FALL;
                n->state = 1;
            }
        }
        if (!in_jump_buffer && !new_jump_check) {
            if (n->state == 2) {
                double projection = fabs(n->yspeed * n->floor_normalized_x - n->xspeed * n->floor_normalized_y);
                if (n->hor_input * projection * n->xspeed > 0) { n->state = 1; return; }
                if (projection < 0.1 && n->floor_normalized_x == 0) { n->state = 0; return; }
                if (n->yspeed < 0 && n->floor_normalized_x != 0) {
                    double speed_scalar = sqrt(SQ(n->xspeed) + SQ(n->yspeed));
                    double fric_force = fabs(n->xspeed * (1 - FRICTION_GROUND) * n->floor_normalized_y);
                    double fric_force2 = speed_scalar - fric_force * SQ(n->floor_normalized_y);
                    n->xspeed = n->xspeed / speed_scalar * fric_force2;
                    n->yspeed = n->yspeed / speed_scalar * fric_force2;
                    return;
                }
                n->xspeed *= FRICTION_GROUND;
                return;
            }
            if (n->state == 1) {
                double projection = fabs(n->yspeed * n->floor_normalized_x - n->xspeed * n->floor_normalized_y);
                if (n->hor_input * projection * n->xspeed > 0) {
                    if (n->hor_input * n->floor_normalized_x >= 0) return;
                    if (fabs(xspeed_new) < MAX_HOR_SPEED) {
                        double boost = GROUND_ACCEL / 2 * n->hor_input;
                        double xboost = boost * n->floor_normalized_y * n->floor_normalized_y;
                        double yboost = boost * n->floor_normalized_y * -n->floor_normalized_x;
                        n->xspeed += xboost;
                        n->yspeed += yboost;
                    }
                    return;
                }
                n->state = 2;
            } else {
                if (n->hor_input) { n->state = 1; return; }
                double projection = fabs(n->yspeed * n->floor_normalized_x - n->xspeed * n->floor_normalized_y);
                if (projection < 0.1) { n->xspeed *= FRICTION_GROUND_SLOW; return; }
                n->state = 2;
            }
            return;
        }
        ninja_floor_jump(n);
        return;
    } else {
        double xspeed_new = n->xspeed + AIR_ACCEL * n->hor_input;
        if (fabs(xspeed_new) < MAX_HOR_SPEED) n->xspeed = xspeed_new;
        if (n->state < 3) { n->state = 4; return; }
        if (n->state == 3) {
            n->jump_duration += 1;
            if (!n->jump_input || n->jump_duration > MAX_JUMP_DURATION) {
                n->applied_gravity = GRAVITY_FALL;
                n->state = 4;
                return;
            }
        }
        if (in_jump_buffer || new_jump_check) {
            if (n->walled || in_wall_buffer) { ninja_wall_jump(n); return; }
            if (in_floor_buffer) { ninja_floor_jump(n); return; }
            if (in_lp_buffer && new_jump_check) { ninja_lp_jump(n); return; }
        }
        if (!n->walled) {
            if (n->state == 5) n->state = 4;
        } else {
            if (n->state == 5) {
                if (n->hor_input * n->wall_normal <= 0) n->yspeed *= FRICTION_WALL;
                else n->state = 4;
            } else {
                if (n->yspeed > 0 && n->hor_input * n->wall_normal < 0) {
                    if (n->state == 3) n->applied_gravity = GRAVITY_FALL;
                    n->state = 5;
                }
            }
        }
    }
}

/* nsim.py:221-292 */
void osim_tick(OSim *S, int hor_input, int jump_input)
{
    Ninja *n = &S->nj;
    S->frame += 1;
    n->hor_input = hor_input;
    n->jump_input = jump_input;
    if (!S->has_zoo) {
        for (int i = 0; i < S->ndic; i++) {
            Entity *e = S->dic_order[i];
            if (e->active && e->thinkable) mine_think(S, e);
        }
    } else {
        /* nsim.py:235-251: both lists are taken from the activity flags at the start of the tick */
        unsigned char *mv = alloca(S->ndic), *th = alloca(S->ndic);
        for (int i = 0; i < S->ndic; i++) {
            Entity *e = S->dic_order[i];
            mv[i] = e->active && e->movable;
            th[i] = e->active && e->thinkable;
        }
        for (int i = 0; i < S->ndic; i++) {
            if (!mv[i]) continue;
            Entity *e = S->dic_order[i];
            switch (e->kind) {
            case K_DRONE: drone_move(S, e); break;
            case K_BOUNCE: bounce_move(S, e); break;
            case K_THWUMP: thwump_move(S, e); break;
            case K_BOOST: boost_move(S, e); break;
            default: break;
            }
        }
        for (int i = 0; i < S->ndic; i++) {
            if (!th[i]) continue;
            Entity *e = S->dic_order[i];
            switch (e->kind) {
            case K_MINE: mine_think(S, e); break;
            case K_DOOR_REG: door_regular_think(S, e); break;
            case K_THWUMP: thwump_think(S, e); break;
            case K_BALL: ball_think(S, e); break;
            case K_SHOVE: shove_think(S, e); break;
            default: break;
            }
        }
    }
    if (n->state != 9) {
        if (n->state != 6) {   /* state 6 -> ragdoll, which is None without the animation file */
            ninja_integrate(n);
            ninja_pre_collision(S);
            for (int k = 0; k < 4; k++) {
                ninja_collide_vs_objects(S);
                ninja_collide_vs_tiles(S);
            }
            ninja_post_collision(S);
        }
        ninja_think(n);
    }
}

/* entity checksum of tests/golden/make_golden_zoo.py:ent_row (entity_dic order) */
void osim_entity_checksum(const OSim *S, double *out)
{
    double sx = 0, sy = 0, svx = 0, svy = 0;
    long code = 0, act = 0;
    for (int i = 0; i < S->ndic; i++) {
        const Entity *e = S->dic_order[i];
        sx += e->x; sy += e->y;
        if (e->kind == K_BOUNCE || e->kind == K_BALL) { svx += e->xspeed; svy += e->yspeed; }
        int c = 0;
        if (e->kind == K_LOCKED || e->kind == K_DOOR_REG || e->kind == K_DOOR_TRAP) c += 3 * e->closed;
        if (e->kind == K_MINE || e->kind == K_THWUMP || e->kind == K_SHOVE) c += 5 * (((e->state % 7) + 7) % 7);
        if (e->kind == K_DRONE) c += 11 * e->dir;
        if (e->kind == K_BOOST) c += 13 * e->touching;
        if (e->kind == K_SHOVE) c += 17 * e->activated;
        code += c;
        act += e->active;
    }
    out[0] = sx; out[1] = sy; out[2] = svx; out[3] = svy; out[4] = (double)code; out[5] = (double)act;
}

/* ------------------------------------------------------------------------------------
 * observation side: nplay_headless.py:735-924, ninja.py:1217-1244, ninja.py:628-839
 * ---------------------------------------------------------------------------------- */
void osim_get_ninja_state(const OSim *S, double *o)
{
    const Ninja *n = &S->nj;
    double velocity_magnitude = POWHALF(SQ(n->xspeed) + SQ(n->yspeed));
    o[0] = pymin(velocity_magnitude / (MAX_HOR_SPEED * 2), 1.0) * 2 - 1;
    if (velocity_magnitude > 1e-6) { o[1] = n->xspeed / velocity_magnitude; o[2] = n->yspeed / velocity_magnitude; }
    else { o[1] = 0; o[2] = 0; }
    o[3] = (n->state >= 0 && n->state <= 2) ? 1 : -1;
    o[4] = (n->state == 3 || n->state == 4) ? 1 : -1;
    o[5] = n->state == 5 ? 1 : -1;
    o[6] = (n->state >= 6 && n->state <= 9) ? 1 : -1;
    o[7] = n->airborn ? 1 : -1;
    o[8] = n->hor_input;
    o[9] = n->jump_input ? 1 : -1;
    o[10] = ((n->jump_buffer > 0 ? n->jump_buffer : 0) / 5.0) * 2 - 1;
    o[11] = ((n->floor_buffer > 0 ? n->floor_buffer : 0) / 5.0) * 2 - 1;
    o[12] = ((n->wall_buffer > 0 ? n->wall_buffer : 0) / 5.0) * 2 - 1;
    o[13] = (n->floor_count < 1 ? n->floor_count : 1) * 2 - 1;
    o[14] = (n->wall_count < 1 ? n->wall_count : 1) * 2 - 1;
    o[15] = (n->ceiling_count < 1 ? n->ceiling_count : 1) * 2 - 1;
    o[16] = POWHALF(SQ(n->floor_normalized_x) + SQ(n->floor_normalized_y)) * 2 - 1;
    o[17] = n->wall_count > 0 ? n->wall_normal : 0.0;
    o[18] = n->floor_normalized_y;
    o[19] = (n->applied_gravity - GRAVITY_JUMP) / (GRAVITY_FALL - GRAVITY_JUMP) * 2 - 1;
    o[20] = n->walled ? 1 : -1;
    o[21] = n->floor_normalized_x;
    o[22] = n->ceiling_normalized_x;
    o[23] = n->ceiling_normalized_y;
    o[24] = (n->applied_drag - DRAG_SLOW) / (DRAG_REGULAR - DRAG_SLOW) * 2 - 1;
    o[25] = (n->applied_friction - FRICTION_GROUND_SLOW) / (FRICTION_GROUND - FRICTION_GROUND_SLOW) * 2 - 1;
    o[26] = pymax(-1.0, pymin(1.0, (n->xspeed - n->xspeed_old) / MAX_HOR_SPEED));
    o[27] = pymax(-1.0, pymin(1.0, (n->yspeed - n->yspeed_old) / MAX_HOR_SPEED));
    o[28] = pymin(sqrt(SQ(n->xspeed) + SQ(n->yspeed)) / (MAX_HOR_SPEED * 1.5), 1.0) * 2 - 1;
    o[29] = pymin(n->frames_airborne / 60.0, 1.0) * 2 - 1;
    o[30] = pymin(n->jump_duration / (double)MAX_JUMP_DURATION, 1.0) * 2 - 1;
    o[31] = pymin(n->state_change_frame / 30.0, 1.0) * 2 - 1;
    double kinetic_energy = 0.5 * (SQ(n->xspeed) + SQ(n->yspeed));
    o[32] = pymin(kinetic_energy / SQ(MAX_HOR_SPEED), 1.0) * 2 - 1;
    o[33] = (n->ypos / 600.0) * 2 - 1;
    double fm = sqrt(SQ(n->applied_gravity) + SQ(!n->airborn ? GROUND_ACCEL : AIR_ACCEL));
    o[34] = pymin(fm / 0.1, 1.0) * 2 - 1;
    double prev_kinetic = 0.5 * (SQ(n->xspeed_old) + SQ(n->yspeed_old));
    double rate = (kinetic_energy - prev_kinetic) / pymax(kinetic_energy + 0.01, 0.01);
    o[35] = pymax(-1.0, pymin(1.0, rate));
    o[36] = pymin(n->floor_count / 5.0, 1.0) * 2 - 1;
    o[37] = pymin(n->wall_count / 3.0, 1.0) * 2 - 1;
    o[38] = atan2(n->floor_normalized_y, n->floor_normalized_x) / M_PI;
    o[39] = n->walled ? n->wall_normal : 0.0;
}

/* ninja.py:628-839 with path-direction masking inert (sim._path_straightness_direction unset) */
int osim_action_mask(const OSim *S)
{
    const Ninja *n = &S->nj;
    int mask = 0x3f;
    int has_active_buffer = (-1 < n->jump_buffer && n->jump_buffer < 5) || (-1 < n->floor_buffer && n->floor_buffer < 5) ||
                            (-1 < n->wall_buffer && n->wall_buffer < 5) || (-1 < n->launch_pad_buffer && n->launch_pad_buffer < 4);
    int jump_useless = 0;
    if (n->airborn && n->state != 3) {
        if (n->jump_input != 0) jump_useless = !has_active_buffer;
    }
    if (jump_useless) mask &= ~(8 | 16 | 32);
    if (n->walled) {
        if (!n->airborn) {
            if (n->wall_normal > 0) mask &= ~2; else if (n->wall_normal < 0) mask &= ~4;
        } else if (n->state == 5) {
            if (n->wall_normal > 0) mask &= ~2; else if (n->wall_normal < 0) mask &= ~4;
        } else {
            int would = n->yspeed >= 0;
            if (n->wall_normal > 0) { if (!would) mask &= ~2; }
            else if (n->wall_normal < 0) { if (!would) mask &= ~4; }
        }
    }
    if (!mask) mask = 1;
    return mask;
}

/* spatial_context (112 f32): NppEnvironment._compute_spatial_context (gym_environment/npp_environment.py:2318-2360)
 * = compute_local_tile_grid (gym_environment/spatial_context.py:113-176) on the INNER 23 x 42 tile array indexed with
 * WORLD tile coordinates (the reference's off-by-one, base_environment.py:3346-3354) + compute_mine_overlay_from_entities
 * (:309-367, :370-508) with its position cache (recomputed only after the ninja moved >= 12 px since the last compute). */
static float sc_clipf(double v) { return (float)(v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v)); }

void osim_spatial_context(OSim *S, float *out)
{
    static const float CAT[38] = {0, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3,
                                  3, 3, 3, 3, 3, 3, 3, 3, 0, 0, 0, 0};
    const Ninja *n = &S->nj;
    int col = (int)floor(n->xpos / 24), row = (int)floor(n->ypos / 24);   /* int(x // 24) */
    for (int gr = 0; gr < 8; gr++)
        for (int gc = 0; gc < 8; gc++) {
            int r = row - 4 + gr, c = col - 4 + gc;
            int t = 1;                                   /* out of bounds -> solid */
            if (r >= 0 && r < 23 && c >= 0 && c < 42) {
                t = S->tiles[c + 1][r + 1];              /* inner array element [r][c] */
                t = t < 0 ? 0 : (t > 37 ? 37 : t);
            }
            out[gr * 8 + gc] = CAT[t] / 4.0f;
        }
    if (S->sc_valid) {
        double dx = n->xpos - S->sc_lx, dy = n->ypos - S->sc_ly;
        if (dx * dx + dy * dy < 144.0) { memcpy(out + 64, S->sc_overlay, sizeof(S->sc_overlay)); return; }
    }
    /* 8 nearest of entity_dic[1] then entity_dic[21], stable sort by distance */
    double bd[8]; const Entity *be[8]; int nb = 0;
    for (int i = 0; i < S->ndic; i++) {
        const Entity *e = S->dic_order[i];
        if (e->kind != K_MINE) continue;
        double dx = e->x - n->xpos, dy = e->y - n->ypos;
        double dist = POWHALF(dx * dx + dy * dy);
        int pos = nb;
        while (pos > 0 && dist < bd[pos - 1]) pos--;
        if (pos >= 8) continue;
        int last = nb < 8 ? nb : 7;
        for (int k = last; k > pos; k--) { bd[k] = bd[k - 1]; be[k] = be[k - 1]; }
        bd[pos] = dist; be[pos] = e;
        if (nb < 8) nb++;
    }
    float *ov = S->sc_overlay;
    memset(ov, 0, sizeof(S->sc_overlay));
    for (int k = 0; k < nb; k++) {
        const Entity *e = be[k];
        double dx = e->x - n->xpos, dy = e->y - n->ypos, dist = bd[k];
        double vd = 0.0, dr = 0.0;
        if (dist > 1e-6) {
            double dirx = dx / dist, diry = dy / dist;
            vd = (n->xspeed * dirx + n->yspeed * diry) / MAX_HOR_SPEED;
            dr = -vd;
        }
        ov[6 * k + 0] = sc_clipf(dx / 1056.0);
        ov[6 * k + 1] = sc_clipf(dy / 600.0);
        ov[6 * k + 2] = e->state == 1 ? 1.0f : (e->state == 2 ? 0.0f : -1.0f);
        ov[6 * k + 3] = (float)(e->state == 1 ? 3.5 / 5.0 : (e->state == 2 ? 4.5 / 5.0 : 4.0 / 5.0));
        ov[6 * k + 4] = sc_clipf(vd);
        ov[6 * k + 5] = sc_clipf(dr);
    }
    S->sc_valid = 1; S->sc_lx = n->xpos; S->sc_ly = n->ypos;
    memcpy(out + 64, ov, sizeof(S->sc_overlay));
}

/* ---- dumps used by the tests ------------------------------------------------------- */
int osim_frame(const OSim *S) { return S->frame; }

void osim_get_core(const OSim *S, double *f, int *d)
{
    const Ninja *n = &S->nj;
    f[0] = n->xpos; f[1] = n->ypos; f[2] = n->xspeed; f[3] = n->yspeed;
    f[4] = n->floor_normalized_x; f[5] = n->floor_normalized_y;
    f[6] = n->ceiling_normalized_x; f[7] = n->ceiling_normalized_y;
    f[8] = n->xspeed_old; f[9] = n->yspeed_old; f[10] = n->applied_gravity; f[11] = n->applied_drag;
    int sw_active = 2;
    for (int i = S->nents - 1; i >= 0; i--)
        if (S->ents[i].kind == K_SWITCH) { sw_active = S->ents[i].active; break; }
    d[0] = n->state; d[1] = n->airborn; d[2] = n->walled; d[3] = (int)n->wall_normal + 1;
    d[4] = n->jump_buffer + 1; d[5] = n->floor_buffer + 1; d[6] = n->wall_buffer + 1; d[7] = n->launch_pad_buffer + 1;
    d[8] = n->floor_count; d[9] = n->ceiling_count; d[10] = n->jump_duration;
    d[11] = n->applied_gravity > 0.05; d[12] = n->applied_drag > 0.95; d[13] = sw_active;
    d[14] = n->gold_collected; d[15] = n->doors_opened; d[16] = n->frames_airborne; d[17] = n->state_change_frame;
    d[18] = n->airborn_old; d[19] = n->jump_input_old; d[20] = n->death_cause; d[21] = n->terminal_impact;
}

/* rows of 8 int16: cell x, cell y, kind, then x1,y1,x2,y2,oriented | cx,cy,hor,ver,convex */
int osim_dump_csr(const OSim *S, int16_t *out, int max_rows)
{
    int r = 0;
    for (int x = 0; x < GW; x++)
        for (int y = 0; y < GH; y++) {
            const IndexCell *c = &S->index[x][y];
            for (int k = 0; k < c->n; k++) {
                if (r >= max_rows) return -1;
                const Seg *s = c->s[k];
                int16_t *o = out + 8 * r++;
                o[0] = x; o[1] = y; o[2] = s->kind;
                if (s->kind == 0) { o[3] = s->x1; o[4] = s->y1; o[5] = s->x2; o[6] = s->y2; o[7] = s->oriented; }
                else { o[3] = s->cx; o[4] = s->cy; o[5] = s->hor; o[6] = s->ver; o[7] = s->convex; }
            }
        }
    return r;
}

/* rows of 8 doubles: dic key, type, x, y, cell x, cell y, state (-1 if none), active */
int osim_dump_entities(const OSim *S, double *out, int max_rows)
{
    int r = 0;
    for (int i = 0; i < S->ndic; i++) {
        const Entity *e = S->dic_order[i];
        if (r >= max_rows) return -1;
        double *o = out + 8 * r++;
        o[0] = e->dic_key; o[1] = e->type; o[2] = e->x; o[3] = e->y; o[4] = e->cx; o[5] = e->cy;
        o[6] = e->kind == K_MINE ? e->state : -1; o[7] = e->active;
    }
    return r;
}

/* what the reference's entity layer would draw (entity_renderer.py:57-150), one row of 13 doubles per entity in entity_dic
 * order: type, x, y, active, state (mines) / switch_hit (exit door), closed (doors), normal x, normal y (oriented kinds),
 * door stroke x1, y1, x2, y2 (0 when the entity has no door segment), draw shape (0 disc of `radius`, 1 oriented stroke,
 * 2 square) with the radius / semi side in the last column packed as shape * 1000 + size */
int osim_dump_draw(const OSim *S, double *out, int max)
{
    int r = 0;
    for (int i = 0; i < S->ndic && r < max; i++) {
        const Entity *e = S->dic_order[i];
        double *o = out + 13 * r++;
        memset(o, 0, sizeof(double) * 13);
        o[0] = e->type; o[1] = e->x; o[2] = e->y; o[3] = e->active;
        o[4] = e->kind == K_EXIT ? e->switch_hit : e->state;
        double size = 0;
        int shape = 0;
        switch (e->kind) {
        case K_MINE: size = e->radius; break;
        case K_GOLD: size = 6; break;
        case K_EXIT: size = 12; break;
        case K_SWITCH: size = 6; break;
        case K_LOCKED: case K_DOOR_TRAP: size = 5; break;
        case K_DOOR_REG: size = 10; break;
        case K_LAUNCH: size = 6; shape = 1; break;
        case K_ONEWAY: size = 12; shape = 1; break;
        case K_DRONE: size = e->drone_radius; break;
        case K_BOUNCE: case K_THWUMP: size = 9; shape = 2; break;
        case K_BOOST: size = 6; break;
        case K_BALL: size = 5; break;
        case K_SHOVE: size = 8; break;   /* hasattr(entity, "RADIUS") wins over SEMI_SIDE (entity_renderer.py:178-185) */
        }
        o[12] = shape * 1000 + size;
        if (e->kind == K_LAUNCH || e->kind == K_ONEWAY) { o[6] = e->nx; o[7] = e->ny; }
        if (e->kind == K_LOCKED || e->kind == K_DOOR_REG || e->kind == K_DOOR_TRAP) {
            o[5] = e->closed;
            /* entity_door_base.py:70-85 */
            if (e->is_vertical) { o[8] = e->door_x; o[9] = e->door_y - 12; o[10] = e->door_x; o[11] = e->door_y + 12; }
            else { o[8] = e->door_x - 12; o[9] = e->door_y; o[10] = e->door_x + 12; o[11] = e->door_y; }
        }
    }
    return r;
}

/* live grid edges (tiles + doors), x-major [89][51]; nonzero = blocked */
void osim_dump_edges(const OSim *S, int *hor, int *ver)
{
    for (int x = 0; x < 89; x++)
        for (int y = 0; y < 51; y++) { hor[x * 51 + y] = S->hor_edge[x][y]; ver[x * 51 + y] = S->ver_edge[x][y]; }
}

int osim_entity_states(const OSim *S, int *out, int max)
{
    int r = 0;
    for (int i = 0; i < S->nents && r < max; i++) {
        const Entity *e = &S->ents[i];
        out[r++] = e->kind == K_MINE ? e->state : (e->kind == K_EXIT ? e->switch_hit : e->active);
    }
    return r;
}

/* per-entity state in entity_dic order (tests/golden/make_golden_fast.py ent_states): mines -> state, exit door ->
 * switch_hit, doors -> closed + 2 * active, others -> active */
int osim_entity_states_dic(const OSim *S, int *out, int max)
{
    int r = 0;
    for (int i = 0; i < S->ndic && r < max; i++) {
        const Entity *e = S->dic_order[i];
        if (e->kind == K_MINE) out[r++] = e->state;
        else if (e->kind == K_EXIT) out[r++] = e->switch_hit;
        else if (e->kind == K_DOOR_REG || e->kind == K_LOCKED || e->kind == K_DOOR_TRAP) out[r++] = e->closed + 2 * e->active;
        else out[r++] = e->active;
    }
    return r;
}

/* ------------------------------------------------------------------------------------
 * Batched driver used by bench.py's cpu_baseline leg ("port") and by parity tests that need
 * many envs: env-step = frame_skip ticks with early stop on win/death
 * (gym_environment/base_environment.py:535-609) and reset on termination.
 * ---------------------------------------------------------------------------------- */
static const int ACT_H[6] = {0, -1, 1, 0, -1, 1};
static const int ACT_J[6] = {0, 0, 0, 1, 1, 1};

/* returns ticks executed; flags bit0 won, bit1 dead */
int osim_env_step(OSim *S, int action, int frame_skip, int *flags)
{
    int h = ACT_H[action], j = ACT_J[action], k = 0;
    *flags = 0;
    for (int i = 0; i < frame_skip; i++) {
        osim_tick(S, h, j);
        k++;
        if (S->nj.state == 8) { *flags = 1; break; }
        if (S->nj.state == 6 || S->nj.state == 7) { *flags = 2; break; }
    }
    return k;
}

/* sims[n_envs]; actions[n_steps][n_envs]; truncation at max_frames; auto reset. Returns total ticks.
 * Parallel over envs when built with OpenMP. */
long long osim_run_batch(OSim **sims, int n_envs, const uint8_t *actions, int n_steps, int frame_skip, int max_frames, int threads)
{
    long long total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : total)
#endif
    for (int e = 0; e < n_envs; e++) {
        OSim *S = sims[e];
        for (int s = 0; s < n_steps; s++) {
            int fl;
            total += osim_env_step(S, actions[(size_t)s * n_envs + e] % 6, frame_skip, &fl);
            if (fl || S->frame >= max_frames) osim_reset(S);
        }
    }
    (void)threads;
    return total;
}
