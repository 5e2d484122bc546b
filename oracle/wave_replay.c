/*
 * wave_replay.c -- CPU replay of the trace kernel's 64-lane rounds (tools/wave_replay.py).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY, like everything under oracle/: nothing in the product links or calls it.
 *
 * What it answers: the oracle's walk counters are per LANE, the kernel pays per WAVE -- a node trip costs
 * its ~72 instructions whether 60 lanes step in it or 6 -- so a scheme that shortens the walk of one class of ray segments
 * (those that start on a sphere inside the tree, on the peeled ground sphere, at the camera) only pays if the number of
 * trips per ROUND falls.  This file replays rounds of 64 lanes with the kernel's own rules (csrc/rtmi_trace_kernel.h):
 * persistent lanes taking chunks of samples of the 64 pixels of one 8x8 tile from a per-wave pool, FETCH -> GEN -> BEGIN ->
 * walk -> SHADE, per trip the vote between a node step and a leaf step, the loop left when `wait_thresh` lanes wait for
 * shading, stragglers carried into the next round -- on the oracle's arithmetic (same segments, same boxes, same hits as
 * rt_oracle.c's instrumented walk) and counts trips per round under walk-start variants:
 *   bit 0  segments that start on a tree sphere begin in that sphere's own leaf (its even-depth ancestor-or-self) with the
 *          WAY pre-loaded on the stack: one record per two levels holding the boxes of the two siblings hanging off the path
 *          there, in the node format (two boxes, two child references), so the unchanged node step tests two levels a trip
 *   bit 1  segments that start on a peeled sphere (the ground) begin at the deepest even-depth node whose box contains the
 *          cell of a 3-d grid over the tree's box that holds the origin, way pre-loaded likewise
 *   bit 2  camera rays begin at their 8x8 tile's ENTRY: the lowest common ancestor of the leaves of every sphere the tile's
 *          beam (lens disk x tile rectangle on the focus plane, grown by a margin) can meet -- nothing else can be hit, so no
 *          way is needed; a tile whose beam meets no tree sphere does not walk at all.  The replay checks every such walk
 *          against the full one (out[20] counts differing hits: must be 0).
 *   bit 4  units handed out costliest tile first (a 2-spp probe of the sampled tiles), as the library does
 * Any start node is exact as long as every subtree hanging off the path above it is tested (the closest hit
 * does not depend on the visiting order); the replay asserts it anyway by comparing each finished walk with the plain one.
 */
#include "rt_oracle.c"

#define RP_END 0x7fffffffu
#define RP_LEAF(r) (((r) & 0x80000000u) != 0u)
enum { RP_FETCH = 0, RP_GEN, RP_BEGIN, RP_TRAV, RP_SHADE, RP_DONE };

typedef struct {
    int phase, cls;
    uint32_t px, py, tile, s, s_end, depth_left;
    orc_rng rng;
    ray_t ray;
    uint32_t origin_obj;
    uint32_t cur, stack[96];
    int sp;
    float best_t;
    uint32_t best;
    float inv[3], oinv[3], ainv[3], pinv[3];
    uint32_t seg_trips;
    int checked; /* this segment did not start at the root: compare with the plain walk */
} rp_lane;

typedef struct {
    const scene_t* sc;
    uint32_t walk_root, pre[4], n_pre;
    int no_walk;
    uint32_t n_nodes, n_el;             /* elements: internal nodes [0, n_nodes), then leaves */
    uint32_t *el_ref, *el_parent, *el_depth, *el_side;
    uint32_t* slot_leaf_el;             /* slot -> element of its leaf (~0u: a peeled leaf) */
    uint32_t* obj_slot;
    orc_bvh_node* vnodes;               /* way records, referenced as n_nodes + index */
    uint32_t n_v, cap_v;
    uint32_t *way_of_el;                /* virtual reference of the way record of an even-depth element, ~0u none yet */
    /* 3-d grid of start elements over the tree's box */
    float glo[3], gcell[3];
    uint32_t gdim[3];
    uint32_t* grid_el;
    int reversed;
} rp_tree;

static uint32_t rp_el_of_ref(const rp_tree* T, uint32_t ref) {
    if (!RP_LEAF(ref)) return ref;
    return T->slot_leaf_el[ref & 0x00ffffffu];
}

static void rp_build_tree(rp_tree* T, const scene_t* sc) {
    memset(T, 0, sizeof(*T));
    T->sc = sc;
    T->n_nodes = sc->n_nodes;
    uint32_t at = 0;
    if (sc->n_nodes != 0) {
        while (!RP_LEAF(at) && T->n_pre < 4u) {
            const uint32_t c0 = sc->nodes[at].child[0], c1 = sc->nodes[at].child[1];
            const int l0 = RP_LEAF(c0), l1 = RP_LEAF(c1);
            if (l0 && l1 && T->n_pre + 2u <= 4u) { T->pre[T->n_pre++] = c0; T->pre[T->n_pre++] = c1; T->no_walk = 1; break; }
            if (l0 == l1) break;
            T->pre[T->n_pre++] = l0 ? c0 : c1;
            at = l0 ? c1 : c0;
        }
    }
    T->walk_root = at;
    const uint32_t max_el = sc->n_nodes * 2u + 2u;
    T->el_ref = calloc(max_el, 4); T->el_parent = calloc(max_el, 4); T->el_depth = calloc(max_el, 4); T->el_side = calloc(max_el, 4);
    T->way_of_el = malloc(max_el * 4u); memset(T->way_of_el, 0xff, max_el * 4u);
    T->slot_leaf_el = malloc((sc->n_slots + 1u) * 4u); memset(T->slot_leaf_el, 0xff, (sc->n_slots + 1u) * 4u);
    T->obj_slot = malloc((sc->n_objs + 1u) * 4u); memset(T->obj_slot, 0xff, (sc->n_objs + 1u) * 4u);
    for (uint32_t q = 0; q < sc->n_slots; ++q) T->obj_slot[sc->slots[q]] = q;
    T->n_el = sc->n_nodes;
    if (T->no_walk || RP_LEAF(at)) return;
    /* depth-first from the walk's root */
    uint32_t* st = malloc(max_el * 4u);
    int sp = 0;
    st[sp++] = at;
    T->el_parent[at] = 0xffffffffu; T->el_depth[at] = 0; T->el_ref[at] = at;
    while (sp) {
        const uint32_t n = st[--sp];
        for (uint32_t k = 0; k < 2; ++k) {
            const uint32_t c = sc->nodes[n].child[k];
            uint32_t e;
            if (RP_LEAF(c)) {
                e = T->n_el++;
                const uint32_t first = c & 0x00ffffffu, cnt = (c >> 24) & 0x7fu;
                for (uint32_t s = 0; s < cnt; ++s) T->slot_leaf_el[first + s] = e;
            } else {
                e = c;
                st[sp++] = c;
            }
            T->el_ref[e] = c; T->el_parent[e] = n; T->el_depth[e] = T->el_depth[n] + 1u; T->el_side[e] = k;
        }
    }
    free(st);
}

/* the way record of even-depth element e (depth >= 2): boxes of the sibling of its parent and of its own sibling */
static uint32_t rp_way(rp_tree* T, uint32_t e) {
    if (T->way_of_el[e] != 0xffffffffu) return T->way_of_el[e];
    if (T->n_v == T->cap_v) { T->cap_v = T->cap_v ? T->cap_v * 2u : 256u; T->vnodes = realloc(T->vnodes, T->cap_v * sizeof(orc_bvh_node)); }
    const uint32_t p = T->el_parent[e], g = T->el_parent[p];
    const orc_bvh_node* np = &T->sc->nodes[p];
    const orc_bvh_node* ng = &T->sc->nodes[g];
    orc_bvh_node* v = &T->vnodes[T->n_v];
    memset(v, 0, sizeof(*v));
    const uint32_t sp_side = 1u - T->el_side[p], se_side = 1u - T->el_side[e];
    memcpy(v->ctr[0], ng->ctr[sp_side], 12); memcpy(v->half[0], ng->half[sp_side], 12); v->child[0] = ng->child[sp_side];
    memcpy(v->ctr[1], np->ctr[se_side], 12); memcpy(v->half[1], np->half[se_side], 12); v->child[1] = np->child[se_side];
    T->way_of_el[e] = T->n_nodes + T->n_v;
    return T->n_nodes + T->n_v++;
}

/* start the lane's walk at element e (moved to its parent when its depth is odd), the way above it pre-loaded: returns levels */
static int rp_start_at(rp_tree* T, rp_lane* L, uint32_t e) {
    if (T->el_depth[e] & 1u) e = T->el_parent[e];
    L->sp = 0;
    L->cur = T->el_ref[e];
    if (T->el_depth[e] == 0u) return 0;
    uint32_t chain[64];
    int m = 0;
    for (uint32_t x = e; T->el_depth[x] >= 2u; x = T->el_parent[T->el_parent[x]]) chain[m++] = rp_way(T, x);
    if (T->reversed) { for (int i = 0; i < m; ++i) L->stack[L->sp++] = chain[i]; } /* (the order a pointer chase from the leaf up would push in) */
    else for (int i = m - 1; i >= 0; --i) L->stack[L->sp++] = chain[i]; /* the top of the tree at the bottom of the stack */
    return m;
}

static const orc_bvh_node* rp_node(const rp_tree* T, uint32_t ref) {
    return ref < T->n_nodes ? &T->sc->nodes[ref] : &T->vnodes[ref - T->n_nodes];
}

static void rp_pop(rp_lane* L) { L->cur = L->sp ? L->stack[--L->sp] : RP_END; }

static void rp_leaf_spheres(const scene_t* sc, rp_lane* L, uint32_t ref) {
    const uint32_t first = ref & 0x00ffffffu, count = (ref >> 24) & 0x7fu;
    for (uint32_t s = 0; s < count; ++s) {
        const uint32_t oi = sc->slots[first + s];
        const float cand = sphere_candidate(&sc->objs[oi], &L->ray, 0.0001f);
        if (cand > 0.0001f && (cand < L->best_t || (cand == L->best_t && oi < L->best))) { L->best_t = cand; L->best = oi; }
    }
}

static void rp_node_step(const rp_tree* T, rp_lane* L) {
    const orc_bvh_node* nd = rp_node(T, L->cur);
    float tn[2], tf[2];
    for (int k = 0; k < 2; ++k) {
        float nmax = 0.0001f, fmin_ = L->best_t;
        for (int i = 0; i < 3; ++i) {
            const float tc = fmaf(nd->ctr[k][i], L->inv[i], L->oinv[i]);
            const float th = fmaf(nd->half[k][i], L->ainv[i], L->pinv[i]);
            nmax = fmaxf(nmax, tc - th);
            fmin_ = fminf(fmin_, tc + th);
        }
        tn[k] = nmax; tf[k] = fmin_;
    }
    const int h0 = tn[0] <= tf[0], h1 = tn[1] <= tf[1];
    if (h0 && h1) {
        const int swap = tn[1] < tn[0];
        L->stack[L->sp++] = nd->child[swap ? 0 : 1];
        L->cur = nd->child[swap ? 1 : 0];
    } else if (h0) L->cur = nd->child[0];
    else if (h1) L->cur = nd->child[1];
    else rp_pop(L);
    L->seg_trips++;
}

static void rp_begin(const rp_tree* T, rp_lane* L) {
    const scene_t* sc = T->sc;
    const float o[3] = {L->ray.o.x, L->ray.o.y, L->ray.o.z}, d[3] = {L->ray.d.x, L->ray.d.y, L->ray.d.z};
    for (int i = 0; i < 3; ++i) { L->inv[i] = 1.0f / d[i]; L->ainv[i] = fabsf(L->inv[i]); L->oinv[i] = -(o[i] * L->inv[i]); }
    L->best_t = INFINITY; L->best = 0xffffffffu;
    for (uint32_t q = 0; q < T->n_pre; ++q) rp_leaf_spheres(sc, L, T->pre[q]);
    const float pad = ray_pad(sc, &L->ray, L->inv, L->oinv, L->best_t);
    for (int i = 0; i < 3; ++i) L->pinv[i] = pad * L->ainv[i];
    L->sp = 0;
    L->cur = T->no_walk ? RP_END : T->walk_root;
    L->seg_trips = 0;
    L->checked = 0;
}

/* the plain walk from the root, for the cross-check */
static void rp_plain(const rp_tree* T, const rp_lane* L, float* bt, uint32_t* b) {
    rp_lane X = *L;
    rp_begin(T, &X);
    while (X.cur != RP_END) {
        if (RP_LEAF(X.cur)) { rp_leaf_spheres(T->sc, &X, X.cur); rp_pop(&X); }
        else rp_node_step(T, &X);
    }
    *bt = X.best_t; *b = X.best;
}

/* ---- camera tiles: which spheres can the beam of an 8x8 tile meet? ------------------------------------------------------ */
static double rp_gap(const double C[3], const double A0[3], const double AD[3], double t, double r_lens, double r_rect) {
    double s = 0.0;
    for (int i = 0; i < 3; ++i) { const double x = C[i] - (A0[i] + t * AD[i]); s += x * x; }
    return sqrt(s) - (fabs(1.0 - t) * r_lens + t * r_rect);
}
static double rp_min_gap(const double C[3], const double A0[3], const double AD[3], double lo, double hi, double r_lens, double r_rect) {
    for (int it = 0; it < 100; ++it) { /* the gap is convex on a piece where the beam radius is linear */
        const double a = lo + (hi - lo) / 3.0, b = hi - (hi - lo) / 3.0;
        if (rp_gap(C, A0, AD, a, r_lens, r_rect) < rp_gap(C, A0, AD, b, r_lens, r_rect)) hi = b; else lo = a;
    }
    return rp_gap(C, A0, AD, 0.5 * (lo + hi), r_lens, r_rect);
}
static uint32_t rp_tile_entry(const rp_tree* T, const orc_camera* cam, uint32_t tx, uint32_t ty, uint32_t* n_cand) {
    const scene_t* sc = T->sc;
    double A0[3], AD[3], du = 0, dv = 0, lu = 0, lv = 0;
    for (int i = 0; i < 3; ++i) {
        A0[i] = cam->cam_center[i];
        const double c1 = (double)cam->pixel00[i] + (double)cam->pixel_delta_u[i] * (8.0 * tx + 3.5) + (double)cam->pixel_delta_v[i] * (8.0 * ty + 3.5);
        AD[i] = c1 - A0[i];
        du += (double)cam->pixel_delta_u[i] * cam->pixel_delta_u[i]; dv += (double)cam->pixel_delta_v[i] * cam->pixel_delta_v[i];
        lu += (double)cam->defocus_disk_u[i] * cam->defocus_disk_u[i]; lv += (double)cam->defocus_disk_v[i] * cam->defocus_disk_v[i];
    }
    const double r_rect = 4.0 * (sqrt(du) + sqrt(dv)) * 1.001, r_lens = cam->defocus_angle <= 0.0f ? 0.0 : (sqrt(lu) + sqrt(lv)) * 1.001;
    const double alen = sqrt(AD[0] * AD[0] + AD[1] * AD[1] + AD[2] * AD[2]);
    uint32_t lca = 0xffffffffu;
    *n_cand = 0;
    for (uint32_t oi = 0; oi < sc->n_objs; ++oi) {
        const uint32_t slot = T->obj_slot[oi];
        if (slot == 0xffffffffu) continue;
        const uint32_t e = T->slot_leaf_el[slot];
        if (e == 0xffffffffu) continue; /* a peeled leaf: tested at set-up anyway */
        const double C[3] = {sc->objs[oi].center[0], sc->objs[oi].center[1], sc->objs[oi].center[2]};
        const double R = fabs((double)sc->objs[oi].radius);
        const double dist0 = sqrt((C[0] - A0[0]) * (C[0] - A0[0]) + (C[1] - A0[1]) * (C[1] - A0[1]) + (C[2] - A0[2]) * (C[2] - A0[2]));
        const double tmax = 2.0 + (dist0 + R) / (alen > 0 ? alen : 1.0) * 2.0;
        const double margin = R * 1e-3 + 1e-4 * (dist0 + R + 1.0);
        double g = rp_min_gap(C, A0, AD, 0.0, 1.0, r_lens, r_rect);
        const double g2 = rp_min_gap(C, A0, AD, 1.0, tmax, r_lens, r_rect);
        if (g2 < g) g = g2;
        if (g > R + margin) continue;
        ++*n_cand;
        if (lca == 0xffffffffu) { lca = e; continue; }
        uint32_t a = lca, b = e;
        while (a != b) { if (T->el_depth[a] >= T->el_depth[b]) a = T->el_parent[a]; else b = T->el_parent[b]; }
        lca = a;
    }
    return lca; /* ~0u: the beam meets no sphere of the tree */
}

/* ---- the replay -------------------------------------------------------------------------------------------------------- */
enum { RO_ROUNDS = 0, RO_SEGMENTS, RO_SAMPLES, RO_NODE_TRIPS, RO_NODE_LANES, RO_LEAF_TRIPS, RO_LEAF_LANES, RO_BEGIN_LANES,
       RO_GEN_ROUNDS, RO_GEN_LANES, RO_SHADE_HIT, RO_SHADE_SKY, RO_PRELOAD_ROUNDS, RO_PRELOAD_LEVELS_MAX, RO_PRELOAD_LANES,
       RO_WAY_RECORDS, RO_CLS_SEG0, RO_CLS_SEG1, RO_CLS_SEG2, RO_MISMATCH_WAY, RO_MISMATCH_CAM, RO_CLS_TRIPS0, RO_CLS_TRIPS1,
       RO_CLS_TRIPS2, RO_TILES, RO_TILES_NO_WALK, RO_TILE_CANDS, RO_ENTRY_DEPTH, RO_FETCH_ROUNDS, RO_CARRIED_LANES, RO_N };

int orc_wave_replay(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats, uint32_t n_mats,
                    const orc_bvh_node* nodes, uint32_t n_nodes, const uint32_t* slots, uint32_t n_slots, const float* pad_classes,
                    uint32_t n_classes, float pad_eps, float pad_floor, uint64_t seed, uint32_t tile_stride, uint32_t chunk,
                    uint32_t n_waves, uint32_t wait_thresh, uint32_t variant, uint64_t* out) {
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.objs = objs; sc.n_objs = n_objs; sc.mats = mats; sc.n_mats = n_mats;
    sc.nodes = nodes; sc.n_nodes = n_nodes; sc.slots = slots; sc.n_slots = n_slots;
    sc.pad_classes = pad_classes; sc.n_classes = n_classes; sc.pad_eps = pad_eps; sc.pad_floor = pad_floor;
    sc.pad_refine = pad_refine_pays(pad_classes, n_classes, pad_eps);
    rp_tree T;
    rp_build_tree(&T, &sc);
    T.reversed = (variant & 32u) != 0u;
    memset(out, 0, RO_N * sizeof(uint64_t));
    const uint32_t W = cam->img_width, H = cam->img_height, spp = cam->samples_per_pixel;
    const uint32_t tiles_x = (W + 7u) / 8u, tiles_y = (H + 7u) / 8u, n_tiles_all = tiles_x * tiles_y;
    if (tile_stride == 0) tile_stride = 1;
    if (chunk == 0 || chunk > spp) chunk = spp;
    const uint32_t n_chunks = (spp + chunk - 1u) / chunk;
    uint32_t n_tiles = 0;
    uint32_t* tiles = malloc((n_tiles_all / tile_stride + 2u) * 4u);
    for (uint32_t t = tile_stride / 2u; t < n_tiles_all; t += tile_stride) tiles[n_tiles++] = t;
    out[RO_TILES] = n_tiles;
    char* obj_pre = calloc(n_objs + 1u, 1);
    for (uint32_t q = 0; q < T.n_pre; ++q) {
        const uint32_t first = T.pre[q] & 0x00ffffffu, cnt = (T.pre[q] >> 24) & 0x7fu;
        for (uint32_t s = 0; s < cnt; ++s) obj_pre[slots[first + s]] = 1;
    }
    /* hand-out order: costliest tile first (segments of a 2-spp probe) */
    if (variant & 16u) {
        uint64_t* key = malloc(n_tiles * 8u);
        orc_rng rng;
        rng_init_counter(&rng, 0x636f7374ull);
        for (uint32_t i = 0; i < n_tiles; ++i) {
            orc_counters c;
            memset(&c, 0, sizeof(c));
            const uint32_t tx = tiles[i] % tiles_x, ty = tiles[i] / tiles_x;
            for (uint32_t j = 0; j < 64; ++j) {
                const uint32_t x = tx * 8u + (j & 7u), y = ty * 8u + (j >> 3);
                if (x >= W || y >= H) continue;
                for (uint32_t s = 0; s < 2u && s < spp; ++s) {
                    counter_begin(&rng, y * W + x, s);
                    const ray_t r = get_ray(cam, x, y, &rng);
                    (void)compute_color(&r, cam->maxdepth, &sc, &rng, &c);
                }
            }
            key[i] = (c.segments << 32) | (0xffffffffu - tiles[i]);
        }
        for (uint32_t i = 1; i < n_tiles; ++i) { /* insertion sort, descending (a few thousand tiles) */
            const uint64_t k = key[i];
            uint32_t j = i;
            while (j > 0 && key[j - 1] < k) { key[j] = key[j - 1]; --j; }
            key[j] = k;
        }
        for (uint32_t i = 0; i < n_tiles; ++i) tiles[i] = 0xffffffffu - (uint32_t)key[i];
        free(key);
    }
    /* camera entries per sampled tile */
    uint32_t* entry = NULL;
    if ((variant & 4u) && !T.no_walk && !RP_LEAF(T.walk_root)) {
        entry = malloc(n_tiles_all * 4u);
        memset(entry, 0xfe, n_tiles_all * 4u);
        for (uint32_t i = 0; i < n_tiles; ++i) {
            uint32_t nc = 0;
            const uint32_t e = rp_tile_entry(&T, cam, tiles[i] % tiles_x, tiles[i] / tiles_x, &nc);
            entry[tiles[i]] = e;
            out[RO_TILE_CANDS] += nc;
            if (e == 0xffffffffu) out[RO_TILES_NO_WALK]++; else out[RO_ENTRY_DEPTH] += T.el_depth[e];
        }
    }
    /* 3-d grid of start elements over the box of the walk's root */
    if ((variant & 2u) && !T.no_walk && !RP_LEAF(T.walk_root)) {
        const orc_bvh_node* r = &nodes[T.walk_root];
        float lo[3], hi[3];
        for (int i = 0; i < 3; ++i) {
            lo[i] = fminf(r->ctr[0][i] - r->half[0][i], r->ctr[1][i] - r->half[1][i]);
            hi[i] = fmaxf(r->ctr[0][i] + r->half[0][i], r->ctr[1][i] + r->half[1][i]);
        }
        static const uint32_t dims[4][3] = {{64, 4, 64}, {32, 2, 32}, {16, 1, 16}, {32, 1, 32}};
        for (int i = 0; i < 3; ++i) T.gdim[i] = dims[(variant >> 8) & 3u][i];
        for (int i = 0; i < 3; ++i) { T.glo[i] = lo[i]; T.gcell[i] = (hi[i] - lo[i]) / (float)T.gdim[i]; if (!(T.gcell[i] > 0)) T.gcell[i] = 1.0f; }
        const uint32_t nc = T.gdim[0] * T.gdim[1] * T.gdim[2];
        T.grid_el = malloc(nc * 4u);
        for (uint32_t c = 0; c < nc; ++c) {
            const uint32_t ix = c % T.gdim[0], iy = (c / T.gdim[0]) % T.gdim[1], iz = c / (T.gdim[0] * T.gdim[1]);
            const float p[3] = {T.glo[0] + (ix + 0.5f) * T.gcell[0], T.glo[1] + (iy + 0.5f) * T.gcell[1], T.glo[2] + (iz + 0.5f) * T.gcell[2]};
            /* descend while exactly one child's box, grown by half a cell, contains the cell's centre */
            uint32_t e = T.walk_root;
            for (;;) {
                if (e >= n_nodes) break;
                const orc_bvh_node* nd = &nodes[e];
                int in[2];
                for (int k = 0; k < 2; ++k) {
                    in[k] = 1;
                    for (int i = 0; i < 3; ++i) if (fabsf(p[i] - nd->ctr[k][i]) > nd->half[k][i] + 0.5f * T.gcell[i]) in[k] = 0;
                }
                if (in[0] == in[1]) break;
                e = rp_el_of_ref(&T, nd->child[in[0] ? 0 : 1]);
            }
            T.grid_el[c] = e;
        }
    }

    rp_lane* lanes = calloc(64, sizeof(rp_lane));
    for (uint32_t w = 0; w < n_waves; ++w) {
        for (int l = 0; l < 64; ++l) { memset(&lanes[l], 0, sizeof(rp_lane)); lanes[l].phase = RP_FETCH; lanes[l].cur = RP_END; rng_init_counter(&lanes[l].rng, seed); }
        uint64_t unit = w; /* this wave's next unit (a unit = the 64 pixels of one tile for one chunk); waves stride the unit space */
        const uint64_t n_units = (uint64_t)n_tiles * n_chunks;
        uint32_t pool_next = 0, pool_end = 0, pool_tile = 0, pool_s0 = 0;
        for (;;) {
            /* FETCH */
            int fetched = 0;
            for (int l = 0; l < 64; ++l) {
                rp_lane* L = &lanes[l];
                while (L->phase == RP_FETCH) {
                    if (pool_next == pool_end) {
                        if (unit >= n_units) { L->phase = RP_DONE; break; }
                        const uint64_t pos = unit / n_chunks;
                        pool_tile = tiles[pos];
                        pool_s0 = (uint32_t)(unit - pos * n_chunks) * chunk;
                        pool_next = 0; pool_end = 64;
                        unit += n_waves;
                    }
                    const uint32_t j = pool_next++;
                    const uint32_t x = (pool_tile % tiles_x) * 8u + (j & 7u), y = (pool_tile / tiles_x) * 8u + (j >> 3);
                    if (x >= W || y >= H) continue;
                    L->px = x; L->py = y; L->tile = pool_tile; L->s = pool_s0; L->s_end = pool_s0 + chunk < spp ? pool_s0 + chunk : spp;
                    L->phase = RP_GEN;
                    fetched = 1;
                }
            }
            if (fetched) out[RO_FETCH_ROUNDS]++;
            int alive = 0;
            for (int l = 0; l < 64; ++l) alive |= lanes[l].phase != RP_DONE;
            if (!alive) break;
            out[RO_ROUNDS]++;
            /* GEN */
            int n_gen = 0;
            for (int l = 0; l < 64; ++l) {
                rp_lane* L = &lanes[l];
                if (L->phase != RP_GEN) continue;
                counter_begin(&L->rng, L->py * W + L->px, L->s);
                L->ray = get_ray(cam, L->px, L->py, &L->rng);
                L->depth_left = cam->maxdepth;
                L->cls = 0; L->origin_obj = 0xffffffffu;
                L->phase = L->depth_left ? RP_BEGIN : RP_SHADE;
                if (!L->depth_left) L->best = 0xfffffffeu;
                ++n_gen;
            }
            if (n_gen) { out[RO_GEN_ROUNDS]++; out[RO_GEN_LANES] += n_gen; }
            /* BEGIN */
            int pre_max = 0, pre_lanes = 0;
            for (int l = 0; l < 64; ++l) {
                rp_lane* L = &lanes[l];
                if (L->phase != RP_BEGIN) continue;
                rp_begin(&T, L);
                out[RO_BEGIN_LANES]++; out[RO_SEGMENTS]++; out[RO_CLS_SEG0 + L->cls]++;
                L->phase = RP_TRAV;
                if (T.no_walk || RP_LEAF(T.walk_root)) continue;
                int m = -1;
                if (L->cls == 2 && (variant & 1u)) {
                    m = rp_start_at(&T, L, T.slot_leaf_el[T.obj_slot[L->origin_obj]]);
                } else if (L->cls == 1 && (variant & 2u)) {
                    int c[3], inside = 1;
                    const float o[3] = {L->ray.o.x, L->ray.o.y, L->ray.o.z};
                    for (int i = 0; i < 3; ++i) {
                        const float f = (o[i] - T.glo[i]) / T.gcell[i];
                        c[i] = (int)floorf(f);
                        if (i == 1) { if (c[i] < 0) c[i] = 0; if (c[i] >= (int)T.gdim[i]) c[i] = (int)T.gdim[i] - 1; } /* (the ground point sits on the box's floor) */
                        if (c[i] < 0 || c[i] >= (int)T.gdim[i]) inside = 0;
                    }
                    if (inside) m = rp_start_at(&T, L, T.grid_el[((uint32_t)c[2] * T.gdim[1] + (uint32_t)c[1]) * T.gdim[0] + (uint32_t)c[0]]);
                } else if (L->cls == 0 && entry) {
                    const uint32_t e = entry[L->tile];
                    if (e == 0xffffffffu) { L->cur = RP_END; L->checked = 2; }
                    else if (e != 0xfefefefeu) { L->sp = 0; L->cur = T.el_ref[e]; L->checked = 2; }
                }
                if (m >= 0) { L->checked = 1; pre_lanes++; if (m > pre_max) pre_max = m; }
            }
            if (pre_lanes) { out[RO_PRELOAD_ROUNDS]++; out[RO_PRELOAD_LEVELS_MAX] += (uint64_t)pre_max; out[RO_PRELOAD_LANES] += (uint64_t)pre_lanes; }
            /* walk */
            int n_trav = 0;
            for (int l = 0; l < 64; ++l) n_trav += lanes[l].phase == RP_TRAV || lanes[l].phase == RP_SHADE;
            const int floor_ = n_trav > (int)wait_thresh ? n_trav - (int)wait_thresh : 0;
            for (;;) {
                int n_node = 0, n_leaf = 0;
                for (int l = 0; l < 64; ++l) {
                    const rp_lane* L = &lanes[l];
                    if (L->phase != RP_TRAV || L->cur == RP_END) continue;
                    if (RP_LEAF(L->cur)) ++n_leaf; else ++n_node;
                }
                if (n_leaf + n_node <= floor_) break;
                if (n_leaf > n_node) {
                    out[RO_LEAF_TRIPS]++; out[RO_LEAF_LANES] += (uint64_t)n_leaf;
                    for (int l = 0; l < 64; ++l) {
                        rp_lane* L = &lanes[l];
                        if (L->phase == RP_TRAV && L->cur != RP_END && RP_LEAF(L->cur)) { rp_leaf_spheres(&sc, L, L->cur); rp_pop(L); }
                    }
                } else {
                    out[RO_NODE_TRIPS]++; out[RO_NODE_LANES] += (uint64_t)n_node;
                    for (int l = 0; l < 64; ++l) {
                        rp_lane* L = &lanes[l];
                        if (L->phase == RP_TRAV && L->cur != RP_END && !RP_LEAF(L->cur)) rp_node_step(&T, L);
                    }
                }
            }
            for (int l = 0; l < 64; ++l) {
                rp_lane* L = &lanes[l];
                if (L->phase == RP_TRAV && L->cur == RP_END) L->phase = RP_SHADE;
                else if (L->phase == RP_TRAV) out[RO_CARRIED_LANES]++;
            }
            /* SHADE */
            for (int l = 0; l < 64; ++l) {
                rp_lane* L = &lanes[l];
                if (L->phase != RP_SHADE) continue;
                int ended = 0;
                if (L->best == 0xfffffffeu) ended = 1;
                else {
                    out[RO_CLS_TRIPS0 + L->cls] += L->seg_trips;
                    if (L->checked) {
                        float bt; uint32_t b;
                        rp_plain(&T, L, &bt, &b);
                        if (b != L->best || (b != 0xffffffffu && bt != L->best_t)) out[L->checked == 1 ? RO_MISMATCH_WAY : RO_MISMATCH_CAM]++;
                    }
                    if (L->best != 0xffffffffu) {
                        out[RO_SHADE_HIT]++;
                        const orc_object* s = &objs[L->best];
                        hit_rec rec;
                        const v3 C = vld(s->center);
                        const v3 p = vadd(L->ray.o, vscale(L->ray.d, L->best_t));
                        const v3 outward = vdivs(vsub(p, C), s->radius);
                        rec.P = p; rec.T = (double)L->best_t; rec.material = s->material;
                        rec.front_face = vdot(L->ray.d, outward) < 0.0f;
                        rec.N = rec.front_face ? outward : vneg(outward);
                        v3 att; ray_t sca;
                        if (material_scatter(&mats[rec.material], &L->ray, &rec, &L->rng, &att, &sca, NULL)) {
                            if (--L->depth_left == 0) ended = 1;
                            else { L->ray = sca; L->origin_obj = L->best; L->cls = obj_pre[L->best] ? 1 : 2; L->phase = RP_BEGIN; }
                        } else ended = 1;
                    } else { out[RO_SHADE_SKY]++; ended = 1; }
                }
                if (ended) {
                    out[RO_SAMPLES]++;
                    L->s++;
                    L->phase = L->s >= L->s_end ? RP_FETCH : RP_GEN;
                    L->cur = RP_END;
                }
            }
        }
    }
    out[RO_WAY_RECORDS] = T.n_v;
    free(lanes); free(tiles); free(obj_pre); free(entry);
    free(T.el_ref); free(T.el_parent); free(T.el_depth); free(T.el_side); free(T.way_of_el); free(T.slot_leaf_el); free(T.obj_slot);
    free(T.vnodes); free(T.grid_el);
    return 0;
}
