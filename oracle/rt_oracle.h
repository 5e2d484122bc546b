/*
 * rt_oracle.h -- CPU oracle for the per-pixel path-tracing hot loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (raytracing.cpp_amd/, include/rtmi.h) never
 * links, imports or calls it.
 *
 * PARITY: the reference (adihodos/raytracing.cpp) ships no tests, no golden vectors and no fixtures for this path,
 * and its hot-path TUs cannot be built in this image (they need glm, tl::optional, strong_type and reflect-cpp, none
 * of which is vendored or installed; stand-in headers are not allowed).  The only outputs of the reference itself
 * that exist are the four RGBA8 pixels recorded in SURVEY.md section 8(c); the mt19937 path of this file reproduces
 * them bit for bit (tests/golden/reference_pixels.json).  Beyond those four vectors PARITY IS UNPINNED: this file is
 * a from-scratch restatement in plain C of the reference's arithmetic, each function citing the reference file:line
 * it follows; the glm operations are restated from glm's published definitions (pinned by the reference's
 * CMakeLists.txt:55 to g-truc/glm@bf71a834).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/camera.parameters.hpp:6-17 */
typedef struct orc_camera_params {
    float aspect_ratio;
    uint32_t image_width;
    uint16_t samples_per_pixel;
    uint16_t max_depth;
    float vertical_fov;
    float defocus_angle;
    float focus_distance;
    float lookfrom[3];
    float lookat[3];
    float world_up[3];
} orc_camera_params;

/* the 14 POD fields of RayTracingCore, src/ray.tracer.core.hpp:19-32 */
typedef struct orc_camera {
    uint32_t img_width;
    uint32_t img_height;
    float defocus_angle;
    float viewport_height;
    float viewport_width;
    uint16_t samples_per_pixel;
    uint16_t maxdepth;
    float pixels_sample_scale;
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float pixel00[3];
    float cam_center[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
} orc_camera;

/* HittableObject, src/ray.tracer.object.defs.hpp:30-57 : 24 bytes */
typedef struct orc_object {
    uint32_t kind; /* 0 = Sphere */
    float center[3];
    float radius;
    uint32_t material;
} orc_object;

/* Material, src/ray.tracer.material.defs.hpp:49-55 : 20 bytes.
 * kind 0 Lambertian {albedo}, 1 Metallic {albedo, fuzz}, 2 Dielectric {p[0] = refraction index} */
typedef struct orc_material {
    uint32_t kind;
    float p[4];
} orc_material;

/* WorldDefinition minus camera/objects, src/ray.tracer.core.cc:67-95 */
typedef struct orc_world_def {
    int32_t a_min, a_max, b_min, b_max;
    float center_offset[3];
    float center_dist_treshold;
    float diffuse_material_treshold;
    float metal_material_treshold;
} orc_world_def;

typedef struct orc_counters {
    uint64_t samples;
    uint64_t segments;     /* compute_color calls that ran an intersection query */
    uint64_t sphere_tests; /* HittableObject_Sphere::intersects calls */
    uint64_t node_tests;   /* BVH box tests (0 on the reference's linear scan) */
    uint64_t rng_doubles;
    uint64_t hit_lambertian, hit_metallic, hit_dielectric;
    uint64_t end_sky, end_depth, end_absorbed;
    /* instrumented BVH walk only (tools/descent_score.py): segments, node trips and leaf trips by where the segment starts --
     * [0] the camera, [1] a sphere peeled off the top of the tree (the ground), [2] a sphere inside the tree; for class 2 the
     * internal nodes on the way from the walk's start to the origin sphere's own leaf (the origin lies inside every one of
     * those boxes: the descent no tree can avoid) and how many of their sibling boxes the ray hits */
    uint64_t seg_class[3], trips_class[3], leaf_trips_class[3];
    uint64_t descent_levels, descent_sibling_hits;
} orc_counters;

/* flat BVH node used only by the instrumented BVH walk (build-side extension, see rt_oracle.c) */
typedef struct orc_bvh_node {
    float ctr[2][3];
    float half[2][3];
    uint32_t child[2]; /* bit31 set: leaf, bits 0..23 first sphere slot, bits 24..30 count */
    float inv2rmin[2];
} orc_bvh_node;

enum { ORC_RNG_MT19937 = 0, ORC_RNG_COUNTER = 1 };

/* --- RNG ---------------------------------------------------------------- */
typedef struct orc_rng orc_rng;
orc_rng* orc_rng_new_mt(uint32_t seed);
void orc_rng_free(orc_rng*);
double orc_rng_double(orc_rng*);
uint32_t orc_mt_next_u32(orc_rng*);
/* the two key words of a counter stream: a bijective mix (splitmix64 finaliser) of the caller's 64-bit seed */
uint64_t orc_mix_seed(uint64_t seed);
/* k-th double of the counter stream of (seed, pixel, sample) */
double orc_counter_double(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox4x32(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_pcg4d(const uint32_t in[4], uint32_t out[4]);
/* block function of the counter stream: 10 / 7 = Philox4x32 rounds, 0 = pcg4d (the product's RTMI_RNG) */
#ifndef ORC_COUNTER_RNG_DEFAULT
#define ORC_COUNTER_RNG_DEFAULT 0
#endif
void orc_set_counter_rng(int kind);
int orc_get_counter_rng(void);
/* box pad of the instrumented BVH walk: 0 = per class from its farthest centre (rounds 1-3), 1 = bounded by the reach of
 * the segment (the product's, round 4), 2 = none (not exact: lower bound of the walk's work, measurements only) */
void orc_set_pad_mode(int mode);
int orc_get_pad_mode(void);
/* camera-ray entries of the instrumented BVH walk (the product's rtmi_tile_entries_build table, rtmi_bvh_node::child format,
 * row-major over gtx tiles per row; NULL = every walk from the root).  The table must outlive the renders that use it. */
void orc_set_tile_entries(const uint32_t* entries, uint32_t gtx);
/* walk starts of scattered rays (the product's rtmi_scene_get_walk_starts records, 16 words per slot; NULL = from the root);
 * `slots` / n_slots / n_objs as passed to orc_render_rect_counter_bvh.  The table must outlive the renders that use it. */
void orc_set_walk_starts(const uint32_t* records, const uint32_t* slots, uint32_t n_slots, uint32_t n_objs);
void orc_counter_block(uint32_t blk, uint32_t sample, uint32_t pixel, const uint32_t key[2], uint32_t out[4]);

/* --- host-side setup ------------------------------------------------------ */
void orc_camera_setup(const orc_camera_params* p, orc_camera* out);
/* returns number of objects written (== number of materials); fixed objects first */
uint32_t orc_make_world_spheres(const orc_world_def* wd, const orc_object* fixed_objs, const orc_material* fixed_mats,
                                uint32_t n_fixed, uint32_t mt_seed, int metal_args_right_to_left, orc_object* objs_out,
                                orc_material* mats_out, uint32_t capacity);
uint32_t orc_pack_rgba(const float rgb[3]);

/* --- unit entry points (edge-case vectors) ------------------------------ */
/* returns 1 on hit; rec = {P[3], N[3], t, front_face} */
int orc_sphere_intersect(const float center[3], float radius, const float origin[3], const float dir[3], double tmin,
                         double tmax, float rec_out[8]);
int orc_world_intersect(const orc_object* objs, uint32_t n, const float origin[3], const float dir[3], float rec_out[8],
                        uint32_t* index_out);
/* returns 1 when scattered; out = {attenuation[3], origin[3], dir[3]} */
int orc_scatter(const orc_material* m, const float ray_o[3], const float ray_d[3], const float P[3], const float N[3],
                int front_face, orc_rng* rng, float out[9]);

/* --- rendering ------------------------------------------------------------ */
/* Reference semantics: ONE mt19937 generator shared by every pixel, visited in list order
 * (RandomNumberGenerator per worker thread, src/main.cc:437). xy = {x0,y0,x1,y1,...}. */
int orc_render_pixels_mt(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats,
                         uint32_t n_mats, uint32_t mt_seed, const uint32_t* xy, uint32_t n_pixels, float* rgb_out,
                         uint32_t* rgba_out, orc_counters* ctr);

/* Counter-RNG semantics: stream keyed by (seed, pixel = y*W+x, sample). Renders rows [y0,y1) x cols [x0,x1)
 * into dense out buffers of (y1-y0)*(x1-x0) pixels. nthreads<=1: single thread. bvh may be NULL (linear scan). */
int orc_render_rect_counter(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats,
                            uint32_t n_mats, uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                            float* rgb_out, uint32_t* rgba_out, orc_counters* ctr, int nthreads);

/* Instrumented BVH walk (build-side extension): same image as the linear scan, counts node/sphere tests.
 * slots[i] = object index stored at leaf slot i; bounds used for the per-ray pad are in pad_classes
 * (n_classes x 8 floats: lo[3], hi[3], inv2rmin, unused). */
int orc_render_rect_counter_bvh(const orc_camera* cam, const orc_object* objs, uint32_t n_objs,
                                const orc_material* mats, uint32_t n_mats, const orc_bvh_node* nodes, uint32_t n_nodes,
                                const uint32_t* slots, uint32_t n_slots, const float* pad_classes, uint32_t n_classes,
                                float pad_eps, float pad_floor, uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1,
                                uint32_t y1, float* rgb_out, uint32_t* rgba_out, orc_counters* ctr, int nthreads);

/* CPU baseline: the reference's job system shape (src/main.cc:608-633): nthreads workers, shuffled 8x8 tiles from
 * a shared queue, one mt19937 per worker; renders every `stride`-th pixel in x and y. Returns wall seconds. */
double orc_bench_mt(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats,
                    uint32_t n_mats, uint32_t mt_seed, uint32_t stride, int nthreads, uint32_t* rgba_out,
                    uint64_t* samples_out);

#ifdef __cplusplus
}
#endif
#endif
