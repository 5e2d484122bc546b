/*
 * rtmi.h -- C-ABI of the MI355X-native path-tracing core (librtmi.so).
 *
 * Drop-in boundary for ONE hot path of adihodos/raytracing.cpp: the per-pixel path tracer
 *   RayTracingCore::raytrace_pixel -> get_ray / compute_color -> HittableObject_Collection::intersects
 *   -> HittableObject_Sphere::intersects -> Material::scatter
 * (reference src/ray.tracer.core.cc:218-265, src/ray.tracer.object.defs.cc:37-81,
 *  src/ray.tracer.material.defs.cc:31-109).
 *
 * The reference has no FFI; its seam is the C++ call `RGBAColor RayTracingCore::raytrace_pixel(x, y, rng)`
 * (src/ray.tracer.core.hpp:41) made per pixel of an 8x8 tile by
 * RayTracingWorker::process_tracing_work_package (src/main.cc:507-519), and the construction seam
 * `RayTracingCore::default_setup()` (src/ray.tracer.core.hpp:36, called at src/main.cc:604).  A per-pixel
 * GPU call is meaningless, so the boundary sits one level up, at row-block granularity, with the reference's
 * own data model: the 14 POD fields of RayTracingCore, the 24-byte HittableObject and the 20-byte Material
 * records, RGBAColor's 0xAABBGGRR packing.  Plain pointers and sizes only; never throws, never aborts.
 *
 * All functions return RTMI_OK (0) or a negative rtmi_status; rtmi_last_error() gives the thread-local text.
 * The library needs a HIP device (gfx950); there is no CPU fallback.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtmi_status {
    RTMI_OK = 0,
    RTMI_ERR_BAD_ARG = -1,   /* null pointer, out-of-range rows, bad material handle, unknown kind */
    RTMI_ERR_HIP = -2,       /* a HIP runtime call failed (no device, launch failure, ...) */
    RTMI_ERR_OOM = -3,       /* host or device allocation failed */
    RTMI_ERR_UNSUPPORTED = -4, /* scene does not fit the selected kernel (e.g. LDS budget) */
    RTMI_ERR_RCCL = -5,       /* an RCCL call of the multi-device frame failed (or librccl could not be loaded) */
    RTMI_ERR_INTERNAL = -6    /* an unexpected C++ exception was caught at the boundary (never propagated) */
} rtmi_status;

/* CameraParameters, reference src/camera.parameters.hpp:6-17 (same fields, order and types). */
typedef struct rtmi_camera_params {
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
} rtmi_camera_params;

/* The 14 POD fields of RayTracingCore, reference src/ray.tracer.core.hpp:19-32 (same order and types). */
typedef struct rtmi_camera {
    uint32_t img_width;          /* rts_img_width */
    uint32_t img_height;         /* rts_img_height */
    float defocus_angle;         /* rts_defocus_angle */
    float viewport_height;       /* rts_viewport_height (stored, unused by the path) */
    float viewport_width;        /* rts_viewport_width  (stored, unused by the path) */
    uint16_t samples_per_pixel;  /* rts_samples_per_pixel */
    uint16_t maxdepth;           /* rts_maxdepth */
    float pixels_sample_scale;   /* rts_pixels_sample_scale */
    float pixel_delta_u[3];      /* rts_pixel_delta_u */
    float pixel_delta_v[3];      /* rts_pixel_delta_v */
    float pixel00[3];            /* rts_pixel00 */
    float cam_center[3];         /* rts_cam_center */
    float defocus_disk_u[3];     /* rts_defocus_disk_u */
    float defocus_disk_v[3];     /* rts_defocus_disk_v */
} rtmi_camera;

/* HittableObject {ObjKind; union {HittableObject_Sphere}}, reference src/ray.tracer.object.defs.hpp:30-57: 24 B. */
typedef struct rtmi_object {
    uint32_t kind;     /* HittableObjectKind: 0 = Sphere (hpp:25-28) */
    float center[3];   /* HittableObject_Sphere::Center */
    float radius;      /* HittableObject_Sphere::Radius */
    uint32_t material; /* MaterialHandleType (src/ray.tracer.material.handle.hpp:6): index into materials */
} rtmi_object;

/* Material {MatKind; union {Lambertian{Albedo}; Metallic{Albedo,Fuzziness}; Dielectric{RefractionIndex}}},
 * reference src/ray.tracer.material.defs.hpp:20-55: 20 B. */
typedef struct rtmi_material {
    uint32_t kind; /* MaterialKind: 0 Lambertian, 1 Metallic, 2 Dielectric (hpp:20-25) */
    float p[4];    /* Lambertian: p[0..2] albedo; Metallic: p[0..2] albedo, p[3] fuzziness; Dielectric: p[0] index */
} rtmi_material;

/* WorldDefinition minus camera and objects, reference src/ray.tracer.core.cc:67-95. */
typedef struct rtmi_world_def {
    int32_t a_min, a_max, b_min, b_max;
    float center_offset[3];
    float center_dist_treshold;
    float diffuse_material_treshold;
    float metal_material_treshold;
} rtmi_world_def;

typedef enum rtmi_accel {
    RTMI_ACCEL_AUTO = 0,  /* BVH walk above 24 objects (measured crossover), the linear scan below */
    RTMI_ACCEL_BRUTE = 1, /* the reference's linear closest-hit scan (object.defs.cc:68-81), spheres in LDS */
    RTMI_ACCEL_BVH = 2    /* exact-equivalent BVH walk (same closest hit, same tie rule), nodes in LDS */
} rtmi_accel;

/* Scheduling knobs.  None of them changes a bit of the image (the draw streams are keyed by absolute pixel and
 * sample); they exist for tuning and for the tests that prove exactly that.  0 = library default everywhere.
 * The library reads no environment variables. */
typedef struct rtmi_tuning {
    uint32_t struct_size;       /* = sizeof(rtmi_tuning) */
    uint32_t block_lanes;       /* lanes per workgroup, multiple of 64 (default 768: 2 x 768 per CU = 6 waves per SIMD) */
    uint32_t blocks_per_cu;     /* cap on resident workgroups per CU (default: the occupancy query) */
    uint32_t wait_thresh;       /* lanes waiting for shading that end a traversal round (default 52; 56 for trees staged into LDS whose camera rays have entries, 44 for trees in HBM with walk starts) */
    uint32_t pad_mode;          /* BVH box pad per ray segment: 1 = the class pad of rounds 1-3 (from the farthest centre of each
                                 * radius class), 2 = that pad bounded by the reach of the segment, 0 = default: 2 on scenes
                                 * much wider than their spheres (where it pays), else 1; same image in every mode.
                                 * (This slot was drain_wait_thresh until round 2 and ignored in round 3.) */
    int32_t chunk_samples;      /* samples per work item: 0 = auto, -1 = whole pixels (no sample records), n > 0 = n */
    int32_t chain_mode;         /* attenuation chains: 0 = auto (packed strings of material handles in LDS, multiplied by the
                                 * resolve pass, when they fit next to the scene; else run-length encoded runs with per-lane
                                 * strips in HBM, multiplied at path end), 1 = always the run-length encoded form */
    uint32_t bvh_passes;        /* reinsertion passes over the BVH after the top-down SAH build: 0 = default (2; none above 8192 objects), n > 0 = n - 1
                                 * (1 = the plain top-down tree of rounds 1-3); any tree gives the same image.
                                 * (This slot was defer_cap until round 2 and ignored in round 3.) */
    uint32_t sample_buf_mb;     /* cap on the sample-record buffer in MiB (default: a third of the device's memory, 96 GB on MI355X;
                                 * 24576 in rounds 2-4); a call that needs more runs in bands of rows, one after the other */
    uint32_t force_hbm_scene;   /* nonzero: leave the scene in HBM even when it fits LDS (the config-4 path) */
    uint32_t top_down;          /* nonzero: hand out tiles top row first instead of bottom row first */
    uint32_t kernel;            /* 0 / 1 = the round-based kernel; 2 asked for round 2's queue-scheduled kernel (an experiment
                                 * that lost on every measured workload, removed in round 4): RTMI_ERR_UNSUPPORTED */
    uint32_t reserved3[3];      /* (were wf_block_lanes / wf_slots / wf_refill, the geometry of that kernel) ignored */
    uint32_t lds_top_nodes;     /* HBM-resident trees: how many breadth-first nodes of the top of the tree each workgroup
                                 * stages into LDS (0 = default: as many as fit next to the stacks; n > 0: at most n - 1) */
    /* ---- added in 0.5 (a caller built against the 0.4 header passes its smaller struct_size and gets the defaults) ---- */
    uint32_t tile_order;        /* order the 8x8 tiles of a launch are handed out in: 0 = costliest first for scenes staged into LDS,
                                 * where a launch has >= 2048 tiles and >= 16 samples per pixel (per-tile costs from one 2-spp probe
                                 * launch per scene, made by the first such call), 1 = row by row, bottom rows first (rounds 1-4),
                                 * 2 = costliest first always */
    uint32_t bands;             /* a call whose sample records pass the cap (sample_buf_mb) is rendered in bands of rows, one after
                                 * the other: 0 / 1 = as many bands as the cap asks for (one if the call fits), n > 1 = at least n */
    uint32_t gen_ahead;         /* ignored since 0.6 (0.5: primary rays of packed-chain scenes generated ahead into per-lane LDS slots; the
                                 * same-box A/B of round 6 showed its code cost the config-5 frame 3 % whether switched on or off) */
    /* ---- added in 0.6 ---- */
    uint32_t cam_entry;         /* camera rays start their walk at the entry of their 8x8 tile -- the lowest common ancestor of every
                                 * sphere the tile's beam (lens disk x tile rectangle on the focus plane) can meet, nowhere when it meets
                                 * none -- instead of the root: 0 = for trees staged into LDS (measured: -3.2 % on the 1080p S-RTOW frame)
                                 * and for trees in HBM whose scattered rays start in their own leaf (walk_start; alone it cost the
                                 * 100k-sphere tree +2.7 %: the levels it skips are the staged, cheap ones), 1 = off, 2 = always */
    uint32_t walk_start;        /* where the walks of scattered rays start: 0 = default -- on trees that stay in HBM, in the leaf of the sphere
                                 * the ray was scattered off, the siblings hanging off the path above it pre-loaded on the stack as way
                                 * records (two levels a record, in the node format); at the root for trees staged into LDS, whose
                                 * 80 KiB have no room for the records --, 1 = always at the root */
    uint32_t stack_cap;         /* HBM-resident trees: entries of a lane's traversal stack kept in LDS (the rest of a deeper walk spills
                                 * to memory): 0 = default, n > 0 = n */
    uint32_t reserved6;
} rtmi_tuning;

typedef struct rtmi_scene_options {
    uint32_t struct_size;  /* = sizeof(rtmi_scene_options) */
    uint32_t accel;        /* rtmi_accel */
    uint32_t leaf_size;    /* BVH: max spheres per leaf, 1..4 (0 = default: 2, or 4 for more than 8192 objects -- trees that stay in HBM) */
    int32_t device;        /* HIP device ordinal, -1 = the caller's current device (also when options == NULL) */
    uint32_t collect_stats;/* nonzero: kernels also count segments / node tests / sphere tests */
    uint32_t reserved[3];
    const rtmi_tuning* tuning; /* NULL = defaults */
} rtmi_scene_options;

typedef struct rtmi_stats {
    uint64_t samples, segments, sphere_tests, node_tests;
} rtmi_stats;

/* flat BVH node as exported by rtmi_scene_get_bvh (for the tests' instrumented CPU walk): 64 B */
typedef struct rtmi_bvh_node {
    float ctr[2][3];
    float half[2][3];
    uint32_t child[2]; /* bit31 set: leaf {bits 0..23 first slot, bits 24..30 count}; else node index */
    float reserved[2];
} rtmi_bvh_node;

typedef struct rtmi_scene rtmi_scene; /* opaque: owns the device copies of camera, objects, materials, BVH */

/* ---- host-side setup (replaces reference src/ray.tracer.core.cc:158-216) -------------------------- */

/* Camera derivation of RayTracingCore::default_setup (core.cc:174-195 + make_camera_frame :158-169). */
int rtmi_camera_setup(const rtmi_camera_params* params, rtmi_camera* out);

/* Scene generator make_world_spheres (core.cc:99-149) with the RNG seeded from `mt_seed` instead of
 * std::random_device (random.number.gen.hpp:45-46).  Writes the `n_fixed` listed objects first, then the
 * a/b grid; returns the object count (== material count) through n_out; capacity is in records. */
int rtmi_make_world_spheres(const rtmi_world_def* def, const rtmi_object* fixed_objects,
                            const rtmi_material* fixed_materials, uint32_t n_fixed, uint32_t mt_seed,
                            rtmi_object* objects_out, rtmi_material* materials_out, uint32_t capacity,
                            uint32_t* n_out);

/* ---- scene life cycle (replaces the ownership of shared_ptr<RayTracingCore>, main.cc:433,604,669-672) ---- */

/* Uploads camera + world + materials (the reference's rts_world / rts_materials, core.hpp:33-34) and builds
 * the acceleration structure.  `options` may be NULL (= BVH/scan chosen by size, the caller's current device).
 * Objects with a non-finite centre or radius are rejected (RTMI_ERR_BAD_ARG): the reference would trace NaNs, a
 * bounding volume cannot hold them.  Every entry point leaves the caller's current HIP device as it found it. */
int rtmi_scene_create(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects,
                      const rtmi_material* materials, uint32_t n_materials, const rtmi_scene_options* options,
                      rtmi_scene** out);
void rtmi_scene_destroy(rtmi_scene* scene);

/* ---- the hot path (replaces RayTracingCore::raytrace_pixel per pixel of a tile, core.cc:259-265) -------- */

/* Renders image rows [y0, y1): for every pixel, samples_per_pixel paths of at most maxdepth segments, summed
 * sequentially in fp32 and scaled by pixels_sample_scale (core.cc:260-264).
 *   rgb_linear_out  nullable, (y1-y0)*W*3 floats: the value handed to RGBAColor{...} (core.cc:264)
 *   rgba8_out       nullable, (y1-y0)*W uint32 0xAABBGGRR: RGBAColor(vec3) (color.hpp:30-36)
 * Random stream: counter-based, keyed by (seed, pixel = y*W + x, sample), independent of tiling, row ranges
 * and GPU count.  Host pointers; blocking; safe to call concurrently on one scene from several threads. */
int rtmi_render_rows(rtmi_scene* scene, uint32_t y0, uint32_t y1, uint64_t seed, float* rgb_linear_out,
                     uint32_t* rgba8_out);

/* Tile-granular entry (0.5): pixels [x0, x1) x [y0, y1) -- one RayTracingWorkPackage{u16vec2 start, end} of the reference
 * (src/main.cc:404-407, consumed pixel by pixel at :507-519) -- into DENSE outputs of (x1 - x0) pixels per row:
 *   rgb_linear_out  nullable, (y1-y0)*(x1-x0)*3 floats;  rgba8_out  nullable, (y1-y0)*(x1-x0) uint32 0xAABBGGRR.
 * Same pixels, bit for bit, as the corresponding part of rtmi_render_rows (the draw streams are keyed by the absolute
 * pixel).  Host pointers; blocking; safe to call concurrently on one scene.  A host that keeps the reference's queue of
 * 8x8 packages unchanged calls this once per package: correct, and slow -- every call is a kernel launch, a resolve pass
 * and a device-to-host copy for 64 pixels; row blocks (rtmi_render_rows) are the fast granularity. */
int rtmi_render_rect(rtmi_scene* scene, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint64_t seed,
                     float* rgb_linear_out, uint32_t* rgba8_out);
/* Same with DEVICE output pointers, asynchronous on `hip_stream` (see rtmi_render_row_blocks_device for the stream rule). */
int rtmi_render_rect_device(rtmi_scene* scene, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint64_t seed,
                            void* d_rgb_linear_out, void* d_rgba8_out, void* hip_stream);

/* Same, row-block sharded and device-resident (multi-GPU path): renders `n_blocks` blocks of `block_rows` rows,
 * block k covering rows [y_first + k*block_stride*block_rows, +block_rows) clipped to the image, into a dense
 * slice of n_blocks*block_rows rows.  Outputs are DEVICE pointers (either may be NULL); the launch is
 * asynchronous on `hip_stream` (a hipStream_t, NULL = default stream).  Launches on one scene share its work
 * counter and sample buffers: issue them on ONE stream (or wait for the previous one) -- use one scene per stream
 * for concurrent frames.
 * What "asynchronous" covers (0.6): the scene's cost probe is made by rtmi_scene_create, so no call probes.  A call only
 * enqueues work on `hip_stream` once the scene has seen its geometry; the FIRST call with a new geometry (another set of rows or
 * columns), or one that needs larger record buffers than the scene holds, also allocates device memory on the host side
 * (hipMalloc, and hipFree -- which waits for the device -- when a buffer grows) and uploads a 4-byte-per-tile order table in
 * front of its kernel: such a call can block for the device and cannot be stream-captured; repeat it once before capturing. */
int rtmi_render_row_blocks_device(rtmi_scene* scene, uint32_t y_first, uint32_t block_rows, uint32_t block_stride,
                                  uint32_t n_blocks, uint64_t seed, void* d_rgb_linear_out, void* d_rgba8_out,
                                  void* hip_stream);

/* Same over a LIST of row blocks (0.6; the cost-balanced shards of a multi-GPU frame, rtmi_shard_plan): block k of the call is
 * image rows [blocks[k] * block_rows, + block_rows) clipped to the image, rendered into rows [k * block_rows, ...) of the dense
 * slice.  block_rows must be a multiple of 8; only the last listed block may be clipped by the image's end.  The image does not
 * depend on how rows are grouped into calls. */
int rtmi_render_block_list_device(rtmi_scene* scene, uint32_t block_rows, const uint32_t* blocks, uint32_t n_blocks,
                                  uint64_t seed, void* d_rgb_linear_out, void* d_rgba8_out, void* hip_stream);

/* Row blocks -> ranks for a frame of `height` rows on `n_ranks` devices (0.6; host only).  block_cost = NULL: block b -> rank
 * b mod n_ranks (rounds 1-5).  With costs (one per block of block_rows rows, e.g. the sums of rtmi_scene_get_tile_costs over
 * the block's tiles): longest processing time first -- blocks in falling cost order, each to the rank with the least cost so far
 * among those that hold fewer than ceil(n_blocks / n_ranks) blocks -- so that every rank's slice has the same size and the
 * ranks' costs differ by a fraction of one block.  rank_of_block_out: ceil(height / block_rows) entries.  Deterministic. */
int rtmi_shard_plan(uint32_t height, uint32_t block_rows, uint32_t n_ranks, const uint64_t* block_cost, uint32_t* rank_of_block_out);
/* The scene's cost map: ray segments per 8x8 tile of the image from the 2-spp probe rtmi_scene_create makes for scenes whose
 * tiles are handed out costliest first (rtmi_tuning::tile_order), row-major over ceil(W / 8) x ceil(H / 8) tiles; *n_tiles = 0
 * when the scene made no probe.  costs_out may be NULL (count only). */
int rtmi_scene_get_tile_costs(const rtmi_scene* scene, uint32_t* costs_out, uint32_t* n_tiles);

/* ---- introspection ------------------------------------------------------------------------------------ */
const char* rtmi_last_error(void);
const char* rtmi_version(void);
/* accumulated since scene creation when options.collect_stats != 0 */
int rtmi_scene_get_stats(rtmi_scene* scene, rtmi_stats* out, int reset);
/* which kernel rtmi_render_* will launch for this scene: RTMI_ACCEL_BRUTE or RTMI_ACCEL_BVH */
int rtmi_scene_get_accel(const rtmi_scene* scene, uint32_t* accel_out);
/* How rtmi_render_* will launch the trace kernel for this scene (what the tuning and the scene's size resolved to). */
typedef struct rtmi_launch_info {
    uint32_t struct_size;   /* in: sizeof(rtmi_launch_info) */
    uint32_t kernel;        /* 1: the round-based kernel (the only one since round 4) */
    uint32_t block_lanes;   /* lanes per workgroup */
    uint32_t grid_blocks;   /* persistent workgroups of one launch */
    uint32_t blocks_per_cu; /* resident workgroups per CU */
    uint32_t lds_bytes;     /* dynamic LDS per workgroup */
    uint32_t scene_in_lds;  /* 1: nodes, spheres and materials are staged into LDS; 0: read from HBM through the caches */
    uint32_t stack_depth;   /* traversal-stack entries per lane */
    uint32_t whole_pixel_fallbacks; /* launches so far whose sample-record buffer (16 B per sample of the call, capped by
                             * rtmi_tuning::sample_buf_mb) could not be allocated: they ran with whole-pixel work items --
                             * the same image, but a longer tail at the end of the launch */
    uint32_t packed_chains; /* 0: run-length encoded attenuation chains; else the words per sample of the packed form the scene
                             * is eligible for (what a launch does when its chain slots can be allocated) */
    /* ---- added in 0.4 (a caller built against the 40-byte struct of 0.3 passes struct_size = 40 and gets the fields above) ---- */
    uint32_t packed_chain_fallbacks; /* launches so far of a scene eligible for packed chains that ran with run-length encoded
                             * chains instead (their chain slots could not be allocated or passed the buffer cap): same image */
    uint32_t lds_top_nodes; /* HBM-resident trees: breadth-first nodes of the top of the tree staged into LDS by every workgroup */
    uint32_t pad_mode;      /* what rtmi_tuning::pad_mode resolved to: 1 class pad, 2 bounded by the segment's reach (0: no BVH) */
    /* ---- added in 0.5: what the most recent call on the scene did ---- */
    uint32_t bands;         /* bands of rows it was rendered in (0: no call yet) */
    uint32_t tile_order;    /* 1: its tiles were handed out costliest first, 0: row by row */
    uint32_t probe_us;      /* duration of the scene's cost probe launch in microseconds (0: none was made) */
    uint32_t gen_ahead;     /* always 0 since 0.6 (see rtmi_tuning::gen_ahead) */
    /* ---- added in 0.6 ---- */
    uint32_t cam_entry;     /* 1: camera rays start at their tile's entry (rtmi_tuning::cam_entry) */
    uint32_t entry_build_us;/* host time rtmi_scene_create spent on the table of entries, microseconds */
    uint32_t walk_start;    /* what rtmi_tuning::walk_start resolved to (0: every scattered ray's walk starts at the root) */
    uint32_t stack_cap;     /* stack entries per lane kept in LDS when the stack is capped (0: the whole stack is in LDS) */
    uint32_t reband_retries;/* times a call was planned again for half the record cap because the device refused a band's buffers
                             * (memory held by other scenes, the caller or another process): more, smaller bands, the same image */
} rtmi_launch_info;
int rtmi_scene_get_launch_info(const rtmi_scene* scene, rtmi_launch_info* out);
/* BVH export: call with NULL buffers to get the counts. pad_classes: n_classes x 8 floats {lo[3], hi[3], 1/(2*rmin), rmax^2}. */
int rtmi_scene_get_bvh(const rtmi_scene* scene, rtmi_bvh_node* nodes_out, uint32_t* n_nodes, uint32_t* slots_out,
                       uint32_t* n_slots, float* pad_classes_out, uint32_t* n_classes, float* pad_eps,
                       float* pad_floor);
/* Host-only BVH build (no device needed): the structure rtmi_scene_create would build for these objects.
 * nodes_out / slots_out need room for n_objects records, pad_classes_out for 32 floats; any output may be NULL. */
int rtmi_bvh_build(const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size, rtmi_bvh_node* nodes_out,
                   uint32_t* n_nodes, uint32_t* slots_out, uint32_t* root_ref, uint32_t* depth,
                   float* pad_classes_out, uint32_t* n_classes, float* pad_eps, float* pad_floor);
/* Same with the number of reinsertion passes chosen as in rtmi_tuning::bvh_passes (0 = default, n > 0 = n - 1). */
int rtmi_bvh_build_passes(const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size, uint32_t bvh_passes,
                          rtmi_bvh_node* nodes_out, uint32_t* n_nodes, uint32_t* slots_out, uint32_t* root_ref,
                          uint32_t* depth, float* pad_classes_out, uint32_t* n_classes, float* pad_eps, float* pad_floor);
/* The scene's table of camera-ray entries (0.6; rtmi_tuning::cam_entry), in the format of rtmi_tile_entries_build below; *n_tiles = 0 when
 * the scene has none (camera rays walk from the root).  entries_out may be NULL (count only). */
int rtmi_scene_get_tile_entries(const rtmi_scene* scene, uint32_t* entries_out, uint32_t* n_tiles);
/* The scene's walk starts of scattered rays (0.6; rtmi_tuning::walk_start, HBM-resident trees): per sphere slot 16 words {reference the
 * walk of a ray scattered off that sphere starts at, n, indices of the n way records pre-loaded on its stack, padding}; way
 * records are the nodes rtmi_scene_get_bvh returns BEHIND the tree's own (two sibling boxes of the path per record, in the node
 * format).  *n_slots = 0 when every walk starts at the root.  records_out may be NULL (count only). */
int rtmi_scene_get_walk_starts(const rtmi_scene* scene, uint32_t* records_out, uint32_t* n_slots);
/* Host-only (0.6): the tree of rtmi_bvh_build_passes for the same arguments with the way records behind its nodes, and the start
 * records (16 words per slot) of rtmi_scene_get_walk_starts.  nodes_out needs room for 3 * n_objects + 2 records, records_out for
 * 16 * n_objects words; any output may be NULL. */
int rtmi_walk_starts_build(const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size, uint32_t bvh_passes,
                           rtmi_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* n_tree_nodes_out, uint32_t* records_out);
/* Host-only (0.6): the table of camera-ray entries rtmi_scene_create builds for this camera and these objects (rtmi_tuning::cam_entry) --
 * per 8x8 tile of the image, row-major over ceil(W / 8) x ceil(H / 8) tiles, the reference (rtmi_bvh_node::child format, of the tree
 * rtmi_bvh_build_passes returns for the same arguments) of the lowest common ancestor of the leaves of every sphere a sample of the
 * tile can hit, 0xffffffff when it can hit none.  entries_out may be NULL (count only). */
int rtmi_tile_entries_build(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size,
                            uint32_t bvh_passes, uint32_t* entries_out, uint32_t* n_tiles);
/* Milliseconds the TRACE kernels of the most recent call on this scene took (the bands of a banded call added up; the
 * ordered resolve passes between them are not included), from HIP events recorded on the launch stream; blocks until that
 * call has finished.  Used by bench.py for the roofline line. */
int rtmi_scene_last_kernel_ms(rtmi_scene* scene, float* ms_out);

/* ---- one frame on several GPUs of one node (replaces the worker fan-out of RayTracer::create, main.cc:586-731,
 *      and the per-frame drain of RayTracer::update, main.cc:733-774) -------------------------------------------
 * One process, n devices.  The image plane is sharded by interleaved row blocks (block b -> device b mod n, so the
 * cheap sky rows and the expensive ground rows are spread evenly); every device holds a replica of the scene and
 * renders its blocks into a dense slice (float RGB and RGBA8 in one buffer); ONE RCCL gather (one ncclGather per rank
 * inside ncclGroupStart/End, over xGMI) brings the slices to devices[0], where a small kernel restores scanline order.
 * The frame is bit-identical for any n (the draw streams are keyed by absolute pixel and sample).  With n == 1 no
 * communicator is created (unless RTMI_FRAME_FORCE_RCCL asks for it). */
typedef struct rtmi_frame rtmi_frame; /* opaque: scene replicas, streams, slices, RCCL communicators */

typedef struct rtmi_frame_timing {
    float total_ms;       /* first launch -> frame in scanline order on devices[0] (host wall clock, blocking call) */
    float gather_ms;      /* RCCL gather + de-interleave on devices[0] (HIP events on its stream) */
    float kernel_ms[16];  /* trace-kernel time of each device (HIP events), first n entries valid */
} rtmi_frame_timing;

/* `options->device` is ignored (the list decides); `devices` = HIP ordinals, distinct, 1 <= n <= 16;
 * `block_rows` = rows per shard block (0 = 8).  Test hook: options->reserved[0] & RTMI_FRAME_REHEARSAL lets a device
 * appear more than once and gathers the slices with plain copies instead of RCCL (which wants one rank per device), so
 * that a one-GPU box can check the shard plan and the scanline order for n > 1.  options->reserved[0] &
 * RTMI_FRAME_FORCE_RCCL creates the communicator and runs the grouped gather for n == 1 as well (rank 0 gathers from
 * itself), so that librccl, ncclCommInitAll, the gather and its stream ordering against the trace and de-interleave
 * kernels execute on a one-GPU box. */
#define RTMI_FRAME_REHEARSAL 1u
#define RTMI_FRAME_FORCE_RCCL 2u
/* options->reserved[0] & RTMI_FRAME_COST_PLAN (0.6): the row blocks are dealt out to the devices by the scene's cost map
 * (rtmi_shard_plan) instead of block b -> device b mod n -- where the scene has one and block_rows is a multiple of 8 */
#define RTMI_FRAME_COST_PLAN 4u
int rtmi_frame_create(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects,
                      const rtmi_material* materials, uint32_t n_materials, const rtmi_scene_options* options,
                      const int32_t* devices, uint32_t n_devices, uint32_t block_rows, rtmi_frame** out);
void rtmi_frame_destroy(rtmi_frame* frame);
/* Renders the whole frame.  Host outputs (either may be NULL): H*W*3 floats / H*W uint32, as rtmi_render_rows.
 * Blocking; externally synchronised (one call at a time per frame object). */
int rtmi_frame_render(rtmi_frame* frame, uint64_t seed, float* rgb_linear_out, uint32_t* rgba8_out);
/* Same, but the frame stays on devices[0]: returns pointers (owned by the frame object, valid until the next
 * render or destroy) to H*W*3 floats and H*W uint32 in scanline order.  Blocking. */
int rtmi_frame_render_device(rtmi_frame* frame, uint64_t seed, void** d_rgb_linear, void** d_rgba8);
int rtmi_frame_get_timing(const rtmi_frame* frame, rtmi_frame_timing* out);
/* The frame's scene replica on devices[index] (owned by the frame, valid until rtmi_frame_destroy): lets a host that
 * also wants per-device row-block workers (rtmi_render_rows from one thread per device) use the replicas the frame
 * already holds instead of creating a second scene -- BVH, strips and sample records -- per device.  A replica must not
 * be rendered through rtmi_render_* while rtmi_frame_render* is running on the frame. */
int rtmi_frame_get_scene(rtmi_frame* frame, uint32_t index, rtmi_scene** scene_out);
/* number of RCCL ranks behind the frame (0 when no communicator exists: n_devices == 1 without RTMI_FRAME_FORCE_RCCL,
 * or a rehearsal frame) */
int rtmi_frame_rccl_ranks(const rtmi_frame* frame, uint32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
