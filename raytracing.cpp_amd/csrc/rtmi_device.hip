// rtmi_device.hip -- the per-pixel path-tracing hot loop as hand-written HIP for gfx950 (CDNA4), and the
// C-ABI entry points that need the HIP runtime.  One translation unit along its seams:
//   rtmi_walk_asm.h      the two hand-scheduled node loops of the BVH walk
//   rtmi_trace_kernel.h  rtmi_trace_kernel<ACCEL, STATS, BIG, MODE>
//   rtmi_resolve.h       the resolve passes (ordered per-pixel sum; packed chains)
//   this file            scene handle, launch scheduling (bands, tile order, probe), entry points
//
// Replaces, from the reference (adihodos/raytracing.cpp):
//   RayTracingCore::raytrace_pixel / get_ray / compute_color      src/ray.tracer.core.cc:218-265
//   HittableObject_Collection::intersects / _Sphere::intersects   src/ray.tracer.object.defs.cc:11-18, 41-81
//   Material::scatter (Lambertian / Metallic / Dielectric)        src/ray.tracer.material.defs.cc:31-109
//   RandomNumberGenerator sampling helpers                        src/random.number.gen.hpp:11-42
//   RGBAColor(vec3), linear_to_gamma, clamp                       src/color.hpp:9-36, src/ray.tracer.math.hpp:10-19
//
// Design (MI355X-first):
//  * persistent lanes: a work item is a chunk of consecutive samples of one pixel; lanes take items from per-wave
//    pools refilled 64 at a time from a global counter (ballot + prefix popcount), store one 16-byte record per
//    sample, and a resolve pass adds the records up in sample order (the reference's sequential fp32 sum,
//    core.cc:260-263).
//  * recursion flattened: a lane is a small state machine FETCH -> GEN -> BEGIN -> TRAVERSE -> SHADE; the attenuation
//    chain A1*(A2*(...*sky)) of the recursive compute_color is multiplied innermost-first, so the colour is bit-identical
//    to the recursion: from run-length encoded material handles at path end, or -- where the whole chain fits LDS as a
//    packed string of handles -- by the resolve pass, one lane per sample (rtmi_resolve_chain_kernel).
//  * traversal: per iteration the wave votes between a node step and a leaf step; it leaves the loop as soon as
//    enough lanes wait for shading (ballot/popcount), shades them, refills them and re-enters traversal.  The node
//    steps are hand-scheduled gfx950 loops (walk_nodes_lds for trees staged into LDS, walk_nodes_hbm for trees read
//    through the caches); the C++ node step beside them serves the statistics and stamp variants.
//  * scenes of up to 24 spheres are scanned linearly, as the reference does (measured crossover).
//  * random_unit_vector: the owner lane makes its first two attempts, the wave shares the retries (coop_draws).
//  * the reference's fp32 divisions and square roots run as the in-range cores of the compiler's own expansions
//    (bit-identical, a third of the instructions), the full expansions behind a branch for operands out of range.
//  * scene (BVH nodes, spheres, materials) staged once per workgroup into LDS with coalesced 16-byte loads; the
//    per-lane traversal stack lives in LDS too.  Scenes that do not fit stay in HBM (BIG variant) behind L2 / Infinity Cache.
//  * counter RNG: a block function (pcg4d; Philox4x32 in the A/B) of (draw block, sample, pixel) keyed by a bijective mix
//    of the seed: the image does not depend on tiling, row sharding or GPU count.
//  * arithmetic of the reference path is kept operation for operation (no FMA contraction, IEEE sqrt/div); only the
//    BVH slab tests, which the reference does not have, use FMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rtmi_internal.h"

#include "rtmi_kernel_common.h"

#include "rtmi_trace_kernel.h"
#include "rtmi_resolve.h"

// ---------------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                                    \
            return e_ == hipErrorOutOfMemory ? RTMI_ERR_OOM : RTMI_ERR_HIP;                                  \
        }                                                                                                    \
    } while (0)

// every entry point leaves the caller's current device as it found it
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() {
        if (hipGetDevice(&prev) != hipSuccess) {
            prev = -1;
            (void)hipGetLastError();
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// nothing is thrown across the C-ABI: std::vector growth in the BVH builder, std::string in set_error ...
template <class F>
static int guarded(const char* where, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { set_error(std::string(where) + ": out of host memory"); } catch (...) {}
        return RTMI_ERR_OOM;
    } catch (const std::exception& e) {
        try { set_error(std::string(where) + ": " + e.what()); } catch (...) {}
        return RTMI_ERR_INTERNAL;
    } catch (...) {
        try { set_error(std::string(where) + ": unknown exception"); } catch (...) {}
        return RTMI_ERR_INTERNAL;
    }
}

// The buffers a launch works in.  A call whose sample records pass the cap is rendered in bands of rows, one after the other on the
// caller's stream (trace, resolve, trace, ...): all bands use this one set.
// Round 5 built what VERDICT r4 #2 asked for -- two, then three of these with a stream each, so that band k's ordered resolve pass and
// tail would run beside band k + 1's trace kernel -- and measured that MI355X does not do it (profiles/r05_band_overlap.txt): while
// a kernel that stores to memory holds 22-24 waves of every CU (the trace kernel: 24) no workgroup of another queue becomes
// resident, whatever its size (tools/ubench/coresidency.hip: a 64-thread kernel of 8 VGPRs waits for the whole kernel; beside a kernel of
// the same footprint that only computes, or at 20 waves per CU, it runs at once), so a resolve pass only ran when the next band
// drained, and two queued trace kernels that are released together share the chip and end together.  Four overlapped bands
// took 140 ms where the 1080p x 512 spp frame takes 133 in one; the machinery is out again.
struct LaunchSlot {
    uint32_t* d_counter = nullptr; // work counter
    uint32_t* d_att = nullptr;     // per-lane strips of attenuation runs (indexed by the lane's position in the grid)
    float4* d_samples = nullptr;   // sample-chunk split: [pixel][sample]
    size_t samples_capacity = 0;   // records
    uint32_t* d_chain = nullptr;   // packed attenuation chains (MODE 4): [pixel][sample][att_words]
    size_t chain_capacity = 0;     // samples
};

struct rtmi_scene {
    rtmi_camera cam{};
    int device = 0;
    uint32_t accel = RTMI_ACCEL_BVH;
    bool collect_stats = false;
    bool big = false; // scene read from HBM instead of LDS
    uint32_t n_objects = 0, n_mats = 0;
    Bvh bvh;
    // device buffers
    uint4* d_spheres = nullptr;
    uint4* d_aux = nullptr;
    uint4* d_mats = nullptr;
    uint4* d_nodes = nullptr;
    uint4* d_walk_starts = nullptr;   // scattered rays of HBM-resident trees: per slot the start record of build_walk_starts (NULL: the root)
    std::vector<uint32_t> walk_starts; // (host copy for rtmi_scene_get_walk_starts: the tests' instrumented CPU walk)
    uint32_t n_tree_nodes = 0;         // nodes of the tree itself; bvh.nodes may hold way records behind them
    uint32_t* d_tile_entry = nullptr; // camera-ray entry per 8x8 tile of the image (build_tile_entries), NULL: walks start at the root
    std::vector<uint32_t> tile_entries; // the same table in the builder's reference format (rtmi_scene_get_tile_entries: the tests' instrumented CPU walk)
    float entry_build_ms = 0.0f;
    unsigned long long* d_stats = nullptr;
    unsigned long long* d_tail = nullptr; // -DRTMI_TAILPROBE builds only
    float* d_rgb = nullptr;     // staging for rtmi_render_rows / rtmi_render_rect (host-pointer entries)
    uint32_t* d_rgba = nullptr; // staging
    size_t staging_pixels = 0;
    LaunchSlot slot[1];
    // packed attenuation chains (MODE 4): eligible when the per-lane strings fit the LDS next to the staged scene
    bool packed_ok = false;
    uint32_t att_bits = 0, att_epw = 0, att_words = 0; // bits per handle, handles per word, words per sample (multiple of 4)
    uint32_t chunk = ~0u;        // samples per work item of the split; ~0u: chosen per launch, 0: split off
    size_t sample_buf_cap_bytes = (size_t)24 << 30; // records (+ chain slots) of one band (scene_create: a third of the device's memory)
    uint32_t min_bands = 0;      // rtmi_tuning::bands: 0 = by the size of the call
    // HIP events around the trace kernel of every band of the most recent call
    std::vector<hipEvent_t> ev_trace; // [2 * band]: before / after that band's trace kernel
    uint32_t n_bands_timed = 0;
    bool ev_valid = false;
    uint32_t whole_pixel_fallbacks = 0; // launches that could not get their sample-record buffer
    uint32_t packed_chain_fallbacks = 0; // launches of a packed-chain scene that ran with run-length encoded chains
    uint32_t reband_retries = 0;         // times a call's plan was made again for half the cap because the device refused a band's buffers
    bool top_down = false;
    bool pad_refine = false; // box pad bounded by the segment's reach (rtmi_tuning::pad_mode; default: where it pays, Bvh::pad_refine)
    // cost-ordered hand-out of the 8x8 tiles (rtmi_tuning::tile_order): segments per tile of the whole image from one probe
    // launch (2 samples per pixel, whole-pixel work items, the counting variant of the kernel), made the first time a call is
    // large enough to gain from it; the reference renders the same scene every frame (main.cc:733-774), so does every caller
    // of a scene handle.  Per launch geometry the order table is computed once and kept.
    uint32_t tile_order_mode = 0;   // 0: where it pays, 1: never, 2: always
    int cost_state = 0;             // 0: no probe yet, 1: tile_cost valid, -1: the probe failed (row-by-row order from then on)
    std::vector<uint32_t> tile_cost;
    float probe_ms = 0.0f;
    double probe_seg_per_sample = 1.0; // ray segments per sample over the probe's frame
    struct OrderEntry { // what a launch geometry needs on the device: the order of its tiles and, for a list of row blocks, their first rows
        uint32_t key[8]; // y_first, block_rows, block_stride, n_blocks, x0, x1, hash of the block list (two words; 0 for a strided set)
        uint32_t* d_order;       // NULL: no cost map, tiles row by row
        uint32_t* d_first_row;   // NULL: a strided set of blocks
        std::vector<uint32_t> h_first_row;
        std::vector<uint32_t> h_order; // (kept: the upload is asynchronous on the stream of the call that made the entry)
        uint32_t n_tiles;
        double cost;     // sum over the entry's tiles
    };
    std::vector<std::unique_ptr<OrderEntry>> orders; // (entries keep their addresses: the bands of a call hold pointers to them)
    uint32_t last_bands = 0, last_tile_order = 0; // of the most recent call
    // launch geometry
    uint32_t block = 768, grid = 0, lds_bytes = 0, stack_depth = 0; // 2 x 768 lanes per CU = 6 waves per SIMD (<= 80 VGPRs)
    uint32_t lds_spheres = 0, lds_aux = 0, lds_mats = 0, lds_nodes = 0, lds_stack = 0;
    // lanes waiting for shading that end a traversal round (A/B on MI355X, round 2: 52 = 56 on the LDS-resident RTOW
    // scene, 3.5 % better than 56 on the HBM-resident 100k-sphere scene; 62 costs that scene 23 %)
    uint32_t wait_thresh = 52; // (scene_create: 56 for trees staged into LDS since camera rays have entries, round 6: 126.6 against 127.8 ms)

    uint32_t lds_att = 0, lds_pool = 0;
    uint32_t lds_top_nodes = 0; // HBM-resident trees: records staged into LDS (48 bytes each, the start of the device array): breadth-first nodes ...
    uint32_t lds_top_ways = 0;  // ... of which this many slots are the block of top way records, placed by path code (rtmi_tuning::walk_start)
    uint32_t way_jtop = 0, way_top_base = 0; // levels of way records in that block (two tree levels each) and its first index
    uint32_t n_dev_nodes = 0;   // records of the device's node array (tree nodes + the sparse block + the deeper way records)
    uint32_t n_cus = 0;
    uint32_t root_ref_dev = 0; // root reference in the form the kernel variant expects
    uint32_t pre_leaf_dev[4] = {}; // leaves peeled off the top of the tree, tested at segment set-up
    uint32_t n_pre_leaves = 0;
    hipStream_t stream = nullptr; // private stream of the blocking entry points
    std::mutex mu;
};

namespace {

using KernelFn = void (*)(const RtmiLaunch);

template <int ACCEL, int MODE>
KernelFn pick_variant(bool stats, bool big) {
    if (big) return stats ? rtmi_trace_kernel<ACCEL, true, true, MODE> : rtmi_trace_kernel<ACCEL, false, true, MODE>;
    return stats ? rtmi_trace_kernel<ACCEL, true, false, MODE> : rtmi_trace_kernel<ACCEL, false, false, MODE>;
}

// mode 3: work items are whole pixels (no sample-record buffer), the lane keeps the pixel's sum; mode 4: sample records
// with packed attenuation chains (LDS-resident scenes only); mode 0: sample records, run-length encoded chains
KernelFn pick_kernel(uint32_t accel, bool stats, bool big, int mode) {
    const bool bvh = accel == RTMI_ACCEL_BVH;
    if (mode == 3) return bvh ? pick_variant<RTMI_ACCEL_BVH, 3>(stats, big) : pick_variant<RTMI_ACCEL_BRUTE, 3>(stats, big);
    if (mode == 4 && !big) {
        if (bvh) return stats ? rtmi_trace_kernel<RTMI_ACCEL_BVH, true, false, 4> : rtmi_trace_kernel<RTMI_ACCEL_BVH, false, false, 4>;
        return stats ? rtmi_trace_kernel<RTMI_ACCEL_BRUTE, true, false, 4> : rtmi_trace_kernel<RTMI_ACCEL_BRUTE, false, false, 4>;
    }
    return bvh ? pick_variant<RTMI_ACCEL_BVH, 0>(stats, big) : pick_variant<RTMI_ACCEL_BRUTE, 0>(stats, big);
}

void free_scene(rtmi_scene* s) {
    if (!s) return;
    hipSetDevice(s->device);
    hipFree(s->d_spheres);
    hipFree(s->d_aux);
    hipFree(s->d_mats);
    hipFree(s->d_nodes);
    hipFree(s->d_tile_entry);
    hipFree(s->d_walk_starts);
    hipFree(s->d_stats);
    hipFree(s->d_tail);
    hipFree(s->d_rgb);
    hipFree(s->d_rgba);
    for (LaunchSlot& sl : s->slot) {
        hipFree(sl.d_counter);
        hipFree(sl.d_att);
        hipFree(sl.d_samples);
        hipFree(sl.d_chain);
    }
    for (auto& e : s->orders) { hipFree(e->d_order); hipFree(e->d_first_row); }
    for (hipEvent_t e : s->ev_trace) hipEventDestroy(e);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

// the part of the launch parameters that belongs to the scene
void fill_scene_params(const rtmi_scene* s, RtmiLaunch& P) {
    P.cam = s->cam;
    P.spheres = s->d_spheres;
    P.aux = s->d_aux;
    P.mats = s->d_mats;
    P.nodes = s->d_nodes;
    P.n_slots = s->n_objects;
    P.n_mats = s->n_mats;
    P.n_nodes = s->n_tree_nodes;
    P.root_ref = s->root_ref_dev;
    std::memcpy(P.pre_leaf, s->pre_leaf_dev, sizeof(P.pre_leaf));
    P.n_pre_leaves = s->n_pre_leaves;
    P.tile_entry = s->d_tile_entry;
    P.walk_starts = s->d_walk_starts;
    P.way_jtop = s->way_jtop;
    P.way_top_base = s->way_top_base;
    std::memcpy(P.pad_classes, s->bvh.pad_classes, sizeof(P.pad_classes));
    P.n_pad_classes = s->bvh.n_pad_classes;
    P.pad_eps = s->bvh.pad_eps;
    P.pad_floor = s->bvh.pad_floor;
    P.pad_refine = s->pad_refine ? 1u : 0u;
    for (uint32_t c = 0; c < s->bvh.n_pad_classes; ++c) P.pad_rmax[c] = std::sqrt(s->bvh.pad_classes[c][7]) * 1.000001f;
    P.lds_spheres = s->lds_spheres;
    P.lds_aux = s->lds_aux;
    P.lds_mats = s->lds_mats;
    P.lds_nodes = s->lds_nodes;
    P.lds_stack = s->lds_stack;
    P.lds_top_nodes = s->lds_top_nodes;
    P.lds_att = s->lds_att;
    P.lds_pool = s->lds_pool;
    P.stack_depth = s->stack_depth;
    P.top_down = s->top_down ? 1u : 0u;
    P.wait_thresh = s->wait_thresh;
    P.div_w = make_fastdiv(s->cam.img_width);
    P.gtiles_x = (s->cam.img_width + 7u) / 8u;
    P.stats = s->d_stats;
    P.tail_probe = s->d_tail;
}

// rows a set of row blocks covers, validated: every block must start inside the image; only the last one may be clipped
int count_rows(const rtmi_scene* s, uint32_t y_first, uint32_t block_rows, uint32_t block_stride, uint32_t n_blocks, uint32_t& n_local_rows,
               const uint32_t* block_list = nullptr) {
    const uint32_t H = s->cam.img_height;
    if (block_rows == 0 || block_stride == 0) {
        set_error("rtmi: block_rows and block_stride must be positive");
        return RTMI_ERR_BAD_ARG;
    }
    n_local_rows = 0;
    for (uint32_t k = 0; k < n_blocks; ++k) {
        const uint64_t y = block_list ? (uint64_t)block_list[k] * block_rows : (uint64_t)y_first + (uint64_t)k * block_stride * block_rows;
        if (y >= H) {
            set_error("rtmi: row block starts outside the image");
            return RTMI_ERR_BAD_ARG;
        }
        n_local_rows += (uint32_t)std::min<uint64_t>(block_rows, H - y);
        if (y + block_rows > H && k + 1 != n_blocks) {
            set_error("rtmi: only the last row block may be clipped");
            return RTMI_ERR_BAD_ARG;
        }
    }
    return RTMI_OK;
}

// The probe behind the cost-ordered hand-out: the whole image at 2 samples per pixel through the counting variant of the kernel
// (whole-pixel work items, no outputs), every work item adding its segment count to its 8x8 tile.  Once per scene, made by
// rtmi_scene_create on the scene's own stream (round 6: the render entries are asynchronous from the first call on; rounds 5 made
// it inside the first large call, on the caller's stream, blocking); a failure only switches the ordering off.
void probe_tile_costs(rtmi_scene* s, hipStream_t stream) {
    s->cost_state = -1;
    const uint32_t W = s->cam.img_width, H = s->cam.img_height;
    const uint32_t gtx = (W + 7u) / 8u, gty = (H + 7u) / 8u;
    const size_t n = (size_t)gtx * gty;
    if (n == 0 || n * 64u > 0xffffffffull - (1ull << 24)) return; // (the bound of launch_one: room for the refills past the end)
    uint32_t* d_cost = nullptr;
    unsigned long long* d_pstats = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] {
        hipFree(d_cost);
        hipFree(d_pstats);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        (void)hipGetLastError();
    };
    if (hipMalloc(reinterpret_cast<void**>(&d_cost), n * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_pstats), 128 * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        cleanup();
        return;
    }
    RtmiLaunch P{};
    fill_scene_params(s, P);
    P.cam.samples_per_pixel = (uint16_t)std::min<uint32_t>(2u, s->cam.samples_per_pixel);
    P.y_first = 0;
    P.block_rows = H;
    P.block_stride = 1;
    P.n_local_rows = H;
    P.x_first = 0;
    P.x_end = W;
    P.local_w = W;
    P.tiles_x = gtx;
    P.tiles_y = gty;
    P.chunk = P.cam.samples_per_pixel;
    P.n_chunks = 1;
    P.n_work = (uint32_t)(n * 64u);
    P.div_tiles_x = make_fastdiv(P.tiles_x);
    P.div_chunks = make_fastdiv(1u);
    P.div_block_rows = make_fastdiv(H);
    P.seed = mix_seed(0x636f7374ull);
    P.work_counter = s->slot[0].d_counter;
    P.att_stack = s->slot[0].d_att;
    P.stats = d_pstats; // (not the caller's counters)
    P.tile_cost = d_cost;
    KernelFn fn = pick_kernel(s->accel, true, s->big, 3);
    void* args[] = {&P};
    bool ok = hipMemsetAsync(d_cost, 0, n * sizeof(uint32_t), stream) == hipSuccess &&
              hipMemsetAsync(d_pstats, 0, 128 * sizeof(unsigned long long), stream) == hipSuccess &&
              hipMemsetAsync(s->slot[0].d_counter, 0, 4 * sizeof(uint32_t), stream) == hipSuccess &&
              hipEventRecord(e0, stream) == hipSuccess &&
              hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(s->grid), dim3(s->block), args, s->lds_bytes, stream) == hipSuccess &&
              hipEventRecord(e1, stream) == hipSuccess;
    std::vector<uint32_t> cost(n);
    unsigned long long pst[2] = {0ull, 0ull}; // {samples, segments} of the probe
    ok = ok && hipMemcpyAsync(cost.data(), d_cost, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
         hipMemcpyAsync(pst, d_pstats, sizeof(pst), hipMemcpyDeviceToHost, stream) == hipSuccess &&
         hipStreamSynchronize(stream) == hipSuccess;
    if (ok) {
        s->probe_seg_per_sample = pst[0] ? (double)pst[1] / (double)pst[0] : 1.0;
        (void)hipEventElapsedTime(&s->probe_ms, e0, e1);
        s->tile_cost.swap(cost);
        s->cost_state = 1;
    }
    cleanup();
}

// first image row of local block k of a launch: a strided set (y_first + k * block_stride * block_rows) or a list of block indices
inline uint32_t block_row(uint32_t k, uint32_t y_first, uint32_t block_rows, uint32_t block_stride, const uint32_t* block_list) {
    return block_list ? block_list[k] * block_rows : y_first + k * block_stride * block_rows;
}

// What a launch geometry needs on the device, cached per geometry: the hand-out order of its 8x8 tiles, costliest first (longest
// processing time first: what is left for the end of the launch are the cheapest items it has; a local tile takes the cost of the
// image tile its first pixel lies in -- row blocks of 8 rows starting at a multiple of 8, the multi-GPU shards, coincide with image
// tiles), and for a launch over a LIST of row blocks the first row of each.  Both tables are uploaded on the stream of the call
// that makes the entry, in front of its kernel; the host copies live as long as the entry.  NULL: nothing to hand to the kernel.
const rtmi_scene::OrderEntry* order_for(rtmi_scene* s, uint32_t y_first, uint32_t block_rows, uint32_t block_stride, uint32_t n_blocks,
                                        const uint32_t* block_list, uint32_t n_local_rows, uint32_t x0, uint32_t x1, bool want_order,
                                        hipStream_t stream) {
    const bool ordered = want_order && s->cost_state == 1;
    if (!ordered && !block_list) return nullptr;
    uint64_t h = 0;
    if (block_list) {
        h = 0xcbf29ce484222325ull; // FNV-1a over the list
        for (uint32_t k = 0; k < n_blocks; ++k) h = (h ^ block_list[k]) * 0x100000001b3ull;
        h |= 1ull;
    }
    const uint32_t key[8] = {block_list ? 0u : y_first, block_rows, block_list ? 0u : block_stride, n_blocks, x0, x1, (uint32_t)h, (uint32_t)(h >> 32)};
    for (const auto& e : s->orders)
        if (std::memcmp(e->key, key, sizeof(key)) == 0 && (e->d_order != nullptr) == ordered &&
            (!block_list || (e->h_first_row.size() == n_blocks && [&] { for (uint32_t k = 0; k < n_blocks; ++k) if (e->h_first_row[k] != block_list[k] * block_rows) return false; return true; }())))
            return e.get();
    const uint32_t W = s->cam.img_width;
    const uint32_t gtx = (W + 7u) / 8u;
    const uint32_t tiles_x = (x1 - x0 + 7u) / 8u, tiles_y = (n_local_rows + 7u) / 8u;
    const size_t n = (size_t)tiles_x * tiles_y;
    auto e = std::make_unique<rtmi_scene::OrderEntry>();
    std::memcpy(e->key, key, sizeof(key));
    e->n_tiles = (uint32_t)n;
    e->cost = 0.0;
    e->d_order = nullptr;
    e->d_first_row = nullptr;
    auto fail = [&]() -> const rtmi_scene::OrderEntry* {
        hipFree(e->d_order);
        hipFree(e->d_first_row);
        (void)hipGetLastError();
        return nullptr;
    };
    if (ordered) {
        std::vector<uint64_t> keyed(n); // cost << 32 | (0xffffffff - tile): descending sort = costliest first, ties in tile order
        for (uint32_t ty = 0; ty < tiles_y; ++ty) {
            const uint32_t ply = ty * 8u, blk = ply / block_rows;
            const uint32_t gy = block_row(blk, y_first, block_rows, block_stride, block_list) + (ply - blk * block_rows);
            for (uint32_t tx = 0; tx < tiles_x; ++tx) {
                const size_t g = (size_t)(gy >> 3) * gtx + ((x0 + tx * 8u) >> 3);
                const uint32_t c = g < s->tile_cost.size() ? s->tile_cost[g] : 0u;
                const uint32_t tile = ty * tiles_x + tx;
                keyed[tile] = ((uint64_t)c << 32) | (0xffffffffu - tile);
                e->cost += c;
            }
        }
        std::sort(keyed.begin(), keyed.end(), std::greater<uint64_t>());
        e->h_order.resize(n);
        for (size_t i = 0; i < n; ++i) e->h_order[i] = 0xffffffffu - (uint32_t)keyed[i];
        if (hipMalloc(reinterpret_cast<void**>(&e->d_order), std::max<size_t>(n, 1) * sizeof(uint32_t)) != hipSuccess ||
            hipMemcpyAsync(e->d_order, e->h_order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream) != hipSuccess)
            return fail();
    }
    if (block_list) {
        e->h_first_row.resize(n_blocks);
        for (uint32_t k = 0; k < n_blocks; ++k) e->h_first_row[k] = block_list[k] * block_rows;
        if (hipMalloc(reinterpret_cast<void**>(&e->d_first_row), std::max<size_t>(n_blocks, 1) * sizeof(uint32_t)) != hipSuccess ||
            hipMemcpyAsync(e->d_first_row, e->h_first_row.data(), n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, stream) != hipSuccess)
            return fail();
    }
    s->orders.push_back(std::move(e));
    return s->orders.back().get();
}

// one launch sequence (trace + resolve) over columns [x0, x1) of a set of row blocks; `band`: position within the call (every
// band's trace kernel sits between its own pair of events)
int launch_one(rtmi_scene* s, uint32_t y_first, uint32_t block_rows, uint32_t block_stride, uint32_t n_blocks, const uint32_t* block_list,
               uint32_t x0, uint32_t x1, uint64_t seed, float* d_rgb, uint32_t* d_rgba, hipStream_t stream, uint32_t band,
               const rtmi_scene::OrderEntry* order, size_t per_slot_cap) {
    LaunchSlot& sl = s->slot[0];
    const uint32_t W = s->cam.img_width;
    uint32_t n_local_rows = 0;
    const int rc_rows = count_rows(s, y_first, block_rows, block_stride, n_blocks, n_local_rows, block_list);
    if (rc_rows != RTMI_OK) return rc_rows;
    if (n_local_rows == 0 || x1 <= x0) return RTMI_OK;
    const uint32_t Wl = x1 - x0;

    RtmiLaunch P{};
    fill_scene_params(s, P);
    P.y_first = y_first;
    P.block_rows = block_rows;
    P.block_stride = block_stride;
    P.n_local_rows = n_local_rows;
    P.x_first = x0;
    P.x_end = x1;
    P.local_w = Wl;
    P.tiles_x = (Wl + 7u) / 8u;
    P.tiles_y = (n_local_rows + 7u) / 8u;
    if (P.tiles_x > 0xffffu || P.tiles_y > 0xffffu) {
        set_error("rtmi: image too large for one launch (more than 65535 tiles of 8 pixels in a row or a column)");
        return RTMI_ERR_UNSUPPORTED;
    }
    // sample-chunk split: the cost of a pixel is heavy-tailed (paths trapped in the ground sphere run 50 bounces), so a
    // launch whose work items are whole pixels ends in a long tail (29 % of a 1080p x 512 spp frame, measured); items
    // of `chunk` samples cut it by spp / chunk at the price of 16 B per sample written once and read once.
    const uint32_t spp = s->cam.samples_per_pixel;
    P.chunk = spp;
    P.n_chunks = 1;
    P.sample_buf = nullptr;
    const size_t sample_floats = (size_t)n_local_rows * Wl * spp; // float4 records
    // chunk size: the end of a launch is a tail of lanes finishing their last item while the others idle, so items must be
    // short next to the launch: ~128 work items per lane of the persistent grid, never below 4 samples (round 4: the eighth of
    // the frame one of 8 GPUs renders, trace-kernel ms for chunks of 2 / 3 / 4 / 5 / 6 / 8: 19.26 / 18.30 / 17.90 / 17.96 /
    // 18.29 / 18.82; whole frame 4: 223.1, 8: 219.1, 16: 217.8, 24: 217.6, 32: 219.0, 86: 235.4 on the round-2 kernel).
    if (block_list && (!order || !order->d_first_row)) {
        set_error("rtmi: the table of a block-list launch could not be allocated");
        return RTMI_ERR_OOM;
    }
    P.block_first_row = block_list ? order->d_first_row : nullptr;
    const rtmi_scene::OrderEntry* ord = (order && order->d_order && order->n_tiles == P.tiles_x * P.tiles_y) ? order : nullptr;
    P.tile_order = ord ? ord->d_order : nullptr;
    // (with the costliest tiles first the heavy items start when the launch does and it ends in the cheapest ones: items can be
    // longer -- ~24 per lane, at most 32 samples; tools/sched_ab.py, trace + resolve ms without / with the order at the best chunk
    // of each: 1080p x 512 spp 134.2 (21) / 132.9 (32), its eighth 18.56 (4) / 17.76 (16), 1200 x 675 x 100 spp 11.80 (4) / 11.41 (8).
    // Cutting the cheap end of the order into shorter items than the rest was built and lost: profiles/r05_two_class_chunks.txt)
    uint32_t chunk = s->chunk;
    if (chunk == ~0u) {
        const uint64_t want_items = (ord ? 24ull : 128ull) * s->grid * s->block;
        const uint64_t pixels = (uint64_t)n_local_rows * Wl;
        const uint32_t n_chunks = (uint32_t)std::min<uint64_t>(spp, (want_items + pixels - 1) / pixels);
        chunk = std::max(4u, (spp + n_chunks - 1u) / std::max(1u, n_chunks));
        if (ord) {
            // ... and no longer than ~130 ray segments (the probe's mean per sample: 4 on S-RTOW, 79 in the box of config 5,
            // where 32 samples would be 2 500 rounds -- 20 ms -- of a lane: 848 ms against 840 with chunks of 4-16 at 1024 spp)
            const double seg_per_sample = s->probe_seg_per_sample; // (the frame's mean: the cost map itself is in time units since round 6)
            const uint32_t by_cost = (uint32_t)std::max(4.0, std::min(32.0, 130.0 / std::max(1.0, seg_per_sample)));
            chunk = std::min(chunk, by_cost);
        }
    }
    if (chunk && spp > chunk && sample_floats * sizeof(float4) > per_slot_cap) s->whole_pixel_fallbacks++; // (not even one unit of rows fits what the device has left)
    if (chunk && spp > chunk && sample_floats * sizeof(float4) <= per_slot_cap) {
        if (sample_floats > sl.samples_capacity) {
            hipFree(sl.d_samples);
            sl.d_samples = nullptr;
            sl.samples_capacity = 0;
            if (hipMalloc(reinterpret_cast<void**>(&sl.d_samples), sample_floats * sizeof(float4)) == hipSuccess) {
                sl.samples_capacity = sample_floats;
            } else {
                (void)hipGetLastError(); // not enough HBM for the split: fall back to whole-pixel work items
                s->whole_pixel_fallbacks++; // (reported by rtmi_scene_get_launch_info: same image, a longer tail)
            }
        }
        if (sl.d_samples) {
            P.chunk = chunk;
            P.n_chunks = (spp + chunk - 1u) / chunk;
            P.sample_buf = sl.d_samples;
        }
    }
    // packed chains travel with the sample records: att_words words per sample next to the 16-byte record
    int mode = P.sample_buf ? 0 : 3;
    const bool chain_slots_fit = (sample_floats * (sizeof(float4) + (size_t)s->att_words * 4u)) <= per_slot_cap;
    if (P.sample_buf && s->packed_ok && !chain_slots_fit) s->packed_chain_fallbacks++;
    if (P.sample_buf && s->packed_ok && chain_slots_fit) {
        if (sample_floats > sl.chain_capacity) {
            hipFree(sl.d_chain);
            sl.d_chain = nullptr;
            sl.chain_capacity = 0;
            if (hipMalloc(reinterpret_cast<void**>(&sl.d_chain), sample_floats * (size_t)s->att_words * 4u) == hipSuccess) {
                sl.chain_capacity = sample_floats;
            } else {
                (void)hipGetLastError(); // no room: the run-length encoded chains of mode 0 instead
                s->packed_chain_fallbacks++; // (reported by rtmi_scene_get_launch_info: same image, chains multiplied at path end)
            }
        }
        if (sl.d_chain) {
            mode = 4;
            P.att_bits = s->att_bits;
            P.att_epw = s->att_epw;
            P.att_words = s->att_words;
            P.chain_buf = sl.d_chain;
        }
    }
    const uint64_t n_work = (uint64_t)P.tiles_x * P.tiles_y * 64u * P.n_chunks;
    if (n_work > 0xffffffffull - (1ull << 24)) { // (room for the refills every wave makes past the end)
        set_error("rtmi: image too large for one launch");
        return RTMI_ERR_UNSUPPORTED;
    }
    P.n_work = (uint32_t)n_work;
    P.div_tiles_x = make_fastdiv(P.tiles_x);
    P.div_chunks = make_fastdiv(P.n_chunks);
    P.div_block_rows = make_fastdiv(P.block_rows);

    P.seed = mix_seed(seed); // two key words, see rng4x32
    P.out_rgb = d_rgb;
    P.out_rgba = d_rgba;
    P.work_counter = sl.d_counter;
    P.att_stack = sl.d_att;

    HIP_TRY(hipMemsetAsync(sl.d_counter, 0, 4 * sizeof(uint32_t), stream));
    while (s->ev_trace.size() < 2u * (band + 1u)) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        s->ev_trace.push_back(e);
    }
    HIP_TRY(hipEventRecord(s->ev_trace[2u * band], stream));
    {
        KernelFn fn = pick_kernel(s->accel, s->collect_stats, s->big, mode);
        void* args[] = {&P};
        // (a call of a few tiles -- the reference's 8x8 work packages through rtmi_render_rect -- does not start 512 workgroups
        // that find the counter empty)
        const uint64_t groups_needed = (n_work + s->block - 1u) / s->block;
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1u, std::min<uint64_t>(s->grid, groups_needed));
        HIP_TRY(hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(grid), dim3(s->block), args, s->lds_bytes, stream));
    }
    HIP_TRY(hipEventRecord(s->ev_trace[2u * band + 1u], stream));
    if (P.sample_buf) {
        ResolveArgs A{};
        A.sample_buf = P.sample_buf;
        A.n_pixels = n_local_rows * Wl;
        A.spp = spp;
        A.scale = s->cam.pixels_sample_scale;
        A.out_rgb = d_rgb;
        A.out_rgba = d_rgba;
        const dim3 rgrid((A.n_pixels + RTMI_RESOLVE_BLOCK - 1u) / RTMI_RESOLVE_BLOCK);
        if (mode == 4) {
            A.chain_buf = P.chain_buf;
            A.mats = s->d_mats;
            A.n_mats = s->n_mats;
            A.bits = s->att_bits;
            A.epw = s->att_epw;
            A.words = s->att_words;
            A.div_epw = make_fastdiv(s->att_epw);
            const size_t lds = (((size_t)s->n_mats * sizeof(uint4) + 15u) & ~(size_t)15u) + 4u * kResPix * kResRow * sizeof(float4);
            const uint32_t groups = (A.n_pixels + kResPix - 1u) / kResPix;
            const uint32_t blocks = std::min<uint32_t>((groups + 3u) / 4u, s->n_cus * 8u);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(rtmi_resolve_chain_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(rtmi_resolve_chain_kernel, dim3(std::max(1u, blocks)), dim3(256), lds, stream, A);
        } else {
            hipLaunchKernelGGL(rtmi_resolve_kernel, rgrid, dim3(RTMI_RESOLVE_BLOCK), 0, stream, A);
        }
        HIP_TRY(hipGetLastError());
    }
    (void)W;
    return RTMI_OK;
}

// A call whose sample records pass the cap is rendered in bands of rows (whole row blocks for a sharded call), one launch
// sequence each, one after the other (config 5: 800 x 800 x 4096 spp = 251 GB of records and chain slots: three bands under
// the 96 GB cap of a 288 GB device).  The draw streams are keyed by the absolute pixel: banding does not change a bit of the image.
int launch(rtmi_scene* s, uint32_t y_first, uint32_t block_rows, uint32_t block_stride, uint32_t n_blocks, uint32_t x0, uint32_t x1,
           uint64_t seed, float* d_rgb, uint32_t* d_rgba, hipStream_t stream, const uint32_t* block_list = nullptr) {
    const uint32_t H = s->cam.img_height, W = s->cam.img_width, spp = s->cam.samples_per_pixel;
    if (x0 > x1 || x1 > W) {
        set_error("rtmi: columns outside the image");
        return RTMI_ERR_BAD_ARG;
    }
    if (block_list && (block_rows == 0u || (block_rows & 7u) != 0u)) {
        set_error("rtmi: a list of row blocks needs block_rows to be a multiple of 8 (the kernel's tiles must not straddle blocks)");
        return RTMI_ERR_BAD_ARG;
    }
    uint32_t rows_total = 0;
    const int rc_rows = count_rows(s, y_first, block_rows, block_stride, n_blocks, rows_total, block_list);
    if (rc_rows != RTMI_OK) return rc_rows;
    if (rows_total == 0 || x1 == x0) return RTMI_OK;
    const uint32_t Wl = x1 - x0;

    // a band is a run of units: 8 rows of a contiguous call (tile rows stay whole), or one row block of a sharded call
    const bool contiguous = n_blocks == 1 && !block_list;
    const uint32_t rows_c = contiguous ? std::min(block_rows, H - y_first) : 0u;
    const uint64_t sample_bytes = sizeof(float4) + (s->packed_ok ? (size_t)s->att_words * 4u : 0u);
    const uint64_t row_bytes = (uint64_t)Wl * spp * sample_bytes;
    struct Band {
        uint32_t y_first, block_rows, n_blocks; // (block_stride: the call's)
        size_t row0;                            // first row of the band in the call's dense output
        uint32_t rows;
        const rtmi_scene::OrderEntry* order;
        const uint32_t* list;                   // the band's part of the call's block list (NULL: strided)
    };
    std::vector<Band> bands;
    uint32_t n_bands = 1;
    // The cap on one band's records (+ chain slots): rtmi_tuning::sample_buf_mb or a third of the device's memory, and never more than
    // the device has left beside what this scene's slot already holds (other scenes' retained buffers, the caller's own allocations,
    // another process): round 5's unrecorded abort was a tree that sized three such buffers from the device's TOTAL
    // memory.  The largest band's buffers are allocated here, before anything is launched; if the device refuses them the plan is
    // made again for half the cap -- more, smaller bands, the same image -- and only a cap below one unit of rows falls back to
    // whole-pixel work items / run-length chains (launch_one; counted in rtmi_launch_info).
    LaunchSlot& sl = s->slot[0];
    size_t cap = s->sample_buf_cap_bytes;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t held = sl.samples_capacity * sizeof(float4) + sl.chain_capacity * (size_t)s->att_words * 4u;
            const size_t reserve = std::max<size_t>((size_t)512 << 20, total_b / 64u);
            const size_t avail = free_b + held;
            cap = std::min(cap, avail > reserve ? avail - reserve : (size_t)0);
        } else {
            (void)hipGetLastError();
        }
    }
    for (int attempt = 0;; ++attempt) {
        const uint64_t max_rows = row_bytes ? cap / row_bytes : 0;
        const bool split_on = s->chunk != 0u && spp > 4u && max_rows >= 1;
        const uint32_t unit_rows = contiguous ? (max_rows >= 8 ? 8u : 1u) : block_rows;
        const uint32_t n_units = contiguous ? (rows_c + unit_rows - 1u) / unit_rows : n_blocks;
        n_bands = 1;
        if (split_on && unit_rows <= max_rows) {
            const uint64_t units_per_band = std::max<uint64_t>(1u, max_rows / unit_rows);
            n_bands = (uint32_t)((n_units + units_per_band - 1) / units_per_band);
            n_bands = std::min(n_units, std::max(n_bands, s->min_bands)); // (rtmi_tuning::bands asks for more than memory does: tests, experiments)
        }
        const uint32_t per_band = (n_units + n_bands - 1u) / n_bands;
        n_bands = (n_units + per_band - 1u) / per_band;
        bands.assign(n_bands, Band{});
        uint32_t max_band_rows = 0;
        for (uint32_t k = 0; k < n_bands; ++k) {
            const uint32_t u0 = k * per_band, nu = std::min(per_band, n_units - u0);
            Band& b = bands[k];
            if (contiguous) {
                const uint32_t r0 = u0 * unit_rows, nr = std::min(nu * unit_rows, rows_c - r0);
                b = Band{y_first + r0, nr, 1u, (size_t)r0, nr, nullptr, nullptr};
            } else {
                uint32_t rows = 0;
                const uint32_t* sub = block_list ? block_list + u0 : nullptr;
                const int rc = count_rows(s, y_first + u0 * block_stride * block_rows, block_rows, block_stride, nu, rows, sub);
                if (rc != RTMI_OK) return rc;
                b = Band{y_first + u0 * block_stride * block_rows, block_rows, nu, (size_t)u0 * block_rows, rows, nullptr, sub};
            }
            max_band_rows = std::max(max_band_rows, b.rows);
        }
        // the buffers of the largest band, now (a band that fits the cap only; launch_one decides the rest exactly as before)
        const size_t need = (size_t)max_band_rows * Wl * spp;
        const bool wants_records = split_on && need * sizeof(float4) <= cap;
        if (!wants_records) break;
        bool ok = true;
        if (need > sl.samples_capacity) {
            hipFree(sl.d_samples); // (hipFree waits for the device: an earlier asynchronous call may still be reading the old buffer)
            sl.d_samples = nullptr;
            sl.samples_capacity = 0;
            if (hipMalloc(reinterpret_cast<void**>(&sl.d_samples), need * sizeof(float4)) == hipSuccess) sl.samples_capacity = need;
            else { (void)hipGetLastError(); ok = false; }
        }
        if (ok && s->packed_ok && need * sample_bytes <= cap && need > sl.chain_capacity) {
            hipFree(sl.d_chain);
            sl.d_chain = nullptr;
            sl.chain_capacity = 0;
            if (hipMalloc(reinterpret_cast<void**>(&sl.d_chain), need * (size_t)s->att_words * 4u) == hipSuccess) sl.chain_capacity = need;
            else { (void)hipGetLastError(); ok = false; }
        }
        if (ok) break;
        if (attempt >= 12 || cap / 2u < row_bytes) break; // (launch_one falls back and counts it)
        cap /= 2u;
        s->reband_retries++;
    }
    // cost-ordered tiles: where a launch has tiles to order and samples enough for the probe to be small next to it, and the
    // scene is in LDS -- a tree read through the caches wants neighbouring tiles in flight together (config 4, 100k spheres, 64 spp:
    // 75.7 ms ordered by cost against 73.7 ms row by row)
    const uint64_t tiles_total = (uint64_t)((Wl + 7u) / 8u) * ((rows_total + 7u) / 8u);
    const bool want_order = s->tile_order_mode == 2u || (s->tile_order_mode == 0u && !s->big && tiles_total >= 2048u && spp >= 16u);
    s->last_tile_order = 0;
    if ((want_order && n_bands <= 256u) || block_list) {
        if (s->orders.size() + n_bands > 512u) { // (a host that walks through many geometries: start over rather than grow)
            HIP_TRY(hipDeviceSynchronize());
            for (auto& e : s->orders) { hipFree(e->d_order); hipFree(e->d_first_row); }
            s->orders.clear();
        }
        for (Band& b : bands)
            b.order = order_for(s, b.y_first, b.block_rows, block_stride, b.n_blocks, b.list, b.rows, x0, x1, want_order && n_bands <= 256u, stream);
        s->last_tile_order = (s->cost_state == 1 && bands[0].order && bands[0].order->d_order) ? 1u : 0u;
    }
    s->last_bands = n_bands;
    s->n_bands_timed = 0;
    s->ev_valid = false;
    auto out_rgb = [&](const Band& b) { return d_rgb ? d_rgb + b.row0 * Wl * 3 : nullptr; };
    auto out_rgba = [&](const Band& b) { return d_rgba ? d_rgba + b.row0 * Wl : nullptr; };
    for (uint32_t k = 0; k < n_bands; ++k) {
        const Band& b = bands[k];
        const int rc = launch_one(s, b.y_first, b.block_rows, block_stride, b.n_blocks, b.list, x0, x1, seed, out_rgb(b), out_rgba(b), stream, k, b.order, cap);
        if (rc != RTMI_OK) return rc;
        s->n_bands_timed = k + 1u;
    }
    s->ev_valid = true;
    return RTMI_OK;
}

} // namespace

// `s` is handed back to the wrapper below, which frees it on any failure (status or exception)
static int scene_create_impl(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects,
                             const rtmi_material* materials, uint32_t n_materials, const rtmi_scene_options* options,
                             rtmi_scene*& s) {
    if (camera->samples_per_pixel == 0) {
        set_error("rtmi_scene_create: samples_per_pixel must be at least 1");
        return RTMI_ERR_BAD_ARG;
    }
    for (uint32_t i = 0; i < n_objects; ++i) {
        if (objects[i].kind != 0u) {
            set_error("rtmi_scene_create: unknown HittableObjectKind (only Sphere = 0 exists)");
            return RTMI_ERR_BAD_ARG;
        }
        if (objects[i].material >= n_materials) { // the reference asserts this, material.defs.hpp:104
            set_error("rtmi_scene_create: object refers to a material handle outside the collection");
            return RTMI_ERR_BAD_ARG;
        }
    }
    for (uint32_t i = 0; i < n_materials; ++i) {
        if (materials[i].kind > 2u) {
            set_error("rtmi_scene_create: unknown MaterialKind");
            return RTMI_ERR_BAD_ARG;
        }
    }
    rtmi_scene_options opt{};
    opt.device = -1; // the caller's current device, also for options == NULL or a short struct
    if (options) std::memcpy(&opt, options, std::min<size_t>(sizeof(opt), options->struct_size));
    rtmi_tuning tune{};
    if (opt.tuning) std::memcpy(&tune, opt.tuning, std::min<size_t>(sizeof(tune), opt.tuning->struct_size));
    for (uint32_t i = 0; i < n_objects; ++i) {
        const rtmi_object& o = objects[i];
        if (!std::isfinite(o.center[0]) || !std::isfinite(o.center[1]) || !std::isfinite(o.center[2]) ||
            !std::isfinite(o.radius)) {
            set_error("rtmi_scene_create: object with a non-finite centre or radius");
            return RTMI_ERR_BAD_ARG;
        }
    }

    s = new (std::nothrow) rtmi_scene();
    if (!s) {
        set_error("rtmi_scene_create: out of host memory");
        return RTMI_ERR_OOM;
    }
    auto fail = [&](int rc) { return rc; };
#define HIP_TRY_S(expr)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                \
            return fail(e_ == hipErrorOutOfMemory ? RTMI_ERR_OOM : RTMI_ERR_HIP);        \
        }                                                                                \
    } while (0)

    int dev = opt.device;
    if (dev < 0) HIP_TRY_S(hipGetDevice(&dev));
    HIP_TRY_S(hipSetDevice(dev));
    s->device = dev;
    s->cam = *camera;
    s->collect_stats = opt.collect_stats != 0;
    s->n_objects = n_objects;
    s->n_mats = n_materials;
    // (the walk overtakes the scan between 20 and 30 spheres: tools/scan_crossover.py on MI355X, random scenes of 4 .. 61
    // spheres: 1.03-1.09x slower up to 21, 0.95x at 29, 0.73x at 61; the 7-sphere box of config 5 is 1.30x slower walked)
    s->accel = opt.accel == RTMI_ACCEL_AUTO ? (n_objects > 24 ? RTMI_ACCEL_BVH : RTMI_ACCEL_BRUTE) : opt.accel;
    if (n_objects == 0) s->accel = RTMI_ACCEL_BRUTE; // an empty world has no tree; the scan over zero spheres misses
    if (s->accel != RTMI_ACCEL_BVH && s->accel != RTMI_ACCEL_BRUTE) {
        set_error("rtmi_scene_create: unknown accel");
        return fail(RTMI_ERR_BAD_ARG);
    }

    // slot order: leaf order for the BVH, insertion order for the linear scan
    std::vector<uint32_t> slot_object(n_objects);
    if (s->accel == RTMI_ACCEL_BVH) {
        // default leaf size: 2 for trees that live in LDS, 4 for trees that stay in HBM (tools/leaf_sweep.py on the 100k-sphere
        // grid: 1: 70.3, 2: 59.3, 3: 60.5, 4: 57.4, 5: 59.9, 6: 58.1, 8: 58.7 ms; on RTOW 1: 43.0, 2: 34.2, 3: 35.4, 4: 35.4 ms).
        // More than 8192 spheres cannot be in LDS, which is known before the build; scenes between ~640 and 8192 spheres, whose
        // residency depends on the tree, keep 2.
        const uint32_t leaf_default = n_objects > 0x2000u ? 4u : 2u;
        build_bvh(objects, n_objects, opt.leaf_size ? opt.leaf_size : leaf_default,
                  tune.bvh_passes ? tune.bvh_passes - 1u : default_bvh_passes(n_objects), s->bvh);
        slot_object = s->bvh.slot_object;
        if (n_objects >= 0x00ffffffu) {
            set_error("rtmi_scene_create: too many objects (leaf references hold 24-bit slots)");
            return fail(RTMI_ERR_UNSUPPORTED);
        }
    } else {
        for (uint32_t i = 0; i < n_objects; ++i) slot_object[i] = i;
        s->bvh.root_ref = 0;
    }
    s->n_tree_nodes = (uint32_t)s->bvh.nodes.size();
    std::vector<uint4> h_spheres(n_objects), h_aux(n_objects), h_mats((size_t)n_materials);
    auto fbits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    for (uint32_t i = 0; i < n_objects; ++i) {
        const rtmi_object& o = objects[slot_object[i]];
        // Radius * Radius of object.defs.cc:46 is the same float for every ray: computed once here
        const float r2 = o.radius * o.radius;
        h_spheres[i] = make_uint4(fbits(o.center[0]), fbits(o.center[1]), fbits(o.center[2]), fbits(r2));
        h_aux[i] = make_uint4(slot_object[i], o.material, fbits(o.radius), materials[o.material].kind);
    }
    for (uint32_t i = 0; i < n_materials; ++i) {
        const rtmi_material& m = materials[i];
        if (m.kind == 2u) {
            // Material_Dielectric::scatter, material.defs.cc:58 and :80-82: eta and the Schlick r1 for both faces, with the
            // operations (and roundings) the per-hit code would use
            const float ri = m.p[0];
            const float eta_front = 1.0f / ri;
            const float r0f = (1.0f - eta_front) / (1.0f + eta_front), r0b = (1.0f - ri) / (1.0f + ri);
            h_mats[i] = make_uint4(fbits(ri), fbits(eta_front), fbits(r0f * r0f), fbits(r0b * r0b));
        } else {
            h_mats[i] = make_uint4(fbits(m.p[0]), fbits(m.p[1]), fbits(m.p[2]), fbits(m.p[3]));
        }
    }

    if (tune.block_lanes) s->block = std::min(1024u, std::max(64u, (tune.block_lanes / 64u) * 64u));
    // LDS carve-up.  Small scenes live in LDS entirely (two workgroups per CU must fit: 80 KiB each); larger ones
    // stay in HBM and only the traversal stack is in LDS.
    auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
    s->stack_depth = s->accel == RTMI_ACCEL_BVH ? std::max(1u, s->bvh.depth) + 2u : 0u; // + the sentinel entry + one spare level (unconditional push store)
    // (the same expression as the carve-up below, per-wave pools and alignment included: a scene within a kilobyte of the
    // limit must not end up with one resident workgroup per CU instead of two)
    constexpr uint32_t kLdsNodeBytes = 48u; // one record format on the device, in LDS and in HBM
    const uint64_t scene_bytes = (uint64_t)s->bvh.nodes.size() * kLdsNodeBytes + (uint64_t)n_objects * 32u + (uint64_t)n_materials * 16u;
    const uint64_t small_total = ((scene_bytes + 15u) & ~15ull) + (((uint64_t)s->stack_depth * s->block * 2u + 15u) & ~15ull) +
                                 (uint64_t)kAttLds * s->block * 4u + (uint64_t)(s->block / 64u) * 80u + 16u;
    s->big = small_total > 80u * 1024u || s->bvh.nodes.size() >= 0x8000u || n_objects > 0x2000u || n_materials > 0x10000u;
    if (tune.force_hbm_scene) s->big = true;
    // HBM-resident scenes: the largest workgroup of which two fit the LDS with their stacks, up to 896 lanes (7 waves per
    // SIMD; 1024 would need 80 KB of stack at the depth of a 100k-sphere tree)
    if (RTMI_WPE_BIG > 6 && s->big && !tune.block_lanes && s->accel == RTMI_ACCEL_BVH) {
        s->block = 896u;
        while (s->block > 768u && (uint64_t)s->stack_depth * s->block * 4u + (s->block / 64u) * 80u + 64u > 80u * 1024u) s->block -= 64u;
    }
    // top of the tree = a spine of (leaf | subtree) nodes: hand up to four such leaves to segment set-up and start every walk below them
    uint32_t walk_root = s->bvh.root_ref, pre[4] = {};
    if (s->accel == RTMI_ACCEL_BVH && !s->bvh.nodes.empty()) walk_root = peel_top_leaves(s->bvh, pre, s->n_pre_leaves);
    // HBM-resident trees: how many 48-byte records the stacks leave room for in the 80 KiB of a workgroup (two per CU): the 100k-sphere
    // tree of config 4 has 18 levels = 61 KB of 32-bit stack entries for 768 lanes, which leaves 384 (rtmi_tuning::lds_top_nodes caps
    // it: n > 0 = at most n - 1)
    uint64_t k_budget = 0;
    if (s->big && s->accel == RTMI_ACCEL_BVH && !s->bvh.nodes.empty()) {
        const uint64_t fixed = (uint64_t)s->stack_depth * s->block * 4u + (uint64_t)(s->block / 64u) * 80u + 64u;
        k_budget = fixed < 80u * 1024u ? (80u * 1024u - fixed) / 48u : 0u;
        if (tune.lds_top_nodes) k_budget = std::min<uint64_t>(k_budget, tune.lds_top_nodes - 1u);
    }
    // Scattered rays of trees that stay in HBM start in their own leaf (rtmi_tuning::walk_start): way records behind the nodes.
    // The way records of the TOP levels (two levels a record: 4, 16, 64, 256 possible records for the pairs down to depth 8) are
    // placed by their path code in a block of the staged records -- `way_jtop` levels, as many as fit while five levels of nodes
    // still do -- so that the device computes their indices and a start record only has to name the (at most four) deeper ones.
    std::vector<uint32_t> starts, node_perm, way_depth, way_code; // node_perm: builder's node index -> position in the device array (empty: identity)
    auto top_slots = [](uint32_t j) { return ((1u << (2u * (j + 1u))) - 4u) / 3u; }; // 4 + 16 + ... + 4^j
    uint32_t way_jtop = 0;
    while (way_jtop < 4u && top_slots(way_jtop + 1u) + std::min<uint64_t>(31u, s->n_tree_nodes) <= k_budget) ++way_jtop;
    if (s->accel == RTMI_ACCEL_BVH && s->big && tune.walk_start != 1u && walk_root != kNoWalkRef && !(walk_root & kLeafBit) &&
        !s->bvh.nodes.empty()) {
        build_walk_starts(s->bvh, walk_root, starts, &way_depth, &way_code, way_jtop + 4u);
        if (s->bvh.nodes.size() == s->n_tree_nodes) starts.clear(); // (no sphere has a way: a tree of two levels)
    }
    uint32_t off = 0;
    if (!s->big) {
        s->lds_nodes = off;
        off += (uint32_t)s->bvh.nodes.size() * kLdsNodeBytes;
        s->lds_spheres = off;
        off += n_objects * 16u;
        s->lds_aux = off;
        off += n_objects * 16u;
        s->lds_mats = off;
        off += n_materials * 16u;
        off = align16(off);
    } else if (s->accel == RTMI_ACCEL_BVH && !s->bvh.nodes.empty()) {
        // The staged records: the tree's first breadth-first nodes -- the levels every walk from the root passes through -- and, with
        // walk starts, the block of top way records (scattered rays never read the top nodes any more: with every way record in memory
        // the scheme measured +2 % SLOWER than walks from the root on config 4, whose first 8 levels had been LDS reads).  The device
        // array is a rearrangement of bvh.nodes: [top nodes | top ways by path code (sparse) | other nodes | deeper ways]; every
        // reference handed to the device goes through node_perm; what the library exports keeps the builder's numbering.
        uint64_t k = std::min<uint64_t>(k_budget, s->n_tree_nodes);
        s->n_dev_nodes = s->n_tree_nodes;
        if (!starts.empty()) {
            const uint32_t n_tree = s->n_tree_nodes, n_ways = (uint32_t)s->bvh.nodes.size() - n_tree;
            const uint32_t k2 = top_slots(way_jtop), k1 = (uint32_t)std::min<uint64_t>(n_tree, k_budget - k2);
            node_perm.assign(n_tree + n_ways, 0u);
            for (uint32_t i = 0; i < n_tree; ++i) node_perm[i] = i < k1 ? i : i + k2;
            uint32_t deep = 0;
            for (uint32_t w = 0; w < n_ways; ++w) {
                const uint32_t level = way_depth[w] / 2u; // 1 = the pair of levels below the walk's root
                if (level <= way_jtop) node_perm[n_tree + w] = k1 + (top_slots(level - 1u)) + way_code[w];
                else node_perm[n_tree + w] = n_tree + k2 + deep++;
            }
            s->n_dev_nodes = n_tree + k2 + deep;
            s->lds_top_ways = k2;
            s->way_jtop = way_jtop;
            s->way_top_base = k1;
            k = k1 + k2;
        }
        s->lds_top_nodes = (uint32_t)k;
        off = align16((uint32_t)k * 48u);
    }
    s->lds_stack = off;
    off += s->stack_depth * s->block * (s->big ? 4u : 2u);
    off = align16(off);
    s->lds_att = off;
    if (!s->big) {
        // packed attenuation chains when the strings of all lanes fit what the scene leaves of the 80 KiB (two workgroups per
        // CU): ceil(log2 n_materials) bits per bounce, no handle straddling a word, a multiple of four words per lane
        uint32_t bits = 1;
        while ((1u << bits) < n_materials) ++bits;
        const uint32_t epw = 32u / bits;
        const uint32_t words = ((std::max<uint32_t>(1u, camera->maxdepth) + epw - 1u) / epw + 3u) & ~3u;
        const uint64_t with_packed = (uint64_t)off + (uint64_t)words * s->block * 4u + (s->block / 64u) * 80u + 16u;
        s->packed_ok = tune.chain_mode != 1 && bits <= 16u && words <= 255u && with_packed <= 80u * 1024u &&
                       (uint64_t)n_materials * sizeof(uint4) <= 64u * 1024u;
        if (s->packed_ok) {
            s->att_bits = bits;
            s->att_epw = epw;
            s->att_words = words;
        }
        off += std::max(kAttLds, s->packed_ok ? words : 0u) * s->block * 4u;
    }
    s->lds_pool = off; // per wave: {work_next, work_end, slot_next, slot_end}
    off += (s->block / 64u) * 80u; // + 64-byte rank table of coop_draws
    s->lds_bytes = align16(off);
    if (s->lds_bytes > 160u * 1024u) {
        set_error("rtmi_scene_create: traversal stack does not fit the 160 KiB LDS of a CU (BVH too deep)");
        return fail(RTMI_ERR_UNSUPPORTED);
    }

    auto upload = [&](uint4** dptr, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(dptr), std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIP_TRY_S(upload(&s->d_spheres, h_spheres.data(), h_spheres.size() * sizeof(uint4)));
    HIP_TRY_S(upload(&s->d_aux, h_aux.data(), h_aux.size() * sizeof(uint4)));
    HIP_TRY_S(upload(&s->d_mats, h_mats.data(), h_mats.size() * sizeof(uint4)));
    {
        // device copy of the nodes: LDS-resident scenes get their references in the 16-bit stack-entry form
        std::vector<rtmi_bvh_node> dn = s->bvh.nodes;
        // (leaf references are stored sign-extended: as 32-bit integers nodes are >= 0, leaves < -1 and the stack's
        // sentinel -1 in both layouts, so the walk's votes are two compares against inline constants)
        auto pack16 = [](uint32_t ref) {
            return (ref & kLeafBit) ? (0xffff8000u | ((((ref >> 24) & 0x7fu) - 1u) << 13) | (ref & 0x1fffu)) : ref;
        };
        s->root_ref_dev = s->bvh.root_ref;
        if (!s->big && s->accel == RTMI_ACCEL_BVH && n_objects > 0) {
            for (auto& nd : dn) {
                nd.child[0] = pack16(nd.child[0]);
                nd.child[1] = pack16(nd.child[1]);
            }
            s->root_ref_dev = pack16(s->bvh.root_ref);
        }
        auto dev_ref = [&](uint32_t ref) {
            if (ref == kNoWalkRef) return kNoWalk;
            if (!s->big) return pack16(ref);
            return (!(ref & kLeafBit) && !node_perm.empty()) ? node_perm[ref] : ref;
        };
        if (s->accel == RTMI_ACCEL_BVH && !dn.empty()) {
            for (uint32_t q = 0; q < s->n_pre_leaves; ++q) s->pre_leaf_dev[q] = dev_ref(pre[q]);
            if (s->n_pre_leaves) s->root_ref_dev = dev_ref(walk_root);
        }
        if (!dn.empty()) {
            // 48-byte records: centres fp32, half extents fp16 rounded up (an extent beyond fp16 becomes +inf: always hit),
            // child references.  The exported tree (rtmi_scene_get_bvh) carries the rounded extents, so that an instrumented
            // CPU walk tests the same boxes.
            auto half_up = [](float v) -> uint16_t {
                _Float16 h = (_Float16)v; // round to nearest
                if ((float)h < v) {        // bump to the next fp16 above
                    uint16_t bits;
                    std::memcpy(&bits, &h, 2);
                    bits = (uint16_t)(bits + 1u); // v > 0 here: the next representable value (or +inf)
                    std::memcpy(&h, &bits, 2);
                }
                uint16_t out;
                std::memcpy(&out, &h, 2);
                return out;
            };
            auto half_to_float = [](uint16_t b) {
                _Float16 h;
                std::memcpy(&h, &b, 2);
                return (float)h;
            };
            std::vector<uint32_t> rec((size_t)std::max<uint32_t>(s->n_dev_nodes, (uint32_t)dn.size()) * 12u, 0u);
            for (size_t i = 0; i < dn.size(); ++i) {
                uint32_t* r = &rec[(node_perm.empty() ? i : (size_t)node_perm[i]) * 12u];
                for (int k = 0; k < 2; ++k)
                    for (int a = 0; a < 3; ++a) r[k * 3 + a] = fbits(dn[i].ctr[k][a]);
                uint16_t hb[6];
                for (int k = 0; k < 2; ++k)
                    for (int a = 0; a < 3; ++a) {
                        hb[k * 3 + a] = half_up(std::max(dn[i].half[k][a], 0.0f));
                        s->bvh.nodes[i].half[k][a] = half_to_float(hb[k * 3 + a]);
                    }
                r[6] = hb[0] | ((uint32_t)hb[1] << 16);
                r[7] = hb[2] | ((uint32_t)hb[3] << 16);
                r[8] = hb[4] | ((uint32_t)hb[5] << 16);
                r[9] = s->big ? dev_ref(dn[i].child[0]) : dn[i].child[0]; // (trees in LDS: packed above)
                r[10] = s->big ? dev_ref(dn[i].child[1]) : dn[i].child[1];
            }
            HIP_TRY_S(upload(&s->d_nodes, rec.data(), rec.size() * sizeof(uint32_t)));
            if (!starts.empty()) {
                // the device's start records: 16 bytes a slot {start reference | n, path code of the top levels, the deeper way indices
                // as 20-bit fields}: one load where the exported form has four, a table of 1.6 MB instead of 6.4 MB for 100k spheres
                const size_t n_sl = starts.size() / 16u;
                std::vector<uint32_t> dev_starts(n_sl * 4u, 0u);
                for (size_t sl = 0; sl < n_sl; ++sl) {
                    const uint32_t* r = &starts[sl * 16u];
                    uint32_t* d = &dev_starts[sl * 4u];
                    uint32_t n = std::min(r[1], s->way_jtop + 4u);
                    const uint32_t n_top = std::min(n, s->way_jtop);
                    uint64_t ids[4] = {0, 0, 0, 0};
                    bool fits = true;
                    for (uint32_t q = 0; n_top + q < n; ++q) {
                        ids[q] = dev_ref(r[2u + n_top + q]);
                        fits = fits && ids[q] < (1u << 20);
                    }
                    if (!fits || r[1] > s->way_jtop + 4u) { // (cannot be named in 20 bits / deeper than the builder was asked for: from the root)
                        d[0] = dev_ref(walk_root);
                        continue;
                    }
                    const uint32_t code_top = n_top ? (r[14] >> (2u * (n - n_top))) & 0xffu : 0u; // the path code at level n_top
                    d[0] = dev_ref(r[0]);
                    d[1] = n | (code_top << 4) | ((uint32_t)ids[0] << 12);
                    d[2] = (uint32_t)ids[1] | ((uint32_t)(ids[2] & 0xfffu) << 20);
                    d[3] = (uint32_t)(ids[2] >> 12) | ((uint32_t)ids[3] << 8);
                }
                HIP_TRY_S(upload(&s->d_walk_starts, dev_starts.data(), dev_starts.size() * sizeof(uint32_t)));
                s->walk_starts.swap(starts);
            }
            // camera rays start at their tile's entry (rtmi_tuning::cam_entry; host: build_tile_entries, csrc/rtmi_host.cpp)
            if (s->accel == RTMI_ACCEL_BVH && (tune.cam_entry == 2u || (tune.cam_entry == 0u && (!s->big || s->d_walk_starts != nullptr))) && walk_root != kNoWalkRef && !(walk_root & kLeafBit)) {
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<uint32_t> entries;
                build_tile_entries(*camera, objects, s->bvh, walk_root, entries);
                s->tile_entries = entries;
                for (uint32_t& e : entries) e = dev_ref(e);
                s->entry_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                uint4* d = nullptr;
                HIP_TRY_S(upload(&d, entries.data(), entries.size() * sizeof(uint32_t)));
                s->d_tile_entry = reinterpret_cast<uint32_t*>(d);
            }
        } else {
            HIP_TRY_S(upload(&s->d_nodes, dn.data(), dn.size() * sizeof(rtmi_bvh_node)));
        }
    }

    // persistent grid: exactly as many workgroups as the device keeps resident
    KernelFn fn = pick_kernel(s->accel, s->collect_stats, s->big, 0);
    HIP_TRY_S(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)s->lds_bytes));
    HIP_TRY_S(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_kernel(s->accel, s->collect_stats, s->big, 3)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    if (s->packed_ok) {
        HIP_TRY_S(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_kernel(s->accel, s->collect_stats, s->big, 4)),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    }
    // (the probe behind the cost-ordered hand-out launches the counting variant with whole-pixel work items)
    HIP_TRY_S(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_kernel(s->accel, true, s->big, 3)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    int per_cu = 0;
    HIP_TRY_S(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)s->block, s->lds_bytes));
    if (per_cu < 1) {
        set_error("rtmi_scene_create: kernel cannot be resident with this scene");
        return fail(RTMI_ERR_UNSUPPORTED);
    }
    hipDeviceProp_t prop;
    HIP_TRY_S(hipGetDeviceProperties(&prop, dev));
    if (tune.blocks_per_cu) per_cu = std::max(1, std::min(per_cu, (int)tune.blocks_per_cu));
    if (!s->big && s->accel == RTMI_ACCEL_BVH && s->d_tile_entry != nullptr) s->wait_thresh = 56u;
    // (trees in HBM whose scattered rays start in their own leaf: those lanes are done after a few trips, and a round that leaves the
    // loop earlier lets them shade while the camera and ground rays carry their walks over -- config 4 at 64 spp, way on, threshold
    // 52 / 48 / 44 / 40: 75.0 / 72.2 / 71.3 / 71.1 ms, against 74.2 - 75.9 for walks from the root at any of them)
    if (s->big && s->accel == RTMI_ACCEL_BVH && s->d_walk_starts != nullptr) s->wait_thresh = 44u;
    if (tune.wait_thresh) s->wait_thresh = std::min(64u, tune.wait_thresh);

    if (tune.chunk_samples) s->chunk = tune.chunk_samples < 0 ? 0u : (uint32_t)tune.chunk_samples; // 0: split off
    // cap on the sample records (+ chain slots) of one band: a third of the device's memory (96 GB on a 288 GB MI355X: the 251 GB
    // of the config-5 frame in three bands, each of which ends in its own tail and resolve pass, where the 24 GB of rounds 2-4 made ten)
    s->sample_buf_cap_bytes = std::max<size_t>((size_t)1 << 30, (size_t)prop.totalGlobalMem / 3u);
    if (tune.sample_buf_mb) s->sample_buf_cap_bytes = (size_t)tune.sample_buf_mb << 20;
    s->top_down = tune.top_down != 0;
    if (tune.tile_order > 2u) {
        set_error("rtmi_scene_create: unknown rtmi_tuning::tile_order");
        return fail(RTMI_ERR_BAD_ARG);
    }
    s->tile_order_mode = tune.tile_order;
    s->min_bands = tune.bands;
    if (tune.pad_mode > 2u) {
        set_error("rtmi_scene_create: unknown rtmi_tuning::pad_mode");
        return fail(RTMI_ERR_BAD_ARG);
    }
    s->pad_refine = tune.pad_mode == 0u ? s->bvh.pad_refine : tune.pad_mode == 2u;
    s->n_cus = (uint32_t)prop.multiProcessorCount;
    s->grid = s->n_cus * (uint32_t)per_cu;
    // the per-lane strips of attenuation runs grow with the bounce limit (8 bytes per bounce and lane): beyond 4 GiB the
    // persistent grid shrinks instead (maxdepth = 65535: 512 KB per lane, ten workgroups) -- slow, correct, never an
    // allocation the device cannot serve
    {
        const uint64_t per_block = (uint64_t)std::max<uint32_t>(1u, camera->maxdepth) * 8u * s->block;
        const uint64_t fit = std::max<uint64_t>(1u, ((uint64_t)4 << 30) / per_block);
        if (s->grid > fit) s->grid = (uint32_t)fit;
    }

    // (rtmi_tuning::kernel = 2 asked for round 2's queue-scheduled kernel: 0.52-0.85x on every BASELINE config, removed in round 4)
    if (tune.kernel > 2u) {
        set_error("rtmi_scene_create: unknown rtmi_tuning::kernel");
        return fail(RTMI_ERR_BAD_ARG);
    }
    if (tune.kernel == 2u) {
        set_error("rtmi_scene_create: the queue-scheduled kernel (rtmi_tuning::kernel = 2) lost on every measured workload "
                  "and was removed in round 4 (git history keeps it)");
        return fail(RTMI_ERR_UNSUPPORTED);
    }

    HIP_TRY_S(hipMalloc(reinterpret_cast<void**>(&s->d_stats), 128 * sizeof(unsigned long long)));
    HIP_TRY_S(hipMemset(s->d_stats, 0, 128 * sizeof(unsigned long long)));
#ifdef RTMI_TAILPROBE
    HIP_TRY_S(hipMalloc(reinterpret_cast<void**>(&s->d_tail), (size_t)s->grid * (s->block / 64u) * 3u * sizeof(unsigned long long)));
#endif
    const size_t att_lanes = (size_t)s->grid * s->block;
    const size_t att_bytes = std::max<size_t>(16, (size_t)camera->maxdepth * att_lanes * 2u * sizeof(uint32_t));
    HIP_TRY_S(hipMalloc(reinterpret_cast<void**>(&s->slot[0].d_counter), 16));
    HIP_TRY_S(hipMalloc(reinterpret_cast<void**>(&s->slot[0].d_att), att_bytes));
    HIP_TRY_S(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // the cost probe of the tile order, here rather than inside the first call that wants it: no call has more tiles than the frame
    {
        const uint64_t frame_tiles = (uint64_t)((camera->img_width + 7u) / 8u) * ((camera->img_height + 7u) / 8u);
        if (s->tile_order_mode == 2u || (s->tile_order_mode == 0u && !s->big && frame_tiles >= 2048u && camera->samples_per_pixel >= 16u))
            probe_tile_costs(s, s->stream);
    }
#undef HIP_TRY_S
    return RTMI_OK;
}

extern "C" int rtmi_scene_create(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects,
                                 const rtmi_material* materials, uint32_t n_materials,
                                 const rtmi_scene_options* options, rtmi_scene** out) {
    if (!camera || !out || (n_objects && !objects) || (n_materials && !materials)) {
        try { set_error("rtmi_scene_create: null argument"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    *out = nullptr;
    DeviceGuard guard;
    rtmi_scene* s = nullptr;
    const int rc = guarded("rtmi_scene_create", [&] {
        return scene_create_impl(camera, objects, n_objects, materials, n_materials, options, s);
    });
    if (rc != RTMI_OK) {
        free_scene(s);
        return rc;
    }
    *out = s;
    return RTMI_OK;
}

extern "C" void rtmi_scene_destroy(rtmi_scene* scene) {
    DeviceGuard guard;
    free_scene(scene);
}

extern "C" int rtmi_render_row_blocks_device(rtmi_scene* s, uint32_t y_first, uint32_t block_rows,
                                             uint32_t block_stride, uint32_t n_blocks, uint64_t seed,
                                             void* d_rgb_linear_out, void* d_rgba8_out, void* hip_stream) {
    DeviceGuard guard;
    return guarded("rtmi_render_row_blocks_device", [&]() -> int {
        if (!s) {
            set_error("rtmi_render_row_blocks_device: null scene");
            return RTMI_ERR_BAD_ARG;
        }
        std::lock_guard<std::mutex> lock(s->mu);
        HIP_TRY(hipSetDevice(s->device));
        return launch(s, y_first, block_rows, block_stride, n_blocks, 0u, s->cam.img_width, seed, static_cast<float*>(d_rgb_linear_out),
                      static_cast<uint32_t*>(d_rgba8_out), static_cast<hipStream_t>(hip_stream));
    });
}

extern "C" int rtmi_render_block_list_device(rtmi_scene* s, uint32_t block_rows, const uint32_t* blocks, uint32_t n_blocks, uint64_t seed,
                                             void* d_rgb_linear_out, void* d_rgba8_out, void* hip_stream) {
    DeviceGuard guard;
    return guarded("rtmi_render_block_list_device", [&]() -> int {
        if (!s || (n_blocks && !blocks)) {
            set_error("rtmi_render_block_list_device: null argument");
            return RTMI_ERR_BAD_ARG;
        }
        if (n_blocks == 0) return RTMI_OK;
        std::lock_guard<std::mutex> lock(s->mu);
        HIP_TRY(hipSetDevice(s->device));
        return launch(s, 0u, block_rows, 1u, n_blocks, 0u, s->cam.img_width, seed, static_cast<float*>(d_rgb_linear_out),
                      static_cast<uint32_t*>(d_rgba8_out), static_cast<hipStream_t>(hip_stream), blocks);
    });
}

extern "C" int rtmi_scene_get_tile_costs(const rtmi_scene* s, uint32_t* costs_out, uint32_t* n_tiles) {
    if (!s || !n_tiles) {
        set_error("rtmi_scene_get_tile_costs: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    *n_tiles = s->cost_state == 1 ? (uint32_t)s->tile_cost.size() : 0u; // (0: the scene made no probe)
    if (costs_out && *n_tiles) std::memcpy(costs_out, s->tile_cost.data(), s->tile_cost.size() * sizeof(uint32_t));
    return RTMI_OK;
}

// pixels [x0, x1) x [y0, y1) into a dense (x1 - x0)-wide output, device pointers, asynchronous on `hip_stream`
extern "C" int rtmi_render_rect_device(rtmi_scene* s, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint64_t seed,
                                       void* d_rgb_linear_out, void* d_rgba8_out, void* hip_stream) {
    DeviceGuard guard;
    return guarded("rtmi_render_rect_device", [&]() -> int {
        if (!s) {
            set_error("rtmi_render_rect_device: null scene");
            return RTMI_ERR_BAD_ARG;
        }
        if (x0 > x1 || x1 > s->cam.img_width || y0 > y1 || y1 > s->cam.img_height) {
            set_error("rtmi_render_rect_device: rectangle outside the image");
            return RTMI_ERR_BAD_ARG;
        }
        if (x0 == x1 || y0 == y1) return RTMI_OK;
        std::lock_guard<std::mutex> lock(s->mu);
        HIP_TRY(hipSetDevice(s->device));
        return launch(s, y0, y1 - y0, 1, 1, x0, x1, seed, static_cast<float*>(d_rgb_linear_out), static_cast<uint32_t*>(d_rgba8_out),
                      static_cast<hipStream_t>(hip_stream));
    });
}

static int render_rect_impl(const char* who, rtmi_scene* s, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint64_t seed,
                            float* rgb_linear_out, uint32_t* rgba8_out) {
    if (!s) {
        set_error(std::string(who) + ": null scene");
        return RTMI_ERR_BAD_ARG;
    }
    if (x0 > x1 || x1 > s->cam.img_width || y0 > y1 || y1 > s->cam.img_height) {
        set_error(std::string(who) + ": pixels outside the image");
        return RTMI_ERR_BAD_ARG;
    }
    if (y0 == y1 || x0 == x1) return RTMI_OK;
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    const size_t pixels = (size_t)(y1 - y0) * (x1 - x0);
    if (pixels > s->staging_pixels) {
        hipFree(s->d_rgb);
        hipFree(s->d_rgba);
        s->d_rgb = nullptr;
        s->d_rgba = nullptr;
        s->staging_pixels = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_rgb), pixels * 3 * sizeof(float)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_rgba), pixels * sizeof(uint32_t)));
        s->staging_pixels = pixels;
    }
    const int rc = launch(s, y0, y1 - y0, 1, 1, x0, x1, seed, rgb_linear_out ? s->d_rgb : nullptr, rgba8_out ? s->d_rgba : nullptr,
                          s->stream);
    if (rc != RTMI_OK) return rc;
    if (rgb_linear_out) {
        HIP_TRY(hipMemcpyAsync(rgb_linear_out, s->d_rgb, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    }
    if (rgba8_out) {
        HIP_TRY(hipMemcpyAsync(rgba8_out, s->d_rgba, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return RTMI_OK;
}

extern "C" int rtmi_render_rows(rtmi_scene* s, uint32_t y0, uint32_t y1, uint64_t seed, float* rgb_linear_out,
                                uint32_t* rgba8_out) {
    DeviceGuard guard;
    return guarded("rtmi_render_rows", [&] {
        return render_rect_impl("rtmi_render_rows", s, 0u, y0, s ? s->cam.img_width : 0u, y1, seed, rgb_linear_out, rgba8_out);
    });
}

extern "C" int rtmi_render_rect(rtmi_scene* s, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint64_t seed,
                                float* rgb_linear_out, uint32_t* rgba8_out) {
    DeviceGuard guard;
    return guarded("rtmi_render_rect", [&] { return render_rect_impl("rtmi_render_rect", s, x0, y0, x1, y1, seed, rgb_linear_out, rgba8_out); });
}

extern "C" int rtmi_scene_get_stats(rtmi_scene* s, rtmi_stats* out, int reset) {
    DeviceGuard guard;
    if (!s || !out) {
        set_error("rtmi_scene_get_stats: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[4];
    HIP_TRY(hipMemcpy(v, s->d_stats, sizeof(v), hipMemcpyDeviceToHost));
    out->samples = v[0];
    out->segments = v[1];
    out->sphere_tests = v[2];
    out->node_tests = v[3];
    if (reset) HIP_TRY(hipMemset(s->d_stats, 0, sizeof(v)));
    return RTMI_OK;
}

extern "C" int rtmi_scene_get_accel(const rtmi_scene* s, uint32_t* accel_out) {
    if (!s || !accel_out) {
        set_error("rtmi_scene_get_accel: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    *accel_out = s->accel;
    return RTMI_OK;
}

extern "C" int rtmi_scene_get_launch_info(const rtmi_scene* s, rtmi_launch_info* out) {
    // (the struct grows at its end: a caller built against an older header passes its own, smaller struct_size and gets the
    // fields it knows -- 40 bytes in 0.3)
    constexpr uint32_t kSizeV03 = 40u;
    if (!s || !out || out->struct_size < kSizeV03) {
        set_error("rtmi_scene_get_launch_info: null argument or struct_size below the 40 bytes of version 0.3");
        return RTMI_ERR_BAD_ARG;
    }
    rtmi_launch_info v{};
    v.struct_size = out->struct_size;
    v.kernel = 1u;
    v.block_lanes = s->block;
    v.grid_blocks = s->grid;
    v.blocks_per_cu = s->n_cus ? v.grid_blocks / s->n_cus : 0u;
    v.lds_bytes = s->lds_bytes;
    v.scene_in_lds = s->big ? 0u : 1u;
    v.stack_depth = s->stack_depth;
    v.whole_pixel_fallbacks = s->whole_pixel_fallbacks;
    v.packed_chains = s->packed_ok ? s->att_words : 0u;
    v.packed_chain_fallbacks = s->packed_chain_fallbacks;
    v.lds_top_nodes = s->lds_top_nodes;
    v.pad_mode = s->accel == RTMI_ACCEL_BVH ? (s->pad_refine ? 2u : 1u) : 0u;
    v.bands = s->last_bands;
    v.tile_order = s->last_tile_order;
    v.probe_us = s->cost_state == 1 ? (uint32_t)(s->probe_ms * 1000.0f + 0.5f) : 0u;
    v.gen_ahead = 0u; // (rounds 4-5; removed in round 6)
    v.cam_entry = s->d_tile_entry != nullptr ? 1u : 0u;
    v.entry_build_us = (uint32_t)(s->entry_build_ms * 1000.0f + 0.5f);
    v.reband_retries = s->reband_retries;
    v.walk_start = s->d_walk_starts != nullptr ? 1u : 0u;
    std::memcpy(out, &v, std::min<size_t>(out->struct_size, sizeof(v)));
    return RTMI_OK;
}

extern "C" int rtmi_scene_get_bvh(const rtmi_scene* s, rtmi_bvh_node* nodes_out, uint32_t* n_nodes,
                                  uint32_t* slots_out, uint32_t* n_slots, float* pad_classes_out,
                                  uint32_t* n_classes, float* pad_eps, float* pad_floor) {
    if (!s) {
        set_error("rtmi_scene_get_bvh: null scene");
        return RTMI_ERR_BAD_ARG;
    }
    if (s->accel != RTMI_ACCEL_BVH) {
        set_error("rtmi_scene_get_bvh: scene has no BVH");
        return RTMI_ERR_UNSUPPORTED;
    }
    if (n_nodes) *n_nodes = (uint32_t)s->bvh.nodes.size();
    if (n_slots) *n_slots = (uint32_t)s->bvh.slot_object.size();
    if (n_classes) *n_classes = s->bvh.n_pad_classes;
    if (pad_eps) *pad_eps = s->bvh.pad_eps;
    if (pad_floor) *pad_floor = s->bvh.pad_floor;
    if (nodes_out) std::memcpy(nodes_out, s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(rtmi_bvh_node));
    if (slots_out) std::memcpy(slots_out, s->bvh.slot_object.data(), s->bvh.slot_object.size() * sizeof(uint32_t));
    if (pad_classes_out) std::memcpy(pad_classes_out, s->bvh.pad_classes, s->bvh.n_pad_classes * 8 * sizeof(float));
    return RTMI_OK;
}

extern "C" int rtmi_scene_get_walk_starts(const rtmi_scene* s, uint32_t* records_out, uint32_t* n_slots) {
    if (!s || !n_slots) {
        set_error("rtmi_scene_get_walk_starts: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    *n_slots = (uint32_t)(s->walk_starts.size() / 16u); // (0: every scattered ray's walk starts at the root)
    if (records_out && !s->walk_starts.empty()) std::memcpy(records_out, s->walk_starts.data(), s->walk_starts.size() * sizeof(uint32_t));
    return RTMI_OK;
}

extern "C" int rtmi_scene_get_tile_entries(const rtmi_scene* s, uint32_t* entries_out, uint32_t* n_tiles) {
    if (!s || !n_tiles) {
        set_error("rtmi_scene_get_tile_entries: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    *n_tiles = (uint32_t)s->tile_entries.size(); // (0: camera rays walk from the root)
    if (entries_out && !s->tile_entries.empty()) std::memcpy(entries_out, s->tile_entries.data(), s->tile_entries.size() * sizeof(uint32_t));
    return RTMI_OK;
}

extern "C" int rtmi_scene_last_kernel_ms(rtmi_scene* s, float* ms_out) {
    if (!s || !ms_out) {
        set_error("rtmi_scene_last_kernel_ms: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    DeviceGuard guard;
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->ev_valid) {
        set_error("rtmi_scene_last_kernel_ms: no launch yet");
        return RTMI_ERR_BAD_ARG;
    }
    HIP_TRY(hipSetDevice(s->device));
    float total = 0.0f; // the trace kernels alone, band by band; the resolve passes between them are not counted
    for (uint32_t b = 0; b < s->n_bands_timed; ++b) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(s->ev_trace[2u * b + 1u]));
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_trace[2u * b], s->ev_trace[2u * b + 1u]));
        total += ms;
    }
    *ms_out = total;
    return RTMI_OK;
}

#ifdef RTMI_TAILPROBE
// per wave {start, first refill past the end of the work, exit} of the most recent trace kernel, 100 MHz ticks (tools/tail_profile.py)
extern "C" int rtmi_prof_tail_read(rtmi_scene* s, unsigned long long* out, uint32_t* n_waves) {
    hipSetDevice(s->device);
    hipDeviceSynchronize();
    *n_waves = s->grid * (s->block / 64u);
    return hipMemcpy(out, s->d_tail, (size_t)*n_waves * 3u * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

#ifdef RTMI_PROF
extern "C" int rtmi_prof_read(rtmi_scene* s, unsigned long long* out128) {
    hipSetDevice(s->device);
    hipDeviceSynchronize();
    return hipMemcpy(out128, s->d_stats, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

