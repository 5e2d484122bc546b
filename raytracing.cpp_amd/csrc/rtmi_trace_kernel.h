// rtmi_trace_kernel.h -- rtmi_trace_kernel<ACCEL, STATS, BIG, MODE>: the per-pixel path-tracing hot loop (persistent lanes, the
// flattened recursion FETCH -> GEN -> BEGIN -> walk / scan -> SHADE).  Included by rtmi_device.hip only.
#pragma once

#include "rtmi_kernel_common.h"

constexpr uint32_t kBlackSample = 0xfffffffeu; // Trav::best of a sample that is finished without tracing (maxdepth == 0)

#ifndef RTMI_WPE
#define RTMI_WPE 6 // waves per SIMD the register allocation aims at (A/B: 7 = 72 VGPRs + 20 B of scratch, 4 % slower)
#endif

#ifndef RTMI_WPE_BIG
#define RTMI_WPE_BIG 6 // HBM-resident scenes: 8 = 64 VGPRs, the compiler's walk, two 896-lane workgroups per CU (round 2);
                       // 6 = the hand-written node loop (v66-v79 are its node registers), two 768-lane workgroups
#endif

#ifndef RTMI_WALK_PRIO
#define RTMI_WALK_PRIO 1 // s_setprio of a wave inside the traversal loop (0 elsewhere)
#endif

#ifndef RTMI_BLOCK_LIST
#define RTMI_BLOCK_LIST 1 // 0: no launches over a LIST of row blocks (rtmi_render_block_list_device): the A/B of what the look-up costs the others
#endif

#ifndef RTMI_ASM_WALK
#define RTMI_ASM_WALK 1 // 0: the compiler's node step everywhere (the A/B and the fallback for a changed register budget)
#endif
static_assert(!RTMI_ASM_WALK || RTMI_WPE <= 6, "the hand-written node loop uses v66-v79 as its node registers");

#include "rtmi_walk_asm.h"

// BIG = false: the whole scene is staged into LDS and stack entries are packed into 16 bits (<= 8192 spheres).
// BIG = true : the scene stays in HBM (read through L1/L2/Infinity Cache), only the traversal stack is in LDS,
//              32-bit entries (config 4: 100k spheres, 2.4 MB of spheres + 6.4 MB of nodes).
template <int ACCEL, bool STATS, bool BIG, int MODE>
// 6 waves per SIMD (<= 80 VGPRs): two workgroups of 768 lanes per CU; that occupancy is worth +17 % over 4 waves per
// SIMD (measured), and one register more would silently halve it -- hence the explicit bound
// (HBM-resident scenes wait on their node reads, not on issue slots: their variants are allocated for 8 waves per SIMD --
// 64 VGPRs, which they fit without spilling -- and run as two 896-lane workgroups per CU, 7 waves per SIMD: -3.4 %)
__global__ void __attribute__((amdgpu_waves_per_eu(BIG ? RTMI_WPE_BIG : RTMI_WPE, BIG ? RTMI_WPE_BIG : RTMI_WPE))) __launch_bounds__(1024) rtmi_trace_kernel(const RtmiLaunch P) {
    // MODE 0: work items are chunks of a pixel's samples, one 16-byte record per sample; 3: whole-pixel work items (no
    // sample records: the lane adds its pixel's samples up itself); 4: as 0, with the attenuation chain as a packed string
    // of material handles in LDS that leaves with the sample record and is multiplied by the resolve pass (scenes whose
    // strings fit the LDS: few materials or a low bounce limit; the box of config 5).  (Modes 1 / 2, the deferred-path
    // queue and its drain launch of rounds 1-2, were measured 8 % slower on the final round-2 kernel and are gone.)
    constexpr bool WHOLE = MODE == 3, PACKED = MODE == 4;
    static_assert(!(PACKED && BIG), "packed chains live next to an LDS-resident scene");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using StackT = typename std::conditional<BIG, uint32_t, uint16_t>::type;
    // traversal stack: entry 0 of every lane holds a sentinel that ends the walk when it is popped.  References are
    // signed: nodes >= 0, leaves < -1, the sentinel -1 (16-bit entries are read back sign-extended); t.sp is an LDS address
    using StackS = typename std::conditional<BIG, int32_t, int16_t>::type;
    typedef __attribute__((address_space(3))) StackS lds_stack_t;
    constexpr uint32_t kStackEnd = 0xffffffffu;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8*)lds_raw;
    const uint32_t sp0 = lds0 + P.lds_stack + threadIdx.x * (uint32_t)sizeof(StackT), sp_stride = blockDim.x * (uint32_t)sizeof(StackT);
    auto stack_at = [](uint32_t addr) -> lds_stack_t* { return (lds_stack_t*)(uintptr_t)addr; };
    if (ACCEL == RTMI_ACCEL_BVH) *stack_at(sp0) = (StackS)-1;
    uint32_t* lds_att = reinterpret_cast<uint32_t*>(lds_raw + P.lds_att);
    // per-wave pools: work indices are taken from the global counter 64 at a time (a single counter word saturates at
    // ~88 returning atomics per microsecond on MI355X; a 1080p x 512 spp frame has 16.6 M work items)
    uint32_t* pool = reinterpret_cast<uint32_t*>(lds_raw + P.lds_pool) + (threadIdx.x >> 6) * 20u;
    lds_u8* rank_tbl = (lds_u8*)(pool + 4); // 64 bytes, see coop_draws
    if ((threadIdx.x & 63u) == 0u) {
        pool[0] = 0u; pool[1] = 0u; pool[2] = 0u; pool[3] = 0u;
    }
    const uint4* lds_spheres;
    const uint4* lds_aux;
    const uint4* lds_mats;
    const uint4* lds_nodes;
    if (BIG) {
        // the scene stays in memory; the first lds_top_nodes nodes of the breadth-first numbering (48-byte records) start the
        // dynamic LDS segment and the node step reads them from there (walk_nodes_hbm)
        lds_spheres = P.spheres;
        lds_aux = P.aux;
        lds_mats = P.mats;
        lds_nodes = P.nodes;
        if (ACCEL == RTMI_ACCEL_BVH && P.lds_top_nodes != 0u) {
            uint4* w_nodes = reinterpret_cast<uint4*>(lds_raw);
            for (uint32_t i = threadIdx.x; i < 3u * P.lds_top_nodes; i += blockDim.x) w_nodes[i] = P.nodes[i];
            __syncthreads();
        }
    } else {
        // ---- stage the scene into LDS: coalesced 16-byte loads, one pass per array -------------------------
        uint4* w_spheres = reinterpret_cast<uint4*>(lds_raw + P.lds_spheres);
        uint4* w_aux = reinterpret_cast<uint4*>(lds_raw + P.lds_aux);
        uint4* w_mats = reinterpret_cast<uint4*>(lds_raw + P.lds_mats);
        uint4* w_nodes = reinterpret_cast<uint4*>(lds_raw); // nodes always start the dynamic LDS segment
        for (uint32_t i = threadIdx.x; i < P.n_slots; i += blockDim.x) {
            w_spheres[i] = P.spheres[i];
            w_aux[i] = P.aux[i];
        }
        for (uint32_t i = threadIdx.x; i < P.n_mats; i += blockDim.x) w_mats[i] = P.mats[i];
        if (ACCEL == RTMI_ACCEL_BVH) {
            for (uint32_t i = threadIdx.x; i < 3u * P.n_nodes; i += blockDim.x) w_nodes[i] = P.nodes[i];
        }
        __syncthreads();
        lds_spheres = w_spheres;
        lds_aux = w_aux;
        lds_mats = w_mats;
        lds_nodes = w_nodes;
    }

#ifdef RTMI_TAILPROBE
    const uint32_t tp_wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0u && P.tail_probe) { P.tail_probe[3u * tp_wave] = wall_clock64(); P.tail_probe[3u * tp_wave + 1u] = 0ull; }
#endif
    // (functions, not values: these are needed once per path or per launch and must not hold a register in between)
    auto glane_of = [&]() -> uint32_t { return blockIdx.x * blockDim.x + threadIdx.x; };
#define glane glane_of()
#define lane lane_id()
    const uint32_t W = P.cam.img_width;
    const uint32_t spp = P.cam.samples_per_pixel;

    uint32_t phase = PH_FETCH;
    uint32_t lpix = 0, s = 0, s_end = 0, depth_left = 0, natt = 0; // lpix: pixel index within this call's rows
    V3 sum = mk(0.0f, 0.0f, 0.0f); // WHOLE only: the pixel's running sum
    Rng rng{};
    Trav t{};
    t.cur = kStackEnd; // "not walking" (see the traversal loop)
    uint32_t st_segments = 0, st_sphere = 0, st_node = 0, st_samples = 0, st_item0 = 0;
    uint32_t ws1 = 0, ws2 = 0, ws3 = 0; // HBM-resident trees with walk starts: the start record of the segment about to begin (see begin_segment)
    PF_DECL
    PB_DECL

    // attenuation chain: material handles of the non-dielectric bounces of the live path, run-length encoded (a path
    // trapped inside the ground sphere bounces 50 times on the same material: one run).  The open run lives in two
    // registers, closed runs in LDS (kAttLds per lane, {handle, count} packed in 32 bits); only a path with more than
    // kAttLds material changes spills to a per-lane strip in HBM.  rocprofv3 on the un-encoded chain: 90 GB of
    // write-backs per 1080p x 512 spp frame, all of it this strip.
    const uint32_t maxdepth = P.cam.maxdepth;
    uint32_t run_h = 0, run_n = 0;
    // LDS-resident scenes: closed runs go through a window of kAttLds (4) entries in LDS; a full window leaves as ONE
    // 16-byte store to the lane's strip in HBM, so a path whose material changes at every bounce (the box of config 5:
    // 79 segments per sample) moves 4 bytes per bounce instead of the two lone 4-byte stores it used to cost
    // (rocprofv3 on config 5: 2.4 TB of write-backs per frame before).
    const uint32_t att_blocks = (maxdepth + 3u) >> 2;
    auto att_store = [&](uint32_t q, uint32_t h, uint32_t n) {
        if (!BIG) {
            const uint32_t e = h | (n << 16);
            const uint32_t j = q & 3u;
            lds_att[j * blockDim.x + threadIdx.x] = e;
            if (j == 3u) {
                reinterpret_cast<uint4*>(P.att_stack)[(size_t)glane * att_blocks + (q >> 2)] =
                    make_uint4(lds_att[threadIdx.x], lds_att[blockDim.x + threadIdx.x], lds_att[2u * blockDim.x + threadIdx.x], e);
            }
        } else {
            P.att_stack[((size_t)glane * maxdepth + q) * 2u] = h;
            P.att_stack[((size_t)glane * maxdepth + q) * 2u + 1u] = n;
        }
    };
    // MODE 4 keeps the chain as a string of handles instead: run_h = the word being filled, run_n = bits used in it |
    // index of that word << 8, natt = handles so far.  A path of the config-5 box changes material at nearly every one
    // of its 79 bounces: run-length encoding buys nothing there, the strips it spilled to were 27x the algorithmic HBM
    // traffic of the launch (r02 profile) and the multiplication at the end of a path ran for one or two lanes of a wave
    // at a time.  Here the string stays in LDS while the path lives, leaves in 16-byte stores next to the sample record
    // when the path reaches the sky, and the resolve pass -- one lane per pixel, every lane busy -- does the multiplying.
    auto att_push = [&](uint32_t h) {
        if (PACKED) {
            run_h |= h << (run_n & 255u);
            run_n += P.att_bits;
            natt++;
            if ((run_n & 255u) + P.att_bits > 32u) { // no room for another handle: the word goes to LDS
                lds_att[(run_n >> 8) * blockDim.x + threadIdx.x] = run_h;
                run_h = 0u;
                run_n = (run_n & ~255u) + 256u;
            }
            return;
        }
        if (run_n != 0u && h == run_h) {
            run_n++;
        } else {
            if (run_n != 0u) att_store(natt++, run_h, run_n);
            run_h = h;
            run_n = 1u;
        }
    };
    auto att_apply = [&](V3 color, uint32_t h, uint32_t n) -> V3 {
        const uint4 m0 = lds_mats[h];
        const V3 a = mk(__uint_as_float(m0.x), __uint_as_float(m0.y), __uint_as_float(m0.z));
        for (uint32_t c = 0; c < n; ++c) color = vmul(a, color); // A*(A*(...)): one multiply per bounce, in order
        return color;
    };
    // the spheres of one leaf against the current segment, two at a time: both discriminants, then the (rare) roots.
    // The first pair is straight code -- with the default leaf size of 2 it is the whole leaf -- larger leaves loop on
    auto test_leaf = [&](uint32_t ref) {
        const uint32_t first = BIG ? (ref & 0x00ffffffu) : (ref & 0x1fffu);
        const uint32_t cnt = BIG ? ((ref >> 24) & 0x7fu) : (((ref >> 13) & 3u) + 1u);
        t.a = vdot(t.d, t.d); // (recomputed here: not a register across the node steps)
        const Recip ra = recip_for(t.a); // shared by every root of this leaf step
        auto pair = [&](uint32_t q) {
            const bool two = q + 1u < cnt;
            const uint4 r0 = lds_spheres[first + q];
            const uint4 r1 = lds_spheres[first + q + (two ? 1u : 0u)];
            float h0, h1, d0, d1;
            bool k0, k1;
            sphere_delta_cand(r0, t, h0, d0, k0);
            sphere_delta_cand(r1, t, h1, d1, k1);
            k1 = k1 & two;
            // Roots only for spheres that can have one ahead of the origin (sphere_delta_cand), and ONE pass of the root
            // arithmetic for the lanes' first such sphere, whichever of the two it is: with a separate branch per sphere the wave
            // ran both (~45 instructions each) whenever any lane needed either -- nearly every leaf trip -- although hardly a lane
            // needs both; the second pass is left for the trips in which one does.  Per lane the spheres are still taken in slot
            // order (and ties go by object index, sphere_root_bvh): the closest hit is the same.
            if (k0 | k1) {
                const bool sec = !k0;
                sphere_root_bvh(sec ? h1 : h0, sec ? d1 : d0, t, ra, first + q + (sec ? 1u : 0u), lds_aux, t.tbest, t.best);
                if (k0 & k1) sphere_root_bvh(h1, d1, t, ra, first + q + 1u, lds_aux, t.tbest, t.best);
            }
        };
        pair(0u);
        if (cnt > 2u) {
            for (uint32_t q = 2u; q < cnt; q += 2u) pair(q);
        }
        if (STATS) st_sphere += cnt;
    };
    auto begin_segment = [&](V3 o, V3 d) {
        t.o = o;
        t.d = d;
        t.a = vdot(d, d);
        t.tbest = __builtin_inff();
        t.best = ~0u;
        t.sp = lds0 + P.lds_stack + threadIdx.x * (uint32_t)sizeof(StackT) + sp_stride; // entry 1 (entry 0: the sentinel)
        if (ACCEL == RTMI_ACCEL_BVH && BIG && P.walk_starts != nullptr) {
            // A ray scattered off a sphere of the tree starts its walk in that sphere's own leaf (t.cur: set with the record, below), the
            // siblings hanging off the path above it pre-loaded on the stack as WAY records -- two levels a record, in the node format,
            // so that the node loop tests them like any node (host: build_walk_starts; exact for any start).  The 16-byte
            // start record was read when the ray was scattered (ws1-3, a round ago: its latency is behind the rest of that round): n,
            // the path code that places the top levels' records in the staged block, the deeper records by index.
            const uint32_t n = ws1 & 15u, code = (ws1 >> 4) & 255u, n_top = min(n, P.way_jtop);
            auto push = [&](uint32_t k, uint32_t id) { if (k < n) *stack_at(t.sp + k * sp_stride) = (StackS)id; };
            if (ballot(n != 0u) != 0ull) {
                // level j = 1 .. n_top at way_top_base + (4^j - 4) / 3 + the first 2 j bits of the code
                const uint32_t sh = 2u * n_top;
                if (n_top >= 1u) push(0u, P.way_top_base + 0u + (code >> ((sh - 2u) & 31u)));
                if (n_top >= 2u) push(1u, P.way_top_base + 4u + (code >> ((sh - 4u) & 31u)));
                if (n_top >= 3u) push(2u, P.way_top_base + 20u + (code >> ((sh - 6u) & 31u)));
                if (n_top >= 4u) push(3u, P.way_top_base + 84u + (code >> ((sh - 8u) & 31u)));
                push(n_top + 0u, ws1 >> 12);
                push(n_top + 1u, ws2 & 0xfffffu);
                push(n_top + 2u, (ws2 >> 20) | ((ws3 & 0xffu) << 12));
                push(n_top + 3u, (ws3 >> 8) & 0xfffffu);
                t.sp += n * sp_stride;
            }
        }
        if (ACCEL == RTMI_ACCEL_BVH) { // (t.cur, where the walk starts, was set by whoever made the segment: the root, or a camera ray's entry)
            t.inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            PF_MARK(17);
            t.oinv = mk(-(o.x * t.inv.x), -(o.y * t.inv.y), -(o.z * t.inv.z));
            // Leaves that hang directly off the top of the tree -- the ground sphere, whose box is the whole scene; the
            // walls of a box made of huge spheres -- are tested here, by all the lanes that start a segment, and the
            // walk begins below them with the far limit already set: one node trip and one leaf trip less per leaf
            // and segment.  A tree that is nothing but such a spine is not walked at all.
            for (uint32_t q = 0; q < P.n_pre_leaves; ++q) {
                const uint32_t ref = P.pre_leaf[q]; // wave-uniform
                const uint32_t cnt = BIG ? ((ref >> 24) & 0x7fu) : (((ref >> 13) & 3u) + 1u);
                if (cnt == 1u) { // a lone sphere (the ground): one discriminant, not the pair routine's two
                    const uint32_t slot = BIG ? (ref & 0x00ffffffu) : (ref & 0x1fffu);
                    float h0, d0;
                    bool k0;
                    sphere_delta_cand(lds_spheres[slot], t, h0, d0, k0);
                    if (k0) sphere_root_bvh(h0, d0, t, recip_for(t.a), slot, lds_aux, t.tbest, t.best);
                    if (STATS) st_sphere += 1u;
                } else {
                    test_leaf(ref);
                }
            }
            PF_MARK(18);
            // The pad of this segment's boxes: every sphere whose root the reference's fp32 arithmetic could
            // accept must be reached.  Per radius class, E0 = e(farthest centre of the class) bounds it for any ray from this
            // origin; on scenes much wider than their spheres (pad_refine: the 316-unit grid of config 4, where E0 is 0.1-0.8
            // units on spheres of radius 0.2) the segment's reach bounds it far better: an accepted root's point lies within
            // G = rmax + E0 of a centre, hence inside the class's centre box grown by G, and before the far limit the peeled
            // leaves left (the ground hit), so L <= t_far |d| + G and E1 = e(L_max) -- the same expression in the oracle's walk.
            float pad = P.pad_floor;
            if (P.pad_refine) pad = fmaxf(pad, 9.5367432e-7f * fmaxf(fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z))); // 16u |O|_inf
            const float pad_floor_o = pad;
            const float dlen = P.pad_refine ? __builtin_amdgcn_sqrtf(t.a) * 1.00001f : 0.0f;
            for (uint32_t c = 0; c < P.n_pad_classes; ++c) {
                const float* k = P.pad_classes[c];
                const float ax = fmaxf((o.x - k[0]) * (o.x - k[0]), (k[3] - o.x) * (k[3] - o.x));
                const float ay = fmaxf((o.y - k[1]) * (o.y - k[1]), (k[4] - o.y) * (k[4] - o.y));
                const float az = fmaxf((o.z - k[2]) * (o.z - k[2]), (k[5] - o.z) * (k[5] - o.z));
                // sqrt(R^2 + x) - R <= min(x / (2R), sqrt(x)): the linear bound explodes for a ray that starts thousands of
                // units away (a path inside the ground sphere), the square root does not
                const float x = P.pad_eps * (((ax + ay) + az) + k[7]); // k[7]: rmax^2 of the class
                float ec = fminf(x * k[6], __builtin_amdgcn_sqrtf(x) * 1.000001f);
                if (P.pad_refine) {
                    // (twice the floor on top of the reach: the exit parameters below are off by at most ~3u (|plane| + |O|) |1/d|)
                    const float g = __builtin_fmaf(2.0f, pad_floor_o, P.pad_rmax[c] + ec);
                    const float ex = fmaxf(__builtin_fmaf(k[0] - g, t.inv.x, t.oinv.x), __builtin_fmaf(k[3] + g, t.inv.x, t.oinv.x));
                    const float ey = fmaxf(__builtin_fmaf(k[1] - g, t.inv.y, t.oinv.y), __builtin_fmaf(k[4] + g, t.inv.y, t.oinv.y));
                    const float ez = fmaxf(__builtin_fmaf(k[2] - g, t.inv.z, t.oinv.z), __builtin_fmaf(k[5] + g, t.inv.z, t.oinv.z));
                    // (fmaxf / fminf drop a NaN operand -- 0 * inf on an axis-parallel ray: that axis does not bound the reach)
                    const float t_far = fmaxf(fminf(fminf(ex, ey), fminf(ez, t.tbest)), 0.0f);
                    const float lmax = __builtin_fmaf(t_far, dlen, g);
                    const float x1 = P.pad_eps * (lmax * lmax + k[7]);
                    ec = fminf(ec, fminf(x1 * k[6], __builtin_amdgcn_sqrtf(x1) * 1.000001f));
                }
                pad = fmaxf(pad, ec);
            }
            t.pinv = mk(pad * __builtin_fabsf(t.inv.x), pad * __builtin_fabsf(t.inv.y), pad * __builtin_fabsf(t.inv.z));
            PF_MARK(19);
        } else {
            t.cur = 0; // next sphere of the linear scan
        }
        if (STATS) st_segments++;
    };

    for (;;) {
        PF_MARK(16);
        // ---- FETCH: one wave-aggregated atomic hands out consecutive indices of the 8x8-tiled pixel space -----
        ISA_MARK("fetch");
        PB(21, true);
        PB(0, phase == PH_FETCH);
        while (phase == PH_FETCH) {
            const uint64_t need = ballot(true);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            uint32_t start = 0, take = 0, txy = 0, s_first = 0;
            if (rank == 0) { // the wave's leader serves the request from the wave's pool, refilling it 64 items at a time
                uint32_t next = pool[0], end = pool[1];
                if (next == end) {
                    next = atomicAdd(P.work_counter, 64u);
                    end = next + 64u;
                    pool[1] = end;
                    // a refill is one unit of 64 consecutive indices: the 64 pixels of ONE 8x8 tile for ONE chunk of samples, so
                    // the tile is looked up here, once per 64 items, and travels to the lanes with the indices.  Hand-out position
                    // -> tile through the launch's order table (costliest tiles first: the heavy-tailed items -- pixels of the
                    // horizon band, most of whose samples run all 50 bounces inside the ground sphere -- start when the launch
                    // does, and its end is made of sky), or row by row, bottom rows first.
#ifdef RTMI_TAILPROBE
                    if (next >= P.n_work && P.tail_probe && P.tail_probe[3u * tp_wave + 1u] == 0ull) P.tail_probe[3u * tp_wave + 1u] = wall_clock64();
#endif
                    const uint32_t unit = next >> 6;
                    const uint32_t pos = fdiv(unit, P.div_chunks);
                    const uint32_t s0 = (unit - pos * P.n_chunks) * P.chunk; // the item's first sample
                    uint32_t tile = pos, flip = P.top_down ? 0u : 1u;
                    if (P.tile_order != nullptr && next < P.n_work) {
                        tile = P.tile_order[pos];
                        flip = 0u;
                    }
                    const uint32_t trow = fdiv(tile, P.div_tiles_x);
                    const uint32_t trow_l = flip ? P.tiles_y - 1u - trow : trow;
                    pool[2] = (tile - trow * P.tiles_x) | (trow_l << 16);
                    // (a launch over a LIST of row blocks -- the cost-balanced shards of a multi-GPU frame: blocks of a multiple of 8
                    // rows that start on multiples of 8 -- looks the tile's image row up once per refill: bits 16.. = that row / 8)
                    uint32_t gy8 = 0u;
                    if (RTMI_BLOCK_LIST && P.block_first_row != nullptr && next < P.n_work) {
                        const uint32_t ply0 = trow_l * 8u, blk0 = fdiv(ply0, P.div_block_rows);
                        gy8 = (P.block_first_row[blk0] + (ply0 - blk0 * P.block_rows)) >> 3;
                    }
                    pool[3] = s0 | (gy8 << 16); // (s0 < samples_per_pixel <= 65535)
                }
                take = min((uint32_t)__popcll(need), end - next);
                start = next;
                pool[0] = next + take;
                txy = pool[2];
                s_first = pool[3];
            }
            const int leader = __ffsll((long long)need) - 1;
            start = __shfl(start, leader);
            take = __shfl(take, leader);
            txy = __shfl(txy, leader);
            s_first = __shfl(s_first, leader);
            if (rank >= take) continue; // pool ran dry mid-request: ask again
            const uint32_t idx = start + rank;
            if (idx >= P.n_work) {
                phase = PH_DONE;
            } else {
                const uint32_t j = idx & 63u;
                const uint32_t px = P.x_first + (txy & 0xffffu) * 8u + (j & 7u), ply = (txy >> 16) * 8u + (j >> 3);
                if (px < P.x_end && ply < P.n_local_rows) {
                    uint32_t gy; // local row -> row of the whole image
                    if (RTMI_BLOCK_LIST && P.block_first_row != nullptr) {
                        gy = ((s_first >> 16) << 3) + (j >> 3);
                    } else {
                        const uint32_t blk = fdiv(ply, P.div_block_rows);
                        gy = P.y_first + blk * P.block_stride * P.block_rows + (ply - blk * P.block_rows);
                    }
                    lpix = ply * P.local_w + (px - P.x_first);
                    rng.pixel = gy * W + px;
                    s = s_first & 0xffffu;
                    s_end = min(spp, s + P.chunk);
                    if (WHOLE) sum = mk(0.0f, 0.0f, 0.0f);
                    phase = PH_GEN;
                }
            }
        }
        if (ballot(phase != PH_DONE) == 0ull) break;
        PF_MARK(0);

        // ---- GEN: RayTracingCore::get_ray, core.cc:218-234 --------------------------------------------------------
        ISA_MARK("gen");
        PB(1, phase == PH_GEN);
        // (a gate on this branch -- run it only when K lanes need a primary ray or one has waited T rounds -- was measured
        // on configs 3, 4 and 5: +-0 at best, slower from K = 8 up; profiles/r03_gating_experiment.txt.  Primary rays generated
        // AHEAD into LDS slots whenever some lane needs one now -- rounds 4 and 5 -- measured -0.6 % switched on against off, and
        // round 6's same-box A/B of whole libraries showed what its code costs the packed-chain variant either way: +3 % on the
        // config-5 frame, profiles/r06_r4_vs_r5.txt; it is out of the source)
        // where a camera ray's walk starts: the entry of its 8x8 tile of the image (host: build_tile_entries -- the lowest common
        // ancestor of every sphere the tile's beam can meet, kStackEnd when it meets none), else the root
        auto camera_entry = [&](uint32_t px, uint32_t gy) -> uint32_t {
            if (ACCEL != RTMI_ACCEL_BVH) return 0u;
            return P.tile_entry != nullptr ? P.tile_entry[(gy >> 3) * P.gtiles_x + (px >> 3)] : P.root_ref;
        };
        if (phase == PH_GEN) {
            const uint32_t gy = fdiv(rng.pixel, P.div_w), px = rng.pixel - gy * W; // (rng.pixel = gy * W + px came with the work item)
            t.cur = camera_entry(px, gy); // (first: the load is in flight behind the ray's arithmetic, its address registers are free again)
            rng.k = 0;
            rng.sample = s;
            Blk gb = rng_block(rng, 0u, P.seed); // draws 0,1: pixel jitter; 2,3: first defocus-disk attempt
            const float offx = draw_centered(gb.w0);
            const float offy = draw_centered(gb.w1);
            rng.k = 2;
            const V3 du = ld3(P.cam.pixel_delta_u), dv = ld3(P.cam.pixel_delta_v);
            const V3 pixel_sample =
                vadd(vadd(ld3(P.cam.pixel00), vscale(du, (float)px + offx)), vscale(dv, (float)gy + offy));
            V3 origin = ld3(P.cam.cam_center);
            if (!(P.cam.defocus_angle <= 0.0f)) {
                // random_vector_on_unit_disk, random.number.gen.hpp:35-42
                float dx = draw_pm1(gb.w2), dy = draw_pm1(gb.w3);
                rng.k = 4;
                ISA_MARK("gen-disk-retry");
                PF_MARK(1);
                while (!(vdot(mk(dx, dy, 0.0f), mk(dx, dy, 0.0f)) < 1.0f)) { // two attempts per further block
                    PB(2, true);
                    if ((rng.k & 3u) == 0u) gb = rng_block(rng, rng.k >> 2, P.seed);
                    dx = draw_pm1((rng.k & 3u) ? gb.w2 : gb.w0);
                    dy = draw_pm1((rng.k & 3u) ? gb.w3 : gb.w1);
                    rng.k += 2u;
                }
                PF_MARK(20);
                ISA_MARK("gen-tail");
                origin = vadd(vadd(ld3(P.cam.cam_center), vscale(ld3(P.cam.defocus_disk_u), dx)),
                              vscale(ld3(P.cam.defocus_disk_v), dy));
            }
            depth_left = P.cam.maxdepth;
            natt = 0;
            run_n = 0;
            if (PACKED) run_h = 0;
            if (depth_left == 0) {
                // compute_color(depth == 0) returns 0 at once (core.cc:238-240): the sample is black
                t.best = kBlackSample; // marker read by SHADE: finish the sample without tracing
                if (ACCEL == RTMI_ACCEL_BVH) t.cur = kStackEnd;
                phase = PH_SHADE;
            } else {
                t.o = origin;
                t.d = vsub(pixel_sample, origin);
                if (ACCEL == RTMI_ACCEL_BVH && BIG) ws1 = 0u; // (a camera ray: no way records; t.cur is its tile's entry or the root)
                phase = PH_BEGIN;
            }
        }
        // every new segment of this round -- primary rays, scattered rays, resumed paths -- is set up here, once
        PF_MARK(1);
        ISA_MARK("begin");
        PB(3, phase == PH_BEGIN);
        if (phase == PH_BEGIN) {
            begin_segment(t.o, t.d);
            phase = (ACCEL == RTMI_ACCEL_BVH && P.root_ref == kNoWalk) ? PH_SHADE : PH_TRAV;
        }

        // waves in the traversal loop issue ahead of waves that shade, draw or fetch: the loop is where the lanes are
        // (A/B on MI355X: +1.3 %; the other way round -0.4 %)
        ISA_MARK("walk");
        __builtin_amdgcn_s_setprio(RTMI_WALK_PRIO);
        PF_MARK(2);
        // ---- TRAVERSE ---------------------------------------------------------------------------------------------
        if (ACCEL == RTMI_ACCEL_BVH) {
            // Two kinds of step: an internal node (two slab tests) or a leaf (its spheres).  Each iteration the wave
            // runs only the kind that holds more of its traversing lanes; the other lanes keep their place.
            // leave when wait_thresh lanes wait for shading; the stragglers keep their state and go on next round
            // (A/B on MI355X: counting finished lanes as waiting too was 0-5 % slower).  Inside the loop lanes only move
            // from TRAV to SHADE, so the test is on the number still traversing: no third vote, no reload per trip.
            // A lane walks iff its t.cur is a node or a leaf reference: a finished walk leaves the sentinel there (and lanes
            // that never walked start with it), so the two votes come straight from t.cur -- no phase compare, no mask
            // algebra in the loop (every instruction of this loop, scalar ones included, is paid ~15 times per round:
            // ten more s_add per trip cost the frame 4.3 %, ten more v_mov 2.8 %, measured).
            const int trav_floor = max(0, (int)__popcll(ballot(phase == PH_TRAV || phase == PH_SHADE)) - (int)P.wait_thresh);
            for (;;) {
#if RTMI_ASM_WALK && !(defined(RTMI_PROF) && RTMI_PROF == 1)
                if (!STATS && (!BIG || RTMI_WPE_BIG <= 6)) {
                    int n_leaf, n_node;
                    if (BIG) walk_nodes_hbm(t, lds_nodes, lds0, P.lds_top_nodes, sp_stride, trav_floor, n_leaf, n_node);
                    else walk_nodes_lds(t, lds0, sp_stride, trav_floor, n_leaf, n_node); // nodes start the dynamic LDS segment
                    if (n_leaf + n_node <= trav_floor) break;
                    if ((int32_t)t.cur < -1) { // the leaf step won the vote
                        test_leaf(t.cur);
                        t.sp -= sp_stride;
                        t.cur = (uint32_t)(int32_t)*stack_at(t.sp);
                    }
                    continue;
                }
#endif
                const bool at_leaf = (int32_t)t.cur < -1, at_node = (int32_t)t.cur >= 0; // (inline constants)
                const uint64_t m_leaf = ballot(at_leaf);
                const uint64_t m_node = ballot(at_node);
#if defined(RTMI_PROF) && RTMI_PROF == 1
                // pfl: 0 leaf trips, 1 lanes stepping in them, 2 lanes parked at a node meanwhile; 3 node trips, 4 lanes stepping, 5 parked at a leaf
                if (__popcll(m_leaf) > __popcll(m_node)) { PF_COUNT(0); PF_LANES(1, m_leaf); PF_LANES(2, m_node); } else if (__popcll(m_leaf) + __popcll(m_node) > trav_floor) { PF_COUNT(3); PF_LANES(4, m_node); PF_LANES(5, m_leaf); }
#endif
                int n_leaf = (int)__popcll(m_leaf), n_node = (int)__popcll(m_node);
                // keep the counts 32-bit scalars: left alone the compiler compares the 64-bit popcounts, for which
                // the scalar unit has no greater-than, and moves the vote's outcome through the vector unit
                asm volatile("" : "+s"(n_leaf), "+s"(n_node));
                if (n_leaf + n_node <= trav_floor) break; // also: nobody walks
                // (one merged pop behind both branches: writing it out in each of them was measured 3 % slower)
                bool pop = false;
                if (n_leaf > n_node) {
                    PF_MARK(3);
                    if (at_leaf) {
                        test_leaf(t.cur);
                        pop = true;
                    }
                    PF_MARK(21);
                } else if (at_node) {
                    NodeFields nd;
                    if (BIG) { // 48-byte records: the staged top of the tree from LDS, the rest through L1 / L2 / Infinity Cache (config 4)
                        uint4 n0, n1, n2;
                        if (t.cur < P.lds_top_nodes) {
                            typedef __attribute__((address_space(3))) uint32_t lds_u32;
                            const lds_u32* np = (const lds_u32*)(uintptr_t)(lds0 + 48u * t.cur);
                            n0 = make_uint4(np[0], np[1], np[2], np[3]);
                            n1 = make_uint4(np[4], np[5], np[6], np[7]);
                            n2 = make_uint4(np[8], np[9], np[10], np[11]);
                        } else {
                            const uint4* np = lds_nodes + 3u * t.cur;
                            n0 = np[0]; n1 = np[1]; n2 = np[2];
                        }
                        nd = unpack_node48(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z);
                    } else { // the same records in LDS
                        const uint4* np = lds_nodes + 3u * t.cur;
                        const uint4 n0 = np[0], n1 = np[1], n2 = np[2];
                        nd = unpack_node48(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z);
                    }
                    const float c0x = nd.c0x, c0y = nd.c0y, c0z = nd.c0z, c1x = nd.c1x, c1y = nd.c1y, c1z = nd.c1z;
                    const float h0x = nd.h0x, h0y = nd.h0y, h0z = nd.h0z, h1x = nd.h1x, h1y = nd.h1y, h1z = nd.h1z;
                    const float ax = __builtin_fabsf(t.inv.x), ay = __builtin_fabsf(t.inv.y), az = __builtin_fabsf(t.inv.z);
                    // slab test in centre/half form: the pad rides in the FMA of the half extent
                    const float tc0x = __builtin_fmaf(c0x, t.inv.x, t.oinv.x), th0x = __builtin_fmaf(h0x, ax, t.pinv.x);
                    const float tc0y = __builtin_fmaf(c0y, t.inv.y, t.oinv.y), th0y = __builtin_fmaf(h0y, ay, t.pinv.y);
                    const float tc0z = __builtin_fmaf(c0z, t.inv.z, t.oinv.z), th0z = __builtin_fmaf(h0z, az, t.pinv.z);
                    const float tc1x = __builtin_fmaf(c1x, t.inv.x, t.oinv.x), th1x = __builtin_fmaf(h1x, ax, t.pinv.x);
                    const float tc1y = __builtin_fmaf(c1y, t.inv.y, t.oinv.y), th1y = __builtin_fmaf(h1y, ay, t.pinv.y);
                    const float tc1z = __builtin_fmaf(c1z, t.inv.z, t.oinv.z), th1z = __builtin_fmaf(h1z, az, t.pinv.z);
                    // fmaxf/fminf drop a NaN operand (0 * inf on an axis-parallel ray), which keeps the test conservative
                    const float tn0 = fmaxf(fmaxf(tc0x - th0x, tc0y - th0y), fmaxf(tc0z - th0z, 0.0001f));
                    // (the far limit goes in through one hand-placed v_min_f32 per box: fminf on a value the compiler cannot
                    // prove quiet costs a canonicalising v_max_f32 per trip; tbest is +inf or a finite root)
                    float zf0 = tc0z + th0z, zf1 = tc1z + th1z;
                    asm("v_min_f32 %0, %1, %2" : "=v"(zf0) : "v"(zf0), "v"(t.tbest));
                    asm("v_min_f32 %0, %1, %2" : "=v"(zf1) : "v"(zf1), "v"(t.tbest));
                    const float tf0 = fminf(fminf(tc0x + th0x, tc0y + th0y), zf0);
                    const float tn1 = fmaxf(fmaxf(tc1x - th1x, tc1y - th1y), fmaxf(tc1z - th1z, 0.0001f));
                    const float tf1 = fminf(fminf(tc1x + th1x, tc1y + th1y), zf1);
                    if (STATS) st_node += 2;
                    const bool hit0 = tn0 <= tf0, hit1 = tn1 <= tf1;
                    const uint32_t ch0 = nd.ch0, ch1 = nd.ch1;
                    asm volatile("" ::"v"(ch0), "v"(ch1)); // keep the child-reference read with the box reads (one LDS round trip)
                    // flat on purpose: selects instead of nested branches (each nesting level is an exec-mask
                    // save / restore and a branch of the wave)
                    const bool nearer1 = tn1 < tn0;
                    const bool take1 = hit1 & (!hit0 | nearer1); // nearer child first (bitwise: no short-circuit branches)
                    // the far child is written above the stack top unconditionally (the stack has one spare level) and only
                    // counts when both boxes are hit: no exec-mask save / restore around the push
                    *stack_at(t.sp) = (StackS)(take1 ? ch0 : ch1);
                    t.sp += (hit0 & hit1) ? sp_stride : 0u;
                    t.cur = take1 ? ch1 : ch0; // overwritten by the pop when neither box is hit
                    pop = !(hit0 | hit1);
                }
                if (pop) {
                    t.sp -= sp_stride;
                    t.cur = (uint32_t)(int32_t)*stack_at(t.sp);
                }
            }
            if (phase == PH_TRAV && t.cur == kStackEnd) phase = PH_SHADE; // popped the sentinel: the walk is over
        } else {
            // the reference's linear closest-hit scan (object.defs.cc:68-81); all lanes of a wave read the same
            // sphere, so every LDS read is a broadcast.
            if (ballot(phase == PH_TRAV) != 0ull) {
                if (phase == PH_TRAV) {
                    uint32_t i = 0;
                    const Recip ra = recip_for(t.a); // shared by every root of this segment
                    for (; i + 4u <= P.n_slots; i += 4u) { // four broadcast reads in flight, four discriminants, then the rare roots
                        const uint4 r0 = lds_spheres[i], r1 = lds_spheres[i + 1u], r2 = lds_spheres[i + 2u], r3 = lds_spheres[i + 3u];
                        float h0, h1, h2, h3, d0, d1, d2, d3;
                        sphere_delta(r0, t, h0, d0);
                        sphere_delta(r1, t, h1, d1);
                        sphere_delta(r2, t, h2, d2);
                        sphere_delta(r3, t, h3, d3);
                        if (fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)) >= 0.0f) { // insertion order, as the reference scans
                            if (d0 >= 0.0f) sphere_root(h0, d0, t, ra, i, t.tbest, t.best);
                            if (d1 >= 0.0f) sphere_root(h1, d1, t, ra, i + 1u, t.tbest, t.best);
                            if (d2 >= 0.0f) sphere_root(h2, d2, t, ra, i + 2u, t.tbest, t.best);
                            if (d3 >= 0.0f) sphere_root(h3, d3, t, ra, i + 3u, t.tbest, t.best);
                        }
                    }
                    for (; i < P.n_slots; ++i) {
                        float h0, d0;
                        sphere_delta(lds_spheres[i], t, h0, d0);
                        if (d0 >= 0.0f) sphere_root(h0, d0, t, ra, i, t.tbest, t.best);
                    }
                    if (STATS) st_sphere += P.n_slots;
                    phase = PH_SHADE;
                }
            }
        }

        PF_MARK(3);
        PF_LANES(6, ballot(phase == PH_DONE));
        // ---- SHADE: compute_color's hit/miss handling (core.cc:242-256) and Material::scatter ------------------
        // unit vectors for every Lambertian / Metallic hit of this round, generated by the whole wave together
        ISA_MARK("request");
        __builtin_amdgcn_s_setprio(0);
        uint32_t rq = RQ_NONE; // Lambertian / Metallic hit: a unit vector; Dielectric hit: one draw
        if (phase == PH_SHADE && t.best < kBlackSample) rq = lds_aux[t.best].w != 2u ? RQ_UNIT : RQ_WORD;
        PF_MARK(4);
        ISA_MARK("draws");
        rng.sample = s; // (defined where it is used: not a register across the walk)
        const V3 unit_vec = coop_draws(rq, rng, P.seed, rank_tbl PB_PASS);
        ISA_MARK("shade");
        PF_MARK(7);
        PB(8, phase == PH_SHADE);
        PB(9, phase == PH_SHADE && t.best < kBlackSample);
        PB(14, phase == PH_SHADE && t.best == ~0u);
        PB(15, phase == PH_SHADE && t.best == ~0u && (natt != 0u || run_n != 0u));
#if defined(RTMI_PROF) && RTMI_PROF == 2
        { // (census votes are taken outside the divergent code they describe: inside it the compiler may move them)
            const uint32_t kind_c = (phase == PH_SHADE && t.best < kBlackSample) ? lds_aux[t.best].w : 3u;
            PB(10, kind_c == 0u);
            PB(11, kind_c == 1u);
            PB(12, kind_c == 2u);
        }
#endif
        PB(22, rq == RQ_WORD || (phase == PH_SHADE && t.best == ~0u)); // the shared normalize(ray.direction)
        if (phase == PH_SHADE) {
            bool ended = false;
            V3 color = mk(0.0f, 0.0f, 0.0f);
            // unit_vector(ray.direction) of the Dielectric scatter (material.defs.cc:60) and of the sky gradient
            // (core.cc:254): one evaluation for the lanes of both branches instead of one per branch (-1.6 %; folding the
            // Metallic branch's normalize(reflect(d, N)) into it as well gained nothing more)
            V3 unit_dir = mk(0.0f, 0.0f, 0.0f);
            if (rq == RQ_WORD || t.best == ~0u) unit_dir = vnormalize(t.d);
            if (t.best == kBlackSample) {
                ended = true; // maxdepth == 0: black sample
            } else if (t.best != ~0u) {
                ISA_MARK("shade-hit-record");
                PF_MARK(8);
                // IntersectionRecord for the winning sphere, object.defs.cc:62-65 and :11-18
                const uint4 sraw = lds_spheres[t.best];
                const uint4 araw = lds_aux[t.best];
                const V3 C = mk(__uint_as_float(sraw.x), __uint_as_float(sraw.y), __uint_as_float(sraw.z));
                const float R = __uint_as_float(araw.z);
                const V3 p = vadd(t.o, vscale(t.d, t.tbest)); // Ray::point_at_param, ray.hpp:9
                const V3 pc = vsub(p, C);
                const V3 outward = vdivs_shared(pc, R, comps_in_range(pc)); // (p - C) / R, object.defs.cc:13
                const bool front = vdot(t.d, outward) < 0.0f;
                const V3 N = front ? outward : vneg(outward);
                const uint32_t mh = araw.y;
                const uint4 m0 = lds_mats[mh]; // {albedo, fuzz} or {refraction index, ...}
                ISA_MARK("shade-material");
                PF_MARK(9);
                const uint32_t kind = araw.w;
                V3 sd = mk(0.0f, 0.0f, 0.0f);
                bool scattered = true;
                if (kind != 2u) {
                    // Lambertian (material.defs.cc:31-42) and Metallic (:44-55) share ONE rejection loop for their
                    // random_unit_vector(): the wave pays the longest run of rejections once, not once per material.
                    V3 rn = mk(0.0f, 0.0f, 0.0f);
                    if (kind == 1u) rn = vnormalize(vreflect(t.d, N));
                    const V3 u = unit_vec; // random_unit_vector(), random.number.gen.hpp:21-29
                    if (kind == 0u) {
                        sd = vadd(N, u);
                        const float eps = 1e-8f; // near_zero, ray.tracer.math.hpp:16-19
                        if (__builtin_fabsf(sd.x) < eps && __builtin_fabsf(sd.y) < eps && __builtin_fabsf(sd.z) < eps) sd = N;
                    } else {
                        sd = vadd(rn, vscale(u, __uint_as_float(m0.w)));
                        scattered = vdot(sd, N) > 0.0f;
                    }
                    PF_MARK(10);
                } else { // Material_Dielectric::scatter, material.defs.cc:57-87
                    ISA_MARK("shade-dielectric");
                    // eta = front ? 1/ri : ri and r1 = ((1 - eta) / (1 + eta))^2 (material.defs.cc:58, 80-82) depend on the
                    // material and the face only: both pairs are computed once on the host with the same fp32 operations
                    const float eta = front ? __uint_as_float(m0.y) : __uint_as_float(m0.x);
                    const float r1 = front ? __uint_as_float(m0.z) : __uint_as_float(m0.w);
                    const float cos_theta = fminf(vdot(vneg(unit_dir), N), 1.0f);
                    const float sin_theta = sqrt_shared(1.0f - cos_theta * cos_theta);
                    bool reflect_it = (eta * sin_theta) > 1.0f;
                    PB(13, !reflect_it);
                    if (!reflect_it) { // short-circuit ||: the draw happens only when refraction is possible
                        // powf(x, 5): x^5 through double is the correctly rounded value except for ties
                        const double xd = (double)(1.0f - cos_theta);
                        const double x2 = xd * xd;
                        const float p5 = (float)((x2 * x2) * xd);
                        const float schlick = r1 + (1.0f - r1) * p5;
                        const double u = (double)__float_as_uint(unit_vec.x) * 2.3283064365386963e-10; // the draw at rng.k
                        rng.k++;
                        reflect_it = (double)schlick > u;
                    }
                    sd = reflect_it ? vreflect(unit_dir, N) : vrefract(unit_dir, N, eta);
                    PF_MARK(11);
                }
                ISA_MARK("shade-continue");
                if (!scattered) {
                    ended = true; // absorbed: compute_color returns 0 (core.cc:251)
                } else {
                    if (kind != 2u) att_push(mh); // dielectric attenuation is (1,1,1): multiplying by it is exact, skip
                    depth_left--;
                    if (depth_left == 0) {
                        ended = true; // the next compute_color call returns 0 (core.cc:238-240)
                    } else {
                        t.o = p;
                        t.d = sd;
                        if (ACCEL == RTMI_ACCEL_BVH) t.cur = P.root_ref;
                        if (ACCEL == RTMI_ACCEL_BVH && BIG && P.walk_starts != nullptr) {
                            // (the start record of the ray just scattered off sphere t.best, read NOW: it is used by the next round's
                            // segment set-up -- a dependent 64-byte read there measured 1.4 % of the config-4 frame)
                            const uint4 ws = P.walk_starts[t.best];
                            t.cur = ws.x;
                            ws1 = ws.y; ws2 = ws.z; ws3 = ws.w;
                        }
                        phase = PH_BEGIN; // set up before the next traversal, together with the new primary rays
                    }
                }
                PF_MARK(12);
            } else {
                ISA_MARK("shade-miss");
                PF_MARK(8);
                // miss: sky gradient (core.cc:254-256), then the attenuations innermost-first (core.cc:247-248)
                const float tt = 0.5f * (unit_dir.y + 1.0f);
                color = vadd(vscale(mk(1.0f, 1.0f, 1.0f), 1.0f - tt), vscale(mk(0.5f, 0.7f, 1.0f), tt));
                ISA_MARK("shade-replay");
                PF_MARK(13);
                if (PACKED) {
                    // the string leaves with the sample: whole 16-byte groups of words (the slot is a multiple of four
                    // words; rows past the last written one are never read back)
                    if (natt != 0u) {
                        uint32_t nw = run_n >> 8;
                        if ((run_n & 255u) != 0u) lds_att[nw++ * blockDim.x + threadIdx.x] = run_h;
                        uint4* dst = reinterpret_cast<uint4*>(P.chain_buf + ((size_t)lpix * spp + s) * P.att_words);
                        for (uint32_t w = 0; w < nw; w += 4u) {
                            const uint32_t* row = lds_att + w * blockDim.x + threadIdx.x;
                            dst[w >> 2] = make_uint4(row[0], row[blockDim.x], row[2u * blockDim.x], row[3u * blockDim.x]);
                        }
                    }
                } else {
                color = att_apply(color, run_h, run_n);
                if (!BIG) {
                    const uint32_t full = natt >> 2; // whole windows that went to HBM; the rest is still in LDS
                    for (uint32_t q = natt; q > 4u * full;) {
                        --q;
                        const uint32_t e = lds_att[(q & 3u) * blockDim.x + threadIdx.x];
                        color = att_apply(color, e & 0xffffu, e >> 16);
                    }
                    // (the strip is read back one block ahead of the multiplies: the next block's load is in flight while
                    // the four runs of this one are applied)
                    const uint4* strip = reinterpret_cast<const uint4*>(P.att_stack) + (size_t)glane * att_blocks;
                    uint4 blk = full != 0u ? strip[full - 1u] : make_uint4(0u, 0u, 0u, 0u);
                    for (uint32_t b = full; b-- > 0u;) {
                        const uint4 cur_blk = blk;
                        if (b != 0u) blk = strip[b - 1u];
                        color = att_apply(color, cur_blk.w & 0xffffu, cur_blk.w >> 16);
                        color = att_apply(color, cur_blk.z & 0xffffu, cur_blk.z >> 16);
                        color = att_apply(color, cur_blk.y & 0xffffu, cur_blk.y >> 16);
                        color = att_apply(color, cur_blk.x & 0xffffu, cur_blk.x >> 16);
                    }
                } else {
                    for (uint32_t q = natt; q-- > 0u;) {
                        const uint32_t h = P.att_stack[((size_t)glane * maxdepth + q) * 2u];
                        const uint32_t n = P.att_stack[((size_t)glane * maxdepth + q) * 2u + 1u];
                        color = att_apply(color, h, n);
                    }
                }
                }
                ended = true;
                PF_MARK(14);
            }
            ISA_MARK("shade-ended");
            PB(16, ended);
            PB(17, ended && t.best != ~0u);
            if (ended) {
                // raytrace_pixel, core.cc:259-265: sequential sum, then scale and pack
                if (!WHOLE) {
                    // one 16-byte record per sample, stored as soon as the sample is finished.  (Round 1 kept an even sample in
                    // three registers until its odd partner could leave with it as one 32-byte sector: half the write-backs at
                    // the fabric, but the path is not bound by HBM and the registers are worth more.)
                    // (.w: 0, or in MODE 4 the length of the chain the resolve pass still has to multiply into the sky colour)
                    const uint32_t pending = (PACKED && t.best == ~0u) ? natt : 0u;
                    P.sample_buf[(size_t)lpix * spp + s] = make_float4(color.x, color.y, color.z, __uint_as_float(pending));
                } else {
                    sum = vadd(sum, color);
                }
                s++;
                if (STATS) st_samples++;
                if (STATS && P.tile_cost != nullptr && s >= s_end) { // probe launch: what this work item cost, into its tile of the whole image
                    // (its ray segments.  Round 6 tried the TIME the lane held the item instead -- a segment of a path trapped inside
                    // the ground sphere costs half an average one -- and it predicted worse: 11.05 against 10.85 ms on config 2, the
                    // 8-way shards 12 % apart instead of 4 %: a lane's rounds last as long as its wave's slowest walk.)
                    const uint32_t gy_c = fdiv(rng.pixel, P.div_w), px_c = rng.pixel - gy_c * W;
                    atomicAdd(&P.tile_cost[(gy_c >> 3) * P.gtiles_x + (px_c >> 3)], st_segments - st_item0);
                    st_item0 = st_segments;
                }
                if (s >= s_end && !WHOLE) {
                    phase = PH_FETCH; // chunk done; rtmi_resolve_kernel finishes the pixel
                } else if (s >= s_end) {
                    const V3 outc = vscale(sum, P.cam.pixels_sample_scale);
                    const size_t o = lpix;
                    if (P.out_rgb) {
                        P.out_rgb[3 * o + 0] = outc.x;
                        P.out_rgb[3 * o + 1] = outc.y;
                        P.out_rgb[3 * o + 2] = outc.z;
                    }
                    if (P.out_rgba) {
                        // RGBAColor(vec3), color.hpp:30-36
                        auto ch = [](float v) -> uint32_t {
                            const float g = v > 0.0f ? __builtin_sqrtf(v) : 0.0f;
                            const float c = g < 0.0f ? 0.0f : (g > 0.999f ? 0.999f : g);
                            return (uint32_t)(uint8_t)(c * 256.0f);
                        };
                        P.out_rgba[o] = ch(outc.x) | (ch(outc.y) << 8) | (ch(outc.z) << 16) | (255u << 24);
                    }
                    phase = PH_FETCH;
                } else {
                    phase = PH_GEN;
                }
                PF_MARK(15);
            }
        }
        ISA_MARK("loop-end");
    }

#ifdef RTMI_TAILPROBE
    if (lane == 0 && P.tail_probe) P.tail_probe[3u * tp_wave + 2u] = wall_clock64();
#endif
#if defined(RTMI_PROF) && RTMI_PROF == 1
    PF_MARK(16);
    if (lane == 0) {
        for (int q = 0; q < PF_SLOTS; ++q) atomicAdd(&P.stats[8 + q], (unsigned long long)pft[q]);
        for (int q = 0; q < 12; ++q) atomicAdd(&P.stats[32 + q], (unsigned long long)pfl[q]);
    }
#elif defined(RTMI_PROF)
    if (lane == 0) {
        for (int q = 0; q < PB_SLOTS; ++q) {
            atomicAdd(&P.stats[64 + 2 * q], (unsigned long long)pb_n[q]);
            atomicAdd(&P.stats[65 + 2 * q], (unsigned long long)pb_l[q]);
        }
    }
#endif
    if (STATS) {
        atomicAdd(&P.stats[0], (unsigned long long)st_samples);
        atomicAdd(&P.stats[1], (unsigned long long)st_segments);
        atomicAdd(&P.stats[2], (unsigned long long)st_sphere);
        atomicAdd(&P.stats[3], (unsigned long long)st_node);
    }
}

#undef glane
#undef lane
