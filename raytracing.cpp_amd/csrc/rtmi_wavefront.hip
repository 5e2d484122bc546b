// rtmi_wavefront.hip -- the path-tracing hot loop as a queue-scheduled persistent kernel for gfx950 (CDNA4).
//
// Same path, same arithmetic and the same draw streams as rtmi_trace_kernel (rtmi_device.hip) -- the frame is
// bit-identical -- but paths are no longer tied to lanes.  What it replaces in the reference is unchanged:
//   RayTracingCore::raytrace_pixel / get_ray / compute_color      src/ray.tracer.core.cc:218-265
//   HittableObject_Collection::intersects / _Sphere::intersects   src/ray.tracer.object.defs.cc:11-18, 41-81
//   Material::scatter (Lambertian / Metallic / Dielectric)        src/ray.tracer.material.defs.cc:31-109
//
// Why: in the round-based kernel a wave's 64 paths are in different phases at any moment, and every phase (primary
// rays, node steps, leaf steps, each material, the sky) runs with the lanes that happen to need it: half of the lanes
// of a vector instruction idle on the RTOW scene (rocprofv3: SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU * 64) = 0.50).
// Here every workgroup keeps a pool of path SLOTS in LDS (structure of arrays, one dword array per field) and four
// rings of slot numbers:
//   Q_T  rays waiting for traversal            Q_H  hits on Lambertian / Metallic spheres (need a unit vector)
//   Q_E  ended or fresh paths (sky colour, sample record, next primary ray)    Q_D  hits on dielectric spheres
// and a wave is whatever the queues need: it pops up to 64 slots of ONE kind, loads their state, runs that kind's code
// with all lanes busy, writes the state back and pushes the slots to the next queue.  A wave that traverses keeps its
// rays in registers and tops its idle lanes up from Q_T whenever `wf_refill` of them are free, so the walk itself
// runs near full width too.  No path state ever leaves the CU: slots, rings and the BVH stack are LDS, the only HBM
// traffic is the scene (staged once), the 16-byte sample records and the framebuffer.
//
// Rings: monotonic head / tail words; a producer reserves positions with one LDS atomic add per wave and kind, a
// consumer claims positions with one compare-and-swap per batch; every cell has exactly one writer and one reader per
// lap (EMPTY -> slot -> EMPTY), so the cells themselves need no atomics.  Every wait is bounded: a watchdog word aborts
// the launch instead of hanging the GPU.
#include "rtmi_kernel_common.h"

#include <string>
#include <type_traits>

namespace {

enum : uint32_t { Q_T = 0, Q_H = 1, Q_D = 2, Q_E = 3, kNumQ = 4 };
// path slot fields (one LDS dword array of wf_slots entries each)
enum : uint32_t {
    F_OX = 0, F_OY, F_OZ, F_DX, F_DY, F_DZ, // the ray of the current segment
    F_T,     // closest root so far (float bits): set by segment set-up (peeled leaves), final after the walk
    F_BEST,  // slot of the closest sphere | kBestNone (sky) | kBestBlack (ended with colour 0) | kBestFresh (no sample yet)
    F_LP,    // pixel index inside the launch's dense slice (sample-record address)
    F_RP,    // absolute pixel = gy * W + px (the key of the draw stream)
    F_S,     // sample | end of the chunk << 16
    F_DEPTH, // segments left | length of the open attenuation run << 16
    F_RUN,   // LDS-resident scene: handle of the open run | closed runs << 16;  HBM-resident scene: closed runs
    F_K,     // draw index
    F_ATT0,  // LDS-resident scene: closed run 0 (handle | count << 16);  HBM-resident scene: handle of the open run
    F_ATT1,  // closed run 1
    F_PAD,   // box pad of this segment (float bits)
    kNumFields
};
static_assert(kNumFields == kWfFields, "the host sizes the slot pool with kWfFields");
constexpr uint32_t kCellEmpty = 0xffffu;
constexpr uint32_t kBestNone = 0xffffffffu, kBestBlack = 0xfffffffeu, kBestFresh = 0xfffffffdu;
enum : uint32_t { C_LIVE = 8, C_ABORT = 9, C_POOLS = 16 }; // control words; [2q] head, [2q + 1] tail of ring q
constexpr uint32_t kSpinLimit = 1u << 22;

// LDS-qualified pointer types: through generic pointers the volatile ring accesses become flat_* instructions with
// system-scope cache bits and full waits (the address-space inference pass leaves volatile accesses alone)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

DEV uint32_t lds_fetch_add(lds_u32* p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV uint32_t lds_fetch_sub(lds_u32* p, uint32_t v) {
    return __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV bool lds_cas(lds_u32* p, uint32_t expected, uint32_t desired) {
    return __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool BIG>
struct Scene { // where the kernel reads the scene from: LDS copies, or HBM behind L1 / L2 / Infinity Cache
    using Ptr = typename std::conditional<BIG, const u32x4*, lds_cu32x4*>::type;
    Ptr spheres, aux, mats, nodes;
    static __device__ __forceinline__ uint4 ld(Ptr p, uint32_t i) {
        const u32x4 v = p[i];
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};

// sphere_root_bvh of rtmi_kernel_common.h with the scene's pointer type for the tie rule's look-up
template <bool BIG>
DEV void sphere_root_tie(float h, float delta, const Trav& t, uint32_t slot, typename Scene<BIG>::Ptr aux, float& tbest,
                         uint32_t& best) {
    const float sqrtd = __builtin_sqrtf(delta);
    float root = (h - sqrtd) / t.a;
    if (!(root > 0.0001f)) root = (h + sqrtd) / t.a;
    if (root > 0.0001f) {
        if (root < tbest) {
            tbest = root;
            best = slot;
        } else if (root == tbest && best != ~0u) {
            if (aux[slot].x < aux[best].x) best = slot;
        }
    }
}

// the spheres of one leaf against the segment in `t` (same routine as the round-based kernel)
template <bool BIG, bool STATS>
DEV void test_leaf(const Scene<BIG>& sc, Trav& t, uint32_t ref, uint32_t& st_sphere) {
    const uint32_t first = BIG ? (ref & 0x00ffffffu) : (ref & 0x1fffu);
    const uint32_t cnt = BIG ? ((ref >> 24) & 0x7fu) : (((ref >> 13) & 3u) + 1u);
    auto pair = [&](uint32_t q) {
        const bool two = q + 1u < cnt;
        const uint4 r0 = Scene<BIG>::ld(sc.spheres, first + q);
        const uint4 r1 = Scene<BIG>::ld(sc.spheres, first + q + (two ? 1u : 0u));
        float h0, h1, d0, d1;
        sphere_delta(r0, t, h0, d0);
        sphere_delta(r1, t, h1, d1);
        if (d0 >= 0.0f) sphere_root_tie<BIG>(h0, d0, t, first + q, sc.aux, t.tbest, t.best);
        if (two & (d1 >= 0.0f)) sphere_root_tie<BIG>(h1, d1, t, first + q + 1u, sc.aux, t.tbest, t.best);
    };
    pair(0u);
    if (cnt > 2u) {
        for (uint32_t q = 2u; q < cnt; q += 2u) pair(q);
    }
    if (STATS) st_sphere += cnt;
}

template <bool STATS, bool BIG, int WPE>
__global__ void __attribute__((amdgpu_waves_per_eu(WPE, WPE))) __launch_bounds__(1024) rtmi_wavefront_kernel(const RtmiLaunch P) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using StackT = typename std::conditional<BIG, uint32_t, uint16_t>::type;
    constexpr uint32_t kStackEnd = BIG ? 0xffffffffu : 0xffffu;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t NS = P.wf_slots, mask = P.wf_cap_mask, cap = mask + 1u;
    const uint32_t W = P.cam.img_width, spp = P.cam.samples_per_pixel, maxdepth = P.cam.maxdepth;
    lds_u32* const fields = (lds_u32*)(lds_raw + P.lds_wf_fields);
    lds_vu16* const cells = (lds_vu16*)(lds_raw + P.lds_wf_rings);
    lds_vu32* const ctrl = (lds_vu32*)(lds_raw + P.lds_wf_ctrl);
    lds_u32* const ctrl_rw = (lds_u32*)(lds_raw + P.lds_wf_ctrl);
    lds_u8* const rank_tbl = (lds_u8*)(ctrl_rw + C_POOLS + wave * 16u); // 64 bytes per wave, see coop_draws
    const uint32_t sp0 = P.lds_stack + threadIdx.x * (uint32_t)sizeof(StackT), sp_stride = blockDim.x * (uint32_t)sizeof(StackT);
    const uint32_t sp1 = sp0 + sp_stride;
    typedef __attribute__((address_space(3))) StackT lds_stack_t;
    *(lds_stack_t*)(lds_raw + sp0) = (StackT)kStackEnd; // entry 0: the sentinel that ends a walk

    // ---- set-up: rings empty, scene staged, every slot FRESH in Q_E -------------------------------------------------
    for (uint32_t i = threadIdx.x; i < kNumQ * cap; i += blockDim.x) cells[i] = (uint16_t)kCellEmpty;
    if (threadIdx.x < C_POOLS) ctrl[threadIdx.x] = 0u;
    Scene<BIG> sc;
    if (BIG) {
        sc.spheres = (typename Scene<BIG>::Ptr)(const u32x4*)P.spheres;
        sc.aux = (typename Scene<BIG>::Ptr)(const u32x4*)P.aux;
        sc.mats = (typename Scene<BIG>::Ptr)(const u32x4*)P.mats;
        sc.nodes = (typename Scene<BIG>::Ptr)(const u32x4*)P.nodes;
    } else {
        lds_u32x4* w_spheres = (lds_u32x4*)(lds_raw + P.lds_spheres);
        lds_u32x4* w_aux = (lds_u32x4*)(lds_raw + P.lds_aux);
        lds_u32x4* w_mats = (lds_u32x4*)(lds_raw + P.lds_mats);
        lds_u32x4* w_nodes = (lds_u32x4*)(lds_raw);
        const u32x4* g_spheres = (const u32x4*)P.spheres;
        const u32x4* g_aux = (const u32x4*)P.aux;
        const u32x4* g_mats = (const u32x4*)P.mats;
        const u32x4* g_nodes = (const u32x4*)P.nodes;
        for (uint32_t i = threadIdx.x; i < P.n_slots; i += blockDim.x) {
            w_spheres[i] = g_spheres[i];
            w_aux[i] = g_aux[i];
        }
        for (uint32_t i = threadIdx.x; i < P.n_mats; i += blockDim.x) w_mats[i] = g_mats[i];
        for (uint32_t i = threadIdx.x; i < 4u * P.n_nodes; i += blockDim.x) w_nodes[i] = g_nodes[i];
        sc.spheres = (typename Scene<BIG>::Ptr)w_spheres;
        sc.aux = (typename Scene<BIG>::Ptr)w_aux;
        sc.mats = (typename Scene<BIG>::Ptr)w_mats;
        sc.nodes = (typename Scene<BIG>::Ptr)w_nodes;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NS; i += blockDim.x) {
        fields[F_BEST * NS + i] = kBestFresh;
        fields[F_S * NS + i] = 0u;
        cells[Q_E * cap + i] = (uint16_t)i;
    }
    if (threadIdx.x == 0u) {
        ctrl[2u * Q_E + 1u] = NS;
        ctrl[C_LIVE] = NS;
    }
    __syncthreads();

    uint32_t st_segments = 0, st_sphere = 0, st_node = 0, st_samples = 0;
    PF_DECL
    const size_t gslot0 = (size_t)blockIdx.x * NS; // first slot of this workgroup in the HBM strip of attenuation runs
    const uint32_t att_blocks = (maxdepth + 1u) >> 1; // 8-byte windows per slot in that strip

    auto fld = [&](uint32_t f, uint32_t slot) -> lds_u32& { return fields[f * NS + slot]; };
    auto raise_abort = [&]() { ctrl[C_ABORT] = 1u; };

    // ---- rings ---------------------------------------------------------------------------------------------------
    // push: wave-uniform call; lanes with `pred` append `slot`.  One atomic add per call reserves the positions.
    auto push = [&](uint32_t q, bool pred, uint32_t slot) {
        const uint64_t m = ballot(pred);
        if (m == 0ull) return;
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        uint32_t base = 0u;
        if (pred && rank == 0u) base = lds_fetch_add(&ctrl_rw[2u * q + 1u], n);
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
        if (pred) {
            lds_vu16* cell = cells + q * cap + ((base + rank) & mask);
            uint32_t guard = 0u;
            while (*cell != (uint16_t)kCellEmpty) { // the reader of the previous lap has not emptied it yet
                __builtin_amdgcn_s_sleep(1);
                if (++guard > kSpinLimit) {
                    raise_abort();
                    break;
                }
            }
            *cell = (uint16_t)slot;
        }
    };
    // pop: wave-uniform call; claims up to `want` entries, lane i < n receives entry i.  Returns n.
    auto pop = [&](uint32_t q, uint32_t want, uint32_t& slot) -> uint32_t {
        uint32_t h = 0u, n = 0u;
        if (lane == 0u) {
#pragma unroll 1
            for (int tries = 0; tries < 8; ++tries) {
                h = ctrl[2u * q];
                const uint32_t t = ctrl[2u * q + 1u];
                n = min(t - h, want);
                if (n == 0u || n > cap) {
                    n = 0u;
                    break;
                }
                if (lds_cas(&ctrl_rw[2u * q], h, h + n)) break;
                n = 0u;
            }
        }
        h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
        slot = 0u;
        if (lane < n) {
            lds_vu16* cell = cells + q * cap + ((h + lane) & mask);
            uint32_t guard = 0u;
            uint32_t v = *cell;
            while (v == kCellEmpty) { // the writer has reserved the position and is about to fill it
                __builtin_amdgcn_s_sleep(1);
                v = *cell;
                if (++guard > kSpinLimit) {
                    raise_abort();
                    v = 0u;
                    break;
                }
            }
            *cell = (uint16_t)kCellEmpty;
            slot = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return n;
    };
    auto qcount = [&](uint32_t q) -> uint32_t {
        const uint32_t h = ctrl[2u * q], t = ctrl[2u * q + 1u];
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)(t - h));
    };

    // ---- attenuation chain of a slot (run-length encoded material handles, see rtmi_device.hip) -----------------------
    struct Chain {
        uint32_t run_h, run_n, natt, att0, att1;
    };
    auto chain_load = [&](uint32_t slot, uint32_t depth_word) -> Chain {
        Chain c;
        c.run_n = depth_word >> 16;
        if (BIG) {
            c.natt = fld(F_RUN, slot);
            c.run_h = fld(F_ATT0, slot);
            c.att0 = 0u;
            c.att1 = 0u;
        } else {
            const uint32_t r = fld(F_RUN, slot);
            c.run_h = r & 0xffffu;
            c.natt = r >> 16;
            c.att0 = fld(F_ATT0, slot);
            c.att1 = fld(F_ATT1, slot);
        }
        return c;
    };
    auto chain_store = [&](uint32_t slot, const Chain& c) {
        if (BIG) {
            fld(F_RUN, slot) = c.natt;
            fld(F_ATT0, slot) = c.run_h;
        } else {
            fld(F_RUN, slot) = c.run_h | (c.natt << 16);
            fld(F_ATT0, slot) = c.att0;
            fld(F_ATT1, slot) = c.att1;
        }
    };
    auto chain_push = [&](uint32_t slot, Chain& c, uint32_t h) {
        if (c.run_n != 0u && h == c.run_h) {
            c.run_n++;
        } else {
            if (c.run_n != 0u) { // close the open run
                const uint32_t q = c.natt++;
                if (!BIG) {
                    // a window of two closed runs lives in the slot; a full window leaves as one 8-byte store
                    const uint32_t e = c.run_h | (c.run_n << 16);
                    if ((q & 1u) == 0u) {
                        c.att0 = e;
                    } else {
                        c.att1 = e;
                        reinterpret_cast<uint2*>(P.att_stack)[(gslot0 + slot) * att_blocks + (q >> 1)] = make_uint2(c.att0, e);
                    }
                } else {
                    P.att_stack[((gslot0 + slot) * maxdepth + q) * 2u] = c.run_h;
                    P.att_stack[((gslot0 + slot) * maxdepth + q) * 2u + 1u] = c.run_n;
                }
            }
            c.run_h = h;
            c.run_n = 1u;
        }
    };
    auto att_apply = [&](V3 color, uint32_t h, uint32_t n) -> V3 {
        const uint4 m0 = Scene<BIG>::ld(sc.mats, h);
        const V3 a = mk(__uint_as_float(m0.x), __uint_as_float(m0.y), __uint_as_float(m0.z));
        for (uint32_t c = 0; c < n; ++c) color = vmul(a, color); // A*(A*(...)): one multiply per bounce, in order
        return color;
    };

    // ---- segment set-up for lanes with `pred`: pad, leaves peeled off the top of the tree, slot fields -----------------
    auto begin_ray = [&](bool pred, uint32_t slot, V3 o, V3 d) {
        if (pred) {
            Trav t{};
            t.o = o;
            t.d = d;
            t.a = vdot(d, d);
            t.tbest = __builtin_inff();
            t.best = ~0u;
            float pad = P.pad_floor;
            for (uint32_t c = 0; c < P.n_pad_classes; ++c) {
                const float* k = P.pad_classes[c];
                const float ax = fmaxf((o.x - k[0]) * (o.x - k[0]), (k[3] - o.x) * (k[3] - o.x));
                const float ay = fmaxf((o.y - k[1]) * (o.y - k[1]), (k[4] - o.y) * (k[4] - o.y));
                const float az = fmaxf((o.z - k[2]) * (o.z - k[2]), (k[5] - o.z) * (k[5] - o.z));
                const float x = P.pad_eps * (((ax + ay) + az) + k[7]); // k[7]: rmax^2 of the class
                pad = fmaxf(pad, fminf(x * k[6], __builtin_amdgcn_sqrtf(x) * 1.000001f));
            }
            for (uint32_t q = 0; q < P.n_pre_leaves; ++q) test_leaf<BIG, STATS>(sc, t, P.pre_leaf[q], st_sphere);
            fld(F_OX, slot) = __float_as_uint(o.x);
            fld(F_OY, slot) = __float_as_uint(o.y);
            fld(F_OZ, slot) = __float_as_uint(o.z);
            fld(F_DX, slot) = __float_as_uint(d.x);
            fld(F_DY, slot) = __float_as_uint(d.y);
            fld(F_DZ, slot) = __float_as_uint(d.z);
            fld(F_T, slot) = __float_as_uint(t.tbest);
            fld(F_BEST, slot) = t.best;
            fld(F_PAD, slot) = __float_as_uint(pad);
            if (STATS) st_segments++;
        }
    };

    // ================================================================================================================
    // job E: ended and fresh paths -- sky colour + attenuation replay (core.cc:247-256), sample record
    // (raytrace_pixel, core.cc:259-265), next work item, next primary ray (get_ray, core.cc:218-234)
    // ================================================================================================================
    auto job_end = [&]() -> bool {
        uint32_t slot;
        const uint32_t n = pop(Q_E, 64u, slot);
        if (n == 0u) return false;
        PF_COUNT(pl0);
        PF_LANES(pl1, ballot(lane < n));
        PF_MARK(pf4);
        const bool valid = lane < n;
        uint32_t best = kBestFresh, sw = 0u, lp = 0u, rp = 0u;
        if (valid) {
            best = fld(F_BEST, slot);
            sw = fld(F_S, slot);
            lp = fld(F_LP, slot);
            rp = fld(F_RP, slot);
        }
        uint32_t s = sw & 0xffffu, s_end = sw >> 16;
        if (valid && best != kBestFresh) {
            V3 color = mk(0.0f, 0.0f, 0.0f);
            if (best == kBestNone) {
                // miss: sky gradient (core.cc:254-256), then the attenuations innermost-first (core.cc:247-248)
                const V3 d = mk(__uint_as_float(fld(F_DX, slot)), __uint_as_float(fld(F_DY, slot)), __uint_as_float(fld(F_DZ, slot)));
                const V3 unit_dir = vnormalize(d);
                const float tt = 0.5f * (unit_dir.y + 1.0f);
                color = vadd(vscale(mk(1.0f, 1.0f, 1.0f), 1.0f - tt), vscale(mk(0.5f, 0.7f, 1.0f), tt));
                const Chain c = chain_load(slot, fld(F_DEPTH, slot));
                color = att_apply(color, c.run_h, c.run_n);
                if (!BIG) {
                    const uint32_t full = c.natt >> 1; // whole windows that went to HBM
                    if (c.natt & 1u) color = att_apply(color, c.att0 & 0xffffu, c.att0 >> 16);
                    for (uint32_t b = full; b-- > 0u;) {
                        const uint2 blk = reinterpret_cast<const uint2*>(P.att_stack)[(gslot0 + slot) * att_blocks + b];
                        color = att_apply(color, blk.y & 0xffffu, blk.y >> 16);
                        color = att_apply(color, blk.x & 0xffffu, blk.x >> 16);
                    }
                } else {
                    for (uint32_t q = c.natt; q-- > 0u;) {
                        const uint32_t h = P.att_stack[((gslot0 + slot) * maxdepth + q) * 2u];
                        const uint32_t cnt = P.att_stack[((gslot0 + slot) * maxdepth + q) * 2u + 1u];
                        color = att_apply(color, h, cnt);
                    }
                }
            }
            P.sample_buf[(size_t)lp * spp + s] = make_float4(color.x, color.y, color.z, 0.0f);
            s++;
            if (STATS) st_samples++;
        }
        PF_MARK(pf5);
        // ---- next work item for the slots whose chunk is finished (one wave-aggregated atomic per round of requests) ---
        bool need = valid && (best == kBestFresh || s >= s_end);
        bool dead = false;
        for (;;) {
            const uint64_t m = ballot(need);
            if (m == 0ull) break;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            // exactly as many indices as slots need them, straight from the global counter: a per-wave reserve could
            // strand indices in a wave that never runs this job again
            const uint32_t take = (uint32_t)__popcll(m);
            uint32_t start = 0u;
            if (need && rank == 0u) start = atomicAdd(P.work_counter, take);
            start = (uint32_t)__shfl((int)start, __ffsll((long long)m) - 1);
            if (need && rank < take) {
                const uint32_t idx = start + rank;
                if (idx >= P.n_work) {
                    dead = true;
                    need = false;
                } else {
                    const uint32_t unit = idx >> 6, j = idx & 63u;
                    const uint32_t tile = fdiv(unit, P.div_chunks), chunk_id = unit - tile * P.n_chunks;
                    const uint32_t trow = fdiv(tile, P.div_tiles_x);
                    const uint32_t tx = tile - trow * P.tiles_x, ty = P.top_down ? trow : P.tiles_y - 1u - trow;
                    const uint32_t px = tx * 8u + (j & 7u), ply = ty * 8u + (j >> 3);
                    if (px < W && ply < P.n_local_rows) {
                        const uint32_t blk = fdiv(ply, P.div_block_rows);
                        const uint32_t gy = P.y_first + blk * P.block_stride * P.block_rows + (ply - blk * P.block_rows);
                        lp = ply * W + px;
                        rp = gy * W + px;
                        s = chunk_id * P.chunk;
                        s_end = min(spp, s + P.chunk);
                        need = false;
                    } // else: a tile position outside the image -- ask again
                }
            }
        }
        {
            const uint32_t ndead = (uint32_t)__popcll(ballot(dead));
            if (ndead && lane == 0u) lds_fetch_sub(&ctrl_rw[C_LIVE], ndead);
        }
        PF_MARK(pf6);
        // ---- GEN: RayTracingCore::get_ray, core.cc:218-234 -------------------------------------------------------------
        const bool alive = valid && !dead;
        bool to_trav = false, to_end = false;
        V3 origin = mk(0.0f, 0.0f, 0.0f), dir = mk(0.0f, 0.0f, 0.0f);
        if (alive) {
            const uint32_t gy = fdiv(rp, P.div_w), px = rp - gy * W;
            Rng rng{0u, rp, s};
            Blk gb = rng_block(rng, 0u, P.seed); // draws 0,1: pixel jitter; 2,3: first defocus-disk attempt
            const float offx = draw_centered(gb.w0);
            const float offy = draw_centered(gb.w1);
            rng.k = 2;
            const V3 du = ld3(P.cam.pixel_delta_u), dv = ld3(P.cam.pixel_delta_v);
            const V3 pixel_sample =
                vadd(vadd(ld3(P.cam.pixel00), vscale(du, (float)px + offx)), vscale(dv, (float)gy + offy));
            origin = ld3(P.cam.cam_center);
            if (!(P.cam.defocus_angle <= 0.0f)) {
                // random_vector_on_unit_disk, random.number.gen.hpp:35-42
                float dx = draw_pm1(gb.w2), dy = draw_pm1(gb.w3);
                rng.k = 4;
                while (!(vdot(mk(dx, dy, 0.0f), mk(dx, dy, 0.0f)) < 1.0f)) { // two attempts per further block
                    if ((rng.k & 3u) == 0u) gb = rng_block(rng, rng.k >> 2, P.seed);
                    dx = draw_pm1((rng.k & 3u) ? gb.w2 : gb.w0);
                    dy = draw_pm1((rng.k & 3u) ? gb.w3 : gb.w1);
                    rng.k += 2u;
                }
                origin = vadd(vadd(ld3(P.cam.cam_center), vscale(ld3(P.cam.defocus_disk_u), dx)),
                              vscale(ld3(P.cam.defocus_disk_v), dy));
            }
            dir = vsub(pixel_sample, origin);
            fld(F_LP, slot) = lp;
            fld(F_RP, slot) = rp;
            fld(F_S, slot) = s | (s_end << 16);
            fld(F_K, slot) = rng.k;
            fld(F_DEPTH, slot) = maxdepth; // run_n = 0
            Chain c{0u, 0u, 0u, 0u, 0u};
            chain_store(slot, c);
            if (maxdepth == 0u) { // compute_color(depth == 0) returns 0 at once (core.cc:238-240): a black sample
                fld(F_BEST, slot) = kBestBlack;
                to_end = true;
            } else {
                to_trav = true;
            }
        }
        PF_MARK(pf7);
        begin_ray(to_trav, slot, origin, dir);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        push(Q_T, to_trav, slot);
        push(Q_E, to_end, slot);
        PF_MARK(pf8);
        return true;
    };

    // ================================================================================================================
    // job H: hits -- IntersectionRecord (object.defs.cc:11-18, 62-65) and Material::scatter (material.defs.cc:31-109)
    // ================================================================================================================
    auto job_hit = [&]() -> bool {
        uint32_t slot;
        uint32_t n = pop(Q_H, 64u, slot);
        if (n < 64u) { // top the batch up with dielectric hits (same code, one draw instead of a unit vector)
            uint32_t slot2;
            const uint32_t n2 = pop(Q_D, 64u - n, slot2);
            const uint32_t got = (uint32_t)__shfl((int)slot2, (int)((lane - n) & 63u));
            if (lane >= n && lane < n + n2) slot = got;
            n += n2;
        }
        if (n == 0u) return false;
        PF_COUNT(pl2);
        PF_LANES(pl3, ballot(lane < n));
        PF_MARK(pf4);
        const bool valid = lane < n;
        V3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 0.0f);
        float tbest = 0.0f;
        uint32_t best = 0u, depth_word = 0u;
        Rng rng{0u, 0u, 0u};
        uint4 sraw = make_uint4(0u, 0u, 0u, 0u), araw = make_uint4(0u, 0u, 0u, 0u);
        uint32_t rq = RQ_NONE;
        if (valid) {
            o = mk(__uint_as_float(fld(F_OX, slot)), __uint_as_float(fld(F_OY, slot)), __uint_as_float(fld(F_OZ, slot)));
            d = mk(__uint_as_float(fld(F_DX, slot)), __uint_as_float(fld(F_DY, slot)), __uint_as_float(fld(F_DZ, slot)));
            tbest = __uint_as_float(fld(F_T, slot));
            best = fld(F_BEST, slot);
            depth_word = fld(F_DEPTH, slot);
            rng.k = fld(F_K, slot);
            rng.pixel = fld(F_RP, slot);
            rng.sample = fld(F_S, slot) & 0xffffu;
            sraw = Scene<BIG>::ld(sc.spheres, best);
            araw = Scene<BIG>::ld(sc.aux, best);
            rq = araw.w != 2u ? RQ_UNIT : RQ_WORD;
        }
        PF_MARK(pf9);
        // unit vectors (Lambertian / Metallic) and the dielectric's draw, generated by the whole wave together
        const V3 unit_vec = coop_draws(rq, rng, P.seed, rank_tbl);
        PF_MARK(pf10);
        bool to_trav = false, to_end = false;
        V3 p = mk(0.0f, 0.0f, 0.0f), sd = mk(0.0f, 0.0f, 0.0f);
        if (valid) {
            // IntersectionRecord for the winning sphere, object.defs.cc:62-65 and :11-18
            const V3 C = mk(__uint_as_float(sraw.x), __uint_as_float(sraw.y), __uint_as_float(sraw.z));
            const float R = __uint_as_float(araw.z);
            p = vadd(o, vscale(d, tbest)); // Ray::point_at_param, ray.hpp:9
            const V3 outward = vdivs(vsub(p, C), R);
            const bool front = vdot(d, outward) < 0.0f;
            const V3 N = front ? outward : vneg(outward);
            const uint32_t mh = araw.y;
            const uint4 m0 = Scene<BIG>::ld(sc.mats, mh); // {albedo, fuzz} or {refraction index, ...}
            const uint32_t kind = araw.w;
            bool scattered = true;
            if (kind != 2u) {
                // Lambertian (material.defs.cc:31-42) and Metallic (:44-55)
                V3 rn = mk(0.0f, 0.0f, 0.0f);
                if (kind == 1u) rn = vnormalize(vreflect(d, N));
                const V3 u = unit_vec; // random_unit_vector(), random.number.gen.hpp:21-29
                if (kind == 0u) {
                    sd = vadd(N, u);
                    const float eps = 1e-8f; // near_zero, ray.tracer.math.hpp:16-19
                    if (__builtin_fabsf(sd.x) < eps && __builtin_fabsf(sd.y) < eps && __builtin_fabsf(sd.z) < eps) sd = N;
                } else {
                    sd = vadd(rn, vscale(u, __uint_as_float(m0.w)));
                    scattered = vdot(sd, N) > 0.0f;
                }
            } else { // Material_Dielectric::scatter, material.defs.cc:57-87
                // eta = front ? 1/ri : ri and r1 = ((1 - eta) / (1 + eta))^2 (material.defs.cc:58, 80-82) depend on the
                // material and the face only: both pairs are computed once on the host with the same fp32 operations
                const float eta = front ? __uint_as_float(m0.y) : __uint_as_float(m0.x);
                const float r1 = front ? __uint_as_float(m0.z) : __uint_as_float(m0.w);
                const V3 unit_dir = vnormalize(d);
                const float cos_theta = fminf(vdot(vneg(unit_dir), N), 1.0f);
                const float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
                bool reflect_it = (eta * sin_theta) > 1.0f;
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
            }
            uint32_t depth_left = depth_word & 0xffffu;
            if (!scattered) {
                to_end = true; // absorbed: compute_color returns 0 (core.cc:251)
            } else {
                if (kind != 2u) { // dielectric attenuation is (1,1,1): multiplying by it is exact, skip
                    Chain c = chain_load(slot, depth_word);
                    chain_push(slot, c, mh);
                    chain_store(slot, c);
                    depth_word = (depth_word & 0xffffu) | (c.run_n << 16);
                }
                depth_left--;
                if (depth_left == 0u) {
                    to_end = true; // the next compute_color call returns 0 (core.cc:238-240)
                } else {
                    to_trav = true;
                    fld(F_DEPTH, slot) = depth_left | (depth_word & 0xffff0000u);
                    fld(F_K, slot) = rng.k;
                }
            }
            if (to_end) fld(F_BEST, slot) = kBestBlack;
        }
        PF_MARK(pf11);
        begin_ray(to_trav, slot, p, sd);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        push(Q_T, to_trav, slot);
        push(Q_E, to_end, slot);
        PF_MARK(pf1);
        return true;
    };

    // ================================================================================================================
    // mode T: the BVH walk (exact-equivalent of HittableObject_Collection::intersects, object.defs.cc:68-81).  The wave
    // keeps 64 rays in registers; finished lanes hand their hit to the slot and the slot to the queue of its kind, idle
    // lanes take new rays from Q_T.  Ends when Q_T is empty and every lane is idle.
    // ================================================================================================================
    auto mode_trav = [&]() {
        uint32_t tst = 0u; // 0 idle, 1 walking, 2 finished (result not handed over yet)
        uint32_t slot = 0u;
        Trav t{};
        const int target = 64 - (int)P.wf_refill;
        for (;;) {
            if (ctrl[C_ABORT] != 0u) break;
            PF_MARK(pf2);
            // ---- hand finished segments over -------------------------------------------------------------------------
            {
                const bool done = tst == 2u;
                uint32_t kind = 3u; // 3: sky
                if (done) {
                    fld(F_T, slot) = __float_as_uint(t.tbest);
                    fld(F_BEST, slot) = t.best;
                    if (t.best != ~0u) kind = sc.aux[t.best].w;
                }
                if (ballot(done) != 0ull) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    push(Q_H, done && kind < 2u, slot);
                    push(Q_E, done && kind == 3u, slot);
                    push(Q_D, done && kind == 2u, slot);
                }
                if (done) tst = 0u;
            }
            // ---- take new rays ---------------------------------------------------------------------------------------
            const uint64_t m_idle = ballot(tst == 0u);
            const uint32_t n_idle = (uint32_t)__popcll(m_idle);
            uint32_t got = 0u;
            if (n_idle != 0u) {
                uint32_t popped;
                got = pop(Q_T, n_idle, popped);
                if (got != 0u) {
                    PF_COUNT(pl6);
#ifdef RTMI_PROF
                    pl7 += got;
#endif
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                    const uint32_t mine = (uint32_t)__shfl((int)popped, (int)(r & 63u));
                    if (tst == 0u && r < got) {
                        slot = mine;
                        t.o = mk(__uint_as_float(fld(F_OX, slot)), __uint_as_float(fld(F_OY, slot)), __uint_as_float(fld(F_OZ, slot)));
                        t.d = mk(__uint_as_float(fld(F_DX, slot)), __uint_as_float(fld(F_DY, slot)), __uint_as_float(fld(F_DZ, slot)));
                        t.tbest = __uint_as_float(fld(F_T, slot));
                        t.best = fld(F_BEST, slot);
                        const float pad = __uint_as_float(fld(F_PAD, slot));
                        t.a = vdot(t.d, t.d);
                        t.inv = mk(__builtin_amdgcn_rcpf(t.d.x), __builtin_amdgcn_rcpf(t.d.y), __builtin_amdgcn_rcpf(t.d.z));
                        t.oinv = mk(-(t.o.x * t.inv.x), -(t.o.y * t.inv.y), -(t.o.z * t.inv.z));
                        t.pinv = mk(pad * __builtin_fabsf(t.inv.x), pad * __builtin_fabsf(t.inv.y), pad * __builtin_fabsf(t.inv.z));
                        t.cur = P.root_ref;
                        t.sp = sp1;
                        tst = P.root_ref == kNoWalk ? 2u : 1u; // kNoWalk: the peeled leaves were the whole tree
                    }
                }
            }
            const int n_act = (int)__popcll(ballot(tst == 1u));
            PF_MARK(pf3);
            if (n_act == 0) {
                if (ballot(tst == 2u) != 0ull) continue; // results to hand over
                if (got == 0u) break;                     // nothing walking, nothing waiting: the episode is over
                continue;
            }
            // walk until enough lanes are free for the next top-up; when Q_T ran dry, until eight more are done
            const int floor = n_act > target ? target : (n_act > 8 ? n_act - 8 : 0);
            for (;;) {
                const bool at_leaf = t.cur >= (BIG ? kLeafBit : 0x8000u);
                const uint64_t m_trav = ballot(tst == 1u);
                const uint64_t m_leaf = ballot(at_leaf) & m_trav;
                const uint64_t m_node = m_trav & ~m_leaf;
                int n_leaf = (int)__popcll(m_leaf), n_node = (int)__popcll(m_node);
                asm volatile("" : "+s"(n_leaf), "+s"(n_node));
                if (n_leaf + n_node <= floor) break;
                PF_COUNT(pl4);
                PF_LANES(pl5, n_leaf > n_node ? m_leaf : m_node);
                bool popit = false;
                if (n_leaf > n_node) {
                    if (tst == 1u && at_leaf) {
                        test_leaf<BIG, STATS>(sc, t, t.cur, st_sphere);
                        popit = true;
                    }
                } else if (tst == 1u && !at_leaf) {
                    NodeFields nd;
                    if (BIG) { // 48-byte records (half extents fp16), see rtmi_kernel_common.h
                        const typename Scene<BIG>::Ptr np = sc.nodes + 3u * t.cur;
                        const u32x4 n0 = np[0], n1 = np[1], n2 = np[2];
                        nd = unpack_node48(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z);
                    } else {
                        const typename Scene<BIG>::Ptr np = sc.nodes + 4u * t.cur;
                        const u32x4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                        nd = unpack_node64(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y);
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
                    const float tf0 = fminf(fminf(tc0x + th0x, tc0y + th0y), fminf(tc0z + th0z, t.tbest));
                    const float tn1 = fmaxf(fmaxf(tc1x - th1x, tc1y - th1y), fmaxf(tc1z - th1z, 0.0001f));
                    const float tf1 = fminf(fminf(tc1x + th1x, tc1y + th1y), fminf(tc1z + th1z, t.tbest));
                    if (STATS) st_node += 2;
                    const bool hit0 = tn0 <= tf0, hit1 = tn1 <= tf1;
                    const uint32_t ch0 = nd.ch0, ch1 = nd.ch1;
                    asm volatile("" ::"v"(ch0), "v"(ch1)); // keep the child-reference read with the box reads
                    const bool nearer1 = tn1 < tn0;
                    const bool take1 = hit1 & (!hit0 | nearer1); // nearer child first
                    if (hit0 & hit1) {
                        *(lds_stack_t*)(lds_raw + t.sp) = (StackT)(take1 ? ch0 : ch1);
                        t.sp += sp_stride;
                    }
                    t.cur = take1 ? ch1 : ch0; // overwritten by the pop when neither box is hit
                    popit = !(hit0 | hit1);
                }
                if (popit) {
                    t.sp -= sp_stride;
                    t.cur = *(lds_stack_t*)(lds_raw + t.sp);
                    if (t.cur == kStackEnd) tst = 2u; // the sentinel: stack empty
                }
            }
        }
    };

    // ================================================================================================================
    // scheduler: a wave takes whatever is plentiful; partial batches only after it found nothing full twice
    // ================================================================================================================
    uint32_t idle = 0u;
    for (;;) {
        if (ctrl[C_ABORT] != 0u) break;
        const uint32_t cE = qcount(Q_E), cH = qcount(Q_H) + qcount(Q_D), cT = qcount(Q_T);
        const uint32_t full = idle < 2u ? 64u : 1u;
        uint32_t job = 0u; // 1: hits, 2: ended / fresh, 3: traversal
        if (cH >= full && cH >= cE) job = 1u;
        else if (cE >= full) job = 2u;
        else if (cT >= full) job = 3u;
        bool did = false;
        PF_MARK(pf4);
        if (job == 1u) {
            did = job_hit();
            PF_MARK(pf1);
        } else if (job == 2u) {
            did = job_end();
            PF_MARK(pf0);
        } else if (job == 3u) {
            mode_trav();
            PF_MARK(pf2);
            did = true;
        }
        if (did) {
            idle = 0u;
            continue;
        }
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)ctrl[C_LIVE]) == 0u) break;
        __builtin_amdgcn_s_sleep(4);
        if (++idle > kSpinLimit) {
            raise_abort();
            break;
        }
    }

#ifdef RTMI_PROF
    PF_MARK(pf4);
    if (lane == 0) {
        unsigned long long* const pst = P.stats + 8;
        atomicAdd(&pst[0], pf0); atomicAdd(&pst[1], pf1); atomicAdd(&pst[2], pf2); atomicAdd(&pst[3], pf3); atomicAdd(&pst[4], pf4);
        atomicAdd(&pst[5], pf5); atomicAdd(&pst[6], pf6); atomicAdd(&pst[7], pf7); atomicAdd(&pst[24], pf8); atomicAdd(&pst[25], pf9);
        atomicAdd(&pst[26], pf10); atomicAdd(&pst[27], pf11);
        atomicAdd(&pst[8], pl0); atomicAdd(&pst[9], pl1); atomicAdd(&pst[10], pl2); atomicAdd(&pst[11], pl3);
        atomicAdd(&pst[12], pl4); atomicAdd(&pst[13], pl5); atomicAdd(&pst[14], pl6); atomicAdd(&pst[15], pl7);
    }
#endif
    if (ctrl[C_ABORT] != 0u && lane == 0u) atomicOr(P.wf_error, 1u);
    if (STATS) {
        atomicAdd(&P.stats[0], (unsigned long long)st_samples);
        atomicAdd(&P.stats[1], (unsigned long long)st_segments);
        atomicAdd(&P.stats[2], (unsigned long long)st_sphere);
        atomicAdd(&P.stats[3], (unsigned long long)st_node);
    }
}

using WfKernel = void (*)(const RtmiLaunch);

WfKernel pick(bool stats, bool big, int wpe) {
    if (wpe >= 6) {
        if (big) return stats ? rtmi_wavefront_kernel<true, true, 6> : rtmi_wavefront_kernel<false, true, 6>;
        return stats ? rtmi_wavefront_kernel<true, false, 6> : rtmi_wavefront_kernel<false, false, 6>;
    }
    if (big) return stats ? rtmi_wavefront_kernel<true, true, 4> : rtmi_wavefront_kernel<false, true, 4>;
    return stats ? rtmi_wavefront_kernel<true, false, 4> : rtmi_wavefront_kernel<false, false, 4>;
}

} // namespace

int rtmi_wavefront_occupancy(bool stats, bool big, int waves_per_eu, uint32_t block, uint32_t lds_bytes, int* per_cu) {
    WfKernel fn = pick(stats, big, waves_per_eu);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, fn, (int)block, lds_bytes);
    if (e != hipSuccess) {
        set_error(std::string("rtmi wavefront kernel: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RTMI_ERR_OOM : RTMI_ERR_HIP;
    }
    return RTMI_OK;
}

int rtmi_wavefront_launch(const RtmiLaunch& P, bool stats, bool big, int waves_per_eu, uint32_t grid, uint32_t block,
                          uint32_t lds_bytes, hipStream_t stream) {
    RtmiLaunch copy = P;
    void* args[] = {&copy};
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(pick(stats, big, waves_per_eu)), dim3(grid), dim3(block),
                                         args, lds_bytes, stream);
    if (e != hipSuccess) {
        set_error(std::string("rtmi wavefront kernel launch: ") + hipGetErrorString(e));
        return RTMI_ERR_HIP;
    }
    return RTMI_OK;
}
