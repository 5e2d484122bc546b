// rtmi_kernel_common.h -- device-side pieces of the trace kernel (rtmi_device.hip: BVH walk and linear scan): launch
// parameters, glm-semantics vec3, the counter RNG, the wave-cooperative draw service and the reference's sphere arithmetic.  Reference lines are cited at each function.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rtmi_internal.h"

using namespace rtmi;

// Division by a launch-invariant 32-bit divisor (Granlund & Montgomery, PLDI'94, fig. 4.1): q = n / d for every
// 32-bit n, five instructions instead of the ~40 of an integer division.
struct FastDiv {
    uint32_t m, sh1, sh2;
};
static FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{1u, 0u, 0u};
    if (d == 0u) d = 1u;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l; // ceil(log2 d)
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
    f.sh1 = l < 1u ? l : 1u;
    f.sh2 = l > 0u ? l - 1u : 0u;
    return f;
}

// ---------------------------------------------------------------------------------------------------------
// launch parameters (kernarg -> SGPRs)
// ---------------------------------------------------------------------------------------------------------
struct RtmiLaunch {
    rtmi_camera cam;
    // scene, global memory (staged into LDS by every workgroup)
    const uint4* spheres;  // [n_slots] {cx, cy, cz, r*r} as bits
    const uint4* aux;      // [n_slots] {object index, material handle, radius bits, MaterialKind of that handle}
    const uint4* mats;     // [n_mats]  {p0, p1, p2, p3} (albedo + fuzz, or refraction index in p0); the kind rides in aux.w
    const uint4* nodes;    // [n_nodes] 3 x uint4 per node (48-byte records, see unpack_node48)
    uint32_t n_slots, n_mats, n_nodes, root_ref;
    uint32_t pre_leaf[4];     // leaves hanging off the top of the tree (the ground sphere): tested at segment set-up
    uint32_t n_pre_leaves;    // root_ref == kNoWalk: they were the whole tree
    // camera rays: where the walks of the samples of each 8x8 tile of the WHOLE image start (tile = (gy / 8) * gtiles_x + px / 8): a
    // node or leaf reference in the kernel variant's format, kStackEnd = the tile's beam meets no sphere of the tree; NULL: the root
    const uint32_t* tile_entry;
    // scattered rays (HBM-resident trees): per sphere slot one 16-byte start record {x: reference the walk of a ray off that sphere
    // starts at; y: n | path code of the top levels << 4 | deeper way index 0 << 12; z, w: deeper way indices 1-3, 20 bits each}: the
    // n way records pre-loaded on the stack are, for the first min(n, way_jtop) levels, at way_top_base + (4^j - 4) / 3 + (code >>
    // 2 (n_top - j)) -- placed by path code in the staged block -- and the named ones below (host: build_walk_starts); NULL: the root
    const uint4* walk_starts;
    uint32_t way_jtop, way_top_base;
    float pad_classes[kMaxPadClasses][8];
    uint32_t n_pad_classes;
    float pad_eps, pad_floor;
    uint32_t pad_refine;             // nonzero: the class pad is bounded by the segment's reach
    float pad_rmax[kMaxPadClasses];  // sqrt(pad_classes[c][7]), rounded up
    // LDS carve-up (byte offsets)
    uint32_t lds_spheres, lds_aux, lds_mats, lds_nodes, lds_stack, stack_depth, lds_att, lds_pool;
    uint32_t lds_top_nodes;   // HBM-resident trees: this many breadth-first nodes (48-byte records) start the LDS segment
    // image rows handled by this launch
    uint32_t y_first, block_rows, block_stride, n_local_rows;
    // a launch over a LIST of row blocks instead of a strided set (rtmi_render_block_list_device): first image row of local block k
    // (multiples of 8; block_rows a multiple of 8), NULL = block k starts at y_first + k * block_stride * block_rows
    const uint32_t* block_first_row;
    uint32_t x_first, x_end, local_w; // columns [x_first, x_end) of every row (rtmi_render_rect; whole rows: 0, W, W); local_w = x_end - x_first
    uint32_t tiles_x, tiles_y, n_work; // work index space = tiles * 64
    FastDiv div_tiles_x, div_chunks, div_block_rows;
    uint32_t top_down;
    // the order the 8x8 tiles of this launch are handed out in: tile_order[k] = local tile (ty * tiles_x + tx) of hand-out
    // position k, the costliest first (host: order_for(), from the per-tile segment counts of a probe launch); NULL: row by
    // row, bottom rows first.  Scheduling only: the draw streams are keyed by the absolute pixel.
    const uint32_t* tile_order;
    // probe launches (STATS variants only): segments per 8x8 tile of the WHOLE image, tile = (gy / 8) * gtiles_x + px / 8
    uint32_t* tile_cost;
    uint32_t gtiles_x;
    unsigned long long* tail_probe; // -DRTMI_TAILPROBE builds only: per wave {start, first refill past the end, exit} in 100 MHz ticks
    // sample-chunk split: a work item is `chunk` consecutive samples of one pixel; their colours go to sample_buf
    // ([pixel][sample] float4) and rtmi_resolve_kernel adds them up in sample order.  n_chunks == 1: a lane owns
    // the whole pixel and sums in registers.
    uint32_t chunk, n_chunks;
    float4* sample_buf;
    uint32_t wait_thresh;     // leave the traversal loop when this many lanes of a wave wait for shading
    uint64_t seed;
    float* out_rgb;
    uint32_t* out_rgba;
    uint32_t* work_counter;
    uint32_t* att_stack; // [lane][maxdepth] {handle, count}: attenuation runs that did not fit LDS
    // packed attenuation chains (MODE 4): every non-dielectric bounce appends its material handle, att_bits wide, to a
    // per-lane bit string in LDS (att_epw handles per 32-bit word, none straddling); a path that reaches the sky copies
    // its string to chain_buf[pixel][sample] (att_words words, a multiple of 4) and rtmi_resolve_kernel multiplies
    uint32_t att_bits, att_epw, att_words;
    uint32_t* chain_buf;
    unsigned long long* stats; // {samples, segments, sphere_tests, node_tests}
    FastDiv div_w;          // by the image width
};

#define DEV static __device__ __forceinline__

DEV uint32_t fdiv(uint32_t n, const FastDiv f) {
    const uint32_t t1 = __umulhi(f.m, n);
    return (t1 + ((n - t1) >> f.sh1)) >> f.sh2;
}

// The node record on the device, in LDS and in HBM, is 48 bytes: both centres fp32, the six half extents fp16 (rounded up on the
// host), two child references -- three 16-byte reads per step.  (Rounds 1-3 kept 64-byte all-fp32 records for trees in LDS;
// see walk_nodes_lds in rtmi_device.hip for the A/B that retired them.)
struct NodeFields {
    float c0x, c0y, c0z, c1x, c1y, c1z, h0x, h0y, h0z, h1x, h1y, h1z;
    uint32_t ch0, ch1;
};
DEV float half_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
DEV float half_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
DEV NodeFields unpack_node48(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b0, uint32_t b1, uint32_t b2,
                             uint32_t b3, uint32_t c0, uint32_t c1, uint32_t c2) {
    NodeFields n;
    n.c0x = __uint_as_float(a0); n.c0y = __uint_as_float(a1); n.c0z = __uint_as_float(a2);
    n.c1x = __uint_as_float(a3); n.c1y = __uint_as_float(b0); n.c1z = __uint_as_float(b1);
    n.h0x = half_lo(b2); n.h0y = half_hi(b2); n.h0z = half_lo(b3);
    n.h1x = half_hi(b3); n.h1y = half_lo(c0); n.h1z = half_hi(c0);
    n.ch0 = c1; n.ch1 = c2;
    return n;
}

// In-kernel stamps (diagnostic build only, -DRTMI_PROF): s_memtime deltas per phase of the v1 kernel, summed per wave
// into stats[8 + i].  Never compiled into the shipped library.
// In-kernel stamps and counters (diagnostic builds only, never the shipped library; tools/branch_census.py):
//   -DRTMI_PROF=1  pft[i]: cycles (s_memtime deltas) of section i of a round; pfl[i]: trips and lanes of the walk
//   -DRTMI_PROF=2  branch census: how often a piece of code runs with at least one lane (pb_n) and with how many (pb_l)
// All of them are wave-uniform 32-bit scalars (two builds, because together they do not fit the scalar registers),
// added into P.stats[8 ..], [32 ..], [64 + 2 i] at the end of the kernel.
#define PF_SLOTS 24
#define PB_SLOTS 24
#if defined(RTMI_PROF) && RTMI_PROF == 1
#define PF_DECL uint32_t pf_t = (uint32_t)__builtin_readcyclecounter(), pft[PF_SLOTS] = {}, pfl[12] = {};
#define PF_MARK(i) do { const uint32_t n_ = (uint32_t)__builtin_readcyclecounter(); pft[i] += n_ - pf_t; pf_t = n_; } while (0)
#define PF_COUNT(i) do { pfl[i] += 1u; } while (0)
#define PF_LANES(i, mask) do { pfl[i] += (uint32_t)__popcll(mask); } while (0)
#define PB_ARGS , uint32_t* pft, uint32_t& pf_t
#define PB_PASS , pft, pf_t
#else
#define PF_DECL
#define PF_MARK(i) do { } while (0)
#define PF_COUNT(i) do { } while (0)
#define PF_LANES(i, mask) do { } while (0)
#endif
#if defined(RTMI_PROF) && RTMI_PROF == 2
#define PB_DECL uint32_t pb_n[PB_SLOTS] = {}, pb_l[PB_SLOTS] = {};
#define PB(i, cond) do { const uint64_t m_ = __builtin_amdgcn_ballot_w64(cond); pb_n[i] += m_ != 0ull ? 1u : 0u; pb_l[i] += (uint32_t)__popcll(m_); } while (0)
#define PB_ARGS , uint32_t* pb_n, uint32_t* pb_l
#define PB_PASS , pb_n, pb_l
#else
#define PB_DECL
#define PB(i, cond) do { } while (0)
#endif
#ifndef PB_ARGS
#define PB_ARGS
#define PB_PASS
#endif

// Static code map (diagnostic build only, -DRTMI_MARKS): comment lines in the ISA at the phase boundaries, so that
// tools/isa_phases.py can count the instructions of each phase.
#ifdef RTMI_MARKS
#define ISA_MARK(name) asm volatile("; @@" name)
#else
#define ISA_MARK(name) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// vec3 with glm's published semantics (glm is an un-vendored dependency of the reference)
// ---------------------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 vmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV V3 vscale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV V3 vdivs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
DEV V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV float vdot(V3 a, V3 b) { // glm::dot: t = a*b; t.x + t.y + t.z
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return (tx + ty) + tz;
}
DEV float sqrt_shared(float x);
DEV float recip_shared(float den);
DEV V3 vnormalize(V3 v) { return vscale(v, recip_shared(sqrt_shared(vdot(v, v)))); } // v * inversesqrt(dot(v,v)) = v * (1.0f / sqrt)
DEV V3 vreflect(V3 I, V3 N) { return vsub(I, vscale(vscale(N, vdot(N, I)), 2.0f)); }
DEV V3 vrefract(V3 I, V3 N, float eta) {
    const float d = vdot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k >= 0.0f) return vsub(vscale(I, eta), vscale(N, eta * d + sqrt_shared(k)));
    return mk(0.0f, 0.0f, 0.0f);
}
DEV V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// Correctly rounded fp32 division the way the compiler expands `/` for this target (v_div_scale x 2, v_rcp, four FMAs,
// v_mul, v_div_fmas, v_div_fixup), minus the three operations that do nothing while the operands are in range: with
// the denominator in [2^-40, 2^40] and a quotient that matters (|q| in [1e-4, 2^40)) v_div_scale passes both operands
// through unscaled, v_div_fmas is a plain FMA and v_div_fixup returns its input, so the sequence below IS that expansion,
// bit for bit -- and the reciprocal (v_rcp + two FMAs) is shared by every division by the same denominator.
struct Recip {
    float den, r; // r = rcp(den) after one Newton step, as in the expansion
    bool in_range;
};
DEV Recip recip_for(float den) {
    Recip q;
    q.den = den;
    const float r0 = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r0, 1.0f);
    q.r = __builtin_fmaf(e, r0, r0);
    q.in_range = __builtin_fabsf(den) >= 0x1p-40f && __builtin_fabsf(den) <= 0x1p40f; // (false for NaN)
    return q;
}
DEV float div_in_range(float num, const Recip& d) {
    const float q0 = num * d.r;
    const float rem0 = __builtin_fmaf(-d.den, q0, num);
    const float q1 = __builtin_fmaf(rem0, d.r, q0);
    const float rem1 = __builtin_fmaf(-d.den, q1, num);
    return __builtin_fmaf(rem1, d.r, q1);
}

// Correctly rounded fp32 square root, likewise: the compiler's expansion of sqrtf is v_sqrt_f32 plus two residual tests
// that move the estimate one ulp down or up; around that it rescales arguments below 2^-96 and passes 0 / inf through.
// For 2^-90 <= x <= 2^90 those wrappers do nothing and the core below is the whole expansion.
DEV float sqrt_core(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float q = r_dn <= 0.0f ? s_dn : s;
    q = r_up > 0.0f ? s_up : q;
    return q;
}
DEV float sqrt_shared(float x) {
    float q = sqrt_core(x);
    if (!(x >= 0x1p-90f && x <= 0x1p90f)) { // (also NaN, 0, negative: the full expansion)
        asm volatile("" : "+v"(x)); // a real branch: left alone the compiler evaluates both square roots and selects
        q = __builtin_sqrtf(x);
    }
    return q;
}

// 1.0f / den: the in-range core of the division expansion, the full expansion for denominators out of range
DEV float recip_shared(float den) {
    const Recip r = recip_for(den);
    float q = div_in_range(1.0f, r);
    if (!r.in_range) {
        asm volatile("" : "+v"(den)); // a real branch (see sqrt_shared)
        q = 1.0f / den;
    }
    return q;
}

// a / s, component by component (glm's vec3 / scalar), on one shared reciprocal.  In range means: |s| in [2^-40, 2^40]
// and every component of `a` is +0 or at least 2^-60 in magnitude (a -0 would come out as +0, a tinier one would be
// rescaled by the expansion); the caller vouches for `comps_ok` or passes the test below (which sends zeros the slow way too).
DEV bool comps_in_range(V3 a) {
    return __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z)) >= 0x1p-60f;
}
DEV V3 vdivs_shared(V3 a, float s, bool comps_ok) {
    const Recip rs = recip_for(s);
    V3 q = mk(div_in_range(a.x, rs), div_in_range(a.y, rs), div_in_range(a.z, rs));
    if (!(rs.in_range && comps_ok)) q = vdivs(a, s); // the full expansion (rare: the wave skips it)
    return q;
}


// ---------------------------------------------------------------------------------------------------------
// counter RNG: draw #k of (seed, pixel, sample) = word (k & 3) of block k >> 2 of that stream (rng4x32 below);
// random_double() = u32 * 2^-32.  The affine maps of random.number.gen.hpp are exact in double for a 32-bit
// draw, so the double -> float narrowing of the reference equals one int -> float conversion here.
// ---------------------------------------------------------------------------------------------------------
struct Rng { // per-lane stream position; no cached block: every consumer asks for the block it needs
    uint32_t k, pixel, sample;
};
struct Blk {
    uint32_t w0, w1, w2, w3;
};

// The block function of the counter RNG: 128 bits from (block, sample, pixel | seed).  RTMI_RNG picks it at build time
// (the oracle has the same switch at run time, oracle/rt_oracle.c):
//   10, 7   Philox4x32 with that many rounds (Salmon et al., SC'11; 7 rounds are the fewest that pass BigCrush there,
//           10 the library default with its safety margin)
//   0       pcg4d (Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 9(3), 2020): 12 multiplies, plus one
//           more xorshift on the way out (without it the low three bits of a word are biased: 25 / 37 / 44 % ones over
//           the pixels of a frame, measured; with it every bit is within 0.1 % of one half)
// Issue cost on MI355X (tools/ubench: v_mul_lo / v_mul_hi 1.7x a v_fma_f32): ~280 / ~196 / ~130 cycles per block;
// whole 1080p x 512 spp frame, same build otherwise: 186.8 / 178.6 / 172.1 ms (tools/rng_ab.py).  Shipped: 0.
#ifndef RTMI_RNG
#define RTMI_RNG 0
#endif
template <int ROUNDS>
DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, Blk& r) {
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r.w0 = c0; r.w1 = c1; r.w2 = c2; r.w3 = c3;
}
// (x + y * w compiles to v_mad_u64_u32; forcing v_mul_lo_u32 + v_add_u32 -- 7.4 issue cycles against 9.2 in the micro-benchmark --
// made the frame 2 % slower: 132.0 against 129.4 ms)
DEV uint32_t muladd(uint32_t a, uint32_t b, uint32_t c) { return a * b + c; }
DEV void pcg4d(uint32_t x, uint32_t y, uint32_t z, uint32_t w, Blk& r) {
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    w = w * 1664525u + 1013904223u;
    x = muladd(y, w, x); y = muladd(z, x, y); z = muladd(x, y, z); w = muladd(y, z, w);
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x = muladd(y, w, x); y = muladd(z, x, y); z = muladd(x, y, z); w = muladd(y, z, w);
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    r.w0 = x; r.w1 = y; r.w2 = z; r.w3 = w;
}
// block `blk` of the stream (seed, pixel, sample).  `seed` arrives pre-mixed (rtmi::mix_seed on the host, a bijection of
// the caller's 64-bit seed): its two halves are two key words that look random whatever the structure of the caller's
// seeds, and neither of them is combined with the sample index -- round 2 fed `sample ^ seed_hi`, so that two seeds
// differing in the high word by less than spp drew the same paths per pixel in another order.
DEV void rng4x32(uint32_t blk, uint32_t sample, uint32_t pixel, uint64_t seed, Blk& r) {
#if RTMI_RNG == 0
    pcg4d(blk ^ (uint32_t)(seed >> 32), sample, pixel, (uint32_t)seed, r);
#else
    philox4x32<RTMI_RNG>(blk, sample, pixel, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#endif
}

DEV Blk rng_block(const Rng& r, uint32_t blk, uint64_t seed) {
    Blk b;
    rng4x32(blk, r.sample, r.pixel, seed, b);
    return b;
}
// (float)(random_double() - 0.5f)   [sample_square, random.number.gen.hpp:16]:  (u - 2^31) * 2^-32, exact in double
DEV float draw_centered(uint32_t u) { return (float)(int32_t)(u ^ 0x80000000u) * 2.3283064365386963e-10f; }
// (float)random_double(-1, 1)       [random.number.gen.hpp:12-14]:  -1 + 2u*2^-32 = (u - 2^31) * 2^-31, exact in double
DEV float draw_pm1(uint32_t u) { return (float)(int32_t)(u ^ 0x80000000u) * 4.6566128730773926e-10f; }

// wave-wide vote straight from the compare (HIP's __ballot goes through an int and a second compare)
DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// The wave's draw service for the shading step.  Every draw is a pure function of (seed, pixel, sample, k), so any
// lane can evaluate any block of any other lane's stream; once per round each shading lane files one request:
//   RQ_UNIT  random_unit_vector (random.number.gen.hpp:21-29; `> 1e-160` on a float is `> 0`).  Every attempt takes
//            one whole block (it starts at a block boundary and skips the fourth word).  A per-lane rejection loop
//            costs the wave its longest run of rejections (6.6 passes for 1.9 attempts per lane at 52 % acceptance),
//            so only the first kSelfAttempts attempts are made by the owner (no table, no shuffles: the cheapest
//            pass there is while most lanes still need one); the ~23 % of lanes still without a vector are then
//            spread over all 64 lanes -- ~5 attempts each in the next pass, 8 in the one after.  Returns the vector.
//   RQ_WORD  the raw draw at the current position (the dielectric's reflectance test, material.defs.cc:71): taken in
//            the owner's first pass.  Returns the bits in .x; the caller advances k if it consumes the draw.
// `tbl` is 64 bytes of LDS private to the wave.
enum : uint32_t { RQ_NONE = 0, RQ_UNIT = 1, RQ_WORD = 2 };
constexpr int kSelfAttempts = 2; // measured on config 3: 0 -> 159.4 ms, 1 -> 153.3, 2 -> 152.5, 3 -> 154.3
constexpr uint32_t kThirdAttemptLanes = 12;

// (an LDS-qualified pointer: through a generic one these accesses become flat_* instructions with full waits)
typedef __attribute__((address_space(3))) uint8_t lds_u8;
DEV V3 coop_draws(uint32_t code, Rng& rng, uint64_t seed, lds_u8* tbl PB_ARGS) {
    V3 out = mk(0.0f, 0.0f, 0.0f);
    if (code == RQ_UNIT) rng.k = (rng.k + 3u) & ~3u; // attempts are block aligned
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    bool pending = code != RQ_NONE;
    PB(7, code == RQ_WORD);
    ISA_MARK("coop-owner");
    // (one more owner attempt when more than kThirdAttemptLanes lanes are still without a vector after the first two -- waves
    // whose lanes nearly all shade Lambertian hits, the box of config 5: 14 of 63 -- is cheaper than the larger shared pass it
    // replaces: -1.6 % there, +-0 on config 3 where 9 lanes are left on average; threshold 8: -0.9 % / +0.5 %)
    for (int self = 0; self < kSelfAttempts + 1; ++self) {
        if (self == kSelfAttempts && (uint32_t)__popcll(ballot(pending)) <= kThirdAttemptLanes) break;
        PB(4 + (self ? 1 : 0), pending);
        if (pending) {
            Blk tmp;
            rng4x32(rng.k >> 2, rng.sample, rng.pixel, seed, tmp);
            if (code == RQ_WORD) { // word (k & 3) of block k >> 2
                const uint32_t j = rng.k & 3u;
                out.x = __uint_as_float(j == 0u ? tmp.w0 : (j == 1u ? tmp.w1 : (j == 2u ? tmp.w2 : tmp.w3)));
                pending = false;
            } else {
                const V3 u = mk(draw_pm1(tmp.w0), draw_pm1(tmp.w1), draw_pm1(tmp.w2));
                const float l2 = vdot(u, u);
                rng.k += 4u;
                if (l2 > 0.0f && l2 <= 1.0f) { // random.number.gen.hpp:25-27
                    out = u;
                    pending = false;
                }
            }
        }
    }
    // from here on every pending lane is an RQ_UNIT one
    uint64_t todo = ballot(pending);
    ISA_MARK("coop-shared");
    PF_MARK(5);
    while (todo != 0ull) {
        PB(6, pending);
        const uint32_t n = (uint32_t)__popcll(todo);
        // attempts per pending lane in this pass: min(64 / n, 8) as a compare chain (an integer division by a run-time
        // value is a dozen instructions) together with the bits c * n, c < per, of the acceptance vote
        uint32_t per = 1u;
        uint64_t stride_mask = 1ull;
        if (n <= 32u) {
            per = n > 21u ? 2u : n > 16u ? 3u : n > 12u ? 4u : n > 10u ? 5u : n > 9u ? 6u : n > 8u ? 7u : 8u;
            stride_mask |= 1ull << n;                                   // c = 0, 1
            if (per > 2u) stride_mask |= stride_mask << (2u * n);        // c = 0..3   (per >= 3 => 4 n <= 84: bits past 63 fall off)
            if (per > 4u) stride_mask |= stride_mask << (4u * n);        // c = 0..7   (per >= 5 => n <= 12)
            if (per * n < 64u) stride_mask &= (1ull << (per * n)) - 1ull; // keep c < per
        }
        const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
        if (pending) tbl[my_rank] = (uint8_t)lane; // rank -> lane of the pending stream
        // this lane evaluates attempt `a` of the pending lane of rank `r`
        const float inv_n = __builtin_amdgcn_rcpf((float)n);
        const uint32_t a = (uint32_t)(((float)lane + 0.5f) * inv_n);
        const uint32_t r = lane - a * n;
        const bool helper = a < per;
        const uint32_t src = tbl[r];
        const uint32_t pix = (uint32_t)__shfl((int)rng.pixel, (int)src);
        const uint32_t smp = (uint32_t)__shfl((int)rng.sample, (int)src);
        const uint32_t kb = (uint32_t)__shfl((int)rng.k, (int)src);
        bool ok = false;
        V3 u = mk(0.0f, 0.0f, 0.0f); // the accepted point, not yet normalised
        if (helper) {
            Blk tmp;
            rng4x32((kb >> 2) + a, smp, pix, seed, tmp);
            u = mk(draw_pm1(tmp.w0), draw_pm1(tmp.w1), draw_pm1(tmp.w2));
            const float l2 = vdot(u, u);
            ok = l2 > 0.0f && l2 <= 1.0f;
        }
        const uint64_t okm = ballot(ok);
        // the pending lane takes its first accepted attempt, in attempt order: its attempts sit at bits
        // my_rank + c * n (c < per) of the vote
        const uint64_t hits = (okm >> my_rank) & stride_mask;
        const bool found = pending && hits != 0ull;
        const uint32_t pos = (uint32_t)__builtin_ctzll(hits | (1ull << 63));
        const uint32_t first = (uint32_t)(((float)pos + 0.5f) * inv_n);
        const uint32_t from = found ? my_rank + pos : lane;
        const float ux = __shfl(u.x, (int)from), uy = __shfl(u.y, (int)from), uz = __shfl(u.z, (int)from);
        if (found) {
            out = mk(ux, uy, uz);
            rng.k += 4u * (first + 1u);
            pending = false;
        } else if (pending) {
            rng.k += 4u * per;
        }
        todo = ballot(pending);
    }
    // p / sqrt(dot(p, p)) once, on the owner's lane: the IEEE square root and divisions are not paid per attempt
    // (components are multiples of 2^-31 in (-1, 1), +0 included, and 2^-31 <= sqrt <= 1: always in range)
    // (and 2^-62 <= dot <= 1 for the square root)
    ISA_MARK("coop-normalize");
    PF_MARK(6);
    if (code == RQ_UNIT) out = vdivs_shared(out, sqrt_core(vdot(out, out)), true);
    return out;
}

// ---------------------------------------------------------------------------------------------------------
// lane state machine
// ---------------------------------------------------------------------------------------------------------
enum : uint32_t { PH_FETCH = 0, PH_GEN = 1, PH_TRAV = 2, PH_SHADE = 3, PH_DONE = 4, PH_BEGIN = 5 };
constexpr uint32_t kNoWalk = 0xffffffffu; // RtmiLaunch::root_ref: every leaf is tested at segment set-up
constexpr uint32_t kAttLds = 4; // closed attenuation runs kept in LDS per lane; more material changes spill to HBM

struct Trav { // per-segment traversal state
    V3 o, d;
    float a;          // dot(d, d), object.defs.cc:44
    float tbest;      // closest accepted root so far (Interval::Max, object.defs.cc:69)
    uint32_t best;    // slot of the closest sphere, ~0u = none
    uint32_t cur;     // BVH: current node/leaf reference; brute force: unused
    uint32_t sp;      // BVH: LDS byte address of the next free entry of this lane's stack ([depth][lane] array)
    V3 inv, oinv, pinv; // BVH slab test: 1/d, -o/d, pad*|1/d|
};

DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// HittableObject_Sphere::intersects (object.defs.cc:41-60) in two halves.  The candidate root of a sphere does not
// depend on the shrinking Max (root1 if it is beyond tmin, else root2); acceptance against Max is done by the caller.
// Spheres are stored as {C, R*R}.
// The discriminant (cheap, every sphere, every lane) and the root (IEEE sqrt and divisions, only where delta >= 0:
// 0.4 % of the tests of the linear scan, which works on four spheres at a time).
DEV void sphere_delta(const uint4 raw, const Trav& t, float& h, float& delta) {
    const V3 oc = mk(__uint_as_float(raw.x) - t.o.x, __uint_as_float(raw.y) - t.o.y, __uint_as_float(raw.z) - t.o.z);
    h = vdot(t.d, oc);
    const float c = vdot(oc, oc) - __uint_as_float(raw.w);
    delta = h * h - t.a * c;
}
// The same, and whether the sphere needs its roots at all: `cand` = the discriminant is not negative AND the sphere is not
// wholly behind the origin.  A sphere with h < 0 (centre behind the origin along the ray) and c > 0 (origin outside it) is met by
// the ray's LINE only at negative parameters, and the reference's fp32 arithmetic agrees: fl(a c) >= 0 (a = dot(d, d) >= 0, c > 0),
// so delta = fl(fl(h h) - fl(a c)) <= fl(h h), sqrt is monotone and sqrt(fl(h h)) rounds to |h| (the relative error of fl(h h) is at
// most 2^-24, halved by the root: less than half an ulp of |h|), hence sqrtd <= |h| and both numerators h - sqrtd, h + sqrtd are <= 0:
// neither root passes `root > 0.0001` (object.defs.cc:52-57) -- skipping the square root and the divisions changes no bit.  (A NaN
// in h or c compares false and takes the full path.)  Half of the spheres whose line a ray crosses are behind it; a ray that
// leaves a sphere has that sphere's own centre behind it, and its origin outside it half of the time.
DEV void sphere_delta_cand(const uint4 raw, const Trav& t, float& h, float& delta, bool& cand) {
    const V3 oc = mk(__uint_as_float(raw.x) - t.o.x, __uint_as_float(raw.y) - t.o.y, __uint_as_float(raw.z) - t.o.z);
    h = vdot(t.d, oc);
    const float c = vdot(oc, oc) - __uint_as_float(raw.w);
    delta = h * h - t.a * c;
    cand = (delta >= 0.0f) & !((h < 0.0f) & (c > 0.0f));
}
DEV void sphere_root(float h, float delta, const Trav& t, const Recip& ra, uint32_t slot, float& tbest, uint32_t& best) {
    // the in-range cores of the square-root and division expansions on a reciprocal shared by the segment's roots (as in
    // the walk, below): inside the box of config 5 every ray's line meets four or five of the seven spheres, and the full
    // expansions (16 + 2 x 11 instructions per root) were a third of the scan
    const float sqrtd = sqrt_shared(delta);
    float root = div_in_range(h - sqrtd, ra);
    if (!(root > 0.0001f)) root = div_in_range(h + sqrtd, ra);
    if (!(ra.in_range && __builtin_fabsf(root) < 0x1p40f)) { // out of range (or NaN): the full expansion
        root = (h - sqrtd) / t.a;
        if (!(root > 0.0001f)) root = (h + sqrtd) / t.a;
    }
    if (root > 0.0001f && root < tbest) { // strict <: the first inserted object wins a tie (object.defs.cc:73)
        tbest = root;
        best = slot;
    }
}

// Root + acceptance for the BVH walk, where leaves are not visited in insertion order: a strictly closer root wins; an
// exactly equal one wins only if its object was inserted earlier (what the reference's in-order scan with `<` yields).
// `ra` = recip_for(t.a), computed once per leaf step.  A root at or below the 1e-4 cut is only compared, never kept: an
// imprecise tiny quotient (numerator below the range the expansion would rescale) takes the same branches as the exact one.
DEV void sphere_root_bvh(float h, float delta, const Trav& t, const Recip& ra, uint32_t slot, const uint4* aux, float& tbest,
                         uint32_t& best) {
    const float sqrtd = sqrt_shared(delta);
    float root = div_in_range(h - sqrtd, ra);
    if (!(root > 0.0001f)) root = div_in_range(h + sqrtd, ra);
    if (!(ra.in_range && __builtin_fabsf(root) < 0x1p40f)) { // out of range (or NaN): the full expansion
        root = (h - sqrtd) / t.a;
        if (!(root > 0.0001f)) root = (h + sqrtd) / t.a;
    }
    if (root > 0.0001f) {
        if (root < tbest) {
            tbest = root;
            best = slot;
        } else if (root == tbest && best != ~0u) {
            if (aux[slot].x < aux[best].x) best = slot;
        }
    }
}

