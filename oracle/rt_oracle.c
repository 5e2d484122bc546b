/*
 * rt_oracle.c -- CPU oracle: plain-C restatement of the reference's per-pixel path tracer.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED beyond the four reference pixels of
 * SURVEY.md 8(c) (reproduced bit for bit): the reference holds no golden vectors for this path and
 * cannot be built in this image; every function below cites the reference file:line whose
 * arithmetic it restates.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math (baseline x86-64, SSE2 scalar fp32/fp64, no FMA),
 * which is what the reference's CMake Release build produces for these TUs.  Every float operation is
 * written out in the reference's evaluation order; nothing here may be re-associated.
 *
 * glm operations (un-vendored dependency, pinned g-truc/glm@bf71a834 by reference CMakeLists.txt:55),
 * restated from glm's published definitions (detail/func_geometric.inl, func_trigonometric.inl):
 *   dot(a,b)        t = a*b; (t.x + t.y) + t.z
 *   normalize(v)    v * (1.0f / sqrt(dot(v,v)))           (inversesqrt, then multiply)
 *   reflect(I,N)    I - N * dot(N,I) * 2.0f
 *   refract(I,N,e)  d = dot(N,I); k = 1 - e*e*(1 - d*d); k >= 0 ? e*I - (e*d + sqrt(k))*N : 0
 *   cross(x,y)      (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
 *   radians(x)      x * 0.01745329251994329576923690768489f
 *   vec / scalar    per-component true division;  vec3::length() == 3 (component count)
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------- */
/* vec3 helpers (glm semantics, see header comment)                                            */
/* ------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vld(const float* p) { return V(p[0], p[1], p[2]); }
static inline void vst(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) {
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return (tx + ty) + tz;
}
static inline v3 vnormalize(v3 v) { return vscale(v, 1.0f / sqrtf(vdot(v, v))); }
static inline v3 vcross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 vreflect(v3 I, v3 N) { return vsub(I, vscale(vscale(N, vdot(N, I)), 2.0f)); }
static inline v3 vrefract(v3 I, v3 N, float eta) {
    const float d = vdot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k >= 0.0f) {
        return vsub(vscale(I, eta), vscale(N, eta * d + sqrtf(k)));
    }
    return V(0.0f, 0.0f, 0.0f);
}
static inline float glm_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

/* ------------------------------------------------------------------------------------------- */
/* RNG: src/random.number.gen.hpp:7-48                                                         */
/* ------------------------------------------------------------------------------------------- */
struct orc_rng {
    int kind;
    /* std::mt19937 (random.number.gen.hpp:46) */
    uint32_t mt[624];
    int mti;
    /* counter stream (build-side replacement of the bit source only) */
    uint32_t key[2];
    uint32_t pixel, sample, k;
    uint32_t cache[4];
    uint64_t n_doubles;
};

static void mt_seed(orc_rng* r, uint32_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < 624; ++i) {
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    }
    r->mti = 624;
}

static uint32_t mt_next(orc_rng* r) {
    if (r->mti >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (r->mt[i] & 0x80000000u) | (r->mt[(i + 1) % 624] & 0x7fffffffu);
            r->mt[i] = r->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        r->mti = 0;
    }
    uint32_t y = r->mt[r->mti++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* Philox4x32-R (Salmon et al., SC'11), the per-lane counter RNG of the GPU path. */
void orc_philox4x32(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < rounds; ++round) {
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
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32(10, ctr, key, out); }

/* pcg4d (Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 9(3), 2020, listing of the 4-d PCG hash) with one more
 * xorshift on the way out: without it the low three bits of a word are biased (tests/test_oracle_cpu.py) */
void orc_pcg4d(const uint32_t in[4], uint32_t out[4]) {
    uint32_t x = in[0], y = in[1], z = in[2], w = in[3];
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

/* Block function of the counter stream -- the same build-time choice the product makes with RTMI_RNG
 * (raytracing.cpp_amd/csrc/rtmi_kernel_common.h): 10 or 7 = Philox4x32 rounds, 0 = pcg4d. */
static int g_counter_rng = ORC_COUNTER_RNG_DEFAULT;
void orc_set_counter_rng(int kind) { g_counter_rng = kind; }
int orc_get_counter_rng(void) { return g_counter_rng; }
void orc_counter_block(uint32_t blk, uint32_t sample, uint32_t pixel, const uint32_t key[2], uint32_t out[4]) {
    if (g_counter_rng == 0) {
        const uint32_t in[4] = {blk ^ key[1], sample, pixel, key[0]};
        orc_pcg4d(in, out);
    } else {
        const uint32_t ctr[4] = {blk, sample, pixel, 0u};
        orc_philox4x32(g_counter_rng, ctr, key, out);
    }
}

/* libstdc++ std::generate_canonical<double,53>(urng32) as used by
 * std::uniform_real_distribution<double>(0,1) (random.number.gen.hpp:11,47):
 * sum = draw1 + draw2 * 2^32 (one rounding to double), / 2^64, clamp below 1. */
static inline double canonical_from_u32_pair(uint32_t lo, uint32_t hi) {
    double sum = (double)lo;
    sum += (double)hi * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) {
        ret = nextafter(1.0, 0.0);
    }
    return ret;
}

static inline void counter_begin(orc_rng* r, uint32_t pixel, uint32_t sample) {
    r->pixel = pixel;
    r->sample = sample;
    r->k = 0;
}

/* random_double(), random.number.gen.hpp:11 */
static inline double rd(orc_rng* r) {
    r->n_doubles++;
    if (r->kind == ORC_RNG_MT19937) {
        const uint32_t lo = mt_next(r);
        const uint32_t hi = mt_next(r);
        return canonical_from_u32_pair(lo, hi);
    }
    /* counter stream (BUILD-SIDE replacement of `_randdist(_randgen)`): draw #k of (seed, pixel, sample) is
     * word (k & 3) of block k >> 2 of that stream (orc_counter_block: pcg4d, or Philox4x32 in the A/B), mapped to [0,1) as
     * u32 * 2^-32 (exact in double).  Everything downstream of random_double() is unchanged. */
    const uint32_t k = r->k++;
    if ((k & 3u) == 0u) {
        orc_counter_block(k >> 2, r->sample, r->pixel, r->key, r->cache);
    }
    return (double)r->cache[k & 3u] * (1.0 / 4294967296.0);
}

/* random_double(r_min, r_max), random.number.gen.hpp:12-14 (double arithmetic) */
static inline double rd_range(orc_rng* r, double r_min, double r_max) { return r_min + (r_max - r_min) * rd(r); }

/* random_vector(rmin, rmax), random.number.gen.hpp:18-20: braces => x, y, z drawn left to right, narrowed to float */
static inline v3 random_vector_range(orc_rng* r, double rmin, double rmax) {
    const float x = (float)rd_range(r, rmin, rmax);
    const float y = (float)rd_range(r, rmin, rmax);
    const float z = (float)rd_range(r, rmin, rmax);
    return V(x, y, z);
}

/* random_unit_vector(), random.number.gen.hpp:21-29.  `length_squared > 1e-160` compares a float
 * against a double below the smallest float denormal, i.e. `> 0`. */
static inline v3 random_unit_vector(orc_rng* r) {
    for (;;) {
        /* counter stream only (build-side): every attempt takes one whole block -- it starts at a block
         * boundary and skips the fourth word -- so that any GPU lane can evaluate any attempt of any stream */
        if (r->kind == ORC_RNG_COUNTER) r->k = (r->k + 3u) & ~3u;
        const v3 p = random_vector_range(r, -1.0, 1.0);
        if (r->kind == ORC_RNG_COUNTER) r->k = (r->k + 3u) & ~3u;
        const float length_squared = vdot(p, p);
        if ((double)length_squared > 1e-160 && length_squared <= 1.0f) {
            return vdivs(p, sqrtf(length_squared));
        }
    }
}

/* random_vector_on_unit_disk(), random.number.gen.hpp:35-42 */
static inline v3 random_vector_on_unit_disk(orc_rng* r) {
    for (;;) {
        const float x = (float)rd_range(r, (double)-1.0f, (double)1.0f);
        const float y = (float)rd_range(r, (double)-1.0f, (double)1.0f);
        const v3 p = V(x, y, 0.0f);
        if (vdot(p, p) < 1.0f) {
            return p;
        }
    }
}

orc_rng* orc_rng_new_mt(uint32_t seed) {
    orc_rng* r = (orc_rng*)calloc(1, sizeof(orc_rng));
    r->kind = ORC_RNG_MT19937;
    mt_seed(r, seed);
    return r;
}
void orc_rng_free(orc_rng* r) { free(r); }
double orc_rng_double(orc_rng* r) { return rd(r); }
uint32_t orc_mt_next_u32(orc_rng* r) { return mt_next(r); }

/* The two key words of a counter stream: a bijective mix of the caller's 64-bit seed (the splitmix64 finaliser), so that
 * seeds differing in one bit or only in the high word give unrelated keys and no key word meets the sample index
 * (build-side: the reference seeds one independent generator per std::random_device value, random.number.gen.hpp:45-46). */
uint64_t orc_mix_seed(uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void rng_init_counter(orc_rng* r, uint64_t seed) {
    memset(r, 0, sizeof(*r));
    r->kind = ORC_RNG_COUNTER;
    const uint64_t k = orc_mix_seed(seed);
    r->key[0] = (uint32_t)k;
    r->key[1] = (uint32_t)(k >> 32);
}

double orc_counter_double(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k) {
    orc_rng r;
    rng_init_counter(&r, seed);
    counter_begin(&r, pixel, sample);
    double v = 0.0;
    for (uint32_t i = 0; i <= k; ++i) {
        v = rd(&r);
    }
    return v;
}

/* ------------------------------------------------------------------------------------------- */
/* camera derivation: src/ray.tracer.core.cc:158-216                                           */
/* ------------------------------------------------------------------------------------------- */
void orc_camera_setup(const orc_camera_params* p, orc_camera* out) {
    /* core.cc:174-175 */
    const uint32_t image_height = (uint32_t)((float)p->image_width / p->aspect_ratio);
    /* core.cc:177-180 */
    const float theta = glm_radians(p->vertical_fov);
    const float h = tanf(theta * 0.5f);
    const float viewport_height = 2.0f * h * p->focus_distance;
    const float viewport_width = viewport_height * ((float)p->image_width / (float)image_height);
    /* make_camera_frame, core.cc:158-169 */
    const v3 lookfrom = vld(p->lookfrom), lookat = vld(p->lookat), up = vld(p->world_up);
    const v3 W = vnormalize(vsub(lookfrom, lookat));
    const v3 U = vnormalize(vcross(up, W));
    const v3 Vv = vcross(W, U);
    /* core.cc:185-189 */
    const v3 viewport_u = vscale(U, viewport_width);
    const v3 viewport_v = vscale(vneg(Vv), viewport_height);
    const v3 pixel_delta_u = vdivs(viewport_u, (float)p->image_width);
    const v3 pixel_delta_v = vdivs(viewport_v, (float)image_height);
    /* core.cc:191-193 */
    const v3 upper_left = vsub(vsub(vsub(lookfrom, vscale(W, p->focus_distance)), vscale(viewport_u, 0.5f)),
                               vscale(viewport_v, 0.5f));
    const v3 pixel00 = vadd(upper_left, vscale(vadd(pixel_delta_u, pixel_delta_v), 0.5f));
    /* core.cc:195 */
    const float defocus_radius = p->focus_distance * tanf(glm_radians(p->defocus_angle * 0.5f));

    /* core.cc:198-215 */
    out->img_width = p->image_width;
    out->img_height = image_height;
    out->defocus_angle = p->defocus_angle;
    out->viewport_height = viewport_height;
    out->viewport_width = viewport_width;
    out->samples_per_pixel = p->samples_per_pixel;
    out->maxdepth = p->max_depth;
    out->pixels_sample_scale = 1.0f / (float)p->samples_per_pixel;
    vst(out->pixel_delta_u, pixel_delta_u);
    vst(out->pixel_delta_v, pixel_delta_v);
    vst(out->pixel00, pixel00);
    vst(out->cam_center, lookfrom);
    vst(out->defocus_disk_u, vscale(U, defocus_radius));
    vst(out->defocus_disk_v, vscale(Vv, defocus_radius));
}

/* ------------------------------------------------------------------------------------------- */
/* scene generator: make_world_spheres, src/ray.tracer.core.cc:99-149                          */
/* ------------------------------------------------------------------------------------------- */
static orc_material make_lambertian(v3 albedo) { /* material.defs.hpp:57-65 */
    orc_material m;
    memset(&m, 0, sizeof(m));
    m.kind = 0;
    m.p[0] = albedo.x; m.p[1] = albedo.y; m.p[2] = albedo.z;
    return m;
}
static orc_material make_metallic(v3 albedo, float fuzziness) { /* material.defs.hpp:67-76, fuzz = min(1, f) */
    orc_material m;
    memset(&m, 0, sizeof(m));
    m.kind = 1;
    m.p[0] = albedo.x; m.p[1] = albedo.y; m.p[2] = albedo.z;
    m.p[3] = (1.0f < fuzziness) ? 1.0f : fuzziness; /* std::min(1.0f, fuzziness) */
    return m;
}
static orc_material make_dielectric(float ri) { /* material.defs.hpp:78-86 */
    orc_material m;
    memset(&m, 0, sizeof(m));
    m.kind = 2;
    m.p[0] = ri;
    return m;
}

uint32_t orc_make_world_spheres(const orc_world_def* wd, const orc_object* fixed_objs, const orc_material* fixed_mats,
                                uint32_t n_fixed, uint32_t mt_seed, int metal_args_right_to_left, orc_object* objs_out,
                                orc_material* mats_out, uint32_t capacity) {
    uint32_t n = 0;
    /* core.cc:104-122: the listed objects, one material each, handle == insertion index */
    for (uint32_t i = 0; i < n_fixed && n < capacity; ++i, ++n) {
        objs_out[n] = fixed_objs[i];
        objs_out[n].kind = 0;
        objs_out[n].material = n;
        mats_out[n] = fixed_mats[i];
        if (mats_out[n].kind == 1) {
            mats_out[n] = make_metallic(V(fixed_mats[i].p[0], fixed_mats[i].p[1], fixed_mats[i].p[2]),
                                        fixed_mats[i].p[3]);
        }
    }
    /* core.cc:124: RandomNumberGenerator rand_gen{}  (seeded here instead of from random_device) */
    orc_rng* rng = orc_rng_new_mt(mt_seed);
    for (int32_t a = wd->a_min; a < wd->a_max; ++a) {     /* core.cc:125 */
        for (int32_t b = wd->b_min; b < wd->b_max; ++b) { /* core.cc:126 */
            const float choose_mat = (float)rd(rng);      /* core.cc:127 */
            /* core.cc:128: {a + 0.9f * rd(), 0.2f, b + 0.9 * rd()} -- double arithmetic, narrowed to float;
             * note 0.9f (float literal promoted) for x, 0.9 (double literal) for z */
            const float cx = (float)((double)a + (double)0.9f * rd(rng));
            const float cy = 0.2f;
            const float cz = (float)((double)b + 0.9 * rd(rng));
            /* core.cc:130: (center - offset).length() is glm's component count == 3 -> `3 > treshold` */
            if (3.0f > wd->center_dist_treshold) {
                if (n >= capacity) {
                    orc_rng_free(rng);
                    return n;
                }
                orc_material m;
                if (choose_mat < wd->diffuse_material_treshold) { /* core.cc:133-135 */
                    /* product of two random_vector(0,1): commutative, so operand evaluation order is immaterial */
                    const v3 c1 = random_vector_range(rng, (double)0.0f, (double)1.0f);
                    const v3 c2 = random_vector_range(rng, (double)0.0f, (double)1.0f);
                    m = make_lambertian(vmul(c1, c2));
                } else if (choose_mat < wd->metal_material_treshold) { /* core.cc:136-138 */
                    /* make_metallic(random_vector(0.5,1), random_double(0,0.5)): argument evaluation order is
                     * unspecified in C++; g++ on x86-64 evaluates right to left (fuzz first). */
                    v3 albedo;
                    float fuzz;
                    if (metal_args_right_to_left) {
                        fuzz = (float)rd_range(rng, (double)0.0f, (double)0.5f);
                        albedo = random_vector_range(rng, (double)0.5f, (double)1.0f);
                    } else {
                        albedo = random_vector_range(rng, (double)0.5f, (double)1.0f);
                        fuzz = (float)rd_range(rng, (double)0.0f, (double)0.5f);
                    }
                    m = make_metallic(albedo, fuzz);
                } else { /* core.cc:139-141 */
                    m = make_dielectric((float)rd_range(rng, (double)1.2f, (double)1.6f));
                }
                mats_out[n] = m;
                objs_out[n].kind = 0;
                objs_out[n].center[0] = cx; objs_out[n].center[1] = cy; objs_out[n].center[2] = cz;
                objs_out[n].radius = 0.2f; /* core.cc:143 */
                objs_out[n].material = n;
                ++n;
            }
        }
    }
    orc_rng_free(rng);
    return n;
}

/* ------------------------------------------------------------------------------------------- */
/* RGBAColor(vec3): src/color.hpp:9-36, clamp: src/ray.tracer.math.hpp:10-14                   */
/* ------------------------------------------------------------------------------------------- */
static inline float linear_to_gamma(float v) { return v > 0.0f ? sqrtf(v) : 0.0f; }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
uint32_t orc_pack_rgba(const float rgb[3]) {
    const uint32_t r = (uint8_t)(clampf(linear_to_gamma(rgb[0]), 0.0f, 0.999f) * 256.0f);
    const uint32_t g = (uint8_t)(clampf(linear_to_gamma(rgb[1]), 0.0f, 0.999f) * 256.0f);
    const uint32_t b = (uint8_t)(clampf(linear_to_gamma(rgb[2]), 0.0f, 0.999f) * 256.0f);
    const uint32_t a = (uint8_t)(clampf(1.0f, 0.0f, 0.999f) * 256.0f);
    return r | (g << 8) | (b << 16) | (a << 24); /* little-endian union, color.hpp:19-27 */
}

/* ------------------------------------------------------------------------------------------- */
/* intersection: src/ray.tracer.object.defs.cc:11-18, 41-81                                    */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    v3 P, N;
    uint32_t material;
    int front_face;
    double T;
} hit_rec;

typedef struct { v3 o, d; } ray_t;

/* Interval::surrounds, src/interval.hpp:14 */
static inline int surrounds(double mn, double mx, double x) { return mn < x && x < mx; }

/* HittableObject_Sphere::intersects, object.defs.cc:41-66 + IntersectionRecord ctor :11-18 */
static inline int sphere_intersects(const orc_object* s, const ray_t* r, double tmin, double tmax, hit_rec* rec) {
    const v3 C = vld(s->center);
    const v3 oc = vsub(C, r->o);
    const float a = vdot(r->d, r->d);
    const float h = vdot(r->d, oc);
    const float c = vdot(oc, oc) - s->radius * s->radius;
    const float delta = h * h - a * c;
    if (delta < 0.0f) {
        return 0;
    }
    const float sqrtd = sqrtf(delta);
    float root = (h - sqrtd) / a;
    if (!surrounds(tmin, tmax, (double)root)) {
        root = (h + sqrtd) / a;
        if (!surrounds(tmin, tmax, (double)root)) {
            return 0;
        }
    }
    const v3 p = vadd(r->o, vscale(r->d, root)); /* Ray::point_at_param, ray.hpp:9 */
    const v3 outward = vdivs(vsub(p, C), s->radius);
    rec->P = p;
    rec->T = (double)root;
    rec->material = s->material;
    rec->front_face = vdot(r->d, outward) < 0.0f;
    rec->N = rec->front_face ? outward : vneg(outward);
    return 1;
}

typedef struct {
    const orc_object* objs;
    uint32_t n_objs;
    const orc_material* mats;
    uint32_t n_mats;
    /* optional BVH (instrumented walk) */
    const orc_bvh_node* nodes;
    uint32_t n_nodes;
    const uint32_t* slots;
    uint32_t n_slots;
    const float* pad_classes;
    uint32_t n_classes;
    float pad_eps, pad_floor;
    int pad_refine; /* pad_refine_pays(pad_classes): what pad mode 0 does */
} scene_t;

/* HittableObject_Collection::intersects, object.defs.cc:68-81: linear scan in insertion order,
 * shrinking Max; strict `<` so the first-inserted object wins an exact tie. */
static inline int world_intersects(const scene_t* sc, const ray_t* r, double tmin, double tmax, hit_rec* rec,
                                   uint32_t* index, orc_counters* ctr) {
    double closest = tmax;
    int any = 0;
    hit_rec tmp;
    for (uint32_t i = 0; i < sc->n_objs; ++i) {
        if (sphere_intersects(&sc->objs[i], r, tmin, closest, &tmp)) {
            *rec = tmp;
            closest = tmp.T;
            any = 1;
            if (index) *index = i;
        }
    }
    if (ctr) ctr->sphere_tests += sc->n_objs;
    return any;
}

/* ------------------------------------------------------------------------------------------- */
/* BVH walk (BUILD-SIDE EXTENSION, not in the reference): returns the same closest hit as the     */
/* linear scan -- the candidate root of a sphere does not depend on Max (root1 if > tmin, else   */
/* root2), acceptance is `candidate < Max`, ties go to the lowest object index -- while visiting */
/* only nodes whose padded box the ray enters.  Mirrors the GPU traversal step for step so the   */
/* node/sphere test counters price the GPU kernel's algorithmic work.                */
/* ------------------------------------------------------------------------------------------- */
static inline float sphere_candidate(const orc_object* s, const ray_t* r, float tmin) {
    const v3 C = vld(s->center);
    const v3 oc = vsub(C, r->o);
    const float a = vdot(r->d, r->d);
    const float h = vdot(r->d, oc);
    const float c = vdot(oc, oc) - s->radius * s->radius;
    const float delta = h * h - a * c;
    if (delta < 0.0f) {
        return -1.0f;
    }
    const float sqrtd = sqrtf(delta);
    float root = (h - sqrtd) / a;
    if (!(root > tmin)) {
        root = (h + sqrtd) / a;
        if (!(root > tmin)) {
            return -1.0f;
        }
    }
    return root;
}

/* Per-segment box pad.  Every sphere whose root the scan could accept must be reached, and the point of
 * an accepted root lies within R + e(L) of the centre, e(L) = min(x / 2R, sqrt(x)), x = 64u (L^2 + R^2), L = |C - O|.
 *   class pad (rounds 1-3): one number per radius class from the FARTHEST centre of the class: E0.
 *   refined (round 4)     : an accepted root's point lies inside the class's centre box grown by G = rmax + E0, and before
 *                           the far limit the segment starts with (the closest root among the peeled leaves), so
 *                           L <= t_far |d| + G with t_far = min(limit, exit parameter from that box): E1 = e(L_max) <= E0.
 *                           The floor follows the origin as well: 16u max(largest box coordinate, |O|_inf).
 * The product refines on scenes much wider than their spheres -- E0 seen from the middle of a class's centre box above 5 % of
 * the class's smallest radius (config 4: 0.44 on radius 0.2; S-RTOW: 0.002, where refining costs more than it culls) --
 * and so does pad mode 0 here, from the same eight floats per class.  Modes: 0 auto (the product's default), 1 class pad,
 * 2 refined, 3 no class pad at all (NOT exact; the lower bound of the walk's work, for measurements only).
 * The kernel computes the same expression with v_rcp_f32 / v_sqrt_f32 where this file divides and calls sqrtf. */
static int g_pad_mode = 0;
void orc_set_pad_mode(int mode) { g_pad_mode = mode; }
int orc_get_pad_mode(void) { return g_pad_mode; }

/* rtmi::pad_refine_pays (csrc/rtmi_host.cpp), operation for operation */
static int pad_refine_pays(const float* classes, uint32_t n_classes, float pad_eps) {
    for (uint32_t c = 0; c < n_classes; ++c) {
        const float* k = classes + 8 * c;
        float far2 = 0.0f;
        for (int i = 0; i < 3; ++i) {
            const float m = 0.5f * k[i] + 0.5f * k[3 + i];
            const float d0 = m - k[i], d1 = k[3 + i] - m;
            far2 = far2 + fmaxf(d0 * d0, d1 * d1);
        }
        const float x = pad_eps * (far2 + k[7]);
        const float e0 = fminf(x * k[6], sqrtf(x) * 1.000001f);
        if (e0 * k[6] * 2.0f > 0.05f) return 1;
    }
    return 0;
}

static inline float ray_pad(const scene_t* sc, const ray_t* r, const float inv[3], const float oinv[3], float limit) {
    const float o[3] = {r->o.x, r->o.y, r->o.z};
    float e = sc->pad_floor;
    const int refine = g_pad_mode == 2 || (g_pad_mode == 0 && sc->pad_refine);
    if (refine) {
        const float oinf = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fabsf(o[2]));
        e = fmaxf(e, 9.5367432e-7f * oinf); /* 16u |O|_inf */
    }
    if (g_pad_mode == 3) return e;
    const float floor_ = e;
    float dlen = 0.0f;
    if (refine) dlen = sqrtf(vdot(r->d, r->d)) * 1.00001f;
    for (uint32_t c = 0; c < sc->n_classes; ++c) {
        const float* k = sc->pad_classes + 8 * c;
        float far2 = 0.0f;
        for (int i = 0; i < 3; ++i) {
            const float d0 = o[i] - k[i];
            const float d1 = k[3 + i] - o[i];
            const float m = fmaxf(d0 * d0, d1 * d1);
            far2 = far2 + m;
        }
        const float x = sc->pad_eps * (far2 + k[7]); /* k[7] = rmax^2 of the class; sqrt(R^2 + x) - R <= min(x / (2R), sqrt(x)) */
        float ec = fminf(x * k[6], sqrtf(x) * 1.000001f);
        if (refine) {
            /* reach of the class around its centre box: rmax + E0, plus twice the floor for the rounding of the exit
             * parameters below (each is off by at most ~3u (|plane| + |O|) |1/d|) */
            const float g = (sqrtf(k[7]) * 1.000001f + ec) + 2.0f * floor_;
            float t_exit = INFINITY;
            for (int i = 0; i < 3; ++i) {
                const float a = fmaf(k[i] - g, inv[i], oinv[i]);
                const float b = fmaf(k[3 + i] + g, inv[i], oinv[i]);
                t_exit = fminf(t_exit, fmaxf(a, b)); /* fmaxf / fminf drop a NaN operand (0 * inf): that axis does not bound */
            }
            const float t_far = fmaxf(fminf(limit, t_exit), 0.0f);
            const float lmax = fmaf(t_far, dlen, g);
            const float x1 = sc->pad_eps * (lmax * lmax + k[7]);
            const float e1 = fminf(x1 * k[6], sqrtf(x1) * 1.000001f);
            ec = fminf(ec, e1);
        }
        e = fmaxf(e, ec);
    }
    return e;
}

/* instrumentation of the walk (counters only, never the image): the object the current segment starts on, ~0u for a camera ray */
static __thread uint32_t g_origin_obj = 0xffffffffu;
/* Camera-ray entries (build-side, round 6; the product's rtmi_tuning::cam_entry): per 8x8 tile of the image the reference of the
 * node or leaf the walks of that tile's camera rays start at (the product's host code proves that no sample of the tile can hit
 * a sphere outside it: csrc/rtmi_host.cpp, build_tile_entries), 0xffffffff = no walk at all.  The oracle only FOLLOWS the table,
 * so that its test counters price what the kernel does and a CPU test can hold the table against the linear scan. */
#define ORC_FROM_ROOT 0xfffffffeu
static const uint32_t* g_tile_entries = NULL;
static uint32_t g_tile_entries_gtx = 0;
static __thread uint32_t g_entry_ref = ORC_FROM_ROOT;
void orc_set_tile_entries(const uint32_t* entries, uint32_t gtx) { g_tile_entries = entries; g_tile_entries_gtx = gtx; }
/* Walk starts of scattered rays (build-side, round 6; the product's rtmi_tuning::walk_start on HBM-resident trees): per sphere slot
 * one record of 16 words {reference the walk of a ray scattered off that sphere starts at, n, indices of the n way records pre-
 * loaded on its stack} -- way records are nodes behind the tree's own in the node array (csrc/rtmi_host.cpp, build_walk_starts).
 * The oracle only FOLLOWS the table (any start is exact as long as every subtree hanging off the path above it is tested). */
static const uint32_t* g_walk_starts = NULL;
static uint32_t* g_obj_slot = NULL;
void orc_set_walk_starts(const uint32_t* records, const uint32_t* slots, uint32_t n_slots, uint32_t n_objs) {
    free(g_obj_slot);
    g_obj_slot = NULL;
    g_walk_starts = records;
    if (!records) return;
    g_obj_slot = (uint32_t*)malloc(((size_t)n_objs + 1u) * sizeof(uint32_t));
    memset(g_obj_slot, 0xff, ((size_t)n_objs + 1u) * sizeof(uint32_t));
    for (uint32_t q = 0; q < n_slots; ++q) if (slots[q] < n_objs) g_obj_slot[slots[q]] = q;
}
static int path_to_slot(const scene_t* sc, uint32_t ref, uint32_t slot, uint32_t* path, int depth) {
    if (ref & 0x80000000u) {
        const uint32_t first = ref & 0x00ffffffu, count = (ref >> 24) & 0x7fu;
        return (slot >= first && slot < first + count) ? depth : -1;
    }
    if (depth >= 64) return -1;
    path[depth] = ref;
    for (int k = 0; k < 2; ++k) {
        const int d = path_to_slot(sc, sc->nodes[ref].child[k], slot, path, depth + 1);
        if (d >= 0) return d;
    }
    return -1;
}

static int bvh_intersects(const scene_t* sc, const ray_t* r, float tmin, hit_rec* rec, uint32_t* index,
                          orc_counters* ctr) {
    const float inv[3] = {1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z};
    const float o[3] = {r->o.x, r->o.y, r->o.z};
    float oinv[3], pinv[3], ainv[3];
    for (int i = 0; i < 3; ++i) {
        ainv[i] = fabsf(inv[i]);
        oinv[i] = -(o[i] * inv[i]);
    }
    float best_t = INFINITY;
    uint32_t best = 0xffffffffu;
    uint32_t stack[64];
    int sp = 0;
    uint32_t cur = (sc->n_nodes == 0) ? (0x80000000u | (sc->n_slots << 24)) : 0u;
    /* as the kernel does: leaves hanging directly off the top of the tree (a spine of (leaf | subtree) nodes, at    */
    /* most four leaves; the ground sphere) are tested at segment set-up and the walk starts below them, without     */
    /* the box tests of the peeled nodes                                                                             */
    uint32_t pre[4];
    uint32_t n_pre = 0;
    int no_walk = 0;
    if (sc->n_nodes != 0) {
        uint32_t at = 0;
        while (!(at & 0x80000000u) && n_pre < 4u) {
            const uint32_t c0 = sc->nodes[at].child[0], c1 = sc->nodes[at].child[1];
            const int l0 = (c0 & 0x80000000u) != 0, l1 = (c1 & 0x80000000u) != 0;
            if (l0 && l1 && n_pre + 2u <= 4u) {
                pre[n_pre++] = c0;
                pre[n_pre++] = c1;
                no_walk = 1;
                break;
            }
            if (l0 == l1) break;
            pre[n_pre++] = l0 ? c0 : c1;
            at = l0 ? c1 : c0;
        }
        if (n_pre) cur = at;
    }
    for (uint32_t q = 0; q < n_pre; ++q) {
        const uint32_t first = pre[q] & 0x00ffffffu, count = (pre[q] >> 24) & 0x7fu;
        for (uint32_t s = 0; s < count; ++s) {
            const uint32_t oi = sc->slots[first + s];
            const float cand = sphere_candidate(&sc->objs[oi], r, tmin);
            if (ctr) ctr->sphere_tests++;
            if (cand > tmin && (cand < best_t || (cand == best_t && oi < best))) {
                best_t = cand;
                best = oi;
            }
        }
    }
    if (g_entry_ref != ORC_FROM_ROOT && !no_walk) { /* a camera ray of a tile with an entry: start there (or nowhere) */
        if (g_entry_ref == 0xffffffffu) no_walk = 1; else cur = g_entry_ref;
    }
    if (g_walk_starts && g_origin_obj != 0xffffffffu && !no_walk) { /* a scattered ray: start where its sphere's record says, way pre-loaded */
        const uint32_t slot = g_obj_slot[g_origin_obj];
        if (slot != 0xffffffffu) {
            const uint32_t* rec = g_walk_starts + (size_t)slot * 16u;
            cur = rec[0];
            for (uint32_t i = 0; i < rec[1] && i < 14u; ++i) stack[sp++] = rec[2u + i];
        }
    }
    /* the pad of this segment's boxes, with the far limit the peeled leaves left */
    const float pad = ray_pad(sc, r, inv, oinv, best_t);
    for (int i = 0; i < 3; ++i) pinv[i] = pad * ainv[i];
    /* (counters) class of the segment and, for one that starts on a sphere inside the tree, the nodes above that sphere's leaf */
    int cls = 0, path_len = -1;
    uint32_t path[64];
    if (ctr) {
        if (g_origin_obj != 0xffffffffu) {
            cls = 1;
            uint32_t slot = 0xffffffffu;
            for (uint32_t q = 0; q < sc->n_slots; ++q)
                if (sc->slots[q] == g_origin_obj) { slot = q; break; }
            if (slot != 0xffffffffu && !no_walk) {
                path_len = path_to_slot(sc, cur, slot, path, 0);
                if (path_len >= 0) cls = 2;
            }
        }
        ctr->seg_class[cls]++;
    }
    for (; !no_walk;) {
        if (cur & 0x80000000u) {
            if (ctr) ctr->leaf_trips_class[cls]++;
            const uint32_t first = cur & 0x00ffffffu, count = (cur >> 24) & 0x7fu;
            for (uint32_t s = 0; s < count; ++s) {
                const uint32_t oi = sc->slots[first + s];
                const float cand = sphere_candidate(&sc->objs[oi], r, tmin);
                if (ctr) ctr->sphere_tests++;
                if (cand > tmin && (cand < best_t || (cand == best_t && oi < best))) {
                    best_t = cand;
                    best = oi;
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const orc_bvh_node* nd = &sc->nodes[cur];
        float tn[2], tf[2];
        for (int k = 0; k < 2; ++k) {
            float nmax = tmin, fmin_ = best_t;
            for (int i = 0; i < 3; ++i) {
                const float tc = fmaf(nd->ctr[k][i], inv[i], oinv[i]);
                const float th = fmaf(nd->half[k][i], ainv[i], pinv[i]);
                const float nr = tc - th, fr = tc + th;
                nmax = fmaxf(nmax, nr); /* fmaxf/fminf ignore a NaN operand, as v_max_f32/v_min_f32 do */
                fmin_ = fminf(fmin_, fr);
            }
            tn[k] = nmax;
            tf[k] = fmin_;
        }
        if (ctr) ctr->node_tests += 2;
        const int h0 = tn[0] <= tf[0], h1 = tn[1] <= tf[1];
        if (ctr) {
            ctr->trips_class[cls]++;
            for (int q = 0; q < path_len; ++q) {
                if (path[q] != cur) continue;
                /* a node above the origin sphere's leaf: which child leads on, and is the other one's box hit? */
                const uint32_t on = q + 1 < path_len ? path[q + 1] : 0xffffffffu;
                const int on_is_0 = q + 1 < path_len ? nd->child[0] == on : -1;
                int sib_hit;
                if (on_is_0 >= 0) sib_hit = on_is_0 ? h1 : h0;
                else { /* the last node of the path: the child that is the origin's leaf */
                    uint32_t slot = 0;
                    for (uint32_t z = 0; z < sc->n_slots; ++z) if (sc->slots[z] == g_origin_obj) { slot = z; break; }
                    const uint32_t c0 = nd->child[0];
                    const int leaf0 = (c0 & 0x80000000u) && slot >= (c0 & 0x00ffffffu) && slot < (c0 & 0x00ffffffu) + ((c0 >> 24) & 0x7fu);
                    sib_hit = leaf0 ? h1 : h0;
                }
                ctr->descent_levels++;
                ctr->descent_sibling_hits += (uint64_t)sib_hit;
            }
        }
        if (h0 && h1) {
            const int swap = tn[1] < tn[0];
            stack[sp++] = nd->child[swap ? 0 : 1];
            cur = nd->child[swap ? 1 : 0];
        } else if (h0) {
            cur = nd->child[0];
        } else if (h1) {
            cur = nd->child[1];
        } else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    if (best == 0xffffffffu) {
        return 0;
    }
    /* rebuild the record of the winning sphere exactly as object.defs.cc:62-65 / :11-18 */
    const orc_object* s = &sc->objs[best];
    const v3 C = vld(s->center);
    const v3 p = vadd(r->o, vscale(r->d, best_t));
    const v3 outward = vdivs(vsub(p, C), s->radius);
    rec->P = p;
    rec->T = (double)best_t;
    rec->material = s->material;
    rec->front_face = vdot(r->d, outward) < 0.0f;
    rec->N = rec->front_face ? outward : vneg(outward);
    if (index) *index = best;
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* materials: src/ray.tracer.material.defs.cc:31-109, near_zero: src/ray.tracer.math.hpp:16-19 */
/* ------------------------------------------------------------------------------------------- */
static inline int near_zero(v3 v) {
    const float s = 1e-8f;
    return fabsf(v.x) < s && fabsf(v.y) < s && fabsf(v.z) < s;
}

/* returns 1 if scattered */
static inline int material_scatter(const orc_material* m, const ray_t* ray_in, const hit_rec* rec, orc_rng* rng,
                                   v3* attenuation, ray_t* scattered, orc_counters* ctr) {
    switch (m->kind) { /* Material::scatter, material.defs.cc:89-109 */
    case 0: {          /* Material_Lambertian::scatter, material.defs.cc:31-42 */
        if (ctr) ctr->hit_lambertian++;
        v3 dir = vadd(rec->N, random_unit_vector(rng));
        if (near_zero(dir)) {
            dir = rec->N;
        }
        *attenuation = V(m->p[0], m->p[1], m->p[2]);
        scattered->o = rec->P;
        scattered->d = dir;
        return 1;
    }
    case 1: { /* Material_Metallic::scatter, material.defs.cc:44-55 */
        if (ctr) ctr->hit_metallic++;
        v3 reflected = vreflect(ray_in->d, rec->N);
        reflected = vadd(vnormalize(reflected), vscale(random_unit_vector(rng), m->p[3]));
        if (vdot(reflected, rec->N) > 0.0f) {
            *attenuation = V(m->p[0], m->p[1], m->p[2]);
            scattered->o = rec->P;
            scattered->d = reflected;
            return 1;
        }
        return 0;
    }
    case 2: { /* Material_Dielectric::scatter, material.defs.cc:57-87 */
        if (ctr) ctr->hit_dielectric++;
        const float ri = m->p[0];
        const float eta = rec->front_face ? (1.0f / ri) : ri;
        const v3 unit_dir = vnormalize(ray_in->d);
        const float cos_theta = fminf(vdot(vneg(unit_dir), rec->N), 1.0f);
        const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        const int cannot_refract = (eta * sin_theta) > 1.0f;
        int reflect_it = cannot_refract;
        if (!reflect_it) { /* short-circuit ||: the RNG draw happens only when refraction is possible */
            const float r0 = (1.0f - eta) / (1.0f + eta);
            const float r1 = r0 * r0;
            const float schlick = r1 + (1.0f - r1) * powf(1.0f - cos_theta, 5.0f);
            reflect_it = (double)schlick > rd(rng);
        }
        const v3 dir = reflect_it ? vreflect(unit_dir, rec->N) : vrefract(unit_dir, rec->N, eta);
        *attenuation = V(1.0f, 1.0f, 1.0f);
        scattered->o = rec->P;
        scattered->d = dir;
        return 1;
    }
    default:
        return 0;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* render core: src/ray.tracer.core.cc:218-265                                                 */
/* ------------------------------------------------------------------------------------------- */
/* RayTracingCore::get_ray, core.cc:218-234 */
static inline ray_t get_ray(const orc_camera* cam, uint32_t x, uint32_t y, orc_rng* rng) {
    /* sample_square(), random.number.gen.hpp:16: {rd() - 0.5f, rd() - 0.5f, 0} in double, narrowed */
    const float offx = (float)(rd(rng) - (double)0.5f);
    const float offy = (float)(rd(rng) - (double)0.5f);
    const v3 du = vld(cam->pixel_delta_u), dv = vld(cam->pixel_delta_v);
    const v3 pixel_sample =
        vadd(vadd(vld(cam->pixel00), vscale(du, (float)x + offx)), vscale(dv, (float)y + offy));
    v3 origin = vld(cam->cam_center);
    if (!(cam->defocus_angle <= 0.0f)) {
        const v3 p = random_vector_on_unit_disk(rng);
        origin = vadd(vadd(vld(cam->cam_center), vscale(vld(cam->defocus_disk_u), p.x)),
                      vscale(vld(cam->defocus_disk_v), p.y));
    }
    ray_t r;
    r.o = origin;
    r.d = vsub(pixel_sample, origin);
    return r;
}

/* RayTracingCore::compute_color, core.cc:236-257 (recursive, attenuation applied innermost-first) */
static v3 compute_color(const ray_t* r, uint32_t depth, const scene_t* sc, orc_rng* rng, orc_counters* ctr) {
    if (depth == 0) {
        if (ctr) ctr->end_depth++;
        return V(0.0f, 0.0f, 0.0f);
    }
    hit_rec rec;
    if (ctr) ctr->segments++;
    uint32_t hit_index = 0xffffffffu;
    const int hit = sc->nodes || sc->n_slots ? bvh_intersects(sc, r, 0.0001f, &rec, &hit_index, ctr)
                                             : world_intersects(sc, r, 0.0001, (double)INFINITY, &rec, NULL, ctr);
    g_origin_obj = hit ? hit_index : 0xffffffffu; /* (counters of the instrumented walk: where the next segment starts) */
    g_entry_ref = ORC_FROM_ROOT;                  /* (every scattered ray walks from the root) */
    if (hit) {
        const orc_material* m = &sc->mats[rec.material]; /* MaterialCollection::operator[], material.defs.hpp:102 */
        v3 att;
        ray_t scattered;
        if (material_scatter(m, r, &rec, rng, &att, &scattered, ctr)) {
            const v3 inner = compute_color(&scattered, depth - 1, sc, rng, ctr);
            return vmul(att, inner);
        }
        if (ctr) ctr->end_absorbed++;
        return V(0.0f, 0.0f, 0.0f);
    }
    if (ctr) ctr->end_sky++;
    const v3 unit_dir = vnormalize(r->d);
    const float t = 0.5f * (unit_dir.y + 1.0f);
    return vadd(vscale(V(1.0f, 1.0f, 1.0f), 1.0f - t), vscale(V(0.5f, 0.7f, 1.0f), t));
}

/* RayTracingCore::raytrace_pixel, core.cc:259-265: sequential fp32 sum over samples, then * 1/spp */
static void raytrace_pixel(const orc_camera* cam, const scene_t* sc, uint32_t x, uint32_t y, orc_rng* rng,
                           float rgb[3], orc_counters* ctr) {
    v3 pixel_color = V(0.0f, 0.0f, 0.0f);
    for (uint32_t s = 0; s < cam->samples_per_pixel; ++s) {
        if (rng->kind == ORC_RNG_COUNTER) {
            counter_begin(rng, y * cam->img_width + x, s);
        }
        const ray_t r = get_ray(cam, x, y, rng);
        g_origin_obj = 0xffffffffu; /* (walk counters: a camera ray) */
        g_entry_ref = g_tile_entries ? g_tile_entries[(y >> 3) * g_tile_entries_gtx + (x >> 3)] : ORC_FROM_ROOT;
        pixel_color = vadd(pixel_color, compute_color(&r, cam->maxdepth, sc, rng, ctr));
        if (ctr) ctr->samples++;
    }
    const v3 scaled = vscale(pixel_color, cam->pixels_sample_scale);
    vst(rgb, scaled);
}

/* ------------------------------------------------------------------------------------------- */
/* exported unit entry points                                                                    */
/* ------------------------------------------------------------------------------------------- */
static void rec_to_floats(const hit_rec* rec, float out[8]) {
    vst(out, rec->P);
    vst(out + 3, rec->N);
    out[6] = (float)rec->T;
    out[7] = rec->front_face ? 1.0f : 0.0f;
}

int orc_sphere_intersect(const float center[3], float radius, const float origin[3], const float dir[3], double tmin,
                         double tmax, float rec_out[8]) {
    orc_object s;
    memset(&s, 0, sizeof(s));
    memcpy(s.center, center, sizeof(s.center));
    s.radius = radius;
    ray_t r = {vld(origin), vld(dir)};
    hit_rec rec;
    memset(&rec, 0, sizeof(rec));
    if (!sphere_intersects(&s, &r, tmin, tmax, &rec)) return 0;
    rec_to_floats(&rec, rec_out);
    return 1;
}

int orc_world_intersect(const orc_object* objs, uint32_t n, const float origin[3], const float dir[3], float rec_out[8],
                        uint32_t* index_out) {
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.objs = objs;
    sc.n_objs = n;
    ray_t r = {vld(origin), vld(dir)};
    hit_rec rec;
    memset(&rec, 0, sizeof(rec));
    if (!world_intersects(&sc, &r, 0.0001, (double)INFINITY, &rec, index_out, NULL)) return 0;
    rec_to_floats(&rec, rec_out);
    return 1;
}

int orc_scatter(const orc_material* m, const float ray_o[3], const float ray_d[3], const float P[3], const float N[3],
                int front_face, orc_rng* rng, float out[9]) {
    ray_t in = {vld(ray_o), vld(ray_d)};
    hit_rec rec;
    memset(&rec, 0, sizeof(rec));
    rec.P = vld(P);
    rec.N = vld(N);
    rec.front_face = front_face;
    v3 att = V(0, 0, 0);
    ray_t sc = {V(0, 0, 0), V(0, 0, 0)};
    const int ok = material_scatter(m, &in, &rec, rng, &att, &sc, NULL);
    vst(out, att);
    vst(out + 3, sc.o);
    vst(out + 6, sc.d);
    return ok;
}

/* ------------------------------------------------------------------------------------------- */
/* render drivers                                                                               */
/* ------------------------------------------------------------------------------------------- */
static void counters_add(orc_counters* a, const orc_counters* b) {
    uint64_t* pa = (uint64_t*)a;
    const uint64_t* pb = (const uint64_t*)b;
    for (size_t i = 0; i < sizeof(orc_counters) / sizeof(uint64_t); ++i) pa[i] += pb[i];
}

int orc_render_pixels_mt(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats,
                         uint32_t n_mats, uint32_t mt_seed, const uint32_t* xy, uint32_t n_pixels, float* rgb_out,
                         uint32_t* rgba_out, orc_counters* ctr) {
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.objs = objs; sc.n_objs = n_objs; sc.mats = mats; sc.n_mats = n_mats;
    orc_rng* rng = orc_rng_new_mt(mt_seed);
    orc_counters local;
    memset(&local, 0, sizeof(local));
    for (uint32_t i = 0; i < n_pixels; ++i) {
        float rgb[3];
        raytrace_pixel(cam, &sc, xy[2 * i], xy[2 * i + 1], rng, rgb, ctr ? &local : NULL);
        if (rgb_out) memcpy(rgb_out + 3 * (size_t)i, rgb, sizeof(rgb));
        if (rgba_out) rgba_out[i] = orc_pack_rgba(rgb);
    }
    local.rng_doubles = rng->n_doubles;
    if (ctr) counters_add(ctr, &local);
    orc_rng_free(rng);
    return 0;
}

typedef struct {
    const orc_camera* cam;
    const scene_t* sc;
    uint64_t seed;
    uint32_t x0, y0, x1, y1;
    float* rgb_out;
    uint32_t* rgba_out;
    volatile uint32_t* next_row;
    orc_counters ctr;
    int want_ctr;
} rect_job;

static void* rect_worker(void* arg) {
    rect_job* j = (rect_job*)arg;
    orc_rng rng;
    rng_init_counter(&rng, j->seed);
    const uint32_t w = j->x1 - j->x0;
    for (;;) {
        const uint32_t y = __sync_fetch_and_add(j->next_row, 1u);
        if (y >= j->y1) break;
        for (uint32_t x = j->x0; x < j->x1; ++x) {
            float rgb[3];
            raytrace_pixel(j->cam, j->sc, x, y, &rng, rgb, j->want_ctr ? &j->ctr : NULL);
            const size_t o = (size_t)(y - j->y0) * w + (x - j->x0);
            if (j->rgb_out) memcpy(j->rgb_out + 3 * o, rgb, sizeof(rgb));
            if (j->rgba_out) j->rgba_out[o] = orc_pack_rgba(rgb);
        }
    }
    j->ctr.rng_doubles = rng.n_doubles;
    return NULL;
}

static int render_rect(const orc_camera* cam, const scene_t* sc, uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1,
                       uint32_t y1, float* rgb_out, uint32_t* rgba_out, orc_counters* ctr, int nthreads) {
    if (x1 > cam->img_width || y1 > cam->img_height || x0 > x1 || y0 > y1) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile uint32_t next_row = y0;
    rect_job* jobs = (rect_job*)calloc((size_t)nthreads, sizeof(rect_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].cam = cam; jobs[t].sc = sc; jobs[t].seed = seed;
        jobs[t].x0 = x0; jobs[t].y0 = y0; jobs[t].x1 = x1; jobs[t].y1 = y1;
        jobs[t].rgb_out = rgb_out; jobs[t].rgba_out = rgba_out;
        jobs[t].next_row = &next_row;
        jobs[t].want_ctr = ctr != NULL;
    }
    /* compute_color recurses maxdepth levels (core.cc:247, up to 65535): every worker -- also a single one, the caller's own
     * stack may be the 8 MiB default -- gets a stack sized from the bounce limit (~1 KiB of frames per level measured; 4 KiB
     * per level + 1 MiB reserved, touched on demand).  Rows are handed out by a shared counter, so the frame is complete
     * with however many workers start; none starting is an error. */
    if (!jobs || !th) {
        free(jobs);
        free(th);
        return -2;
    }
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    size_t stack_bytes = ((size_t)cam->maxdepth * 4096u + ((size_t)1 << 20) + 65535u) & ~(size_t)65535u;
    int started = 0;
    for (int attempt = 0; attempt < 2 && started == 0; ++attempt) {
        if (pthread_attr_setstacksize(&attr, stack_bytes) != 0) pthread_attr_setstacksize(&attr, (size_t)8 << 20);
        for (int t = 0; t < nthreads; ++t) {
            if (pthread_create(&th[started], &attr, rect_worker, &jobs[started]) != 0) break; /* (address space / cgroup limits) */
            ++started;
        }
        if (started == 0) nthreads = 1; /* second attempt: one worker */
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    pthread_attr_destroy(&attr);
    if (started == 0) {
        free(jobs);
        free(th);
        return -3;
    }
    nthreads = started;
    if (ctr) {
        for (int t = 0; t < nthreads; ++t) counters_add(ctr, &jobs[t].ctr);
    }
    free(jobs);
    free(th);
    return 0;
}

int orc_render_rect_counter(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats,
                            uint32_t n_mats, uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                            float* rgb_out, uint32_t* rgba_out, orc_counters* ctr, int nthreads) {
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.objs = objs; sc.n_objs = n_objs; sc.mats = mats; sc.n_mats = n_mats;
    return render_rect(cam, &sc, seed, x0, y0, x1, y1, rgb_out, rgba_out, ctr, nthreads);
}

int orc_render_rect_counter_bvh(const orc_camera* cam, const orc_object* objs, uint32_t n_objs,
                                const orc_material* mats, uint32_t n_mats, const orc_bvh_node* nodes, uint32_t n_nodes,
                                const uint32_t* slots, uint32_t n_slots, const float* pad_classes, uint32_t n_classes,
                                float pad_eps, float pad_floor, uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1,
                                uint32_t y1, float* rgb_out, uint32_t* rgba_out, orc_counters* ctr, int nthreads) {
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.objs = objs; sc.n_objs = n_objs; sc.mats = mats; sc.n_mats = n_mats;
    sc.nodes = nodes; sc.n_nodes = n_nodes; sc.slots = slots; sc.n_slots = n_slots;
    sc.pad_classes = pad_classes; sc.n_classes = n_classes; sc.pad_eps = pad_eps; sc.pad_floor = pad_floor;
    sc.pad_refine = pad_refine_pays(pad_classes, n_classes, pad_eps);
    if (n_slots == 0) return -1;
    return render_rect(cam, &sc, seed, x0, y0, x1, y1, rgb_out, rgba_out, ctr, nthreads);
}

/* ------------------------------------------------------------------------------------------- */
/* CPU baseline in the shape of the reference's job system (src/main.cc:404-519, 608-633)      */
/* ------------------------------------------------------------------------------------------- */
typedef struct { uint16_t sx, sy, ex, ey; } work_pkg; /* RayTracingWorkPackage, main.cc:404-407 */

typedef struct {
    const orc_camera* cam;
    const scene_t* sc;
    const work_pkg* queue;
    uint32_t n_pkgs;
    volatile uint32_t* next;
    uint32_t stride;
    uint32_t seed;
    uint32_t* rgba_out;
    uint64_t samples;
} bench_job;

static void* bench_worker(void* arg) {
    bench_job* j = (bench_job*)arg;
    orc_rng* rng = orc_rng_new_mt(j->seed); /* one RandomNumberGenerator per worker, main.cc:437 */
    for (;;) {
        const uint32_t i = __sync_fetch_and_add(j->next, 1u); /* MonkaGigaQueue::pop_pkg, main.cc:413-421 */
        if (i >= j->n_pkgs) break;
        const work_pkg p = j->queue[i];
        /* process_tracing_work_package, main.cc:507-519 */
        for (uint32_t y = p.sy; y < p.ey; ++y) {
            for (uint32_t x = p.sx; x < p.ex; ++x) {
                if ((x % j->stride) || (y % j->stride)) continue;
                float rgb[3];
                raytrace_pixel(j->cam, j->sc, x, y, rng, rgb, NULL);
                if (j->rgba_out) j->rgba_out[(size_t)y * j->cam->img_width + x] = orc_pack_rgba(rgb);
                j->samples += j->cam->samples_per_pixel;
            }
        }
    }
    orc_rng_free(rng);
    return NULL;
}

double orc_bench_mt(const orc_camera* cam, const orc_object* objs, uint32_t n_objs, const orc_material* mats,
                    uint32_t n_mats, uint32_t mt_seed, uint32_t stride, int nthreads, uint32_t* rgba_out,
                    uint64_t* samples_out) {
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.objs = objs; sc.n_objs = n_objs; sc.mats = mats; sc.n_mats = n_mats;
    if (stride < 1) stride = 1;
    if (nthreads < 1) nthreads = 1;
    /* 8x8 tiles, main.cc:615-631 */
    const uint32_t W = cam->img_width, H = cam->img_height, T = 8;
    const uint32_t ntx = (W + T - 1) / T, nty = (H + T - 1) / T;
    const uint32_t n_pkgs = ntx * nty;
    work_pkg* q = (work_pkg*)malloc(sizeof(work_pkg) * n_pkgs);
    for (uint32_t ty = 0; ty < nty; ++ty) {
        for (uint32_t tx = 0; tx < ntx; ++tx) {
            work_pkg p;
            p.sx = (uint16_t)(tx * T); p.sy = (uint16_t)(ty * T);
            p.ex = (uint16_t)((tx + 1) * T < W ? (tx + 1) * T : W);
            p.ey = (uint16_t)((ty + 1) * T < H ? (ty + 1) * T : H);
            q[ty * ntx + tx] = p;
        }
    }
    /* std::shuffle, main.cc:633 (Fisher-Yates with our own mt19937 draw; the order is scheduling only) */
    orc_rng* sh = orc_rng_new_mt(mt_seed ^ 0x5bd1e995u);
    for (uint32_t i = n_pkgs; i > 1; --i) {
        const uint32_t k = mt_next(sh) % i;
        const work_pkg t = q[i - 1];
        q[i - 1] = q[k];
        q[k] = t;
    }
    orc_rng_free(sh);

    volatile uint32_t next = 0;
    bench_job* jobs = (bench_job*)calloc((size_t)nthreads, sizeof(bench_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* only the workers that started are joined (a thread limit of the cgroup can refuse one: pthread_join on a thread that
     * was never created is undefined); the packages they leave are rendered by this thread, so every sample is counted */
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].cam = cam; jobs[t].sc = &sc; jobs[t].queue = q; jobs[t].n_pkgs = n_pkgs; jobs[t].next = &next;
        jobs[t].stride = stride; jobs[t].seed = mt_seed + (uint32_t)t; jobs[t].rgba_out = rgba_out;
        if (pthread_create(&th[started], NULL, bench_worker, &jobs[t]) != 0) break;
        ++started;
    }
    if (started < nthreads) bench_worker(&jobs[started]); /* (returns when the queue is empty) */
    uint64_t samples = 0;
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; ++t) samples += jobs[t].samples;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (samples_out) *samples_out = samples;
    free(jobs);
    free(th);
    free(q);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
