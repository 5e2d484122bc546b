// rtmi_host.cpp -- host side of librtmi: camera derivation, scene generator, BVH builder, error text.
//
// Everything here runs once per scene on the CPU; the per-pixel hot path is in rtmi_device.hip.
// Compiled with -ffp-contract=off: the camera constants feed every ray and must come out bit-identical to
// the reference's baseline-x86-64 build (reference src/ray.tracer.core.cc:158-216).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <random>

#include "rtmi_internal.h"

namespace rtmi {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

namespace {

struct Vec3 {
    float x, y, z;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm::dot / normalize / cross / radians as published by glm (pinned by reference CMakeLists.txt:55)
inline float dot(Vec3 a, Vec3 b) {
    const Vec3 t = a * b;
    return t.x + t.y + t.z;
}
inline Vec3 normalize(Vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline Vec3 cross(Vec3 x, Vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline Vec3 load3(const float* p) { return {p[0], p[1], p[2]}; }
inline void store3(float* p, Vec3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

// RandomNumberGenerator (reference src/random.number.gen.hpp:7-48) with an explicit seed instead of
// std::random_device; same engine and distribution types.
class HostRng {
public:
    explicit HostRng(uint32_t seed) : _randgen{seed} {}
    double random_double() { return _randdist(_randgen); }
    double random_double(double r_min, double r_max) { return r_min + (r_max - r_min) * random_double(); }
    Vec3 random_vector(double rmin, double rmax) {
        // braces in the reference: x, y, z are drawn left to right and narrowed to float
        const float x = static_cast<float>(random_double(rmin, rmax));
        const float y = static_cast<float>(random_double(rmin, rmax));
        const float z = static_cast<float>(random_double(rmin, rmax));
        return {x, y, z};
    }

private:
    std::mt19937 _randgen;
    std::uniform_real_distribution<> _randdist{0.0, 1.0};
};

} // namespace
} // namespace rtmi

using namespace rtmi;

extern "C" const char* rtmi_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* rtmi_version(void) {
    return "rtmi 0.6 (gfx950)";
}

// RayTracingCore::default_setup camera block, reference core.cc:171-216.
extern "C" int rtmi_camera_setup(const rtmi_camera_params* cp, rtmi_camera* out) {
    if (!cp || !out) {
        set_error("rtmi_camera_setup: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    const uint32_t image_height = static_cast<uint32_t>(static_cast<float>(cp->image_width) / cp->aspect_ratio);

    const float theta = radians(cp->vertical_fov);
    const float h = std::tan(theta * 0.5f);
    const float viewport_height = 2.0f * h * cp->focus_distance;
    const float viewport_width =
        viewport_height * (static_cast<float>(cp->image_width) / static_cast<float>(image_height));

    // make_camera_frame, core.cc:158-169
    const Vec3 lookfrom = load3(cp->lookfrom), lookat = load3(cp->lookat), world_up = load3(cp->world_up);
    const Vec3 W = normalize(lookfrom - lookat);
    const Vec3 U = normalize(cross(world_up, W));
    const Vec3 V = cross(W, U);

    const Vec3 viewport_u = U * viewport_width;
    const Vec3 viewport_v = -V * viewport_height;
    const Vec3 pixel_delta_u = viewport_u / static_cast<float>(cp->image_width);
    const Vec3 pixel_delta_v = viewport_v / static_cast<float>(image_height);
    const Vec3 viewport_upper_left = lookfrom - cp->focus_distance * W - viewport_u * 0.5f - viewport_v * 0.5f;
    const Vec3 pixel00_loc = viewport_upper_left + 0.5f * (pixel_delta_u + pixel_delta_v);
    const float defocus_radius = cp->focus_distance * std::tan(radians(cp->defocus_angle * 0.5f));

    out->img_width = cp->image_width;
    out->img_height = image_height;
    out->defocus_angle = cp->defocus_angle;
    out->viewport_height = viewport_height;
    out->viewport_width = viewport_width;
    out->samples_per_pixel = cp->samples_per_pixel;
    out->maxdepth = cp->max_depth;
    out->pixels_sample_scale = 1.0f / static_cast<float>(cp->samples_per_pixel);
    store3(out->pixel_delta_u, pixel_delta_u);
    store3(out->pixel_delta_v, pixel_delta_v);
    store3(out->pixel00, pixel00_loc);
    store3(out->cam_center, lookfrom);
    store3(out->defocus_disk_u, U * defocus_radius);
    store3(out->defocus_disk_v, V * defocus_radius);
    return RTMI_OK;
}

// make_world_spheres, reference core.cc:99-149.
extern "C" int rtmi_make_world_spheres(const rtmi_world_def* def, const rtmi_object* fixed_objects,
                                       const rtmi_material* fixed_materials, uint32_t n_fixed, uint32_t mt_seed,
                                       rtmi_object* objects_out, rtmi_material* materials_out, uint32_t capacity,
                                       uint32_t* n_out) {
    if (!def || !objects_out || !materials_out || !n_out || (n_fixed && (!fixed_objects || !fixed_materials))) {
        set_error("rtmi_make_world_spheres: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    uint32_t n = 0;
    auto push = [&](const rtmi_object& o, const rtmi_material& m) -> bool {
        if (n >= capacity) return false;
        objects_out[n] = o;
        objects_out[n].material = n; // MaterialCollection::add returns the insertion index (material.defs.hpp:96-100)
        materials_out[n] = m;
        ++n;
        return true;
    };
    auto lambertian = [](Vec3 albedo) {
        rtmi_material m{};
        m.kind = 0;
        store3(m.p, albedo);
        return m;
    };
    auto metallic = [](Vec3 albedo, float fuzziness) {
        rtmi_material m{};
        m.kind = 1;
        store3(m.p, albedo);
        m.p[3] = std::min(1.0f, fuzziness); // Material::make_metallic, material.defs.hpp:73
        return m;
    };
    auto dielectric = [](float ri) {
        rtmi_material m{};
        m.kind = 2;
        m.p[0] = ri;
        return m;
    };

    for (uint32_t i = 0; i < n_fixed; ++i) { // core.cc:104-122
        rtmi_object o = fixed_objects[i];
        o.kind = 0;
        rtmi_material m = fixed_materials[i];
        if (m.kind == 1) m = metallic(load3(m.p), m.p[3]);
        if (m.kind > 2) {
            set_error("rtmi_make_world_spheres: unknown material kind");
            return RTMI_ERR_BAD_ARG;
        }
        if (!push(o, m)) {
            set_error("rtmi_make_world_spheres: capacity too small");
            return RTMI_ERR_BAD_ARG;
        }
    }

    HostRng rand_gen{mt_seed}; // core.cc:124
    for (int32_t a = def->a_min; a < def->a_max; ++a) {
        for (int32_t b = def->b_min; b < def->b_max; ++b) {
            const float choose_mat = static_cast<float>(rand_gen.random_double());
            // core.cc:128: `a + 0.9f * rd()` and `b + 0.9 * rd()` are double expressions narrowed by the braces
            const float cx = static_cast<float>(a + 0.9f * rand_gen.random_double());
            const float cy = 0.2f;
            const float cz = static_cast<float>(b + 0.9 * rand_gen.random_double());
            // core.cc:130: glm's vec3::length() is the component count, so the test reads `3 > treshold`
            // (bug-compatible: no grid sphere is ever skipped for tresholds below 3).
            if (3.0f > def->center_dist_treshold) {
                rtmi_material m{};
                if (choose_mat < def->diffuse_material_treshold) {
                    const Vec3 c1 = rand_gen.random_vector(0.0f, 1.0f);
                    const Vec3 c2 = rand_gen.random_vector(0.0f, 1.0f);
                    m = lambertian(c1 * c2);
                } else if (choose_mat < def->metal_material_treshold) {
                    // core.cc:137-138: argument evaluation order of make_metallic(random_vector, random_double) is
                    // unspecified by C++; g++/x86-64 (the reference's toolchain) evaluates right to left.
                    const float fuzz = static_cast<float>(rand_gen.random_double(0.0f, 0.5f));
                    const Vec3 albedo = rand_gen.random_vector(0.5f, 1.0f);
                    m = metallic(albedo, fuzz);
                } else {
                    m = dielectric(static_cast<float>(rand_gen.random_double(1.2f, 1.6f)));
                }
                rtmi_object o{};
                o.kind = 0;
                o.center[0] = cx;
                o.center[1] = cy;
                o.center[2] = cz;
                o.radius = 0.2f;
                if (!push(o, m)) {
                    set_error("rtmi_make_world_spheres: capacity too small");
                    return RTMI_ERR_BAD_ARG;
                }
            }
        }
    }
    *n_out = n;
    return RTMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BVH builder (build-side extension; the reference scans linearly, object.defs.cc:68-81).
//
// Binary tree, each node carrying both children's boxes in centre/half-extent form so one 64-byte LDS record
// feeds two slab tests.  Boxes are rounded outward; the kernel adds a per-ray pad (pad_classes) that covers
// the fp32 error of the reference's sphere discriminant, so that every sphere the linear scan would accept
// is reached by the walk (DESIGN.md "Exactness of the BVH").
// ---------------------------------------------------------------------------------------------------------
namespace rtmi {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int i = 0; i < 3; ++i) {
            lo[i] = std::numeric_limits<float>::infinity();
            hi[i] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box& b) {
        for (int i = 0; i < 3; ++i) {
            lo[i] = std::min(lo[i], b.lo[i]);
            hi[i] = std::max(hi[i], b.hi[i]);
        }
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float c[3];
    uint32_t object;
};

inline float down(float v) { return std::nextafter(v, -std::numeric_limits<float>::infinity()); }
inline float up(float v) { return std::nextafter(v, std::numeric_limits<float>::infinity()); }

// The tree while it is built and optimised: explicit nodes (leaves included) with parent links and exact float boxes.
struct TNode {
    Box box;
    int32_t parent = -1;
    int32_t c[2] = {-1, -1};       // children; -1: this node is a leaf
    uint32_t first = 0, count = 0; // leaf: its objects are leaf_objs[first, first + count)
    int32_t height = 0;            // internal levels of the subtree (leaf: 0) -- the walk's stack needs one entry per level
};

struct Builder {
    std::vector<Prim> prims;
    std::vector<TNode> tree;
    std::vector<uint32_t> leaf_objs;
    uint32_t leaf_size;

    static void set_child(rtmi_bvh_node& nd, int k, const Box& b, uint32_t ref) {
        for (int i = 0; i < 3; ++i) {
            const float c = 0.5f * b.lo[i] + 0.5f * b.hi[i];
            // half extent rounded up so [c-h, c+h] contains [lo, hi] despite the rounding of c
            const float h = up(std::max(b.hi[i] - c, c - b.lo[i]));
            nd.ctr[k][i] = c;
            nd.half[k][i] = h;
        }
        nd.child[k] = ref;
    }

    // top-down greedy SAH over prims[begin, end); returns the index of the subtree's node
    int32_t build(uint32_t begin, uint32_t end, uint32_t level) {
        Box box;
        box.reset();
        Box cbox;
        cbox.reset();
        for (uint32_t i = begin; i < end; ++i) {
            box.grow(prims[i].box);
            for (int a = 0; a < 3; ++a) {
                cbox.lo[a] = std::min(cbox.lo[a], prims[i].c[a]);
                cbox.hi[a] = std::max(cbox.hi[a], prims[i].c[a]);
            }
        }
        const uint32_t n = end - begin;
        const int32_t me = static_cast<int32_t>(tree.size());
        tree.emplace_back();
        tree[me].box = box;
        if (n <= leaf_size) {
            tree[me].first = static_cast<uint32_t>(leaf_objs.size());
            tree[me].count = n;
            for (uint32_t i = begin; i < end; ++i) leaf_objs.push_back(prims[i].object);
            return me;
        }

        // split: full-sweep SAH on the three axes for small ranges, binned SAH (32 bins per axis) for large ones, a median
        // split where neither finds a plane (coincident centres) or the tree gets too deep
        uint32_t mid = begin + n / 2;
        int axis = 0;
        {
            float ext = -1.0f;
            for (int a = 0; a < 3; ++a) {
                const float e = cbox.hi[a] - cbox.lo[a];
                if (e > ext) {
                    ext = e;
                    axis = a;
                }
            }
        }
        bool done = false;
        if (level < 24 && n <= 4096) {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1;
            uint32_t best_split = 0;
            std::vector<float> right_area(n);
            for (int a = 0; a < 3; ++a) {
                std::sort(prims.begin() + begin, prims.begin() + end, [a](const Prim& p, const Prim& q) {
                    return p.c[a] < q.c[a] || (p.c[a] == q.c[a] && p.object < q.object);
                });
                Box acc;
                acc.reset();
                for (uint32_t i = n; i-- > 1;) {
                    acc.grow(prims[begin + i].box);
                    right_area[i] = acc.half_area();
                }
                acc.reset();
                for (uint32_t i = 1; i < n; ++i) {
                    acc.grow(prims[begin + i - 1].box);
                    const float cost = acc.half_area() * static_cast<float>(i) + right_area[i] * static_cast<float>(n - i);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_split = i;
                    }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis;
                std::sort(prims.begin() + begin, prims.begin() + end, [a](const Prim& p, const Prim& q) {
                    return p.c[a] < q.c[a] || (p.c[a] == q.c[a] && p.object < q.object);
                });
                mid = begin + best_split;
                done = true;
            }
        } else if (level < 24) {
            constexpr int kBins = 32;
            float best_cost = std::numeric_limits<float>::infinity();
            int best_k = -1, best_axis = -1;
            for (int a = 0; a < 3; ++a) {
                const float lo = cbox.lo[a], ext = cbox.hi[a] - cbox.lo[a];
                if (!(ext > 0.0f)) continue;
                Box bins[kBins];
                uint32_t cnt[kBins] = {};
                for (auto& bb : bins) bb.reset();
                for (uint32_t i = begin; i < end; ++i) {
                    const int k = std::min(std::max(static_cast<int>(kBins * ((prims[i].c[a] - lo) / ext)), 0), kBins - 1);
                    bins[k].grow(prims[i].box);
                    cnt[k]++;
                }
                float ra[kBins];
                uint32_t rc[kBins];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int k = kBins - 1; k > 0; --k) {
                    acc.grow(bins[k]);
                    c += cnt[k];
                    ra[k] = acc.half_area();
                    rc[k] = c;
                }
                acc.reset();
                c = 0;
                for (int k = 1; k < kBins; ++k) {
                    acc.grow(bins[k - 1]);
                    c += cnt[k - 1];
                    if (c == 0 || rc[k] == 0) continue;
                    const float cost = acc.half_area() * static_cast<float>(c) + ra[k] * static_cast<float>(rc[k]);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_k = k;
                        best_axis = a;
                    }
                }
            }
            if (best_k > 0) {
                const int a = best_axis;
                const float lo = cbox.lo[a], ext = cbox.hi[a] - cbox.lo[a];
                auto it = std::partition(prims.begin() + begin, prims.begin() + end, [&](const Prim& p) {
                    return std::min(std::max(static_cast<int>(kBins * ((p.c[a] - lo) / ext)), 0), kBins - 1) < best_k;
                });
                mid = static_cast<uint32_t>(it - prims.begin());
                done = mid > begin && mid < end;
            }
        }
        if (!done) {
            std::nth_element(prims.begin() + begin, prims.begin() + begin + n / 2, prims.begin() + end,
                             [axis](const Prim& p, const Prim& q) {
                                 return p.c[axis] < q.c[axis] || (p.c[axis] == q.c[axis] && p.object < q.object);
                             });
            mid = begin + n / 2;
        }

        const int32_t l = build(begin, mid, level + 1);
        const int32_t r = build(mid, end, level + 1);
        tree[me].c[0] = l;
        tree[me].c[1] = r;
        tree[l].parent = me;
        tree[r].parent = me;
        tree[me].height = 1 + std::max(tree[l].height, tree[r].height);
        return me;
    }

    // ---- post-pass: reinsertion (Bittner, Hapala & Havran, "Fast Insertion-Based Optimization of Bounding Volume
    // Hierarchies", CGF 32(1), 2013).  A greedy top-down build fixes the upper levels before it has seen what ends up below them;
    // here every subtree in turn (largest boxes first) is taken out of the tree -- its parent goes with it, the sibling moves
    // up -- and put back where it adds the least surface area: the area of the new parent's box (the subtree's box joined with
    // the box of the node it is put next to) plus what that enlarges the ancestors by, found by a best-first search over the
    // tree bounded from below by the area already induced.  The leaves keep their spheres; any binary tree over them yields
    // the same frame (the walk is exact for every valid tree), so this only changes how many boxes a ray meets:
    // the expected number of boxes a random ray tests is proportional to the sum of the internal nodes' areas.
    // The tree may not get deeper than `max_height`: the walk's LDS stack is sized from it.
    double inner_area() const {
        double sum = 0.0;
        for (const TNode& t : tree) {
            if (t.c[0] >= 0) sum += t.box.half_area();
        }
        return sum;
    }
    void refit_up(int32_t at) {
        for (; at >= 0; at = tree[at].parent) {
            TNode& t = tree[at];
            t.box = tree[t.c[0]].box;
            t.box.grow(tree[t.c[1]].box);
            t.height = 1 + std::max(tree[t.c[0]].height, tree[t.c[1]].height);
        }
    }
    static float joined_area(const Box& a, const Box& b) {
        Box u = a;
        u.grow(b);
        return u.half_area();
    }
    uint32_t optimise(int32_t root, int32_t max_height, uint32_t max_passes) {
        struct Cand {
            float induced;
            int32_t node, depth;
            bool operator<(const Cand& o) const { return induced > o.induced; } // min-heap on the induced area
        };
        std::vector<Cand> heap;
        std::vector<std::pair<float, int32_t>> order;
        uint32_t moved_total = 0;
        double area_before = inner_area();
        for (uint32_t pass = 0; pass < max_passes; ++pass) {
            order.clear();
            for (int32_t i = 0; i < static_cast<int32_t>(tree.size()); ++i) {
                if (i != root && tree[i].parent != root) order.emplace_back(-tree[i].box.half_area(), i);
            }
            std::sort(order.begin(), order.end()); // largest area first, ties by index: deterministic
            uint32_t moved = 0;
            for (const auto& oc : order) {
                const int32_t nd = oc.second;
                const int32_t par = tree[nd].parent;
                if (par < 0 || par == root) continue; // (became a child of the root meanwhile)
                const int32_t grand = tree[par].parent;
                const int32_t sib = tree[par].c[0] == nd ? tree[par].c[1] : tree[par].c[0];
                // take nd and its parent out: the sibling moves up
                tree[grand].c[tree[grand].c[0] == par ? 0 : 1] = sib;
                tree[sib].parent = grand;
                refit_up(grand);
                // best-first search for the node to put it next to
                const Box& nb = tree[nd].box;
                const float n_area = nb.half_area();
                const int32_t n_height = tree[nd].height;
                float best_cost = std::numeric_limits<float>::infinity();
                int32_t best = -1;
                heap.clear();
                heap.push_back(Cand{0.0f, root, 0});
                while (!heap.empty()) {
                    std::pop_heap(heap.begin(), heap.end());
                    const Cand c = heap.back();
                    heap.pop_back();
                    if (c.induced + n_area >= best_cost) break; // nothing left can be cheaper
                    const TNode& x = tree[c.node];
                    const float direct = joined_area(x.box, nb);
                    const float total = c.induced + direct;
                    if (total < best_cost && c.node != root && c.depth + 1 + std::max(n_height, x.height) <= max_height) {
                        best_cost = total;
                        best = c.node;
                    }
                    if (x.c[0] >= 0) {
                        const float child_induced = total - x.box.half_area();
                        if (child_induced + n_area < best_cost) {
                            for (int k = 0; k < 2; ++k) {
                                heap.push_back(Cand{child_induced, x.c[k], c.depth + 1});
                                std::push_heap(heap.begin(), heap.end());
                            }
                        }
                    }
                }
                if (best < 0) best = sib; // (cannot happen: the old place is always admissible) put it back
                if (best != sib) ++moved;
                // the freed parent node takes `best`'s place, with `best` and nd as its children
                const int32_t bp = tree[best].parent;
                tree[bp].c[tree[bp].c[0] == best ? 0 : 1] = par;
                tree[par].parent = bp;
                tree[par].c[0] = best;
                tree[par].c[1] = nd;
                tree[best].parent = par;
                tree[nd].parent = par;
                refit_up(par);
            }
            moved_total += moved;
            const double area_after = inner_area();
            const bool converged = area_after > 0.998 * area_before || moved == 0;
            area_before = area_after;
            if (converged) break;
        }
        return moved_total;
    }
};

} // namespace

// Whether bounding the class pad by the reach of each segment (rtmi_device.hip, begin_segment) pays: it costs
// every segment ~30 instructions per class and removes work only where the class pad is large next to the spheres.  The
// class pad seen from the middle of the class's centre box, relative to the class's smallest radius: 2.2 on the 316-unit
// grid of config 4 (refined: -50 % sphere tests, -22 % box tests, frame 1.47x faster), 0.01 on S-RTOW (refined: same test
// counts, frame 1.5 % slower).  Threshold 0.05.  (oracle/rt_oracle.c repeats this rule operation for operation.)
bool pad_refine_pays(const float (*classes)[8], uint32_t n_classes, float pad_eps) {
    for (uint32_t c = 0; c < n_classes; ++c) {
        const float* k = classes[c];
        float far2 = 0.0f;
        for (int i = 0; i < 3; ++i) {
            const float m = 0.5f * k[i] + 0.5f * k[3 + i];
            const float d0 = m - k[i], d1 = k[3 + i] - m;
            far2 = far2 + std::max(d0 * d0, d1 * d1);
        }
        const float x = pad_eps * (far2 + k[7]);
        const float e0 = std::min(x * k[6], std::sqrt(x) * 1.000001f);
        if (e0 * k[6] * 2.0f > 0.05f) return true;
    }
    return false;
}

void build_bvh(const rtmi_object* objects, uint32_t n, uint32_t leaf_size, uint32_t optimise_passes, Bvh& out) {
    Builder b;
    b.leaf_size = std::min(std::max(leaf_size, 1u), kMaxLeafSize);
    b.prims.resize(n);
    float maxabs = 0.0f;
    float rmin_all = std::numeric_limits<float>::infinity(), rmax_all = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        Prim& p = b.prims[i];
        const float r = std::fabs(objects[i].radius);
        for (int a = 0; a < 3; ++a) {
            p.c[a] = objects[i].center[a];
            p.box.lo[a] = down(objects[i].center[a] - r);
            p.box.hi[a] = up(objects[i].center[a] + r);
            maxabs = std::max(maxabs, std::max(std::fabs(p.box.lo[a]), std::fabs(p.box.hi[a])));
        }
        p.object = i;
        rmin_all = std::min(rmin_all, r);
        rmax_all = std::max(rmax_all, r);
    }
    b.tree.reserve(2 * static_cast<size_t>(n) + 1);
    b.leaf_objs.reserve(n);
    out.nodes.clear();
    out.slot_object.clear();
    out.depth = 0;
    if (n == 0) {
        out.root_ref = make_leaf_ref(0, 0);
    } else {
        const int32_t root = b.build(0, n, 0);
        // (the post-pass may not deepen the tree: the stack of every lane is sized from its height, and on S-RTOW three more
        // levels would push the scene out of LDS)
        if (optimise_passes && b.tree[root].c[0] >= 0) b.optimise(root, b.tree[root].height, optimise_passes);
        out.depth = static_cast<uint32_t>(b.tree[root].height);
        // slots in depth-first leaf order: the spheres of neighbouring leaves are neighbours in memory
        std::vector<uint32_t> leaf_ref(b.tree.size(), 0u);
        out.slot_object.reserve(n);
        {
            std::vector<int32_t> stack{root};
            while (!stack.empty()) {
                const int32_t at = stack.back();
                stack.pop_back();
                const TNode& t = b.tree[at];
                if (t.c[0] < 0) {
                    leaf_ref[at] = make_leaf_ref(static_cast<uint32_t>(out.slot_object.size()), t.count);
                    for (uint32_t q = 0; q < t.count; ++q) out.slot_object.push_back(b.leaf_objs[t.first + q]);
                } else {
                    stack.push_back(t.c[1]);
                    stack.push_back(t.c[0]);
                }
            }
        }
        if (b.tree[root].c[0] < 0) {
            out.root_ref = leaf_ref[root];
        } else {
            // Internal nodes numbered breadth-first: the levels every ray walks through sit next to each other -- the first
            // few hundred are what the kernels of HBM-resident scenes stage into LDS (rtmi_device.hip, lds_top_nodes).
            std::vector<int32_t> order;
            std::vector<uint32_t> index(b.tree.size(), 0u);
            order.push_back(root);
            for (size_t head = 0; head < order.size(); ++head) {
                const TNode& t = b.tree[order[head]];
                for (int k = 0; k < 2; ++k) {
                    if (b.tree[t.c[k]].c[0] >= 0) order.push_back(t.c[k]);
                }
            }
            for (size_t i = 0; i < order.size(); ++i) index[order[i]] = static_cast<uint32_t>(i);
            out.nodes.resize(order.size());
            for (size_t i = 0; i < order.size(); ++i) {
                const TNode& t = b.tree[order[i]];
                rtmi_bvh_node nd{};
                for (int k = 0; k < 2; ++k) {
                    const TNode& ch = b.tree[t.c[k]];
                    Builder::set_child(nd, k, ch.box, ch.c[0] >= 0 ? index[t.c[k]] : leaf_ref[t.c[k]]);
                }
                out.nodes[i] = nd;
            }
            out.root_ref = 0;
        }
    }

    // Per-ray pad (derivation: DESIGN.md, "Exactness of the BVH").  With u = 2^-24, L = |C - O| and M = max(L, R), the
    // fp32 arithmetic of object.defs.cc:43-60 accepts a sphere only if the ray's line passes within sqrt(R^2 + 25 u M^2) of
    // the centre, and the point of every root it accepts lies within sqrt(R^2 + 50 u M^2) of it.  The walk must reach the
    // sphere whenever that point is inside the search interval, i.e. the point must lie inside the sphere's (and every
    // ancestor's) box: boxes are padded per ray segment by
    //     max_c min(x_c / (2 rmin_c), sqrt(x_c)),   x_c = pad_eps * (far^2(origin, centres of class c) + rmax_c^2),
    // over radius classes c (sqrt(R^2 + x) - R <= min(x / 2R, sqrt(x))), with pad_eps = 64 u >= 50 u plus the
    // second-order terms, and never by less than pad_floor = 16 u * (largest box coordinate), which covers the rounding
    // of the slab test itself (v_rcp_f32 and two FMAs per plane on coordinates of that size).
    out.pad_eps = 64.0f * 5.9604645e-8f;
    out.pad_floor = std::max(16.0f * 5.9604645e-8f * maxabs, 1e-6f);
    out.n_pad_classes = 0;
    // Spheres in the leaves that the kernels peel off the top of the tree (the same spine rule as rtmi_scene_create: up to
    // four leaves hanging directly off the root path) are tested at segment set-up, never through a box: they need no pad,
    // and leaving them out usually removes a whole radius class (the ground sphere of the RTOW scene) from the per-segment
    // pad computation.
    std::vector<char> peeled(n, 0);
    if (n && !out.nodes.empty()) {
        uint32_t cur = out.root_ref, n_pre = 0;
        auto mark = [&](uint32_t leaf) {
            const uint32_t first = leaf & 0x00ffffffu, cnt = (leaf >> 24) & 0x7fu;
            for (uint32_t q = 0; q < cnt; ++q) peeled[out.slot_object[first + q]] = 1;
        };
        while (!(cur & kLeafBit) && n_pre < 4u) {
            const rtmi_bvh_node& nd = out.nodes[cur];
            const bool l0 = (nd.child[0] & kLeafBit) != 0u, l1 = (nd.child[1] & kLeafBit) != 0u;
            if (l0 && l1 && n_pre + 2u <= 4u) {
                mark(nd.child[0]);
                mark(nd.child[1]);
                break;
            }
            if (l0 == l1) break;
            mark(l0 ? nd.child[0] : nd.child[1]);
            ++n_pre;
            cur = l0 ? nd.child[1] : nd.child[0];
        }
    }
    rmin_all = std::numeric_limits<float>::infinity();
    rmax_all = 0.0f;
    uint32_t n_boxed = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (peeled[i]) continue;
        const float r = std::fabs(objects[i].radius);
        rmin_all = std::min(rmin_all, r);
        rmax_all = std::max(rmax_all, r);
        ++n_boxed;
    }
    if (n_boxed) {
        const float l0 = std::log2(std::max(rmin_all, 1e-30f));
        const float l1 = std::log2(std::max(rmax_all, 1e-30f));
        // one class per three octaves of radius, at most kMaxPadClasses: every class costs each ray segment ~23 vector
        // instructions, and within a factor 8 the smallest radius of a class pads the others' boxes by little more
        const uint32_t n_cls = std::min<uint32_t>(kMaxPadClasses, std::max(1u, static_cast<uint32_t>(std::ceil((l1 - l0) / 3.0f))));
        const float step = (l1 - l0) / static_cast<float>(n_cls);
        Box cb[kMaxPadClasses];
        float rmin[kMaxPadClasses];
        float rmax[kMaxPadClasses] = {};
        bool used[kMaxPadClasses] = {};
        for (uint32_t k = 0; k < kMaxPadClasses; ++k) {
            cb[k].reset();
            rmin[k] = std::numeric_limits<float>::infinity();
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (peeled[i]) continue;
            const float r = std::max(std::fabs(objects[i].radius), 1e-30f);
            uint32_t k = 0;
            if (step > 0.0f) {
                k = static_cast<uint32_t>(std::min(std::max((std::log2(r) - l0) / step, 0.0f), static_cast<float>(n_cls - 1)));
            }
            used[k] = true;
            rmin[k] = std::min(rmin[k], r);
            rmax[k] = std::max(rmax[k], r);
            for (int a = 0; a < 3; ++a) {
                cb[k].lo[a] = std::min(cb[k].lo[a], objects[i].center[a]);
                cb[k].hi[a] = std::max(cb[k].hi[a], objects[i].center[a]);
            }
        }
        for (uint32_t k = 0; k < kMaxPadClasses; ++k) {
            if (!used[k]) continue;
            float* pc = out.pad_classes[out.n_pad_classes++];
            for (int a = 0; a < 3; ++a) {
                pc[a] = cb[k].lo[a];
                pc[3 + a] = cb[k].hi[a];
            }
            pc[6] = 1.0f / (2.0f * rmin[k]);
            pc[7] = up(rmax[k] * rmax[k]);
        }
    }
    out.pad_refine = pad_refine_pays(out.pad_classes, out.n_pad_classes, out.pad_eps);
}

} // namespace rtmi

// ---------------------------------------------------------------------------------------------------------
// Leaves peeled off the top of the tree and camera-ray entry points
// ---------------------------------------------------------------------------------------------------------
namespace rtmi {

// The top of the tree as a spine of (leaf | subtree) nodes: up to four such leaves (the ground sphere, whose box is the whole
// scene) are tested at segment set-up and every walk starts below them.  Returns where walks start (a reference in the builder's
// format) or kNoWalkRef when the peeled leaves were the whole tree.
uint32_t peel_top_leaves(const Bvh& bvh, uint32_t pre[4], uint32_t& n_pre) {
    n_pre = 0;
    uint32_t cur = bvh.root_ref;
    if (bvh.nodes.empty()) return cur;
    while (!(cur & kLeafBit) && n_pre < 4u) {
        const rtmi_bvh_node& nd = bvh.nodes[cur];
        const bool l0 = (nd.child[0] & kLeafBit) != 0u, l1 = (nd.child[1] & kLeafBit) != 0u;
        if (l0 && l1 && n_pre + 2u <= 4u) { // the spine ends in two leaves: nothing left to walk
            pre[n_pre++] = nd.child[0];
            pre[n_pre++] = nd.child[1];
            return kNoWalkRef;
        }
        if (l0 == l1) break;
        pre[n_pre++] = l0 ? nd.child[0] : nd.child[1];
        cur = l0 ? nd.child[1] : nd.child[0];
    }
    return cur;
}

// Walks of scattered rays that start where the ray does (round 6; trees that stay in HBM).  A ray scattered off a sphere starts
// INSIDE the box of every ancestor of that sphere's leaf, so a walk from the root descends to that leaf whatever the tree: on the
// 100k-sphere grid of config 4, 15.1 of the 15.8 node trips of such a segment, with the sibling culled on 97 % of the levels
// (tools/descent_score.py).  Any start node is exact as long as every subtree hanging off the path above it is still tested
// (the closest hit does not depend on the visiting order), so: the walk starts at the sphere's own leaf (its
// even-depth ancestor-or-self, at most kMaxWays * 2 levels down) with the WAY pre-loaded on its stack -- one record per two levels
// of the path, holding the boxes and references of the two siblings that hang off the path there, IN THE NODE FORMAT: the
// unchanged hand-written node step tests two levels a trip, pushes what it hits and pops the next record.  Round 5 built the same
// idea with per-level sibling lists tested by a compiled loop at segment set-up and lost 14-31 %; the replay of the kernel's rounds
// (tools/wave_replay.py grid) scored this form first: node trips per round 17.5 -> 13.2 on the grid.
// Output: way records appended to `nodes` (indices >= the tree's node count), and per sphere SLOT one 16-word start record
// {start reference, n <= max_ways, way record indices top of the tree first (popped last), ..., word 14: the start's path code}; n = 0
// and the walk's root where there is no way (leaves one or two levels below the root, peeled leaves).  This is the form the library
// exports (rtmi_scene_get_walk_starts); the device reads a 16-byte packing of it (csrc/rtmi_device.hip).
void build_walk_starts(Bvh& bvh, uint32_t walk_root, std::vector<uint32_t>& start_records, std::vector<uint32_t>* way_depth,
                       std::vector<uint32_t>* way_code, uint32_t max_ways) {
    const uint32_t n_slots = (uint32_t)bvh.slot_object.size();
    start_records.assign((size_t)n_slots * 16u, 0u);
    for (uint32_t sl = 0; sl < n_slots; ++sl) start_records[(size_t)sl * 16u] = walk_root;
    if (walk_root == kNoWalkRef || (walk_root & kLeafBit) || bvh.nodes.empty()) return;
    const uint32_t n_nodes = (uint32_t)bvh.nodes.size();
    // elements: internal nodes [0, n_nodes), then leaves in discovery order
    struct El { uint32_t ref, parent, depth, side; };
    std::vector<El> el(n_nodes);
    std::vector<uint32_t> leaf_el; // per leaf element: index into el
    std::vector<uint32_t> stack{walk_root};
    el[walk_root] = El{walk_root, 0xffffffffu, 0u, 0u};
    std::vector<uint32_t> slot_el(n_slots, 0xffffffffu);
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        for (uint32_t k = 0; k < 2; ++k) {
            const uint32_t c = bvh.nodes[n].child[k];
            uint32_t e;
            if (c & kLeafBit) {
                e = (uint32_t)el.size();
                el.push_back(El{});
                const uint32_t first = c & 0x00ffffffu, cnt = (c >> 24) & 0x7fu;
                for (uint32_t q = 0; q < cnt && first + q < n_slots; ++q) slot_el[first + q] = e;
            } else {
                e = c;
                stack.push_back(c);
            }
            el[e] = El{c, n, el[n].depth + 1u, k};
        }
    }
    const uint32_t kMaxWays = std::min(std::max(max_ways, 1u), 12u); // (12 ids + two words of path code fit one 64-byte start record)
    // path code of an element: two bits per pair of levels -- (side at depth 2j - 1) << 1 | side at depth 2j -- the top pair in the
    // most significant position: the device places the way records of the top levels BY this code (csrc/rtmi_device.hip)
    auto path_code = [&](uint32_t e) {
        uint32_t code = 0, shift = 0;
        for (uint32_t x = e; el[x].depth >= 2u; x = el[el[x].parent].parent, shift += 2u)
            code |= ((el[el[x].parent].side << 1) | el[x].side) << shift;
        return code;
    };
    std::vector<uint32_t> way_of(el.size(), 0xffffffffu);
    auto way = [&](uint32_t e) { // the way record of even-depth element e (depth >= 2): the sibling of its parent, its own sibling
        if (way_of[e] != 0xffffffffu) return way_of[e];
        const uint32_t p = el[e].parent, g = el[p].parent;
        rtmi_bvh_node v{};
        const uint32_t sp = 1u - el[p].side, se = 1u - el[e].side;
        // (copies: bvh.nodes grows below)
        const rtmi_bvh_node ng = bvh.nodes[g], np = bvh.nodes[p];
        std::memcpy(v.ctr[0], ng.ctr[sp], 12); std::memcpy(v.half[0], ng.half[sp], 12); v.child[0] = ng.child[sp];
        std::memcpy(v.ctr[1], np.ctr[se], 12); std::memcpy(v.half[1], np.half[se], 12); v.child[1] = np.child[se];
        way_of[e] = (uint32_t)bvh.nodes.size();
        bvh.nodes.push_back(v);
        if (way_depth) way_depth->push_back(el[e].depth); // (the deeper of the two levels the record covers)
        if (way_code) way_code->push_back(path_code(e));
        return way_of[e];
    };
    for (uint32_t sl = 0; sl < n_slots; ++sl) {
        uint32_t e = slot_el[sl];
        if (e == 0xffffffffu) continue; // a peeled leaf
        if (el[e].depth & 1u) e = el[e].parent;
        while (el[e].depth > 2u * kMaxWays) e = el[el[e].parent].parent;
        if (el[e].depth == 0u) continue;
        uint32_t chain[12], m = 0;
        for (uint32_t x = e; el[x].depth >= 2u; x = el[el[x].parent].parent) chain[m++] = way(x);
        uint32_t* r = &start_records[(size_t)sl * 16u];
        r[0] = el[e].ref;
        r[1] = m;
        for (uint32_t i = 0; i < m; ++i) r[2u + i] = chain[m - 1u - i]; // top of the tree first: the deepest record is popped first
        r[14] = path_code(e); // (2 m bits)
    }
}

// Camera rays get an entry point (round 6).  Every sample of a pixel of one 8x8 tile is a ray from a point of the lens disk
// through a point of the tile's rectangle on the focus plane (RayTracingCore::get_ray, core.cc:218-234: origin = cam_center +
// dx disk_u + dy disk_v with dx^2 + dy^2 < 1, target = pixel00 + (x + ox) du + (y + oy) dv with ox, oy in [-0.5, 0.5)): at ray
// parameter t its point lies within rho(t) = |1 - t| r_lens + t r_rect of the AXIS point A(t) = cam_center + t (tile centre -
// cam_center).  A sphere whose root the reference's fp32 arithmetic could accept has the root's point within R + e(L) of its
// centre, so a sphere the beam never comes within R + e(L) + (rounding of get_ray itself) of can be hit by no
// sample of the tile, and the tile's walks may start at the LOWEST COMMON ANCESTOR of the leaves of the spheres that remain --
// or not walk at all when none does (the sky).  The oracle's replay of the kernel's rounds (tools/wave_replay.py) scored it
// before it was built: camera rays 8.9 -> 2.0 node trips each on S-RTOW, 10.5 -> 9.3 node trips per round of a wave.
// Exactness does not rest on the order of the walk, only on this list being complete; all of it in double,
// with the margins below.  entries[tile of the WHOLE image, row-major] = reference in the builder's format, kNoWalkRef = none.
namespace {
struct Beam {
    double a0[3], ax[3], alen, r_lens, r_rect, coord_max;
    // does the beam come within `reach` of point c (for some t >= 0)?  Two pieces on which rho is linear; on each the squared
    // distance to the axis minus (reach + rho)^2 is a convex parabola in the arc length s: its minimum over the piece decides.
    bool reaches(const double c[3], double reach) const {
        const double w[3] = {c[0] - a0[0], c[1] - a0[1], c[2] - a0[2]};
        const double h = w[0] * ax[0] + w[1] * ax[1] + w[2] * ax[2];
        const double q2 = std::max(0.0, w[0] * w[0] + w[1] * w[1] + w[2] * w[2] - h * h);
        auto piece = [&](double b, double k, double lo, double hi) {
            const double a = 1.0 - k * k;
            if (!(a > 1e-6)) return true; // (a beam that opens at 45 degrees or more: no claim)
            double s = (h + b * k) / a;
            s = std::min(std::max(s, lo), hi);
            const double f = q2 + (h - s) * (h - s) - (b + k * s) * (b + k * s);
            return f <= 1e-9 * (q2 + h * h + b * b);
        };
        return piece(reach + r_lens, (r_rect - r_lens) / alen, 0.0, alen) ||
               piece(reach - r_lens, (r_rect + r_lens) / alen, alen, std::numeric_limits<double>::infinity());
    }
    // e(L) of DESIGN.md 5.4 for a sphere of radius r whose centre is at most `l` from any lens point (r = 0: the square-root form),
    // plus the rounding of get_ray: the fp32 origin and direction are each within ~18u of the ideal ones relative to the largest
    // coordinate in play, so the fp32 ray's point at parameter t within 64u S (1 + t) of an ideal ray's -- three times over
    double margin(double l, double r) const {
        const double u = 5.9604644775390625e-8, x = 64.0 * u * (l * l + r * r);
        const double e = r > 0.0 ? std::min(x / (2.0 * r), std::sqrt(x)) : std::sqrt(x);
        const double t_reach = (l + r) / alen + 1.0;
        return 1.001 * e + 64.0 * u * std::max(coord_max, l + r) * (1.0 + t_reach);
    }
};
} // namespace

void build_tile_entries(const rtmi_camera& cam, const rtmi_object* objects, const Bvh& bvh, uint32_t walk_root,
                        std::vector<uint32_t>& entries) {
    const uint32_t gtx = (cam.img_width + 7u) / 8u, gty = (cam.img_height + 7u) / 8u;
    entries.assign((size_t)gtx * gty, walk_root);
    if (walk_root == kNoWalkRef || bvh.nodes.empty()) return;
    double du = 0.0, dv = 0.0, lu = 0.0, lv = 0.0, cmax = 0.0;
    for (int i = 0; i < 3; ++i) {
        du += (double)cam.pixel_delta_u[i] * cam.pixel_delta_u[i];
        dv += (double)cam.pixel_delta_v[i] * cam.pixel_delta_v[i];
        lu += (double)cam.defocus_disk_u[i] * cam.defocus_disk_u[i];
        lv += (double)cam.defocus_disk_v[i] * cam.defocus_disk_v[i];
        const double span = std::fabs((double)cam.pixel_delta_u[i]) * (cam.img_width + 1.0) + std::fabs((double)cam.pixel_delta_v[i]) * (cam.img_height + 1.0);
        cmax = std::max(cmax, std::fabs((double)cam.pixel00[i]) + span);
        cmax = std::max(cmax, std::fabs((double)cam.cam_center[i]) + std::fabs((double)cam.defocus_disk_u[i]) + std::fabs((double)cam.defocus_disk_v[i]));
    }
    if (!std::isfinite(du + dv + lu + lv + cmax)) return; // (lookfrom == lookat: an all-NaN camera; every walk from the root)
    Beam b;
    b.r_rect = 4.0 * (std::sqrt(du) + std::sqrt(dv)) * 1.001; // |a du + b dv|, |a|, |b| <= 4
    b.r_lens = cam.defocus_angle <= 0.0f ? 0.0 : (std::sqrt(lu) + std::sqrt(lv)) * 1.001;
    b.coord_max = cmax;
    struct Frame { uint32_t ref; int state; uint32_t e0; };
    std::vector<Frame> st;
    for (uint32_t ty = 0; ty < gty; ++ty) {
        for (uint32_t tx = 0; tx < gtx; ++tx) {
            double alen2 = 0.0;
            for (int i = 0; i < 3; ++i) {
                b.a0[i] = cam.cam_center[i];
                b.ax[i] = (double)cam.pixel00[i] + (double)cam.pixel_delta_u[i] * (8.0 * tx + 3.5) + (double)cam.pixel_delta_v[i] * (8.0 * ty + 3.5) - b.a0[i];
                alen2 += b.ax[i] * b.ax[i];
            }
            b.alen = std::sqrt(alen2);
            if (!(b.alen > 1e-12) || !std::isfinite(b.alen)) continue; // degenerate camera: from the root
            for (int i = 0; i < 3; ++i) b.ax[i] /= b.alen;
            // does any sphere of leaf `ref` survive?
            auto leaf_reached = [&](uint32_t ref) {
                const uint32_t first = ref & 0x00ffffffu, cnt = (ref >> 24) & 0x7fu;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const rtmi_object& o = objects[bvh.slot_object[first + k]];
                    const double c[3] = {o.center[0], o.center[1], o.center[2]}, r = std::fabs((double)o.radius);
                    const double l = std::sqrt((c[0] - b.a0[0]) * (c[0] - b.a0[0]) + (c[1] - b.a0[1]) * (c[1] - b.a0[1]) + (c[2] - b.a0[2]) * (c[2] - b.a0[2])) + b.r_lens;
                    if (b.reaches(c, r + b.margin(l, r))) return true;
                }
                return false;
            };
            // can the box of child k of node n hold such a sphere?  (its bounding ball: every sphere below lies inside the box)
            auto box_reached = [&](const rtmi_bvh_node& nd, int k) {
                const double c[3] = {nd.ctr[k][0], nd.ctr[k][1], nd.ctr[k][2]};
                const double rb = std::sqrt((double)nd.half[k][0] * nd.half[k][0] + (double)nd.half[k][1] * nd.half[k][1] + (double)nd.half[k][2] * nd.half[k][2]);
                if (!std::isfinite(rb)) return true;
                const double l = std::sqrt((c[0] - b.a0[0]) * (c[0] - b.a0[0]) + (c[1] - b.a0[1]) * (c[1] - b.a0[1]) + (c[2] - b.a0[2]) * (c[2] - b.a0[2])) + rb + b.r_lens;
                return b.reaches(c, rb + b.margin(l, 0.0)); // (r = 0: the square-root form bounds e for every radius below)
            };
            // entry(ref) = the lowest common ancestor of the leaves with a surviving sphere below ref (kNoWalkRef: none), iteratively
            st.clear();
            st.push_back({walk_root, 0, kNoWalkRef});
            uint32_t ret = kNoWalkRef;
            while (!st.empty()) {
                Frame& f = st.back();
                if (f.ref & kLeafBit) {
                    ret = leaf_reached(f.ref) ? f.ref : kNoWalkRef;
                    st.pop_back();
                    continue;
                }
                const rtmi_bvh_node& nd = bvh.nodes[f.ref];
                if (f.state == 0) {
                    f.state = 1;
                    if (box_reached(nd, 0)) { st.push_back({nd.child[0], 0, kNoWalkRef}); continue; }
                    ret = kNoWalkRef;
                }
                if (f.state == 1) {
                    f.e0 = ret;
                    f.state = 2;
                    if (box_reached(nd, 1)) { st.push_back({nd.child[1], 0, kNoWalkRef}); continue; }
                    ret = kNoWalkRef;
                }
                const uint32_t e1 = ret, e0 = f.e0, self = f.ref;
                ret = (e0 != kNoWalkRef && e1 != kNoWalkRef) ? self : (e0 != kNoWalkRef ? e0 : e1);
                st.pop_back();
            }
            entries[(size_t)ty * gtx + tx] = ret;
        }
    }
}

} // namespace rtmi

extern "C" int rtmi_bvh_build(const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size,
                              rtmi_bvh_node* nodes_out, uint32_t* n_nodes, uint32_t* slots_out, uint32_t* root_ref,
                              uint32_t* depth, float* pad_classes_out, uint32_t* n_classes, float* pad_eps,
                              float* pad_floor) {
    return rtmi_bvh_build_passes(objects, n_objects, leaf_size, 0u, nodes_out, n_nodes, slots_out, root_ref, depth,
                                 pad_classes_out, n_classes, pad_eps, pad_floor);
}

extern "C" int rtmi_bvh_build_passes(const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size, uint32_t bvh_passes,
                                     rtmi_bvh_node* nodes_out, uint32_t* n_nodes, uint32_t* slots_out, uint32_t* root_ref,
                                     uint32_t* depth, float* pad_classes_out, uint32_t* n_classes, float* pad_eps,
                                     float* pad_floor) {
    if (n_objects && !objects) {
        set_error("rtmi_bvh_build: null objects");
        return RTMI_ERR_BAD_ARG;
    }
    for (uint32_t i = 0; i < n_objects; ++i) {
        const rtmi_object& o = objects[i];
        // a NaN centre breaks the strict weak ordering of the builder's sorts, an infinite radius gives NaN boxes
        if (!std::isfinite(o.center[0]) || !std::isfinite(o.center[1]) || !std::isfinite(o.center[2]) ||
            !std::isfinite(o.radius)) {
            set_error("rtmi_bvh_build: object with a non-finite centre or radius");
            return RTMI_ERR_BAD_ARG;
        }
    }
    Bvh bvh;
    try {
        build_bvh(objects, n_objects, leaf_size ? leaf_size : (n_objects > 0x2000u ? 4u : 2u), // (the defaults of rtmi_scene_create)
                  bvh_passes ? bvh_passes - 1u : default_bvh_passes(n_objects), bvh);
    } catch (const std::bad_alloc&) {
        set_error("rtmi_bvh_build: out of host memory");
        return RTMI_ERR_OOM;
    } catch (...) {
        set_error("rtmi_bvh_build: unexpected exception");
        return RTMI_ERR_INTERNAL;
    }
    if (n_nodes) *n_nodes = static_cast<uint32_t>(bvh.nodes.size());
    if (root_ref) *root_ref = bvh.root_ref;
    if (depth) *depth = bvh.depth;
    if (n_classes) *n_classes = bvh.n_pad_classes;
    if (pad_eps) *pad_eps = bvh.pad_eps;
    if (pad_floor) *pad_floor = bvh.pad_floor;
    if (nodes_out && !bvh.nodes.empty()) std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(rtmi_bvh_node));
    if (slots_out && n_objects) std::memcpy(slots_out, bvh.slot_object.data(), n_objects * sizeof(uint32_t));
    if (pad_classes_out) std::memcpy(pad_classes_out, bvh.pad_classes, sizeof(bvh.pad_classes));
    return RTMI_OK;
}

extern "C" int rtmi_tile_entries_build(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size,
                                       uint32_t bvh_passes, uint32_t* entries_out, uint32_t* n_tiles) {
    if (!camera || (n_objects && !objects)) {
        set_error("rtmi_tile_entries_build: null argument");
        return RTMI_ERR_BAD_ARG;
    }
    for (uint32_t i = 0; i < n_objects; ++i) {
        const rtmi_object& o = objects[i];
        if (!std::isfinite(o.center[0]) || !std::isfinite(o.center[1]) || !std::isfinite(o.center[2]) || !std::isfinite(o.radius)) {
            set_error("rtmi_tile_entries_build: object with a non-finite centre or radius");
            return RTMI_ERR_BAD_ARG;
        }
    }
    try {
        Bvh bvh;
        build_bvh(objects, n_objects, leaf_size ? leaf_size : (n_objects > 0x2000u ? 4u : 2u),
                  bvh_passes ? bvh_passes - 1u : default_bvh_passes(n_objects), bvh);
        uint32_t pre[4], n_pre = 0;
        const uint32_t walk_root = peel_top_leaves(bvh, pre, n_pre);
        std::vector<uint32_t> entries;
        build_tile_entries(*camera, objects, bvh, walk_root, entries);
        if (n_tiles) *n_tiles = static_cast<uint32_t>(entries.size());
        if (entries_out && !entries.empty()) std::memcpy(entries_out, entries.data(), entries.size() * sizeof(uint32_t));
    } catch (const std::bad_alloc&) {
        set_error("rtmi_tile_entries_build: out of host memory");
        return RTMI_ERR_OOM;
    } catch (...) {
        set_error("rtmi_tile_entries_build: unexpected exception");
        return RTMI_ERR_INTERNAL;
    }
    return RTMI_OK;
}

extern "C" int rtmi_shard_plan(uint32_t height, uint32_t block_rows, uint32_t n_ranks, const uint64_t* block_cost, uint32_t* rank_of_block_out) {
    if (block_rows == 0 || n_ranks == 0 || (height && !rank_of_block_out)) {
        set_error("rtmi_shard_plan: null argument, or block_rows / n_ranks of zero");
        return RTMI_ERR_BAD_ARG;
    }
    const uint32_t nb = (height + block_rows - 1u) / block_rows;
    if (!block_cost) {
        for (uint32_t b = 0; b < nb; ++b) rank_of_block_out[b] = b % n_ranks;
        return RTMI_OK;
    }
    try {
        std::vector<uint32_t> order(nb);
        for (uint32_t b = 0; b < nb; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return block_cost[a] > block_cost[b]; });
        std::vector<uint64_t> load(n_ranks, 0u);
        std::vector<uint32_t> count(n_ranks, 0u);
        const uint32_t cap = (nb + n_ranks - 1u) / n_ranks;
        for (uint32_t b : order) {
            uint32_t best = n_ranks;
            for (uint32_t r = 0; r < n_ranks; ++r)
                if (count[r] < cap && (best == n_ranks || load[r] < load[best] || (load[r] == load[best] && count[r] < count[best]))) best = r;
            rank_of_block_out[b] = best;
            load[best] += block_cost[b];
            count[best]++;
        }
    } catch (...) {
        set_error("rtmi_shard_plan: out of host memory");
        return RTMI_ERR_OOM;
    }
    return RTMI_OK;
}

extern "C" int rtmi_walk_starts_build(const rtmi_object* objects, uint32_t n_objects, uint32_t leaf_size, uint32_t bvh_passes,
                                      rtmi_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* n_tree_nodes_out, uint32_t* records_out) {
    if (n_objects && !objects) {
        set_error("rtmi_walk_starts_build: null objects");
        return RTMI_ERR_BAD_ARG;
    }
    for (uint32_t i = 0; i < n_objects; ++i) {
        const rtmi_object& o = objects[i];
        if (!std::isfinite(o.center[0]) || !std::isfinite(o.center[1]) || !std::isfinite(o.center[2]) || !std::isfinite(o.radius)) {
            set_error("rtmi_walk_starts_build: object with a non-finite centre or radius");
            return RTMI_ERR_BAD_ARG;
        }
    }
    try {
        Bvh bvh;
        build_bvh(objects, n_objects, leaf_size ? leaf_size : (n_objects > 0x2000u ? 4u : 2u),
                  bvh_passes ? bvh_passes - 1u : default_bvh_passes(n_objects), bvh);
        const uint32_t n_tree = static_cast<uint32_t>(bvh.nodes.size());
        uint32_t pre[4], n_pre = 0;
        const uint32_t walk_root = peel_top_leaves(bvh, pre, n_pre);
        std::vector<uint32_t> starts;
        build_walk_starts(bvh, walk_root, starts, nullptr, nullptr, 12u);
        if (n_nodes_out) *n_nodes_out = static_cast<uint32_t>(bvh.nodes.size());
        if (n_tree_nodes_out) *n_tree_nodes_out = n_tree;
        if (nodes_out && !bvh.nodes.empty()) std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(rtmi_bvh_node));
        if (records_out && !starts.empty()) std::memcpy(records_out, starts.data(), starts.size() * sizeof(uint32_t));
    } catch (const std::bad_alloc&) {
        set_error("rtmi_walk_starts_build: out of host memory");
        return RTMI_ERR_OOM;
    } catch (...) {
        set_error("rtmi_walk_starts_build: unexpected exception");
        return RTMI_ERR_INTERNAL;
    }
    return RTMI_OK;
}
