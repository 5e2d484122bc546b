// rtmi_internal.h -- shared between the host side (rtmi_host.cpp) and the HIP side (rtmi_device.hip).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "rtmi.h"

namespace rtmi {

static_assert(sizeof(rtmi_object) == 24, "HittableObject layout (reference object.defs.hpp:30-57) is 24 bytes");
static_assert(sizeof(rtmi_material) == 20, "Material layout (reference material.defs.hpp:49-55) is 20 bytes");
static_assert(sizeof(rtmi_bvh_node) == 64, "BVH node is one 64-byte LDS record");
static_assert(sizeof(rtmi_camera) == 100, "14 POD fields of RayTracingCore (reference core.hpp:19-32)");

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kMaxPadClasses = 4;
constexpr uint32_t kMaxLeafSize = 4;
constexpr uint32_t kNoWalkRef = 0xffffffffu; // "nothing to walk": every leaf was peeled, or (camera entries) the tile's beam meets no sphere

// The counter streams are keyed by a bijective mix of the caller's 64-bit seed (the finaliser of splitmix64, Steele,
// Lea & Flood, OOPSLA'14): seeds that differ in one bit, or only in their high word, give unrelated key words
// (oracle/rt_oracle.c: orc_mix_seed).
inline uint64_t mix_seed(uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) { return kLeafBit | (count << 24) | first; }

// Host-built acceleration structure.  Slots are the spheres in leaf order; slot_object[i] is the index of the
// object (in insertion order, i.e. the order HittableObject_Collection::add_object was called) stored at slot i.
struct Bvh {
    std::vector<rtmi_bvh_node> nodes; // empty when the whole scene is one leaf
    std::vector<uint32_t> slot_object;
    uint32_t root_ref = 0;
    uint32_t depth = 0; // number of internal levels on the longest root-to-leaf path (== max stack entries)
    float pad_classes[kMaxPadClasses][8] = {};
    uint32_t n_pad_classes = 0;
    float pad_eps = 0.0f;
    float pad_floor = 0.0f;
    bool pad_refine = false; // pad_refine_pays(): bound the class pad by the segment's reach (scenes much wider than their spheres)
};

void set_error(const std::string& msg);
// Reinsertion passes after the top-down SAH build (rtmi_tuning::bvh_passes = 0).  Measured with the oracle's instrumented walk
// (tools/tree_score.py, profiles/r04_tree_quality.txt): the greedy full-sweep SAH trees of these sphere fields are already at a
// local optimum of the surface-area sum -- one pass moves 3 of 574 subtrees on S-RTOW (84.90 -> 84.68 box tests per sample)
// and 351 of 65 838 on the 100k-sphere grid (461.12 -> 461.22, +40 ms of build) -- so: two passes where they cost a
// millisecond, none on trees that stay in HBM.
inline uint32_t default_bvh_passes(uint32_t n_objects) { return n_objects > 0x2000u ? 0u : 2u; }
void build_bvh(const rtmi_object* objects, uint32_t n, uint32_t leaf_size, uint32_t optimise_passes, Bvh& out);
bool pad_refine_pays(const float (*classes)[8], uint32_t n_classes, float pad_eps);
uint32_t peel_top_leaves(const Bvh& bvh, uint32_t pre[4], uint32_t& n_pre);
void build_walk_starts(Bvh& bvh, uint32_t walk_root, std::vector<uint32_t>& start_records, std::vector<uint32_t>* way_depth,
                       std::vector<uint32_t>* way_code, uint32_t max_ways);
void build_tile_entries(const rtmi_camera& cam, const rtmi_object* objects, const Bvh& bvh, uint32_t walk_root, std::vector<uint32_t>& entries);

} // namespace rtmi
