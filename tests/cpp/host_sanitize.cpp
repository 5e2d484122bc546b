// Host-only part of librtmi (csrc/rtmi_host.cpp: camera derivation, scene generator, SAH BVH builder) and the C oracle
// under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the pool; this is the CPU
// build the task statement asks sanitizers to run on.  Built and run by tests/test_sanitizers_cpu.py; exits non-zero on
// a wrong status, the sanitizers abort on anything else.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <limits>
#include <random>
#include <vector>

#include "rtmi.h"
extern "C" {
#include "rt_oracle.h"
}

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static int build(const std::vector<rtmi_object>& objs, uint32_t leaf, int expect) {
    const uint32_t n = (uint32_t)objs.size();
    std::vector<rtmi_bvh_node> nodes(n + 1);
    std::vector<uint32_t> slots(n + 1);
    float classes[32], eps = 0, floor_ = 0;
    uint32_t n_nodes = 0, root = 0, depth = 0, n_cls = 0;
    const int rc = rtmi_bvh_build(objs.data(), n, leaf, nodes.data(), &n_nodes, slots.data(), &root, &depth, classes, &n_cls,
                                  &eps, &floor_);
    if (rc != expect) {
        std::fprintf(stderr, "rtmi_bvh_build(n=%u, leaf=%u) = %d, expected %d (%s)\n", n, leaf, rc, expect, rtmi_last_error());
        return 1;
    }
    if (rc == RTMI_OK && n > 0) {
        if (n_nodes > n || n_cls > 4) return 1;
        std::vector<int> seen(n, 0); // every object sits in exactly one slot
        for (uint32_t i = 0; i < n; ++i) {
            if (slots[i] >= n || seen[slots[i]]++) return 1;
        }
    }
    return 0;
}

int main() {
    // camera derivations, including degenerate ones (zero width, narrow, huge)
    for (uint32_t w : {0u, 1u, 17u, 400u, 1920u, 70000u}) {
        rtmi_camera_params cp{16.0f / 9.0f, w, 8, 5, 20.0f, 0.6f, 10.0f, {13, 2, 3}, {0, 0, 0}, {0, 1, 0}};
        rtmi_camera cam;
        CHECK(rtmi_camera_setup(&cp, &cam) == RTMI_OK);
        CHECK(rtmi_camera_setup(nullptr, &cam) == RTMI_ERR_BAD_ARG);
    }
    // the reference's scene generator at several seeds, and with too small a capacity
    rtmi_world_def wd{-11, 11, -11, 11, {4.0f, 0.2f, 0.0f}, 0.9f, 0.8f, 0.95f};
    for (uint32_t seed : {0u, 1u, 12345u, 0xffffffffu}) {
        std::vector<rtmi_object> objs(600);
        std::vector<rtmi_material> mats(600);
        uint32_t n = 0;
        CHECK(rtmi_make_world_spheres(&wd, nullptr, nullptr, 0, seed, objs.data(), mats.data(), 600, &n) == RTMI_OK);
        CHECK(n > 0 && n <= 600);
        objs.resize(n);
        for (uint32_t leaf : {0u, 1u, 2u, 3u, 4u}) CHECK(build(objs, leaf, RTMI_OK) == 0);
        uint32_t m = 0;
        CHECK(rtmi_make_world_spheres(&wd, nullptr, nullptr, 0, seed, objs.data(), mats.data(), 3, &m) != RTMI_OK);
    }
    // builder on degenerate and adversarial inputs
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-50.0f, 50.0f);
    for (uint32_t n : {0u, 1u, 2u, 3u, 5u, 64u, 1000u, 20000u}) {
        std::vector<rtmi_object> objs(n);
        for (auto& o : objs) o = rtmi_object{0, {u(rng), u(rng), u(rng)}, std::fabs(u(rng)) * 0.05f + 0.01f, 0};
        CHECK(build(objs, 2, RTMI_OK) == 0);
        for (auto& o : objs) o = rtmi_object{0, {1.0f, 2.0f, 3.0f}, 0.5f, 0}; // all identical
        CHECK(build(objs, 2, RTMI_OK) == 0);
        for (uint32_t i = 0; i < n; ++i) objs[i] = rtmi_object{0, {(float)i * 1e6f, 0, 0}, i % 2 ? 1e-6f : 1e5f, 0}; // a line, wild radii
        CHECK(build(objs, 4, RTMI_OK) == 0);
        if (n > 0) {
            objs[n / 2].radius = std::numeric_limits<float>::quiet_NaN();
            CHECK(build(objs, 2, RTMI_ERR_BAD_ARG) == 0);
            objs[n / 2].radius = 1.0f;
            objs[0].center[1] = std::numeric_limits<float>::infinity();
            CHECK(build(objs, 2, RTMI_ERR_BAD_ARG) == 0);
        }
    }
    // round 6: camera entries, walk starts with their way records and the shard plan on the same inputs; the oracle's walk following
    // both tables against its own linear scan
    for (uint32_t n : {0u, 1u, 2u, 3u, 7u, 200u, 3000u}) {
        std::vector<rtmi_object> objs(n);
        std::vector<rtmi_material> mats(std::max(n, 1u));
        for (auto& m : mats) m = rtmi_material{0, {0.5f, 0.6f, 0.7f, 0.0f}};
        for (uint32_t i = 0; i < n; ++i) objs[i] = rtmi_object{0, {u(rng) * 0.2f, std::fabs(u(rng)) * 0.05f, u(rng) * 0.2f}, std::fabs(u(rng)) * 0.01f + 0.05f, 0};
        if (n > 2) objs[0] = rtmi_object{0, {0.0f, -1000.0f, 0.0f}, 1000.0f, 0};
        rtmi_camera_params cp{1.5f, 60, 3, 6, 40.0f, n % 2 ? 0.0f : 1.0f, 10.0f, {13, 2, 3}, {0, 0, 0}, {0, 1, 0}};
        rtmi_camera cam;
        CHECK(rtmi_camera_setup(&cp, &cam) == RTMI_OK);
        const uint32_t gtx = (cam.img_width + 7) / 8, gty = (cam.img_height + 7) / 8;
        for (uint32_t leaf : {1u, 2u, 4u}) {
            std::vector<uint32_t> entries((size_t)gtx * gty + 1, 0xdeadbeefu);
            uint32_t n_tiles = 0;
            CHECK(rtmi_tile_entries_build(&cam, objs.data(), n, leaf, 0, entries.data(), &n_tiles) == RTMI_OK);
            CHECK(n_tiles == gtx * gty && entries[n_tiles] == 0xdeadbeefu);
            std::vector<rtmi_bvh_node> nodes(3 * (size_t)n + 2);
            std::vector<uint32_t> rec(16 * (size_t)std::max(n, 1u) + 1, 0xdeadbeefu), slots(n + 1);
            uint32_t n_all = 0, n_tree = 0;
            CHECK(rtmi_walk_starts_build(objs.data(), n, leaf, 0, nodes.data(), &n_all, &n_tree, rec.data()) == RTMI_OK);
            CHECK(n_all >= n_tree && n_all <= 3 * (size_t)n + 2 && rec[16 * (size_t)std::max(n, 1u)] == 0xdeadbeefu);
            if (n == 0) continue;
            float classes[32], eps = 0, floor_ = 0;
            uint32_t nn = 0, root = 0, depth = 0, n_cls = 0;
            std::vector<rtmi_bvh_node> tree(n + 1);
            CHECK(rtmi_bvh_build(objs.data(), n, leaf, tree.data(), &nn, slots.data(), &root, &depth, classes, &n_cls, &eps, &floor_) == RTMI_OK);
            CHECK(nn == n_tree);
            orc_camera ocam;
            static_assert(sizeof(orc_camera) == sizeof(rtmi_camera), "same 14 PODs");
            std::memcpy(&ocam, &cam, sizeof(ocam));
            const size_t px = (size_t)cam.img_width * cam.img_height;
            std::vector<float> want(px * 3), got(px * 3);
            const orc_object* oo = reinterpret_cast<const orc_object*>(objs.data());
            const orc_material* om = reinterpret_cast<const orc_material*>(mats.data());
            CHECK(orc_render_rect_counter(&ocam, oo, n, om, (uint32_t)mats.size(), 5ull, 0, 0, cam.img_width, cam.img_height, want.data(), nullptr, nullptr, 2) == 0);
            orc_set_tile_entries(entries.data(), gtx);
            orc_set_walk_starts(rec.data(), slots.data(), n, n);
            const int rc = orc_render_rect_counter_bvh(&ocam, oo, n, om, (uint32_t)mats.size(), reinterpret_cast<const orc_bvh_node*>(nodes.data()), n_all,
                                                       slots.data(), n, classes, n_cls, eps, floor_, 5ull, 0, 0, cam.img_width, cam.img_height,
                                                       got.data(), nullptr, nullptr, 2);
            orc_set_tile_entries(nullptr, 0);
            orc_set_walk_starts(nullptr, nullptr, 0, 0);
            CHECK(rc == 0);
            CHECK(std::memcmp(want.data(), got.data(), px * 3 * sizeof(float)) == 0);
        }
    }
    for (uint32_t h : {0u, 1u, 7u, 64u, 1080u}) {
        for (uint32_t ranks : {1u, 3u, 8u}) {
            const uint32_t nb = (h + 7) / 8;
            std::vector<uint64_t> cost(nb + 1);
            for (auto& c : cost) c = (uint64_t)(std::fabs(u(rng)) * 1000.0f);
            std::vector<uint32_t> rank_of(nb + 1, 0xdeadbeefu);
            CHECK(rtmi_shard_plan(h, 8, ranks, nullptr, rank_of.data()) == RTMI_OK);
            for (uint32_t b = 0; b < nb; ++b) CHECK(rank_of[b] == b % ranks);
            CHECK(rtmi_shard_plan(h, 8, ranks, cost.data(), rank_of.data()) == RTMI_OK);
            std::vector<uint32_t> cnt(ranks, 0);
            for (uint32_t b = 0; b < nb; ++b) { CHECK(rank_of[b] < ranks); cnt[rank_of[b]]++; }
            for (uint32_t r = 0; r < ranks; ++r) CHECK(cnt[r] <= (nb + ranks - 1) / ranks);
            CHECK(rank_of[nb] == 0xdeadbeefu);
        }
        CHECK(rtmi_shard_plan(h, 0, 2, nullptr, nullptr) == RTMI_ERR_BAD_ARG);
    }
    // the oracle: one small frame per generator (mt19937 per worker, counter streams), linear scan and BVH walk
    {
        orc_camera_params cp{16.0f / 9.0f, 48, 4, 12, 20.0f, 0.6f, 10.0f, {13, 2, 3}, {0, 0, 0}, {0, 1, 0}};
        orc_camera cam;
        orc_camera_setup(&cp, &cam);
        std::vector<rtmi_object> objs(600);
        std::vector<rtmi_material> mats(600);
        uint32_t n = 0;
        CHECK(rtmi_make_world_spheres(&wd, nullptr, nullptr, 0, 12345u, objs.data(), mats.data(), 600, &n) == RTMI_OK);
        std::vector<float> rgb((size_t)cam.img_width * cam.img_height * 3);
        std::vector<uint32_t> rgba((size_t)cam.img_width * cam.img_height);
        CHECK(orc_render_rect_counter(&cam, reinterpret_cast<const orc_object*>(objs.data()), n,
                                      reinterpret_cast<const orc_material*>(mats.data()), n, 9ull, 0, 0, cam.img_width,
                                      cam.img_height, rgb.data(), rgba.data(), nullptr, 2) == 0);
        double sum = 0;
        for (float v : rgb) sum += v;
        CHECK(sum > 0.0 && std::isfinite(sum));
    }
    std::puts("sanitizers: ok");
    return 0;
}
