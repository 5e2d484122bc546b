// rtmi_frame.hip -- one frame on several GPUs of one node, behind the C-ABI (include/rtmi.h, rtmi_frame_*).
//
// Replaces the worker fan-out of the reference's RayTracer::create (src/main.cc:586-731: one std::thread per core,
// a shuffled queue of 8x8 tiles) and the per-frame drain of RayTracer::update (src/main.cc:733-774: ZeroMQ inproc
// mailboxes, one message per pixel) by: one scene replica per device, the image plane sharded by row blocks (dealt out by the
// scene's cost map, longest processing time first; block b -> device b mod n without one), ONE RCCL gather of the dense per-device slices (float RGB and RGBA8 of a device
// packed into one buffer, so that a frame is one ncclGather per rank) over xGMI to devices[0], and a small kernel that
// restores scanline order.  No exchange happens during rendering: pixels are independent and the
// draw streams are keyed by absolute (pixel, sample), so the frame is bit-identical for any n.
//
// Built on the single-device entry points (rtmi_scene_create / rtmi_render_row_blocks_device); librccl is opened
// at run time, and only when n > 1 (or when RTMI_FRAME_FORCE_RCCL asks for a one-rank communicator): a host that already
// carries an RCCL (torch does) shares its copy, and a single-GPU host never loads the 570 MB library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rtmi_internal.h"

using namespace rtmi;

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// one attempt per process; the result (or the reason it failed) is kept
RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        a.handle = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD); // the copy the host process already has, if any
        for (size_t i = 0; !a.handle && i < sizeof(names) / sizeof(names[0]); ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) {
            const char* e = dlerror();
            a.error = std::string("cannot load librccl: ") + (e ? e : "?");
            return a;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(a.handle, n);
            if (!p && a.error.empty()) a.error = std::string("librccl lacks ") + n;
            return p;
        };
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.Gather = reinterpret_cast<decltype(a.Gather)>(sym("ncclGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        return a;
    }();
    return api;
}

// The gathered buffer is rank-major: rank r's slice = [max_rows * W * 3 floats | max_rows * W RGBA8 words], `slice_words`
// 32-bit words apart; image row y sits at row index[y] % max_rows of rank index[y] / max_rows.
__global__ void __launch_bounds__(256) rtmi_deinterleave_kernel(const uint32_t* __restrict__ gathered, const uint32_t* __restrict__ index,
                                                                uint32_t W, uint32_t H, uint32_t max_rows, size_t slice_words,
                                                                float* __restrict__ rgb, uint32_t* __restrict__ rgba) {
    const uint32_t y = blockIdx.y;
    if (y >= H) return;
    const uint32_t r = index[y] / max_rows, lr = index[y] - r * max_rows;
    const uint32_t* g_rgb = gathered + (size_t)r * slice_words + (size_t)lr * W * 3;
    const uint32_t* g_rgba = gathered + (size_t)r * slice_words + (size_t)max_rows * W * 3 + (size_t)lr * W;
    const size_t dst = (size_t)y * W;
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < W; x += gridDim.x * blockDim.x) {
        rgba[dst + x] = g_rgba[x];
        rgb[3 * (dst + x) + 0] = __uint_as_float(g_rgb[3 * x + 0]);
        rgb[3 * (dst + x) + 1] = __uint_as_float(g_rgb[3 * x + 1]);
        rgb[3 * (dst + x) + 2] = __uint_as_float(g_rgb[3 * x + 2]);
    }
}

struct Shard {
    uint32_t y_first = 0, n_blocks = 0, rows = 0;
    std::vector<uint32_t> blocks; // cost-balanced plan: the rank's row blocks, ascending (empty: the strided set y_first, n_blocks)
};

} // namespace

struct rtmi_frame {
    uint32_t n = 0, W = 0, H = 0, block_rows = 8, max_rows = 0;
    bool rehearsal = false; // test hook: devices may repeat, slices are gathered with copies instead of RCCL
    bool force_rccl = false; // test hook: a communicator and the gather even for n == 1 (rank 0 gathers from itself)
    bool by_cost = false;    // the row blocks were dealt out by the scene's cost map (rtmi_shard_plan) instead of block b -> device b mod n
    size_t slice_words = 0;  // 32-bit words per slice: max_rows * W * (3 + 1)
    std::vector<int> devices;
    std::vector<rtmi_scene*> scenes;
    std::vector<hipStream_t> streams;
    std::vector<uint32_t*> d_slice;       // per device: [max_rows * W * 3 floats | max_rows * W RGBA8]
    std::vector<Shard> shards;
    std::vector<ncclComm_t> comms;        // empty when no communicator exists (n == 1 without RTMI_FRAME_FORCE_RCCL)
    // on devices[0]
    uint32_t* d_gather = nullptr;         // n * slice_words (when a gather happens; otherwise the slice itself)
    bool own_gather = false;
    float* d_rgb_frame = nullptr;         // H * W * 3, scanline order
    uint32_t* d_rgba_frame = nullptr;
    uint32_t* d_index = nullptr;          // H
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;
    rtmi_frame_timing timing{};
};

namespace {

#define HIPF(expr)                                                                           \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
            return e_ == hipErrorOutOfMemory ? RTMI_ERR_OOM : RTMI_ERR_HIP;                  \
        }                                                                                    \
    } while (0)
#define NCCLF(expr)                                                                          \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + rccl_api().GetErrorString(r_));           \
            return RTMI_ERR_RCCL;                                                            \
        }                                                                                    \
    } while (0)

struct PrevDevice {
    int prev = -1;
    PrevDevice() {
        if (hipGetDevice(&prev) != hipSuccess) {
            prev = -1;
            (void)hipGetLastError();
        }
    }
    ~PrevDevice() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

void free_frame(rtmi_frame* f) {
    if (!f) return;
    for (size_t i = 0; i < f->comms.size(); ++i) {
        if (f->comms[i]) rccl_api().CommDestroy(f->comms[i]);
    }
    for (uint32_t i = 0; i < f->devices.size(); ++i) {
        (void)hipSetDevice(f->devices[i]);
        if (i < f->scenes.size() && f->scenes[i]) rtmi_scene_destroy(f->scenes[i]);
        if (i < f->d_slice.size()) (void)hipFree(f->d_slice[i]);
        if (i < f->streams.size() && f->streams[i]) (void)hipStreamDestroy(f->streams[i]);
    }
    if (!f->devices.empty()) {
        (void)hipSetDevice(f->devices[0]);
        if (f->own_gather) (void)hipFree(f->d_gather);
        (void)hipFree(f->d_rgb_frame);
        (void)hipFree(f->d_rgba_frame);
        (void)hipFree(f->d_index);
        if (f->ev_g0) (void)hipEventDestroy(f->ev_g0);
        if (f->ev_g1) (void)hipEventDestroy(f->ev_g1);
    }
    delete f;
}

int frame_create_impl(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects,
                      const rtmi_material* materials, uint32_t n_materials, const rtmi_scene_options* options,
                      const int32_t* devices, uint32_t n, uint32_t block_rows, rtmi_frame*& f) {
    int n_visible = 0;
    HIPF(hipGetDeviceCount(&n_visible));
    rtmi_scene_options opt{};
    opt.device = -1;
    if (options) std::memcpy(&opt, options, std::min<size_t>(sizeof(opt), options->struct_size));
    opt.struct_size = sizeof(opt);
    const bool rehearsal = (opt.reserved[0] & RTMI_FRAME_REHEARSAL) != 0u;
    const bool force_rccl = (opt.reserved[0] & RTMI_FRAME_FORCE_RCCL) != 0u && !rehearsal;
    for (uint32_t i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= n_visible) {
            set_error("rtmi_frame_create: device ordinal outside the visible devices");
            return RTMI_ERR_BAD_ARG;
        }
        for (uint32_t j = 0; j < i && !rehearsal; ++j) {
            if (devices[j] == devices[i]) {
                set_error("rtmi_frame_create: the same device listed twice (RCCL needs one rank per device)");
                return RTMI_ERR_BAD_ARG;
            }
        }
    }
    f = new (std::nothrow) rtmi_frame();
    if (!f) {
        set_error("rtmi_frame_create: out of host memory");
        return RTMI_ERR_OOM;
    }
    f->n = n;
    f->rehearsal = rehearsal;
    f->force_rccl = force_rccl;
    f->W = camera->img_width;
    f->H = camera->img_height;
    f->block_rows = block_rows ? block_rows : 8u;
    f->devices.assign(devices, devices + n);
    f->scenes.assign(n, nullptr);
    f->streams.assign(n, nullptr);
    f->d_slice.assign(n, nullptr);
    f->shards.assign(n, Shard{});

    // one scene replica per device (each makes its own cost probe: the same integers on every device)
    const uint32_t H = f->H, W = f->W, B = f->block_rows;
    for (uint32_t i = 0; i < n; ++i) {
        HIPF(hipSetDevice(devices[i]));
        opt.device = devices[i];
        const int rc = rtmi_scene_create(camera, objects, n_objects, materials, n_materials, &opt, &f->scenes[i]);
        if (rc != RTMI_OK) return rc;
        HIPF(hipStreamCreateWithFlags(&f->streams[i], hipStreamNonBlocking));
    }
    // Row blocks -> devices: block b -> device b mod n, or (RTMI_FRAME_COST_PLAN) by the scene's cost map where it
    // has one and the blocks are whole tile rows: longest processing time first, the same number to every device (rtmi_shard_plan),
    // every device rendering its LIST of blocks into a dense slice of max_rows rows.  The frame is bit-identical for any assignment.
    // Opt-in because it measured nothing: the eighth-frame shards of the 1080p S-RTOW frame are 4 % apart under either plan
    // (slowest 17.88 / 17.86 ms: profiles/r06_shard_perf.txt) -- segment counts balance to 0.1 %, shard times do not follow them.
    const uint32_t n_blocks_total = (H + B - 1) / B;
    std::vector<uint32_t> rank_of_block(std::max(1u, n_blocks_total), 0u);
    bool by_cost = false;
    if (n > 1 && (B & 7u) == 0u && n_blocks_total >= n && (opt.reserved[0] & RTMI_FRAME_COST_PLAN) != 0u) {
        uint32_t n_tiles = 0;
        if (rtmi_scene_get_tile_costs(f->scenes[0], nullptr, &n_tiles) == RTMI_OK && n_tiles != 0u) {
            std::vector<uint32_t> tc(n_tiles);
            std::vector<uint64_t> bc(n_blocks_total, 0u);
            const uint32_t gtx = (W + 7u) / 8u;
            if (rtmi_scene_get_tile_costs(f->scenes[0], tc.data(), &n_tiles) == RTMI_OK && gtx != 0u) {
                for (uint32_t t = 0; t < n_tiles; ++t) bc[std::min(n_blocks_total - 1u, ((t / gtx) * 8u) / B)] += tc[t];
                by_cost = rtmi_shard_plan(H, B, n, bc.data(), rank_of_block.data()) == RTMI_OK;
            }
        }
    }
    if (!by_cost) {
        const int rc = rtmi_shard_plan(H, B, n, nullptr, rank_of_block.data());
        if (rc != RTMI_OK) return rc;
    }
    f->by_cost = by_cost;
    std::vector<uint32_t> index(H, 0u);
    for (uint32_t r = 0; r < n; ++r) {
        Shard& sh = f->shards[r];
        sh.y_first = r * B;
        for (uint32_t b = 0; b < n_blocks_total; ++b) {
            if (rank_of_block[b] != r) continue;
            if (by_cost) sh.blocks.push_back(b);
            sh.n_blocks++;
            sh.rows += std::min(B, H - b * B);
        }
        f->max_rows = std::max(f->max_rows, sh.rows);
    }
    {
        std::vector<uint32_t> k_of_rank(n, 0u);
        for (uint32_t b = 0; b < n_blocks_total; ++b) {
            const uint32_t r = rank_of_block[b], k = k_of_rank[r]++;
            for (uint32_t y = b * B; y < std::min(H, (b + 1u) * B); ++y) index[y] = r * f->max_rows + k * B + (y - b * B);
        }
    }

    f->max_rows = std::max(1u, f->max_rows);
    const size_t slice_px = (size_t)f->max_rows * std::max(1u, W);
    f->slice_words = slice_px * 4;
    for (uint32_t i = 0; i < n; ++i) {
        HIPF(hipSetDevice(devices[i]));
        HIPF(hipMalloc(reinterpret_cast<void**>(&f->d_slice[i]), f->slice_words * sizeof(uint32_t)));
        HIPF(hipMemset(f->d_slice[i], 0, f->slice_words * sizeof(uint32_t)));
    }
    HIPF(hipSetDevice(devices[0]));
    const size_t frame_px = (size_t)std::max(1u, H) * std::max(1u, W);
    HIPF(hipMalloc(reinterpret_cast<void**>(&f->d_rgb_frame), frame_px * 3 * sizeof(float)));
    HIPF(hipMalloc(reinterpret_cast<void**>(&f->d_rgba_frame), frame_px * sizeof(uint32_t)));
    HIPF(hipMalloc(reinterpret_cast<void**>(&f->d_index), std::max<size_t>(1, H) * sizeof(uint32_t)));
    if (H) HIPF(hipMemcpy(f->d_index, index.data(), H * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPF(hipEventCreate(&f->ev_g0));
    HIPF(hipEventCreate(&f->ev_g1));
    if (n > 1 || force_rccl) {
        HIPF(hipMalloc(reinterpret_cast<void**>(&f->d_gather), f->slice_words * n * sizeof(uint32_t)));
        f->own_gather = true;
    } else {
        f->d_gather = f->d_slice[0];
    }
    if ((n > 1 && !rehearsal) || force_rccl) {
        RcclApi& api = rccl_api();
        if (!api.error.empty() || !api.Gather) {
            set_error("rtmi_frame_create: " + (api.error.empty() ? std::string("librccl unusable") : api.error));
            return RTMI_ERR_RCCL;
        }
        f->comms.assign(n, nullptr);
        NCCLF(api.CommInitAll(f->comms.data(), (int)n, f->devices.data()));
    }
    return RTMI_OK;
}

int frame_render_impl(rtmi_frame* f, uint64_t seed) {
    const uint32_t n = f->n, W = f->W, H = f->H;
    const auto t0 = std::chrono::steady_clock::now();
    if (W == 0 || H == 0) return RTMI_OK;
    for (uint32_t i = 0; i < n; ++i) {
        const Shard& sh = f->shards[i];
        if (!sh.n_blocks) continue;
        const int rc = sh.blocks.empty()
                           ? rtmi_render_row_blocks_device(f->scenes[i], sh.y_first, f->block_rows, n, sh.n_blocks, seed, f->d_slice[i],
                                                           f->d_slice[i] + (size_t)f->max_rows * W * 3, f->streams[i])
                           : rtmi_render_block_list_device(f->scenes[i], f->block_rows, sh.blocks.data(), sh.n_blocks, seed, f->d_slice[i],
                                                           f->d_slice[i] + (size_t)f->max_rows * W * 3, f->streams[i]);
        if (rc != RTMI_OK) return rc;
    }
    HIPF(hipSetDevice(f->devices[0]));
    HIPF(hipEventRecord(f->ev_g0, f->streams[0]));
    if (n > 1 && f->rehearsal) {
        // test hook: the same plan with plain copies, so that one box can check the shard and scanline-order arithmetic
        for (uint32_t i = 0; i < n; ++i) {
            HIPF(hipSetDevice(f->devices[i]));
            HIPF(hipStreamSynchronize(f->streams[i]));
            HIPF(hipMemcpyAsync(f->d_gather + (size_t)i * f->slice_words, f->d_slice[i], f->slice_words * sizeof(uint32_t),
                                hipMemcpyDeviceToDevice, f->streams[0]));
        }
        HIPF(hipSetDevice(f->devices[0]));
    } else if (!f->comms.empty()) {
        // ONE gather of the dense slices to devices[0]: every rank's call sits in one group, on its own stream, behind its
        // own kernels (a slice is float RGB followed by RGBA8, moved as 32-bit words)
        RcclApi& api = rccl_api();
        NCCLF(api.GroupStart());
        for (uint32_t i = 0; i < n; ++i) {
            NCCLF(api.Gather(f->d_slice[i], f->d_gather, f->slice_words, ncclUint32, 0, f->comms[i], f->streams[i]));
        }
        NCCLF(api.GroupEnd());
        HIPF(hipSetDevice(f->devices[0]));
    }
    rtmi_deinterleave_kernel<<<dim3((W + 255u) / 256u, H), dim3(256), 0, f->streams[0]>>>(
        f->d_gather, f->d_index, W, H, f->max_rows, f->slice_words, f->d_rgb_frame, f->d_rgba_frame);
    HIPF(hipGetLastError());
    HIPF(hipEventRecord(f->ev_g1, f->streams[0]));
    for (uint32_t i = 0; i < n; ++i) {
        HIPF(hipSetDevice(f->devices[i]));
        HIPF(hipStreamSynchronize(f->streams[i]));
    }
    f->timing.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIPF(hipSetDevice(f->devices[0]));
    HIPF(hipEventElapsedTime(&f->timing.gather_ms, f->ev_g0, f->ev_g1));
    for (uint32_t i = 0; i < n && i < 16u; ++i) {
        f->timing.kernel_ms[i] = 0.0f;
        if (f->shards[i].n_blocks) {
            const int rc = rtmi_scene_last_kernel_ms(f->scenes[i], &f->timing.kernel_ms[i]);
            if (rc != RTMI_OK) return rc;
        }
    }
    return RTMI_OK;
}

template <class F>
int guarded(const char* where, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { set_error(std::string(where) + ": out of host memory"); } catch (...) {}
        return RTMI_ERR_OOM;
    } catch (...) {
        try { set_error(std::string(where) + ": unexpected exception"); } catch (...) {}
        return RTMI_ERR_INTERNAL;
    }
}

} // namespace

extern "C" int rtmi_frame_create(const rtmi_camera* camera, const rtmi_object* objects, uint32_t n_objects,
                                 const rtmi_material* materials, uint32_t n_materials, const rtmi_scene_options* options,
                                 const int32_t* devices, uint32_t n_devices, uint32_t block_rows, rtmi_frame** out) {
    if (!camera || !out || !devices || n_devices == 0 || n_devices > 16 || (n_objects && !objects) ||
        (n_materials && !materials)) {
        try { set_error("rtmi_frame_create: null argument or device count outside 1..16"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    *out = nullptr;
    PrevDevice guard;
    rtmi_frame* f = nullptr;
    const int rc = guarded("rtmi_frame_create", [&] {
        return frame_create_impl(camera, objects, n_objects, materials, n_materials, options, devices, n_devices,
                                 block_rows, f);
    });
    if (rc != RTMI_OK) {
        // keep the message of the failure, not of the clean-up
        std::string msg;
        try { msg = rtmi_last_error(); } catch (...) {}
        free_frame(f);
        try { set_error(msg); } catch (...) {}
        return rc;
    }
    *out = f;
    return RTMI_OK;
}

extern "C" void rtmi_frame_destroy(rtmi_frame* frame) {
    PrevDevice guard;
    free_frame(frame);
}

extern "C" int rtmi_frame_render_device(rtmi_frame* f, uint64_t seed, void** d_rgb_linear, void** d_rgba8) {
    if (!f) {
        try { set_error("rtmi_frame_render_device: null frame"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    PrevDevice guard;
    const int rc = guarded("rtmi_frame_render_device", [&] { return frame_render_impl(f, seed); });
    if (rc != RTMI_OK) return rc;
    if (d_rgb_linear) *d_rgb_linear = f->d_rgb_frame;
    if (d_rgba8) *d_rgba8 = f->d_rgba_frame;
    return RTMI_OK;
}

extern "C" int rtmi_frame_render(rtmi_frame* f, uint64_t seed, float* rgb_linear_out, uint32_t* rgba8_out) {
    if (!f) {
        try { set_error("rtmi_frame_render: null frame"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    PrevDevice guard;
    return guarded("rtmi_frame_render", [&]() -> int {
        const int rc = frame_render_impl(f, seed);
        if (rc != RTMI_OK) return rc;
        const size_t px = (size_t)f->W * f->H;
        if (!px) return RTMI_OK;
        HIPF(hipSetDevice(f->devices[0]));
        if (rgb_linear_out) HIPF(hipMemcpy(rgb_linear_out, f->d_rgb_frame, px * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (rgba8_out) HIPF(hipMemcpy(rgba8_out, f->d_rgba_frame, px * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return RTMI_OK;
    });
}

extern "C" int rtmi_frame_get_timing(const rtmi_frame* f, rtmi_frame_timing* out) {
    if (!f || !out) {
        try { set_error("rtmi_frame_get_timing: null argument"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    *out = f->timing;
    return RTMI_OK;
}

extern "C" int rtmi_frame_get_scene(rtmi_frame* f, uint32_t index, rtmi_scene** scene_out) {
    if (!f || !scene_out || index >= f->scenes.size()) {
        try { set_error("rtmi_frame_get_scene: null argument or index outside the device list"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    *scene_out = f->scenes[index];
    return RTMI_OK;
}

extern "C" int rtmi_frame_rccl_ranks(const rtmi_frame* f, uint32_t* n_out) {
    if (!f || !n_out) {
        try { set_error("rtmi_frame_rccl_ranks: null argument"); } catch (...) {}
        return RTMI_ERR_BAD_ARG;
    }
    *n_out = (uint32_t)f->comms.size();
    return RTMI_OK;
}
