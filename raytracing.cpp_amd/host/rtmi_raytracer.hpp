// rtmi_raytracer.hpp -- the reference's job-system seam and display contract on top of the GPU render core
// ("next" rows of SURVEY 8f: progressive streaming into the display contract, host adapter for the job system,
// image file output).
//
//   RaytracedPixel            the per-pixel message of the reference (src/main.cc:252-256)
//   RayTracedImageSSBOData    the mapped buffer the fragment shader reads (src/ray.tracer.image.display.hpp:10-14)
//   RayTracedImageTarget      write_pixel() with the reference's centring + y-flip (src/ray.tracer.image.display.cc:108-117)
//                             over a host-side buffer of that layout (the GL object itself stays in the reference)
//   RayTracer                 same public surface as the reference's class (src/main.cc:526-585): create / update /
//                             shutdown / pixels_count / pixels_raytraced / image_size / render_time.  One GPU worker
//                             thread per attached device (RayTracingCore::attach_devices; one worker on the core's own
//                             scene otherwise) replaces the N CPU workers: they pop shuffled ROW BLOCKS instead of 8x8 tiles
//                             (main.cc:615-633), renders each with RayTracingCore::raytrace_rows and posts the finished
//                             block; update() drains a bounded number of blocks per frame into write_pixel, exactly where
//                             the reference drains its ZeroMQ inproc mailboxes (main.cc:733-774).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <optional>
#include <random>
#include <thread>

#include "rtmi_host.hpp"

namespace rtmi {

struct RaytracedPixel { // src/main.cc:252-256
    uint32_t rtp_x;
    uint32_t rtp_y;
    uint32_t rtp_color;
};

struct alignas(16) RayTracedImageSSBOData { // src/ray.tracer.image.display.hpp:10-14
    uint32_t rti_width;
    uint32_t rti_height;
    RGBAColor rti_pixels[1];
};

// Host-side stand-in for the persistently mapped SSBO of RayTracedImageDisplay.
class RayTracedImageTarget {
public:
    RayTracedImageTarget(uint32_t surface_w, uint32_t surface_h, uint32_t img_w, uint32_t img_h)
        : _surface_w{surface_w}, _surface_h{surface_h}, _img_w{img_w}, _img_h{img_h},
          _storage(sizeof(RayTracedImageSSBOData) + size_t(surface_w) * surface_h * sizeof(RGBAColor), 0) {
        ssbo()->rti_width = surface_w;
        ssbo()->rti_height = surface_h;
    }
    RayTracedImageSSBOData* ssbo() noexcept { return reinterpret_cast<RayTracedImageSSBOData*>(_storage.data()); }
    const RayTracedImageSSBOData* ssbo() const noexcept {
        return reinterpret_cast<const RayTracedImageSSBOData*>(_storage.data());
    }
    // RayTracedImageDisplay::write_pixel, image.display.cc:108-117
    void write_pixel(const uint32_t x, const uint32_t y, const RGBAColor color) {
        const uint32_t tx = (_surface_w - _img_w) / 2u, ty = (_surface_h - _img_h) / 2u; // centre the image
        const uint32_t px = x + tx, py = y + ty;
        ssbo()->rti_pixels[size_t(_surface_h - 1 - py) * _surface_w + px] = color; // GL view coords: lower-left origin
    }
    uint32_t surface_width() const noexcept { return _surface_w; }
    uint32_t surface_height() const noexcept { return _surface_h; }

private:
    uint32_t _surface_w, _surface_h, _img_w, _img_h;
    std::vector<unsigned char> _storage;
};

// Binary PPM (P6) of an RGBAColor image in scanline order: the reference keeps its image only in the SSBO
// (stb_image_write is vendored but unused), so this is the build's own regression-image writer.
inline bool write_ppm(const std::string& path, uint32_t w, uint32_t h, const RGBAColor* pixels) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    std::vector<unsigned char> row(size_t(w) * 3);
    for (uint32_t y = 0; y < h; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            const RGBAColor c = pixels[size_t(y) * w + x];
            row[3 * x + 0] = c.r;
            row[3 * x + 1] = c.g;
            row[3 * x + 2] = c.b;
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return true;
}

class RayTracer {
public:
    struct RowBlock { // what the GPU worker posts instead of one message per pixel
        uint32_t y0, y1;
        std::vector<RGBAColor> pixels; // (y1 - y0) * width, scanline order
    };

    // RayTracer::create, main.cc:586-731.  `block_rows` rows per work package; `max_blocks_per_update` mirrors the
    // reference's "at most 64 messages per signalled worker per frame" (main.cc:752).
    static std::optional<RayTracer> create(std::shared_ptr<RayTracingCore> core, uint64_t frame_seed,
                                           uint32_t block_rows = 8, uint32_t max_blocks_per_update = 4) {
        if (!core || !core->rts_gpu_scene || block_rows == 0) return std::nullopt;
        return std::optional<RayTracer>{std::in_place, PrivateConstructionToken{}, std::move(core), frame_seed, block_rows,
                                        max_blocks_per_update};
    }

    struct PrivateConstructionToken {};
    RayTracer(PrivateConstructionToken, std::shared_ptr<RayTracingCore> core, uint64_t frame_seed, uint32_t block_rows,
              uint32_t max_blocks_per_update)
        : _core{std::move(core)}, _state{std::make_unique<Shared>()}, _max_blocks{max_blocks_per_update} {
        const uint32_t h = _core->rts_img_height;
        std::vector<std::pair<uint32_t, uint32_t>> blocks;
        for (uint32_t y = 0; y < h; y += block_rows) blocks.emplace_back(y, std::min(h, y + block_rows));
        std::mt19937 shuffler{static_cast<uint32_t>(frame_seed)}; // std::shuffle of the work packages, main.cc:633
        std::shuffle(blocks.begin(), blocks.end(), shuffler);
        Shared* st = _state.get();
        std::shared_ptr<RayTracingCore> c = _core;
        st->blocks = std::move(blocks);
        // one worker per attached GPU, all pulling from the same shuffled queue (MonkaGigaQueue::pop_pkg, main.cc:413-421)
        const size_t n_workers = std::max<size_t>(1, c->rts_gpu_replicas.size());
        st->live_workers.store(static_cast<uint32_t>(n_workers));
        for (size_t w = 0; w < n_workers; ++w) {
            _workers.emplace_back([st, c, w, frame_seed]() {
                for (;;) { // RayTracingWorker::worker_loop, main.cc:443-505
                    if (st->quit.load()) break; // ThreadQuitMessage, main.cc:776-782
                    const size_t i = st->next_block.fetch_add(1);
                    if (i >= st->blocks.size()) break;
                    const auto [y0, y1] = st->blocks[i];
                    RowBlock rb{y0, y1, std::vector<RGBAColor>(size_t(y1 - y0) * c->rts_img_width)};
                    const int rc = c->rts_gpu_replicas.empty() ? c->raytrace_rows(y0, y1, frame_seed, rb.pixels.data())
                                                               : c->raytrace_rows_on(w, y0, y1, frame_seed, rb.pixels.data());
                    if (rc != RTMI_OK) {
                        st->failed.store(true); // setup/launch failures end the worker, as in main.cc:685-706
                        break;
                    }
                    st->pixels_processed += (y1 - y0) * c->rts_img_width; // g_pixels_processed, main.cc:516
                    std::lock_guard<std::mutex> lock(st->mu);
                    st->mailbox.push_back(std::move(rb));
                }
                if (st->live_workers.fetch_sub(1) == 1) st->done.store(true);
            });
        }
    }
    RayTracer(RayTracer&&) = default;
    RayTracer(const RayTracer&) = delete;
    ~RayTracer() {
        if (_state) _state->quit.store(true);
        for (auto& w : _workers) {
            if (w.joinable()) w.join();
        }
    }
    size_t worker_count() const noexcept { return _workers.size(); }

    // RayTracer::update, main.cc:733-774: drain what the worker has posted (bounded per call) into the display.
    template <typename Display>
    void update(Display* img_output) {
        bool any = false;
        for (uint32_t n = 0; n < _max_blocks; ++n) {
            RowBlock rb;
            {
                std::lock_guard<std::mutex> lock(_state->mu);
                if (_state->mailbox.empty()) break;
                rb = std::move(_state->mailbox.front());
                _state->mailbox.pop_front();
            }
            const uint32_t w = _core->rts_img_width;
            for (uint32_t y = rb.y0; y < rb.y1; ++y)
                for (uint32_t x = 0; x < w; ++x) {
                    img_output->write_pixel(x, y, rb.pixels[size_t(y - rb.y0) * w + x]);
                    _pixels_raytraced += 1;
                }
            any = true;
        }
        if (any && _state->pixels_processed.load() <= pixels_count()) {
            _end_timepoint = std::chrono::high_resolution_clock::now(); // freezes at the last received pixel
        }
    }
    void shutdown() { _state->quit.store(true); }
    uint32_t pixels_count() const noexcept { return _core->rts_img_width * _core->rts_img_height; }
    uint32_t pixels_raytraced() const noexcept { return _pixels_raytraced; }
    std::pair<uint16_t, uint16_t> image_size() const noexcept {
        return {static_cast<uint16_t>(_core->rts_img_width), static_cast<uint16_t>(_core->rts_img_height)};
    }
    std::chrono::duration<double> render_time() const noexcept { return _end_timepoint - _start_timepoint; }
    bool worker_failed() const noexcept { return _state->failed.load(); }

private:
    struct Shared {
        std::mutex mu;
        std::deque<RowBlock> mailbox; // stands in for the ZMQ_CHANNEL inproc pair (main.cc:642-712)
        std::atomic<bool> quit{false}, done{false}, failed{false};
        std::atomic<uint32_t> pixels_processed{0};
        std::vector<std::pair<uint32_t, uint32_t>> blocks; // the shuffled work packages, read-only once the workers run
        std::atomic<size_t> next_block{0};
        std::atomic<uint32_t> live_workers{0};
    };
    std::shared_ptr<RayTracingCore> _core;
    std::unique_ptr<Shared> _state;
    std::vector<std::thread> _workers;
    uint32_t _max_blocks;
    uint32_t _pixels_raytraced{};
    std::chrono::time_point<std::chrono::high_resolution_clock> _start_timepoint{
        std::chrono::high_resolution_clock::now()};
    std::chrono::time_point<std::chrono::high_resolution_clock> _end_timepoint{
        std::chrono::high_resolution_clock::now()};
};

} // namespace rtmi
