// Exercises the C++ host mirror (raytracing.cpp_amd/host/rtmi_host.hpp) the way the reference's own host would:
//   host_mirror_check setup  <world.config.json> <seed>            -> dumps camera + scene records (no GPU needed)
//   host_mirror_check render <world.config.json> <seed> <rseed> <out.bin>  -> default_setup + raytrace_rows on the GPU
//   host_mirror_check display <surface_w> <surface_h> <img_w> <img_h> <out.bin>  -> write_pixel mapping (no GPU needed)
//   host_mirror_check stream <world.config.json> <seed> <rseed> <surface_w> <surface_h> <out.bin> <out.ppm>
//                            -> RayTracer (job-system adapter) streaming row blocks into the display contract on the GPU
//   host_mirror_check frame  <world.config.json> <seed> <rseed> <out.bin> [n_devices]
//                            -> attach_devices({0..n-1}) + raytrace_frame (rtmi_frame_*: shards + RCCL gather) and the
//                               multi-worker RayTracer; must equal raytrace_rows of the single scene
#include <cstdio>
#include <cstring>
#include <string>

#include "rtmi_host.hpp"
#include "rtmi_raytracer.hpp"

using namespace rtmi;

static void dump(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) std::printf("%02x", b[i]);
    std::printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    if (mode == "display" && argc >= 7) {
        const uint32_t sw = std::atoi(argv[2]), sh = std::atoi(argv[3]), iw = std::atoi(argv[4]), ih = std::atoi(argv[5]);
        RayTracedImageTarget target(sw, sh, iw, ih);
        for (uint32_t y = 0; y < ih; ++y)
            for (uint32_t x = 0; x < iw; ++x) target.write_pixel(x, y, RGBAColor{0xff000000u | (y << 12) | x});
        FILE* f = std::fopen(argv[6], "wb");
        if (!f) return 8;
        std::fwrite(target.ssbo(), 1, sizeof(RayTracedImageSSBOData) - sizeof(RGBAColor) + size_t(sw) * sh * sizeof(RGBAColor), f);
        std::fclose(f);
        return 0;
    }
    try {
        const WorldDefinition wd = load_world_definition(argv[2]);
        const uint32_t seed = static_cast<uint32_t>(std::strtoul(argv[3], nullptr, 10));
        if (mode == "setup") {
            auto [cam_params, world, mats] = make_world_spheres(wd, seed);
            rtmi_camera_params cp;
            std::memcpy(&cp, &cam_params, sizeof(cp));
            rtmi_camera cam;
            if (rtmi_camera_setup(&cp, &cam) != RTMI_OK) return 3;
            std::printf("camera ");
            dump(&cam, sizeof(cam));
            std::printf("objects %zu ", world.size());
            dump(world.data(), world.size() * sizeof(HittableObject));
            std::printf("materials %zu ", mats.size());
            dump(mats.data(), mats.size() * sizeof(Material));
            // handle-indexed access, as the reference's MaterialCollection::operator[]
            std::printf("mat3kind %u\n", static_cast<unsigned>(mats[MaterialHandleType{3}].MatKind));
            return 0;
        }
        if (mode == "render" && argc >= 6) {
            auto core = RayTracingCore::setup(wd, seed);
            const uint64_t rseed = std::strtoull(argv[4], nullptr, 10);
            const size_t n = size_t(core->rts_img_width) * core->rts_img_height;
            std::vector<RGBAColor> rgba(n);
            std::vector<float> rgb(n * 3);
            if (core->raytrace_rows(0, core->rts_img_height, rseed, rgba.data(), rgb.data()) != RTMI_OK) {
                std::fprintf(stderr, "raytrace_rows: %s\n", rtmi_last_error());
                return 4;
            }
            // one 8x8 work package, as RayTracingWorker::process_tracing_work_package would ask for
            std::vector<RGBAColor> tile(64);
            if (core->raytrace_tile(8, 16, 16, 24, rseed, tile.data()) != RTMI_OK) return 5;
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x)
                    if (tile[y * 8 + x].color != rgba[size_t(16 + y) * core->rts_img_width + 8 + x].color) return 6;
            // error behaviour: status codes, no exceptions
            if (core->raytrace_rows(0, core->rts_img_height + 1, rseed, rgba.data()) != RTMI_ERR_BAD_ARG) return 7;
            FILE* f = std::fopen(argv[5], "wb");
            if (!f) return 8;
            const uint32_t hdr[2] = {core->rts_img_width, core->rts_img_height};
            std::fwrite(hdr, sizeof(hdr), 1, f);
            std::fwrite(rgb.data(), sizeof(float), rgb.size(), f);
            std::fwrite(rgba.data(), sizeof(RGBAColor), rgba.size(), f);
            std::fclose(f);
            return 0;
        }
        if (mode == "frame" && argc >= 6) {
            auto core = RayTracingCore::setup(wd, seed);
            const uint64_t rseed = std::strtoull(argv[4], nullptr, 10);
            // n_devices > 0: devices 0..n-1 (RCCL gather for n > 1); n_devices < 0: device 0 listed |n| times through the
            // rehearsal hook (copies instead of RCCL), what a one-GPU box can check of the n > 1 plumbing
            const int n_arg = argc >= 7 ? std::atoi(argv[6]) : 1;
            const int n_dev = n_arg < 0 ? -n_arg : n_arg;
            std::vector<int32_t> devices;
            for (int d = 0; d < n_dev; ++d) devices.push_back(n_arg < 0 ? 0 : d);
            rtmi_scene_options fopt{};
            fopt.struct_size = sizeof(fopt);
            fopt.device = -1;
            if (n_arg < 0) fopt.reserved[0] = RTMI_FRAME_REHEARSAL;
            const size_t n = size_t(core->rts_img_width) * core->rts_img_height;
            std::vector<RGBAColor> want(n), got(n);
            std::vector<float> want_rgb(n * 3), got_rgb(n * 3);
            if (core->raytrace_rows(0, core->rts_img_height, rseed, want.data(), want_rgb.data()) != RTMI_OK) return 4;
            if (core->attach_devices(devices, 8, &fopt) != RTMI_OK) {
                std::fprintf(stderr, "attach_devices: %s\n", rtmi_last_error());
                return 5;
            }
            if (core->raytrace_frame(rseed, got.data(), got_rgb.data()) != RTMI_OK) {
                std::fprintf(stderr, "raytrace_frame: %s\n", rtmi_last_error());
                return 6;
            }
            if (std::memcmp(want.data(), got.data(), n * sizeof(RGBAColor)) != 0) return 7;
            if (std::memcmp(want_rgb.data(), got_rgb.data(), n * 3 * sizeof(float)) != 0) return 7;
            // the job-system adapter with one worker per attached device
            RayTracedImageTarget target(core->rts_img_width, core->rts_img_height, core->rts_img_width, core->rts_img_height);
            auto tracer = RayTracer::create(core, rseed, 8, 64);
            if (!tracer || tracer->worker_count() != devices.size()) return 8;
            const auto t0 = std::chrono::steady_clock::now();
            while (tracer->pixels_raytraced() < tracer->pixels_count()) {
                tracer->update(&target);
                if (tracer->worker_failed()) return 9;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 10;
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            for (uint32_t y = 0; y < core->rts_img_height; ++y)
                for (uint32_t x = 0; x < core->rts_img_width; ++x)
                    if (target.ssbo()->rti_pixels[size_t(core->rts_img_height - 1 - y) * core->rts_img_width + x].color !=
                        want[size_t(y) * core->rts_img_width + x].color)
                        return 11;
            FILE* f = std::fopen(argv[5], "wb");
            if (!f) return 8;
            std::fwrite(got.data(), sizeof(RGBAColor), got.size(), f);
            std::fclose(f);
            std::printf("frame ok devices %d\n", n_dev);
            return 0;
        }
        if (mode == "stream" && argc >= 9) {
            auto core = RayTracingCore::setup(wd, seed);
            const uint64_t rseed = std::strtoull(argv[4], nullptr, 10);
            const uint32_t sw = std::atoi(argv[5]), sh = std::atoi(argv[6]);
            RayTracedImageTarget target(sw, sh, core->rts_img_width, core->rts_img_height);
            auto tracer = RayTracer::create(core, rseed, 8, 2);
            if (!tracer) return 3;
            uint32_t last = 0, progress_steps = 0, frames = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (tracer->pixels_raytraced() < tracer->pixels_count()) { // the render_event delegate, main.cc:880
                tracer->update(&target);
                if (tracer->pixels_raytraced() != last) {
                    ++progress_steps; // the UI's progress bar would move here (main.cc:378-388)
                    last = tracer->pixels_raytraced();
                }
                ++frames;
                if (tracer->worker_failed()) return 4;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 5;
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            tracer->shutdown();
            if (progress_steps < 3) return 6; // the image must arrive progressively, not in one piece
            if (!(tracer->render_time().count() > 0.0)) return 7;
            std::vector<RGBAColor> frame(size_t(core->rts_img_width) * core->rts_img_height);
            if (core->raytrace_rows(0, core->rts_img_height, rseed, frame.data()) != RTMI_OK) return 4;
            if (!write_ppm(argv[8], core->rts_img_width, core->rts_img_height, frame.data())) return 8;
            FILE* f = std::fopen(argv[7], "wb");
            if (!f) return 8;
            std::fwrite(target.ssbo(), 1, sizeof(RayTracedImageSSBOData) - sizeof(RGBAColor) + size_t(sw) * sh * sizeof(RGBAColor), f);
            std::fclose(f);
            std::printf("frames %u progress_steps %u render_time %.4f\n", frames, progress_steps, tracer->render_time().count());
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 9;
    }
    return 2;
}
