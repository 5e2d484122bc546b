// Exercises the C++ host mirror (raytracing.cpp_amd/host/rtmi_host.hpp) the way the reference's own host would:
//   host_mirror_check setup  <world.config.json> <seed>            -> dumps camera + scene records (no GPU needed)
//   host_mirror_check render <world.config.json> <seed> <rseed> <out.bin>  -> default_setup + raytrace_rows on the GPU
#include <cstdio>
#include <cstring>
#include <string>

#include "rtmi_host.hpp"

using namespace rtmi;

static void dump(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) std::printf("%02x", b[i]);
    std::printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string mode = argv[1];
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
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 9;
    }
    return 2;
}
