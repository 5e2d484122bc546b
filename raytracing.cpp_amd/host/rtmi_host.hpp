// rtmi_host.hpp -- C++ host mirror of the reference's render-core interface, on top of the C-ABI (include/rtmi.h).
//
// Same names, argument meaning and error behaviour as the reference's types on this path, so that the reference's
// host (job system, GL/Nuklear front end, scene loader) can drive the GPU renderer as a drop-in:
//   CameraParameters            src/camera.parameters.hpp:6-17
//   MaterialHandleType          src/ray.tracer.material.handle.hpp:6
//   HittableObject(_Sphere), HittableObject_Collection   src/ray.tracer.object.defs.hpp:30-67
//   Material(_Lambertian/_Metallic/_Dielectric), MaterialCollection   src/ray.tracer.material.defs.hpp:27-110
//   RGBAColor                   src/color.hpp:15-37
//   WorldDefinition, make_world_spheres, RayTracingCore::default_setup   src/ray.tracer.core.cc:47-216
//   RayTracingCore (14 PODs + rts_world + rts_materials)   src/ray.tracer.core.hpp:18-42
// What changes: the collections expose data()/size() (the reference keeps _objects/_materials private with no
// accessor, object.defs.hpp:65-66, material.defs.hpp:108-109), and the per-pixel `raytrace_pixel(x, y, rng)` seam
// becomes `raytrace_rows(y0, y1, seed, ...)` / `raytrace_tile(...)`: a per-pixel GPU call is meaningless.
// No exceptions cross this header except std::runtime_error from the explicitly throwing helpers (`.value()`-style,
// as core.cc:102 does for a bad JSON file).
//
// Header-only; link with librtmi.so.  vec3 is a plain struct here; inside the reference tree define RTMI_VEC3 to
// glm::vec3 before including (same layout, three floats).
#pragma once

#include <array>
#include <cassert>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "rtmi.h"

namespace rtmi {

#ifdef RTMI_VEC3
using vec3 = RTMI_VEC3;
#else
struct vec3 {
    float x, y, z;
};
#endif
static_assert(sizeof(vec3) == 12, "vec3 must be three packed floats");

// ---- src/camera.parameters.hpp:6-17 -----------------------------------------------------------------------------
struct CameraParameters {
    float aspect_ratio;
    uint32_t image_width;
    uint16_t samples_per_pixel;
    uint16_t max_depth;
    float vertical_fov;
    float defocus_angle;
    float focus_distance;
    std::array<float, 3> lookfrom;
    std::array<float, 3> lookat;
    std::array<float, 3> world_up;
};
static_assert(sizeof(CameraParameters) == sizeof(rtmi_camera_params), "CameraParameters layout");

// ---- src/ray.tracer.material.handle.hpp:6 (strong::type<uint32_t, ...>) -------------------------------------------
struct MaterialHandleType {
    uint32_t v;
    MaterialHandleType() = default; // trivial, so that it can sit inside HittableObject's union
    explicit MaterialHandleType(uint32_t x) : v{x} {}
    friend uint32_t value_of(MaterialHandleType h) noexcept { return h.v; }
};

// ---- src/color.hpp:15-37 --------------------------------------------------------------------------------------------
struct RGBAColor {
    union {
        struct {
            uint8_t r, g, b, a;
        };
        uint32_t color;
    };
    RGBAColor() noexcept = default;
    explicit RGBAColor(const uint32_t c) noexcept { color = c; }
};

// ---- src/ray.tracer.object.defs.hpp:25-67 ---------------------------------------------------------------------------
enum class HittableObjectKind : uint32_t { Sphere, Count };

struct HittableObject_Sphere {
    vec3 Center;
    float Radius;
    MaterialHandleType Material;
};

struct HittableObject {
    HittableObjectKind ObjKind;
    union {
        HittableObject_Sphere Sphere;
    };
    static HittableObject make_sphere(const vec3& center, const float radius, MaterialHandleType mtl) noexcept {
        HittableObject o;
        o.ObjKind = HittableObjectKind::Sphere;
        o.Sphere = HittableObject_Sphere{center, radius, mtl};
        return o;
    }
};
static_assert(sizeof(HittableObject) == sizeof(rtmi_object), "HittableObject is the 24-byte record of the C-ABI");

class HittableObject_Collection {
public:
    void add_object(const HittableObject& obj) { _objects.push_back(obj); }
    void clear() { _objects.clear(); }
    // added accessors (the reference has none): contiguous 24-byte records for rtmi_scene_create
    const HittableObject* data() const noexcept { return _objects.data(); }
    size_t size() const noexcept { return _objects.size(); }

private:
    std::vector<HittableObject> _objects;
};

// ---- src/ray.tracer.material.defs.hpp:20-110 ------------------------------------------------------------------------
enum class MaterialKind : uint32_t { Lambertian, Metallic, Dielectric, Count };

struct Material_Lambertian {
    vec3 Albedo;
};
struct Material_Metallic {
    vec3 Albedo;
    float Fuzziness;
};
struct Material_Dielectric {
    float RefractionIndex;
};

struct Material {
    MaterialKind MatKind;
    union {
        Material_Lambertian Lambertian;
        Material_Metallic Metallic;
        Material_Dielectric Dielectric;
    };
    static Material zero() noexcept {
        Material m;
        std::memset(static_cast<void*>(&m), 0, sizeof(m));
        return m;
    }
    static Material make_lambertian(vec3 albedo) noexcept {
        Material m = zero();
        m.MatKind = MaterialKind::Lambertian;
        m.Lambertian = Material_Lambertian{albedo};
        return m;
    }
    static Material make_metallic(vec3 albedo, const float fuzziness) noexcept {
        Material m = zero();
        m.MatKind = MaterialKind::Metallic;
        m.Metallic = Material_Metallic{albedo, fuzziness < 1.0f ? fuzziness : 1.0f}; // std::min(1.0f, fuzziness)
        return m;
    }
    static Material make_dielectric(const float refraction_index) noexcept {
        Material m = zero();
        m.MatKind = MaterialKind::Dielectric;
        m.Dielectric = Material_Dielectric{refraction_index};
        return m;
    }
};
static_assert(sizeof(Material) == sizeof(rtmi_material), "Material is the 20-byte record of the C-ABI");

class MaterialCollection {
public:
    MaterialCollection() = default;
    MaterialHandleType add(const Material& mtl) {
        const MaterialHandleType mtl_handle{static_cast<uint32_t>(_materials.size())};
        _materials.push_back(mtl);
        return mtl_handle;
    }
    const Material& operator[](const MaterialHandleType mtl) const {
        const uint32_t idx = value_of(mtl);
        assert(idx < _materials.size());
        return _materials[idx];
    }
    const Material* data() const noexcept { return _materials.data(); }
    size_t size() const noexcept { return _materials.size(); }

private:
    std::vector<Material> _materials;
};

// ---- src/ray.tracer.core.cc:47-95 : the scene file's schema ------------------------------------------------------
struct SphereDef {
    std::array<float, 3> center;
    float radius;
};
struct AlbedoMatDef {
    std::array<float, 3> albedo;
};
struct DielectricMatDef {
    float refindex;
};
struct MetallicMatDef {
    std::array<float, 3> albedo;
    float fuzzines;
};
using MaterialDef = std::variant<AlbedoMatDef, DielectricMatDef, MetallicMatDef>; // rfl::TaggedUnion<"material_def",...>

struct WorldDefinition { // defaults as core.cc:67-95
    CameraParameters camera{16.0f / 9.0f, 1200, 100, 50, 20.0f, 10.0f, 3.4f,
                            {-2.0f, 2.0f, 1.0f}, {0.0f, 0.0f, -1.0f}, {0.0f, 1.0f, 0.0f}};
    int32_t a_min{-11};
    int32_t a_max{11};
    int32_t b_min{-11};
    int32_t b_max{11};
    std::array<float, 3> center{0.2f, 0.9f, 0.2f};
    std::array<float, 3> center_offset{4.0f, 0.2f, 0.0f};
    float center_dist_treshold{0.9f};
    float diffuse_material_treshold{0.85f};
    float metal_material_treshold{0.95f};
    std::vector<std::pair<SphereDef, MaterialDef>> objects{
        {SphereDef{{0.0f, -1000.0f, 0.0f}, 1000.0f}, AlbedoMatDef{{0.5f, 0.5f, 0.5f}}},
        {SphereDef{{0.0f, 1.0f, 0.0f}, 1.0f}, DielectricMatDef{1.5f}},
        {SphereDef{{-4.0f, -1.0f, 0.0f}, 1.0f}, AlbedoMatDef{{0.4f, 0.2f, 0.1f}}},
        {SphereDef{{4.0f, -1.0f, 0.0f}, 1.0f}, AlbedoMatDef{{0.7f, 0.6f, 0.5f}}},
    };
};

namespace detail {
// Minimal JSON reader for the reference's world.config.json (data/config/world.config.json): objects, arrays,
// numbers, strings.  Throws std::runtime_error on malformed input (the reference's `.value()` aborts, core.cc:102).
struct Json {
    enum Kind { Null, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::map<std::string, Json> obj;
    const Json& at(const std::string& k) const {
        auto it = obj.find(k);
        if (kind != Obj || it == obj.end()) throw std::runtime_error("world config: missing field '" + k + "'");
        return it->second;
    }
    bool has(const std::string& k) const { return kind == Obj && obj.count(k) != 0; }
};
class JsonParser {
public:
    explicit JsonParser(const std::string& s) : _s(s) {}
    Json parse() {
        Json v = value();
        ws();
        if (_i != _s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& _s;
    size_t _i = 0;
    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("world config: JSON error (") + what + ") at offset " + std::to_string(_i));
    }
    void ws() {
        while (_i < _s.size() && std::isspace(static_cast<unsigned char>(_s[_i]))) ++_i;
    }
    Json value() {
        ws();
        if (_i >= _s.size()) fail("unexpected end");
        const char c = _s[_i];
        Json v;
        if (c == '{') {
            v.kind = Json::Obj;
            ++_i;
            ws();
            if (_i < _s.size() && _s[_i] == '}') {
                ++_i;
                return v;
            }
            for (;;) {
                ws();
                Json k = string();
                ws();
                if (_i >= _s.size() || _s[_i] != ':') fail("expected ':'");
                ++_i;
                v.obj[k.str] = value();
                ws();
                if (_i < _s.size() && _s[_i] == ',') {
                    ++_i;
                    continue;
                }
                if (_i < _s.size() && _s[_i] == '}') {
                    ++_i;
                    return v;
                }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Json::Arr;
            ++_i;
            ws();
            if (_i < _s.size() && _s[_i] == ']') {
                ++_i;
                return v;
            }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (_i < _s.size() && _s[_i] == ',') {
                    ++_i;
                    continue;
                }
                if (_i < _s.size() && _s[_i] == ']') {
                    ++_i;
                    return v;
                }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') return string();
        if (c == '-' || c == '+' || std::isdigit(static_cast<unsigned char>(c))) {
            char* end = nullptr;
            v.kind = Json::Num;
            v.num = std::strtod(_s.c_str() + _i, &end);
            if (end == _s.c_str() + _i) fail("bad number");
            _i = static_cast<size_t>(end - _s.c_str());
            return v;
        }
        fail("unsupported token");
    }
    Json string() {
        if (_i >= _s.size() || _s[_i] != '"') fail("expected string");
        ++_i;
        Json v;
        v.kind = Json::Str;
        while (_i < _s.size() && _s[_i] != '"') {
            if (_s[_i] == '\\' && _i + 1 < _s.size()) ++_i;
            v.str.push_back(_s[_i++]);
        }
        if (_i >= _s.size()) fail("unterminated string");
        ++_i;
        return v;
    }
};
inline float num(const Json& j) {
    if (j.kind != Json::Num) throw std::runtime_error("world config: number expected");
    return static_cast<float>(j.num);
}
inline std::array<float, 3> arr3(const Json& j) {
    if (j.kind != Json::Arr || j.arr.size() != 3) throw std::runtime_error("world config: array of 3 numbers expected");
    return {num(j.arr[0]), num(j.arr[1]), num(j.arr[2])};
}
inline void check(int rc, const char* what) {
    if (rc != RTMI_OK) throw std::runtime_error(std::string(what) + ": " + rtmi_last_error());
}
} // namespace detail

// rfl::json::load<WorldDefinition>(path) of core.cc:102: every field is required, as reflect-cpp does for
// non-optional members; the tagged union is encoded as {"material_def": "<type name>", ...fields}.
inline WorldDefinition load_world_definition(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("world config: cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    const detail::Json j = detail::JsonParser(text).parse();
    using detail::arr3;
    using detail::num;
    WorldDefinition wd;
    const detail::Json& c = j.at("camera");
    wd.camera.aspect_ratio = num(c.at("aspect_ratio"));
    wd.camera.image_width = static_cast<uint32_t>(c.at("image_width").num);
    wd.camera.samples_per_pixel = static_cast<uint16_t>(c.at("samples_per_pixel").num);
    wd.camera.max_depth = static_cast<uint16_t>(c.at("max_depth").num);
    wd.camera.vertical_fov = num(c.at("vertical_fov"));
    wd.camera.defocus_angle = num(c.at("defocus_angle"));
    wd.camera.focus_distance = num(c.at("focus_distance"));
    wd.camera.lookfrom = arr3(c.at("lookfrom"));
    wd.camera.lookat = arr3(c.at("lookat"));
    wd.camera.world_up = arr3(c.at("world_up"));
    wd.a_min = static_cast<int32_t>(j.at("a_min").num);
    wd.a_max = static_cast<int32_t>(j.at("a_max").num);
    wd.b_min = static_cast<int32_t>(j.at("b_min").num);
    wd.b_max = static_cast<int32_t>(j.at("b_max").num);
    wd.center = arr3(j.at("center"));
    wd.center_offset = arr3(j.at("center_offset"));
    wd.center_dist_treshold = num(j.at("center_dist_treshold"));
    wd.diffuse_material_treshold = num(j.at("diffuse_material_treshold"));
    wd.metal_material_treshold = num(j.at("metal_material_treshold"));
    wd.objects.clear();
    const detail::Json& objs = j.at("objects");
    if (objs.kind != detail::Json::Arr) throw std::runtime_error("world config: 'objects' must be an array");
    for (const detail::Json& pair : objs.arr) {
        if (pair.kind != detail::Json::Arr || pair.arr.size() != 2)
            throw std::runtime_error("world config: each object is a [SphereDef, MaterialDef] pair");
        SphereDef sd{arr3(pair.arr[0].at("center")), num(pair.arr[0].at("radius"))};
        const detail::Json& m = pair.arr[1];
        const std::string tag = m.at("material_def").str;
        if (tag == "AlbedoMatDef") wd.objects.emplace_back(sd, AlbedoMatDef{arr3(m.at("albedo"))});
        else if (tag == "DielectricMatDef") wd.objects.emplace_back(sd, DielectricMatDef{num(m.at("refindex"))});
        else if (tag == "MetallicMatDef") wd.objects.emplace_back(sd, MetallicMatDef{arr3(m.at("albedo")), num(m.at("fuzzines"))});
        else throw std::runtime_error("world config: unknown material_def '" + tag + "'");
    }
    return wd;
}

inline vec3 to_vec3(const std::array<float, 3>& a) noexcept { return vec3{a[0], a[1], a[2]}; }

// make_world_spheres, core.cc:99-149 (the generator itself runs inside librtmi: rtmi_make_world_spheres).
// `mt_seed` replaces std::random_device (random.number.gen.hpp:45-46).
inline std::tuple<CameraParameters, HittableObject_Collection, MaterialCollection>
make_world_spheres(const WorldDefinition& world_def, uint32_t mt_seed) {
    std::vector<rtmi_object> fo;
    std::vector<rtmi_material> fm;
    for (const auto& [sphere_def, mtl_def] : world_def.objects) {
        const Material mtl = std::visit(
            [](const auto& d) {
                using T = std::decay_t<decltype(d)>;
                if constexpr (std::is_same_v<T, AlbedoMatDef>) return Material::make_lambertian(to_vec3(d.albedo));
                else if constexpr (std::is_same_v<T, DielectricMatDef>) return Material::make_dielectric(d.refindex);
                else return Material::make_metallic(to_vec3(d.albedo), d.fuzzines);
            },
            mtl_def);
        rtmi_object o{};
        std::memcpy(o.center, sphere_def.center.data(), sizeof(o.center));
        o.radius = sphere_def.radius;
        rtmi_material m;
        std::memcpy(&m, &mtl, sizeof(m));
        fo.push_back(o);
        fm.push_back(m);
    }
    rtmi_world_def def{};
    def.a_min = world_def.a_min;
    def.a_max = world_def.a_max;
    def.b_min = world_def.b_min;
    def.b_max = world_def.b_max;
    std::memcpy(def.center_offset, world_def.center_offset.data(), sizeof(def.center_offset));
    def.center_dist_treshold = world_def.center_dist_treshold;
    def.diffuse_material_treshold = world_def.diffuse_material_treshold;
    def.metal_material_treshold = world_def.metal_material_treshold;
    const int64_t na = std::max<int64_t>(0, int64_t(def.a_max) - def.a_min), nb = std::max<int64_t>(0, int64_t(def.b_max) - def.b_min);
    const uint32_t cap = static_cast<uint32_t>(fo.size() + na * nb);
    std::vector<rtmi_object> objs(cap);
    std::vector<rtmi_material> mats(cap);
    uint32_t n = 0;
    detail::check(rtmi_make_world_spheres(&def, fo.data(), fm.data(), static_cast<uint32_t>(fo.size()), mt_seed,
                                          objs.data(), mats.data(), cap, &n),
                  "rtmi_make_world_spheres");
    HittableObject_Collection world;
    MaterialCollection material_coll;
    for (uint32_t i = 0; i < n; ++i) {
        Material m;
        std::memcpy(static_cast<void*>(&m), &mats[i], sizeof(m));
        const MaterialHandleType h = material_coll.add(m);
        world.add_object(HittableObject::make_sphere(vec3{objs[i].center[0], objs[i].center[1], objs[i].center[2]},
                                                     objs[i].radius, h));
    }
    return {world_def.camera, std::move(world), std::move(material_coll)};
}

// ---- src/ray.tracer.core.hpp:18-42 ----------------------------------------------------------------------------------
struct RayTracingCore {
    uint32_t rts_img_width;
    uint32_t rts_img_height;
    float rts_defocus_angle;
    float rts_viewport_height;
    float rts_viewport_width;
    uint16_t rts_samples_per_pixel;
    uint16_t rts_maxdepth;
    float rts_pixels_sample_scale;
    vec3 rts_pixel_delta_u;
    vec3 rts_pixel_delta_v;
    vec3 rts_pixel00;
    vec3 rts_cam_center;
    vec3 rts_defocus_disk_u;
    vec3 rts_defocus_disk_v;
    HittableObject_Collection rts_world;
    MaterialCollection rts_materials;

    // device side (owned): created by default_setup / setup, released by the destructor
    std::shared_ptr<rtmi_scene> rts_gpu_scene;
    // more GPUs of the node (optional, attach_devices): one rtmi_frame for whole frames (interleaved row-block shards + one
    // RCCL gather inside the library); its per-device scene replicas also serve workers that pull row blocks from a
    // shared queue (the reference's N workers, main.cc:608-712) -- one scene per device, borrowed from the frame
    // (rtmi_frame_get_scene), not a second copy.  Footprint per device: BVH + strips + the sample records of the largest
    // call so far (16 B per sample, capped by rtmi_tuning::sample_buf_mb).  raytrace_frame and raytrace_rows_on must not
    // run at the same time.
    std::vector<rtmi_scene*> rts_gpu_replicas; // owned by rts_gpu_frame
    std::shared_ptr<rtmi_frame> rts_gpu_frame;
    rtmi_camera rts_camera_pod{}; // the 14 fields above as the C-ABI record

    // RayTracingCore::default_setup(), core.cc:171-216: loads data/config/world.config.json relative to the CWD.
    static std::shared_ptr<RayTracingCore> default_setup(uint32_t mt_seed = 12345,
                                                         const char* config = "data/config/world.config.json",
                                                         const rtmi_scene_options* options = nullptr) {
        return setup(load_world_definition(config), mt_seed, options);
    }

    static std::shared_ptr<RayTracingCore> setup(const WorldDefinition& wd, uint32_t mt_seed,
                                                 const rtmi_scene_options* options = nullptr) {
        auto [cam_params, world, mtl_coll] = make_world_spheres(wd, mt_seed);
        return setup(cam_params, std::move(world), std::move(mtl_coll), options);
    }

    static std::shared_ptr<RayTracingCore> setup(const CameraParameters& cam_params, HittableObject_Collection world,
                                                 MaterialCollection mtl_coll, const rtmi_scene_options* options = nullptr) {
        rtmi_camera_params cp;
        std::memcpy(&cp, &cam_params, sizeof(cp));
        rtmi_camera cam;
        detail::check(rtmi_camera_setup(&cp, &cam), "rtmi_camera_setup");
        auto core = std::make_shared<RayTracingCore>();
        auto v3 = [](const float* p) { return vec3{p[0], p[1], p[2]}; };
        core->rts_img_width = cam.img_width; // core.cc:198-215, same 14 fields
        core->rts_img_height = cam.img_height;
        core->rts_defocus_angle = cam.defocus_angle;
        core->rts_viewport_height = cam.viewport_height;
        core->rts_viewport_width = cam.viewport_width;
        core->rts_samples_per_pixel = cam.samples_per_pixel;
        core->rts_maxdepth = cam.maxdepth;
        core->rts_pixels_sample_scale = cam.pixels_sample_scale;
        core->rts_pixel_delta_u = v3(cam.pixel_delta_u);
        core->rts_pixel_delta_v = v3(cam.pixel_delta_v);
        core->rts_pixel00 = v3(cam.pixel00);
        core->rts_cam_center = v3(cam.cam_center);
        core->rts_defocus_disk_u = v3(cam.defocus_disk_u);
        core->rts_defocus_disk_v = v3(cam.defocus_disk_v);
        core->rts_world = std::move(world);
        core->rts_materials = std::move(mtl_coll);
        rtmi_scene* scene = nullptr;
        detail::check(rtmi_scene_create(&cam, reinterpret_cast<const rtmi_object*>(core->rts_world.data()),
                                        static_cast<uint32_t>(core->rts_world.size()),
                                        reinterpret_cast<const rtmi_material*>(core->rts_materials.data()),
                                        static_cast<uint32_t>(core->rts_materials.size()), options, &scene),
                      "rtmi_scene_create");
        core->rts_gpu_scene = std::shared_ptr<rtmi_scene>(scene, rtmi_scene_destroy);
        core->rts_camera_pod = cam;
        return core;
    }

    // Replicates the scene on `devices` (HIP ordinals).  Returns an rtmi_status; nothing is attached on failure.
    int attach_devices(const std::vector<int32_t>& devices, uint32_t block_rows = 8,
                       const rtmi_scene_options* options = nullptr) {
        const auto* objs = reinterpret_cast<const rtmi_object*>(rts_world.data());
        const auto* mats = reinterpret_cast<const rtmi_material*>(rts_materials.data());
        const auto n_objs = static_cast<uint32_t>(rts_world.size()), n_mats = static_cast<uint32_t>(rts_materials.size());
        rtmi_frame* frame = nullptr;
        int rc = rtmi_frame_create(&rts_camera_pod, objs, n_objs, mats, n_mats, options, devices.data(),
                                   static_cast<uint32_t>(devices.size()), block_rows, &frame);
        if (rc != RTMI_OK) return rc;
        std::shared_ptr<rtmi_frame> holder(frame, rtmi_frame_destroy);
        std::vector<rtmi_scene*> replicas(devices.size(), nullptr);
        for (uint32_t i = 0; i < devices.size(); ++i) {
            rc = rtmi_frame_get_scene(frame, i, &replicas[i]);
            if (rc != RTMI_OK) return rc;
        }
        rts_gpu_replicas = std::move(replicas);
        rts_gpu_frame = std::move(holder);
        return RTMI_OK;
    }

    // The whole frame on all attached devices in one call (rtmi_frame_render): scanline order, as raytrace_rows.
    int raytrace_frame(uint64_t seed, RGBAColor* rgba, float* rgb_linear = nullptr) const noexcept {
        if (!rts_gpu_frame) return raytrace_rows(0, rts_img_height, seed, rgba, rgb_linear);
        return rtmi_frame_render(rts_gpu_frame.get(), seed, rgb_linear, reinterpret_cast<uint32_t*>(rgba));
    }

    // raytrace_rows on one of the attached replicas (worker w of a multi-GPU job system)
    int raytrace_rows_on(size_t replica, uint32_t y0, uint32_t y1, uint64_t seed, RGBAColor* rgba,
                         float* rgb_linear = nullptr) const noexcept {
        if (replica >= rts_gpu_replicas.size()) return RTMI_ERR_BAD_ARG;
        return rtmi_render_rows(rts_gpu_replicas[replica], y0, y1, seed, rgb_linear, reinterpret_cast<uint32_t*>(rgba));
    }

    // Replaces the per-pixel loop over RayTracingCore::raytrace_pixel (core.cc:259-265, called from
    // process_tracing_work_package, main.cc:507-519): all pixels of rows [y0, y1), RGBAColor per pixel in scanline
    // order.  rgb_linear (optional) receives the value handed to RGBAColor{...}.  Returns an rtmi_status.
    int raytrace_rows(uint32_t y0, uint32_t y1, uint64_t seed, RGBAColor* rgba, float* rgb_linear = nullptr) const noexcept {
        static_assert(sizeof(RGBAColor) == sizeof(uint32_t), "RGBAColor packs to 0xAABBGGRR");
        return rtmi_render_rows(rts_gpu_scene.get(), y0, y1, seed, rgb_linear, reinterpret_cast<uint32_t*>(rgba));
    }

    // One RayTracingWorkPackage{start, end} (main.cc:404-407) through the tile-granular entry (rtmi_render_rect, 0.5): the
    // tile's pixels and nothing else, row-major, (ex-sx)*(ey-sy) RGBAColor.  (Rounds 1-4 rendered every column of the tile's
    // rows and threw all but the tile away: 240 times the work for the reference's 8x8 package at 1920 wide.)  Correct for a
    // host that keeps the reference's queue of 8x8 packages unchanged, and slow -- a launch, a resolve pass and a copy per 64
    // pixels; raytrace_rows is the fast granularity.
    int raytrace_tile(uint16_t sx, uint16_t sy, uint16_t ex, uint16_t ey, uint64_t seed, RGBAColor* tile_out,
                      float* rgb_linear = nullptr) const noexcept {
        if (ex < sx || ey < sy || ex > rts_img_width || ey > rts_img_height) return RTMI_ERR_BAD_ARG;
        return rtmi_render_rect(rts_gpu_scene.get(), sx, sy, ex, ey, seed, rgb_linear, reinterpret_cast<uint32_t*>(tile_out));
    }
};

} // namespace rtmi
