"""raytracing.cpp_amd -- MI355X-native path-tracing core (librtmi.so) and its Python plumbing.

The product is the C-ABI shared library built from csrc/ (hand-written HIP for gfx950, see include/rtmi.h)
plus the C++ host mirror in host/.  This module is the thin ctypes layer that tests, bench.py and
__graft_entry__.py use to drive that library from Python (device buffers and torch.distributed are plumbing,
not the product).  It never imports anything from oracle/ and has no CPU fallback: without the HIP library or
without a GPU every compute call fails loudly.

The directory name contains a dot, so it is loaded through `rtmi_loader.load()` (repo root) rather than a
plain `import`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "librtmi.so")
CSRC = [os.path.join(_HERE, "csrc", f) for f in ("rtmi_host.cpp", "rtmi_device.hip", "rtmi_frame.hip")]
HEADERS = [os.path.join(_HERE, "csrc", f) for f in ("rtmi_internal.h", "rtmi_kernel_common.h", "rtmi_walk_asm.h",
                                                     "rtmi_trace_kernel.h", "rtmi_resolve.h")] + [
    os.path.join(_ROOT, "include", "rtmi.h")]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # parity: no FMA contraction, IEEE sqrt/div, denormals kept -- the reference path is plain x86-64 fp32
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-gpu-flush-denormals-to-zero", "-Wno-unused-value",
    # SLP packing of the scalar fp32 math into v_pk_* costs more v_mov shuffles than it saves (A/B on MI355X: +3.4 %)
    "-fno-slp-vectorize",
]

RTMI_OK = 0
RTMI_ERR_BAD_ARG, RTMI_ERR_HIP, RTMI_ERR_OOM, RTMI_ERR_UNSUPPORTED, RTMI_ERR_RCCL, RTMI_ERR_INTERNAL = -1, -2, -3, -4, -5, -6
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_BVH = 0, 1, 2


def build_library(force=False, verbose=False, out=None, defines=()):
    """Compile csrc/ for gfx950 with hipcc into librtmi.so (in-tree, so it travels to the GPU box).
    `out` / `defines`: another file name and -D switches, for the A/B and diagnostic builds of tools/."""
    out = out or LIB_PATH
    csrc = CSRC
    srcs = csrc + HEADERS
    if (not force and os.path.exists(out)
            and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs)):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = ([hipcc] + HIPCC_FLAGS + ["-D" + d for d in defines]
           + ["-I", os.path.join(_ROOT, "include"), "-o", out] + csrc + ["-ldl"])
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


class CameraParams(C.Structure):  # rtmi_camera_params == reference src/camera.parameters.hpp:6-17
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_uint32), ("samples_per_pixel", C.c_uint16),
                ("max_depth", C.c_uint16), ("vertical_fov", C.c_float), ("defocus_angle", C.c_float),
                ("focus_distance", C.c_float), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3),
                ("world_up", C.c_float * 3)]


class Camera(C.Structure):  # rtmi_camera == the 14 PODs of RayTracingCore, reference src/ray.tracer.core.hpp:19-32
    _fields_ = [("img_width", C.c_uint32), ("img_height", C.c_uint32), ("defocus_angle", C.c_float),
                ("viewport_height", C.c_float), ("viewport_width", C.c_float), ("samples_per_pixel", C.c_uint16),
                ("maxdepth", C.c_uint16), ("pixels_sample_scale", C.c_float), ("pixel_delta_u", C.c_float * 3),
                ("pixel_delta_v", C.c_float * 3), ("pixel00", C.c_float * 3), ("cam_center", C.c_float * 3),
                ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3)]


class WorldDef(C.Structure):  # rtmi_world_def == reference src/ray.tracer.core.cc:67-95 minus camera/objects
    _fields_ = [("a_min", C.c_int32), ("a_max", C.c_int32), ("b_min", C.c_int32), ("b_max", C.c_int32),
                ("center_offset", C.c_float * 3), ("center_dist_treshold", C.c_float),
                ("diffuse_material_treshold", C.c_float), ("metal_material_treshold", C.c_float)]


class Tuning(C.Structure):  # rtmi_tuning: scheduling knobs, 0 = default; none of them changes the image
    _fields_ = [("struct_size", C.c_uint32), ("block_lanes", C.c_uint32), ("blocks_per_cu", C.c_uint32),
                ("wait_thresh", C.c_uint32), ("pad_mode", C.c_uint32), ("chunk_samples", C.c_int32),
                ("chain_mode", C.c_int32), ("bvh_passes", C.c_uint32), ("sample_buf_mb", C.c_uint32),
                ("force_hbm_scene", C.c_uint32), ("top_down", C.c_uint32), ("kernel", C.c_uint32),
                ("reserved3", C.c_uint32 * 3), ("lds_top_nodes", C.c_uint32), ("tile_order", C.c_uint32),
                ("bands", C.c_uint32), ("gen_ahead", C.c_uint32),
                ("cam_entry", C.c_uint32), ("walk_start", C.c_uint32), ("stack_cap", C.c_uint32), ("reserved6", C.c_uint32)]


class SceneOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("accel", C.c_uint32), ("leaf_size", C.c_uint32), ("device", C.c_int32),
                ("collect_stats", C.c_uint32), ("reserved", C.c_uint32 * 3), ("tuning", C.POINTER(Tuning))]


class LaunchInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "kernel", "block_lanes", "grid_blocks", "blocks_per_cu",
                                          "lds_bytes", "scene_in_lds", "stack_depth", "whole_pixel_fallbacks",
                                          "packed_chains", "packed_chain_fallbacks", "lds_top_nodes", "pad_mode",
                                          "bands", "tile_order", "probe_us", "gen_ahead",
                                          "cam_entry", "entry_build_us", "walk_start", "stack_cap", "reband_retries")]


class FrameTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("gather_ms", C.c_float), ("kernel_ms", C.c_float * 16)]


def make_tuning(**kw):
    """rtmi_tuning from keyword arguments (field names of include/rtmi.h); unknown names are an error."""
    t = Tuning()
    t.struct_size = C.sizeof(Tuning)
    names = {n for n, _ in Tuning._fields_} - {"struct_size", "reserved3", "reserved6"}
    for k, v in kw.items():
        if k not in names:
            raise KeyError(f"unknown tuning knob {k!r}")
        setattr(t, k, int(v))
    return t


def _options(accel, leaf_size, device, collect_stats, tuning):
    opt = SceneOptions()
    opt.struct_size = C.sizeof(SceneOptions)
    opt.accel, opt.leaf_size, opt.device, opt.collect_stats = accel, leaf_size, device, int(collect_stats)
    tun = make_tuning(**tuning) if tuning else None
    if tun is not None:
        opt.tuning = C.pointer(tun)
    return opt, tun  # keep `tun` alive while `opt` is in use


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("node_tests", C.c_uint64)]


from . import workloads  # noqa: E402  (scene + camera generators of the BASELINE configs)

OBJECT_DTYPE, MATERIAL_DTYPE = workloads.OBJECT_DTYPE, workloads.MATERIAL_DTYPE
BVH_NODE_DTYPE = np.dtype([("ctr", "<f4", (2, 3)), ("half", "<f4", (2, 3)), ("child", "<u4", 2),
                           ("reserved", "<f4", 2)])
assert OBJECT_DTYPE.itemsize == 24 and MATERIAL_DTYPE.itemsize == 20 and BVH_NODE_DTYPE.itemsize == 64

# every symbol include/rtmi.h declares
EXPORTS = ("rtmi_camera_setup", "rtmi_make_world_spheres", "rtmi_scene_create", "rtmi_scene_destroy",
           "rtmi_render_rows", "rtmi_render_row_blocks_device", "rtmi_render_rect", "rtmi_render_rect_device",
           "rtmi_render_block_list_device", "rtmi_shard_plan", "rtmi_scene_get_tile_costs",
           "rtmi_last_error", "rtmi_version",
           "rtmi_scene_get_stats", "rtmi_scene_get_accel", "rtmi_scene_get_launch_info", "rtmi_scene_get_bvh", "rtmi_scene_last_kernel_ms",
           "rtmi_bvh_build", "rtmi_bvh_build_passes", "rtmi_tile_entries_build", "rtmi_scene_get_tile_entries", "rtmi_scene_get_walk_starts", "rtmi_walk_starts_build", "rtmi_frame_create", "rtmi_frame_destroy", "rtmi_frame_render",
           "rtmi_frame_render_device", "rtmi_frame_get_timing", "rtmi_frame_rccl_ranks", "rtmi_frame_get_scene")

_lib = None


class RtmiError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"rtmi error {code}: {text}")
        self.code = code


def lib():
    """Load librtmi.so.  Fails loudly when the HIP extension is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950) first; "
                           "the product has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rtmi_camera_setup.argtypes = [C.POINTER(CameraParams), C.POINTER(Camera)]
    L.rtmi_make_world_spheres.argtypes = [C.POINTER(WorldDef), vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32,
                                          C.POINTER(C.c_uint32)]
    L.rtmi_scene_create.argtypes = [C.POINTER(Camera), vp, C.c_uint32, vp, C.c_uint32, C.POINTER(SceneOptions),
                                    C.POINTER(vp)]
    L.rtmi_scene_destroy.argtypes = [vp]
    L.rtmi_scene_destroy.restype = None
    L.rtmi_render_rows.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp]
    L.rtmi_render_row_blocks_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp,
                                                vp, vp]
    if hasattr(L, "rtmi_render_rect"):  # (absent from older builds that tools/ A/B against the current one)
        L.rtmi_render_rect.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp]
        L.rtmi_render_rect_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp]
    L.rtmi_last_error.restype = C.c_char_p
    L.rtmi_version.restype = C.c_char_p
    L.rtmi_scene_get_stats.argtypes = [vp, C.POINTER(Stats), C.c_int]
    L.rtmi_scene_get_accel.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.rtmi_scene_get_launch_info.argtypes = [vp, C.POINTER(LaunchInfo)]
    L.rtmi_scene_get_bvh.argtypes = [vp, vp, C.POINTER(C.c_uint32), vp, C.POINTER(C.c_uint32), vp,
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rtmi_scene_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    u32p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.rtmi_bvh_build.argtypes = [vp, C.c_uint32, C.c_uint32, vp, u32p, vp, u32p, u32p, vp, u32p, f32p, f32p]
    if hasattr(L, "rtmi_bvh_build_passes"):  # (absent from older builds that tools/ A/B against the current one)
        L.rtmi_bvh_build_passes.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p, vp, u32p, u32p, vp, u32p, f32p, f32p]
    if hasattr(L, "rtmi_tile_entries_build"):  # (absent from older builds that tools/ A/B against the current one)
        L.rtmi_tile_entries_build.argtypes = [C.POINTER(Camera), vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p]
        L.rtmi_scene_get_tile_entries.argtypes = [vp, vp, u32p]
        L.rtmi_scene_get_walk_starts.argtypes = [vp, vp, u32p]
        L.rtmi_walk_starts_build.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p, u32p, vp]
        L.rtmi_render_block_list_device.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint64, vp, vp, vp]
        L.rtmi_shard_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
        L.rtmi_scene_get_tile_costs.argtypes = [vp, vp, u32p]
    L.rtmi_frame_create.argtypes = [C.POINTER(Camera), vp, C.c_uint32, vp, C.c_uint32, C.POINTER(SceneOptions),
                                    C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.rtmi_frame_destroy.argtypes = [vp]
    L.rtmi_frame_destroy.restype = None
    L.rtmi_frame_render.argtypes = [vp, C.c_uint64, vp, vp]
    L.rtmi_frame_render_device.argtypes = [vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
    L.rtmi_frame_get_timing.argtypes = [vp, C.POINTER(FrameTiming)]
    L.rtmi_frame_rccl_ranks.argtypes = [vp, C.POINTER(C.c_uint32)]
    if hasattr(L, "rtmi_frame_get_scene"):  # (absent from older builds that tools/ A/B against the current one)
        L.rtmi_frame_get_scene.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    for name in EXPORTS:
        if name not in ("rtmi_last_error", "rtmi_version", "rtmi_scene_destroy", "rtmi_frame_destroy") and hasattr(L, name):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _check(rc):
    if rc != RTMI_OK:
        raise RtmiError(rc, lib().rtmi_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def camera_params(aspect_ratio=16.0 / 9.0, image_width=1200, samples_per_pixel=100, max_depth=50, vertical_fov=20.0,
                  defocus_angle=0.6, focus_distance=10.0, lookfrom=(13.0, 2.0, 3.0), lookat=(0.0, 0.0, 0.0),
                  world_up=(0.0, 1.0, 0.0)):
    """CameraParameters of the S-RTOW workload: data/config/world.config.json:2-25 at aspect 16/9."""
    p = CameraParams()
    p.aspect_ratio, p.image_width = aspect_ratio, image_width
    p.samples_per_pixel, p.max_depth = samples_per_pixel, max_depth
    p.vertical_fov, p.defocus_angle, p.focus_distance = vertical_fov, defocus_angle, focus_distance
    p.lookfrom[:], p.lookat[:], p.world_up[:] = lookfrom, lookat, world_up
    return p


def camera_setup(params):
    cam = Camera()
    _check(lib().rtmi_camera_setup(C.byref(params), C.byref(cam)))
    return cam


# the listed objects of data/config/world.config.json:43-112
RTOW_FIXED = [
    ((0.0, -1000.0, 0.0), 1000.0, (0, (0.5, 0.5, 0.5, 0.0))),
    ((0.0, 1.0, 0.0), 1.0, (2, (1.5, 0.0, 0.0, 0.0))),
    ((-4.0, 1.0, 0.0), 1.0, (0, (0.4, 0.2, 0.1, 0.0))),
    ((4.0, 1.0, 0.0), 1.0, (0, (0.7, 0.6, 0.5, 0.0))),
]


def world_def(a_min=-11, a_max=11, b_min=-11, b_max=11, center_offset=(4.0, 0.2, 0.0), center_dist_treshold=0.9,
              diffuse=0.8, metal=0.95):
    wd = WorldDef()
    wd.a_min, wd.a_max, wd.b_min, wd.b_max = a_min, a_max, b_min, b_max
    wd.center_offset[:] = center_offset
    wd.center_dist_treshold = center_dist_treshold
    wd.diffuse_material_treshold, wd.metal_material_treshold = diffuse, metal
    return wd


def fixed_arrays(fixed):
    objs = np.zeros(len(fixed), OBJECT_DTYPE)
    mats = np.zeros(len(fixed), MATERIAL_DTYPE)
    for i, (c, r, (k, p)) in enumerate(fixed):
        objs[i] = (0, c, r, i)
        mats[i] = (k, p)
    return objs, mats


def make_world_spheres(seed=12345, wd=None, fixed=None):
    """S-RTOW(seed): make_world_spheres of the reference (src/ray.tracer.core.cc:99-149), RNG seeded explicitly."""
    wd = wd or world_def()
    fobjs, fmats = fixed_arrays(RTOW_FIXED if fixed is None else fixed)
    cap = len(fobjs) + max(0, wd.a_max - wd.a_min) * max(0, wd.b_max - wd.b_min)
    objs = np.zeros(cap, OBJECT_DTYPE)
    mats = np.zeros(cap, MATERIAL_DTYPE)
    n = C.c_uint32(0)
    _check(lib().rtmi_make_world_spheres(C.byref(wd), _ptr(fobjs), _ptr(fmats), len(fobjs), seed, _ptr(objs),
                                         _ptr(mats), cap, C.byref(n)))
    return objs[:n.value].copy(), mats[:n.value].copy()


class Scene:
    """Owns an rtmi_scene handle (device copies of camera, world, materials, BVH)."""

    def __init__(self, cam, objs, mats, accel=ACCEL_AUTO, leaf_size=0, device=-1, collect_stats=False, tuning=None):
        objs = np.ascontiguousarray(objs, dtype=OBJECT_DTYPE)
        mats = np.ascontiguousarray(mats, dtype=MATERIAL_DTYPE)
        opt, _tun = _options(accel, leaf_size, device, collect_stats, tuning)
        self._h = C.c_void_p()
        self.cam = cam
        self.width, self.height = cam.img_width, cam.img_height
        _check(lib().rtmi_scene_create(C.byref(cam), _ptr(objs), len(objs), _ptr(mats), len(mats), C.byref(opt),
                                       C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rtmi_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def accel(self):
        v = C.c_uint32(0)
        _check(lib().rtmi_scene_get_accel(self._h, C.byref(v)))
        return v.value

    def launch_info(self):
        """rtmi_scene_get_launch_info as a dict."""
        v = LaunchInfo()
        v.struct_size = C.sizeof(LaunchInfo)
        _check(lib().rtmi_scene_get_launch_info(self._h, C.byref(v)))
        return {n: getattr(v, n) for n, _ in LaunchInfo._fields_ if n != "struct_size"}

    def render_rows(self, y0, y1, seed, rgb=True, rgba=True):
        """rtmi_render_rows: host buffers, blocking."""
        n = max(0, y1 - y0) * self.width
        rgb_a = np.zeros((max(0, y1 - y0), self.width, 3), np.float32) if rgb else None
        rgba_a = np.zeros((max(0, y1 - y0), self.width), np.uint32) if rgba else None
        _check(lib().rtmi_render_rows(self._h, y0, y1, seed, _ptr(rgb_a) if n else None,
                                      _ptr(rgba_a) if n else None))
        return rgb_a, rgba_a

    def render_rect(self, x0, y0, x1, y1, seed, rgb=True, rgba=True):
        """rtmi_render_rect: pixels [x0, x1) x [y0, y1) into dense host buffers, blocking."""
        h, w = max(0, y1 - y0), max(0, x1 - x0)
        rgb_a = np.zeros((h, w, 3), np.float32) if rgb else None
        rgba_a = np.zeros((h, w), np.uint32) if rgba else None
        _check(lib().rtmi_render_rect(self._h, x0, y0, x1, y1, seed, _ptr(rgb_a) if h * w else None,
                                      _ptr(rgba_a) if h * w else None))
        return rgb_a, rgba_a

    def render_rect_device(self, x0, y0, x1, y1, seed, d_rgb=0, d_rgba=0, stream=0):
        """rtmi_render_rect_device: raw device pointers (ints), asynchronous on `stream`."""
        _check(lib().rtmi_render_rect_device(self._h, x0, y0, x1, y1, seed, C.c_void_p(d_rgb or None),
                                             C.c_void_p(d_rgba or None), C.c_void_p(stream or None)))

    def render_row_blocks_device(self, y_first, block_rows, block_stride, n_blocks, seed, d_rgb=0, d_rgba=0,
                                 stream=0):
        """rtmi_render_row_blocks_device: raw device pointers (ints), asynchronous on `stream`."""
        _check(lib().rtmi_render_row_blocks_device(self._h, y_first, block_rows, block_stride, n_blocks, seed,
                                                   C.c_void_p(d_rgb or None), C.c_void_p(d_rgba or None),
                                                   C.c_void_p(stream or None)))

    def render_block_list_device(self, block_rows, blocks, seed, d_rgb=0, d_rgba=0, stream=0):
        """rtmi_render_block_list_device: image rows [b * block_rows, + block_rows) for b in `blocks`, into a dense slice in list
        order; raw device pointers (ints), asynchronous on `stream`."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint32)
        _check(lib().rtmi_render_block_list_device(self._h, block_rows, _ptr(blocks), len(blocks), seed, C.c_void_p(d_rgb or None),
                                                   C.c_void_p(d_rgba or None), C.c_void_p(stream or None)))

    def tile_costs(self):
        """rtmi_scene_get_tile_costs: ray segments per 8x8 tile of the image from the scene's probe, shape (tiles_y, tiles_x), or
        None when the scene made no probe."""
        n = C.c_uint32(0)
        _check(lib().rtmi_scene_get_tile_costs(self._h, None, C.byref(n)))
        if not n.value:
            return None
        out = np.zeros(((self.height + 7) // 8, (self.width + 7) // 8), np.uint32)
        assert out.size == n.value
        _check(lib().rtmi_scene_get_tile_costs(self._h, _ptr(out), C.byref(n)))
        return out

    def stats(self, reset=False):
        st = Stats()
        _check(lib().rtmi_scene_get_stats(self._h, C.byref(st), int(reset)))
        return {n: int(getattr(st, n)) for n, _ in Stats._fields_}

    def last_kernel_ms(self):
        v = C.c_float(0)
        _check(lib().rtmi_scene_last_kernel_ms(self._h, C.byref(v)))
        return v.value

    def bvh(self):
        nn, ns, nc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        eps, floor = C.c_float(0), C.c_float(0)
        _check(lib().rtmi_scene_get_bvh(self._h, None, C.byref(nn), None, C.byref(ns), None, C.byref(nc),
                                        C.byref(eps), C.byref(floor)))
        nodes = np.zeros(nn.value, BVH_NODE_DTYPE)
        slots = np.zeros(ns.value, np.uint32)
        pc = np.zeros((nc.value, 8), np.float32)
        _check(lib().rtmi_scene_get_bvh(self._h, _ptr(nodes), None, _ptr(slots), None, _ptr(pc), None, None, None))
        entries = None  # camera-ray entries per 8x8 tile (rtmi_tuning::cam_entry), for an instrumented CPU walk that follows them
        if hasattr(lib(), "rtmi_scene_get_tile_entries"):
            nt = C.c_uint32(0)
            _check(lib().rtmi_scene_get_tile_entries(self._h, None, C.byref(nt)))
            if nt.value:
                entries = np.zeros(((self.height + 7) // 8, (self.width + 7) // 8), np.uint32)
                assert entries.size == nt.value
                _check(lib().rtmi_scene_get_tile_entries(self._h, _ptr(entries), C.byref(nt)))
        starts = None  # walk starts of scattered rays (rtmi_tuning::walk_start): 16 words per slot; way records sit behind the tree's nodes
        if hasattr(lib(), "rtmi_scene_get_walk_starts"):
            nsl = C.c_uint32(0)
            _check(lib().rtmi_scene_get_walk_starts(self._h, None, C.byref(nsl)))
            if nsl.value:
                starts = np.zeros((nsl.value, 16), np.uint32)
                _check(lib().rtmi_scene_get_walk_starts(self._h, _ptr(starts), C.byref(nsl)))
        n_tree = len(nodes)  # (way records sit behind the tree's own nodes, numbered without a gap)
        if starts is not None:
            ways = starts[:, 2:][np.arange(14)[None, :] < starts[:, 1:2]]
            if ways.size:
                n_tree = int(ways.min())
        return dict(nodes=nodes, slots=slots, pad_classes=pc, pad_eps=eps.value, pad_floor=floor.value, entries=entries,
                    walk_starts=starts, n_tree_nodes=n_tree)


class Frame:
    """Owns an rtmi_frame handle: one frame on several GPUs of one node, one process (rtmi_frame_*): scene replicas,
    interleaved row-block shards, one RCCL gather to devices[0]."""

    def __init__(self, cam, objs, mats, devices=(0,), block_rows=8, accel=ACCEL_AUTO, leaf_size=0, tuning=None,
                 rehearsal=False, force_rccl=False, cost_plan=False):
        objs = np.ascontiguousarray(objs, dtype=OBJECT_DTYPE)
        mats = np.ascontiguousarray(mats, dtype=MATERIAL_DTYPE)
        opt, _tun = _options(accel, leaf_size, -1, False, tuning)
        # RTMI_FRAME_REHEARSAL (1): repeated devices, copies instead of RCCL; RTMI_FRAME_FORCE_RCCL (2): a communicator and
        # the grouped gather even for one device; RTMI_FRAME_COST_PLAN (4): row blocks dealt out by the scene's cost map
        opt.reserved[0] = (1 if rehearsal else 0) | (2 if force_rccl else 0) | (4 if cost_plan else 0)
        devs = (C.c_int32 * len(devices))(*devices)
        self._h = C.c_void_p()
        self.width, self.height, self.n_devices = cam.img_width, cam.img_height, len(devices)
        _check(lib().rtmi_frame_create(C.byref(cam), _ptr(objs), len(objs), _ptr(mats), len(mats), C.byref(opt), devs,
                                       len(devices), block_rows, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rtmi_frame_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def render(self, seed):
        """rtmi_frame_render: the whole frame into host buffers."""
        rgb = np.zeros((self.height, self.width, 3), np.float32)
        rgba = np.zeros((self.height, self.width), np.uint32)
        _check(lib().rtmi_frame_render(self._h, seed, _ptr(rgb), _ptr(rgba)))
        return rgb, rgba

    def render_device(self, seed):
        """rtmi_frame_render_device: the frame stays on devices[0]; returns the two device pointers (ints)."""
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().rtmi_frame_render_device(self._h, seed, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing(self):
        t = FrameTiming()
        _check(lib().rtmi_frame_get_timing(self._h, C.byref(t)))
        return dict(total_ms=t.total_ms, gather_ms=t.gather_ms, kernel_ms=list(t.kernel_ms)[:self.n_devices])

    @property
    def rccl_ranks(self):
        v = C.c_uint32(0)
        _check(lib().rtmi_frame_rccl_ranks(self._h, C.byref(v)))
        return v.value


def bvh_build(objs, leaf_size=0, bvh_passes=0):
    """rtmi_bvh_build_passes: the host-side BVH of a scene (no device needed); bvh_passes as rtmi_tuning::bvh_passes."""
    objs = np.ascontiguousarray(objs, dtype=OBJECT_DTYPE)
    n = len(objs)
    nodes = np.zeros(max(n, 1), BVH_NODE_DTYPE)
    slots = np.zeros(max(n, 1), np.uint32)
    pc = np.zeros((4, 8), np.float32)
    nn, root, depth, nc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    eps, floor = C.c_float(0), C.c_float(0)
    _check(lib().rtmi_bvh_build_passes(_ptr(objs), n, leaf_size, bvh_passes, _ptr(nodes), C.byref(nn), _ptr(slots),
                                       C.byref(root), C.byref(depth), _ptr(pc), C.byref(nc), C.byref(eps), C.byref(floor)))
    return dict(nodes=nodes[:nn.value].copy(), slots=slots[:n].copy(), pad_classes=pc[:nc.value].copy(),
                pad_eps=eps.value, pad_floor=floor.value, root_ref=root.value, depth=depth.value)


def walk_starts_build(objs, leaf_size=0, bvh_passes=0):
    """rtmi_walk_starts_build (no device needed): (nodes, n_tree_nodes, records) -- the tree of bvh_build for the same arguments with
    the way records behind its nodes, and the (n_objects, 16) start records of scattered rays."""
    objs = np.ascontiguousarray(objs, dtype=OBJECT_DTYPE)
    n = len(objs)
    nodes = np.zeros(3 * n + 2, BVH_NODE_DTYPE)
    rec = np.zeros((max(n, 1), 16), np.uint32)
    nn, nt = C.c_uint32(0), C.c_uint32(0)
    _check(lib().rtmi_walk_starts_build(_ptr(objs), n, leaf_size, bvh_passes, _ptr(nodes), C.byref(nn), C.byref(nt), _ptr(rec)))
    return nodes[:nn.value].copy(), nt.value, rec[:n].copy()


def tile_entries_build(cam, objs, leaf_size=0, bvh_passes=0):
    """rtmi_tile_entries_build: the camera-ray entry of every 8x8 tile of the image (no device needed), shape (tiles_y, tiles_x);
    references into the tree bvh_build returns for the same arguments, 0xffffffff = the tile's beam meets no sphere of the tree."""
    objs = np.ascontiguousarray(objs, dtype=OBJECT_DTYPE)
    gtx, gty = (cam.img_width + 7) // 8, (cam.img_height + 7) // 8
    out = np.zeros((gty, gtx), np.uint32)
    n = C.c_uint32(0)
    _check(lib().rtmi_tile_entries_build(C.byref(cam), _ptr(objs), len(objs), leaf_size, bvh_passes, _ptr(out), C.byref(n)))
    assert n.value == gtx * gty
    return out


def row_block_shards(height, block_rows, world_size):
    """Interleaved row-block sharding of the image plane: block b -> rank b % world_size.

    Returns per rank (y_first, n_blocks, rows) with rows the number of image rows it renders; blocks of a rank are
    `world_size` blocks apart (rtmi_render_row_blocks_device's block_stride)."""
    n_blocks_total = (height + block_rows - 1) // block_rows
    out = []
    for r in range(world_size):
        nb = (n_blocks_total - r + world_size - 1) // world_size if n_blocks_total > r else 0
        rows = 0
        for k in range(nb):
            y = (r + k * world_size) * block_rows
            rows += min(block_rows, height - y)
        out.append((r * block_rows, nb, rows))
    return out


def deinterleave_rows(height, block_rows, world_size):
    """Index map for the gathered [rank-major] slices -> scanline order.

    gathered row index g (rank r's dense slice padded to `max_rows` rows, concatenated) for image row y."""
    shards = row_block_shards(height, block_rows, world_size)
    max_rows = max(s[2] for s in shards) if shards else 0
    idx = np.zeros(height, np.int64)
    for y in range(height):
        b = y // block_rows
        r, k = b % world_size, b // world_size
        idx[y] = r * max_rows + k * block_rows + (y - b * block_rows)
    return idx, max_rows


# ---------------------------------------------------------------------------------------------------------
# multi-GPU plumbing: interleaved row-block sharding + one gather of the per-rank framebuffer slices
# (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests)
# ---------------------------------------------------------------------------------------------------------
class RowShardPlan:
    """Row-block sharding of an image of `height` rows over `world_size` ranks.

    Block b (block_rows rows) belongs to rank b % world_size, so that the expensive lower rows and the cheap sky
    rows are spread evenly.  Every rank renders its blocks into a dense slice padded to `max_rows` rows; rank 0
    gathers the slices and `index` maps the rank-major gathered rows back to scanline order."""

    def __init__(self, height, block_rows, world_size):
        self.height, self.block_rows, self.world_size = height, block_rows, world_size
        self.shards = row_block_shards(height, block_rows, world_size)
        self.index, self.max_rows = deinterleave_rows(height, block_rows, world_size)

    def shard(self, rank):
        """(y_first, n_blocks, rows) of `rank`; block stride is world_size."""
        return self.shards[rank]


def block_costs(tile_costs, height, block_rows):
    """Cost of every block of `block_rows` rows from a per-8x8-tile cost map (Scene.tile_costs): a tile row counts for the block
    its first row lies in (blocks of a multiple of 8 rows, the multi-GPU shards, are whole tile rows)."""
    nb = (height + block_rows - 1) // block_rows
    out = np.zeros(nb, np.uint64)
    rows = tile_costs.astype(np.uint64).sum(axis=1)
    for ty, c in enumerate(rows):
        out[min(nb - 1, (ty * 8) // block_rows)] += c
    return out


class CostShardPlan:
    """Row blocks of an image -> ranks by cost (rtmi_shard_plan): block b -> rank b mod N ignores that the rows of a
    frame differ in cost -- on the 1080p S-RTOW frame the eighth-frame shards of 8 GPUs were 4 % apart -- so the blocks are dealt
    out longest processing time first, every rank the same number of them.  Same interface as RowShardPlan (`max_rows`, `index`),
    with `blocks(rank)` the list for rtmi_render_block_list_device.  `block_cost=None` reproduces block b -> rank b mod N.
    The frame is bit-identical for any assignment (the draw streams are keyed by the absolute pixel)."""

    def __init__(self, height, block_rows, world_size, block_cost=None):
        self.height, self.block_rows, self.world_size = height, block_rows, world_size
        nb = (height + block_rows - 1) // block_rows
        self.rank_of_block = np.zeros(max(nb, 1), np.uint32)
        cost = None if block_cost is None else np.ascontiguousarray(block_cost, dtype=np.uint64)
        assert cost is None or len(cost) == nb
        _check(lib().rtmi_shard_plan(height, block_rows, world_size, _ptr(cost), _ptr(self.rank_of_block)))
        self.rank_of_block = self.rank_of_block[:nb]
        self._blocks = [np.flatnonzero(self.rank_of_block == r).astype(np.uint32) for r in range(world_size)]  # (ascending: a clipped last block stays last)
        rows = [int(sum(min(block_rows, height - int(b) * block_rows) for b in bl)) for bl in self._blocks]
        self.shards = [(bl, len(bl), rw) for bl, rw in zip(self._blocks, rows)]
        self.max_rows = max(rows) if rows else 0
        self.index = np.zeros(height, np.int64)
        for r, bl in enumerate(self._blocks):
            for k, b in enumerate(bl):
                y0 = int(b) * block_rows
                n = min(block_rows, height - y0)
                self.index[y0:y0 + n] = r * self.max_rows + k * block_rows + np.arange(n)

    def blocks(self, rank):
        return self._blocks[rank]

    def rows(self, rank):
        return self.shards[rank][2]


def gather_frame(local_slice, plan, rank, dst=0, group=None):
    """Gathers the per-rank slices ([max_rows, W, C] tensors) to `dst` and returns the de-interleaved
    [height, W, C] frame there (None elsewhere).  One collective, no data-path exchange otherwise."""
    import torch
    import torch.distributed as dist
    if plan.world_size == 1:
        return local_slice[:plan.height]
    if rank == dst:
        parts = [torch.empty_like(local_slice) for _ in range(plan.world_size)]
        dist.gather(local_slice, parts, dst=dst, group=group)
        stacked = torch.cat(parts, dim=0)
        idx = torch.as_tensor(plan.index, device=stacked.device)
        return stacked.index_select(0, idx)
    dist.gather(local_slice, None, dst=dst, group=group)
    return None
