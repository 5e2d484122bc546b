"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  Parity is pinned only by the four reference pixels of
SURVEY.md 8(c); otherwise unpinned (see rt_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librt_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rt_oracle.c", "rt_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


class CameraParams(C.Structure):  # src/camera.parameters.hpp:6-17
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_uint32), ("samples_per_pixel", C.c_uint16),
                ("max_depth", C.c_uint16), ("vertical_fov", C.c_float), ("defocus_angle", C.c_float),
                ("focus_distance", C.c_float), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3),
                ("world_up", C.c_float * 3)]


class Camera(C.Structure):  # RayTracingCore PODs, src/ray.tracer.core.hpp:19-32
    _fields_ = [("img_width", C.c_uint32), ("img_height", C.c_uint32), ("defocus_angle", C.c_float),
                ("viewport_height", C.c_float), ("viewport_width", C.c_float), ("samples_per_pixel", C.c_uint16),
                ("maxdepth", C.c_uint16), ("pixels_sample_scale", C.c_float), ("pixel_delta_u", C.c_float * 3),
                ("pixel_delta_v", C.c_float * 3), ("pixel00", C.c_float * 3), ("cam_center", C.c_float * 3),
                ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3)]


class WorldDef(C.Structure):  # src/ray.tracer.core.cc:67-95
    _fields_ = [("a_min", C.c_int32), ("a_max", C.c_int32), ("b_min", C.c_int32), ("b_max", C.c_int32),
                ("center_offset", C.c_float * 3), ("center_dist_treshold", C.c_float),
                ("diffuse_material_treshold", C.c_float), ("metal_material_treshold", C.c_float)]


class Counters(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("samples", "segments", "sphere_tests", "node_tests", "rng_doubles",
                                           "hit_lambertian", "hit_metallic", "hit_dielectric", "end_sky",
                                           "end_depth", "end_absorbed")]
                # instrumented BVH walk only (tools/descent_score.py): by where a segment starts -- camera, peeled sphere, tree sphere
                + [(n, C.c_uint64 * 3) for n in ("seg_class", "trips_class", "leaf_trips_class")]
                + [(n, C.c_uint64) for n in ("descent_levels", "descent_sibling_hits")])

    def as_dict(self):
        return {n: (int(getattr(self, n)) if t is C.c_uint64 else [int(v) for v in getattr(self, n)]) for n, t in self._fields_}


OBJECT_DTYPE = np.dtype([("kind", "<u4"), ("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4")])  # 24 B
MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("p", "<f4", 4)])  # 20 B
BVH_NODE_DTYPE = np.dtype([("ctr", "<f4", (2, 3)), ("half", "<f4", (2, 3)), ("child", "<u4", 2),
                           ("inv2rmin", "<f4", 2)])  # 64 B
assert OBJECT_DTYPE.itemsize == 24 and MATERIAL_DTYPE.itemsize == 20 and BVH_NODE_DTYPE.itemsize == 64

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.orc_rng_new_mt.restype = vp
    L.orc_rng_new_mt.argtypes = [C.c_uint32]
    L.orc_rng_free.argtypes = [vp]
    L.orc_rng_double.restype = C.c_double
    L.orc_rng_double.argtypes = [vp]
    L.orc_mt_next_u32.restype = C.c_uint32
    L.orc_mt_next_u32.argtypes = [vp]
    L.orc_counter_double.restype = C.c_double
    L.orc_counter_double.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_mix_seed.restype = C.c_uint64
    L.orc_mix_seed.argtypes = [C.c_uint64]
    L.orc_philox4x32_10.argtypes = [vp, vp, vp]
    L.orc_philox4x32.argtypes = [C.c_int, vp, vp, vp]
    L.orc_pcg4d.argtypes = [vp, vp]
    L.orc_set_counter_rng.argtypes = [C.c_int]
    L.orc_get_counter_rng.restype = C.c_int
    L.orc_set_pad_mode.argtypes = [C.c_int]
    L.orc_get_pad_mode.restype = C.c_int
    L.orc_camera_setup.argtypes = [C.POINTER(CameraParams), C.POINTER(Camera)]
    L.orc_make_world_spheres.restype = C.c_uint32
    L.orc_make_world_spheres.argtypes = [C.POINTER(WorldDef), vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp,
                                         C.c_uint32]
    L.orc_pack_rgba.restype = C.c_uint32
    L.orc_pack_rgba.argtypes = [vp]
    L.orc_sphere_intersect.restype = C.c_int
    L.orc_sphere_intersect.argtypes = [vp, C.c_float, vp, vp, C.c_double, C.c_double, vp]
    L.orc_world_intersect.restype = C.c_int
    L.orc_world_intersect.argtypes = [vp, C.c_uint32, vp, vp, vp, vp]
    L.orc_scatter.restype = C.c_int
    L.orc_scatter.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp]
    L.orc_render_pixels_mt.restype = C.c_int
    L.orc_render_pixels_mt.argtypes = [C.POINTER(Camera), vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32,
                                       vp, vp, C.POINTER(Counters)]
    L.orc_render_rect_counter.restype = C.c_int
    L.orc_render_rect_counter.argtypes = [C.POINTER(Camera), vp, C.c_uint32, vp, C.c_uint32, C.c_uint64, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(Counters), C.c_int]
    L.orc_render_rect_counter_bvh.restype = C.c_int
    L.orc_render_rect_counter_bvh.argtypes = [C.POINTER(Camera), vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp,
                                              C.c_uint32, vp, C.c_uint32, C.c_float, C.c_float, C.c_uint64,
                                              C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp,
                                              C.POINTER(Counters), C.c_int]
    L.orc_bench_mt.restype = C.c_double
    L.orc_bench_mt.argtypes = [C.POINTER(Camera), vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp,
                               C.POINTER(C.c_uint64)]
    _lib = L
    return L


def set_counter_rng(kind):
    """Block function of the counter stream: 10 / 7 = Philox4x32 rounds, 0 = pcg4d (must match the product's RTMI_RNG)."""
    lib().orc_set_counter_rng(int(kind))


def get_counter_rng():
    return int(lib().orc_get_counter_rng())


def set_pad_mode(mode):
    """Box pad of the instrumented walk: 0 farthest-centre class pad (rounds 1-3), 1 bounded by the segment's reach (the
    product's), 2 none (not exact; measurements only)."""
    lib().orc_set_pad_mode(int(mode))


_tile_entries = None  # (kept alive while the oracle points at it)


def set_tile_entries(entries):
    """Camera-ray entries of the instrumented BVH walk: the (tiles_y, tiles_x) uint32 table of the product's
    rtmi_tile_entries_build, or None = every walk from the root."""
    global _tile_entries
    L = lib()
    L.orc_set_tile_entries.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_set_tile_entries.restype = None
    if entries is None:
        L.orc_set_tile_entries(None, 0)
        _tile_entries = None
    else:
        _tile_entries = np.ascontiguousarray(entries, dtype=np.uint32)
        L.orc_set_tile_entries(_tile_entries.ctypes.data_as(C.c_void_p), _tile_entries.shape[1])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def camera_params(aspect_ratio=16.0 / 9.0, image_width=1200, samples_per_pixel=100, max_depth=50, vertical_fov=20.0,
                  defocus_angle=0.6, focus_distance=10.0, lookfrom=(13.0, 2.0, 3.0), lookat=(0.0, 0.0, 0.0),
                  world_up=(0.0, 1.0, 0.0)):
    """S-RTOW camera: the JSON camera of data/config/world.config.json:2-25 with aspect 16/9."""
    p = CameraParams()
    p.aspect_ratio = aspect_ratio
    p.image_width = image_width
    p.samples_per_pixel = samples_per_pixel
    p.max_depth = max_depth
    p.vertical_fov = vertical_fov
    p.defocus_angle = defocus_angle
    p.focus_distance = focus_distance
    p.lookfrom[:] = lookfrom
    p.lookat[:] = lookat
    p.world_up[:] = world_up
    return p


def camera_setup(params):
    cam = Camera()
    lib().orc_camera_setup(C.byref(params), C.byref(cam))
    return cam


# the four listed objects of data/config/world.config.json:43-112
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
    wd.diffuse_material_treshold = diffuse
    wd.metal_material_treshold = metal
    return wd


def fixed_arrays(fixed):
    objs = np.zeros(len(fixed), OBJECT_DTYPE)
    mats = np.zeros(len(fixed), MATERIAL_DTYPE)
    for i, (c, r, (k, p)) in enumerate(fixed):
        objs[i] = (0, c, r, i)
        mats[i] = (k, p)
    return objs, mats


def make_world_spheres(seed=12345, wd=None, fixed=None, metal_args_right_to_left=True):
    """S-RTOW(seed): restated generator (src/ray.tracer.core.cc:99-149)."""
    wd = wd or world_def()
    fobjs, fmats = fixed_arrays(RTOW_FIXED if fixed is None else fixed)
    cap = len(fobjs) + max(0, wd.a_max - wd.a_min) * max(0, wd.b_max - wd.b_min)
    objs = np.zeros(cap, OBJECT_DTYPE)
    mats = np.zeros(cap, MATERIAL_DTYPE)
    n = lib().orc_make_world_spheres(C.byref(wd), _ptr(fobjs), _ptr(fmats), len(fobjs), seed,
                                     1 if metal_args_right_to_left else 0, _ptr(objs), _ptr(mats), cap)
    return objs[:n].copy(), mats[:n].copy()


def render_pixels_mt(cam, objs, mats, mt_seed, xy, counters=False):
    xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
    n = len(xy)
    rgb = np.zeros((n, 3), np.float32)
    rgba = np.zeros(n, np.uint32)
    ctr = Counters()
    rc = lib().orc_render_pixels_mt(C.byref(cam), _ptr(objs), len(objs), _ptr(mats), len(mats), mt_seed, _ptr(xy), n,
                                    _ptr(rgb), _ptr(rgba), C.byref(ctr) if counters else None)
    assert rc == 0
    return (rgb, rgba, ctr.as_dict()) if counters else (rgb, rgba)


def render_rect_counter(cam, objs, mats, seed, x0, y0, x1, y1, nthreads=1, counters=False, bvh=None):
    """bvh = dict(nodes, slots, pad_classes, pad_eps, pad_floor) switches to the instrumented BVH walk."""
    w, h = x1 - x0, y1 - y0
    rgb = np.zeros((h, w, 3), np.float32)
    rgba = np.zeros((h, w), np.uint32)
    ctr = Counters()
    cp = C.byref(ctr) if counters else None
    if bvh is None:
        rc = lib().orc_render_rect_counter(C.byref(cam), _ptr(objs), len(objs), _ptr(mats), len(mats), seed, x0, y0,
                                           x1, y1, _ptr(rgb), _ptr(rgba), cp, nthreads)
    else:
        nodes = np.ascontiguousarray(bvh["nodes"])
        slots = np.ascontiguousarray(bvh["slots"], dtype=np.uint32)
        set_tile_entries(bvh.get("entries"))  # (the camera-ray entries of the scene the tree came from, if it has any)
        starts = bvh.get("walk_starts")  # (and the walk starts of its scattered rays; way records sit behind the tree's nodes)
        L = lib()
        L.orc_set_walk_starts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_set_walk_starts.restype = None
        if starts is not None:
            starts = np.ascontiguousarray(starts, dtype=np.uint32)
            L.orc_set_walk_starts(_ptr(starts), _ptr(slots), len(slots), len(objs))
        pc = np.ascontiguousarray(bvh["pad_classes"], dtype=np.float32).reshape(-1, 8)
        assert nodes.dtype.itemsize == 64
        rc = lib().orc_render_rect_counter_bvh(C.byref(cam), _ptr(objs), len(objs), _ptr(mats), len(mats),
                                               _ptr(nodes), len(nodes), _ptr(slots), len(slots), _ptr(pc), len(pc),
                                               bvh["pad_eps"], bvh["pad_floor"], seed, x0, y0, x1, y1, _ptr(rgb),
                                               _ptr(rgba), cp, nthreads)
        set_tile_entries(None)
        L.orc_set_walk_starts(None, None, 0, 0)
    assert rc == 0, rc
    return (rgb, rgba, ctr.as_dict()) if counters else (rgb, rgba)


def bench_mt(cam, objs, mats, mt_seed=12345, stride=8, nthreads=1):
    """Reference-shaped CPU baseline; returns (seconds, samples)."""
    samples = C.c_uint64(0)
    secs = lib().orc_bench_mt(C.byref(cam), _ptr(objs), len(objs), _ptr(mats), len(mats), mt_seed, stride, nthreads,
                              None, C.byref(samples))
    return secs, int(samples.value)
