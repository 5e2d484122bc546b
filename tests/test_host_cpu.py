"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol of include/rtmi.h, the
host logic (camera derivation, scene generator, BVH builder, row sharding) matches the oracle, and the compute entry
points fail loudly without a GPU.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    declared = set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pkg.EXPORTS), declared ^ set(pkg.EXPORTS)
    lib = pkg.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.rtmi_version()


def test_struct_layouts_match_reference_records(pkg):
    # HittableObject 24 B (object.defs.hpp:30-57), Material 20 B (material.defs.hpp:49-55),
    # 14 PODs of RayTracingCore = 100 B (core.hpp:19-32), CameraParameters 60 B (camera.parameters.hpp:6-17)
    assert pkg.OBJECT_DTYPE.itemsize == 24 and pkg.MATERIAL_DTYPE.itemsize == 20
    assert C.sizeof(pkg.Camera) == 100 and C.sizeof(pkg.CameraParams) == 60
    assert pkg.BVH_NODE_DTYPE.itemsize == 64


@pytest.mark.parametrize("kw", [
    dict(), dict(image_width=400), dict(image_width=1920, samples_per_pixel=512),
    dict(aspect_ratio=1.7, samples_per_pixel=8, max_depth=8),  # data/config/world.config.json:3-8
    dict(aspect_ratio=1.0, image_width=800, vertical_fov=40.0, defocus_angle=0.0, lookfrom=(0, 0, 18)),
    dict(image_width=801, vertical_fov=90.0, defocus_angle=10.0, focus_distance=3.4, lookfrom=(-2, 2, 1),
         lookat=(0, 0, -1)),  # WorldDefinition defaults, core.cc:68-79
])
def test_camera_setup_matches_oracle_bit_for_bit(pkg, ob, kw):
    a = pkg.camera_setup(pkg.camera_params(**kw))
    b = ob.camera_setup(ob.camera_params(**kw))
    assert bytes(a) == bytes(b)


def test_world_generator_matches_oracle_bit_for_bit(pkg, ob):
    """Product: std::mt19937 + std::uniform_real_distribution<double>, as the reference; oracle: plain-C restatement."""
    for seed, kw in ((12345, {}), (1, {}), (777, dict(a_min=-3, a_max=5, b_min=0, b_max=2, diffuse=0.5, metal=0.7))):
        o1, m1 = pkg.make_world_spheres(seed, pkg.world_def(**kw))
        o2, m2 = ob.make_world_spheres(seed, ob.world_def(**kw))
        assert o1.tobytes() == o2.tobytes() and m1.tobytes() == m2.tobytes()
    o, _ = pkg.make_world_spheres(12345)
    assert len(o) == 488


def test_compute_entry_points_fail_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cam = pkg.camera_setup(pkg.camera_params(image_width=64))
    objs, mats = pkg.make_world_spheres(1)
    with pytest.raises(pkg.RtmiError) as e:
        pkg.Scene(cam, objs, mats)
    assert e.value.code == pkg.RTMI_ERR_HIP


def test_scene_create_argument_errors(pkg):
    """Bad arguments are reported as RTMI_ERR_BAD_ARG with a message, never by aborting (main.cc's worker setup
    failures just return; material.defs.hpp:104 asserts on bad handles)."""
    cam = pkg.camera_setup(pkg.camera_params(image_width=64))
    objs, mats = pkg.make_world_spheres(1)
    bad = objs.copy()
    bad["material"][3] = len(mats)
    for o, m, c in ((bad, mats, cam), (objs, mats[:10], cam)):
        with pytest.raises(pkg.RtmiError) as e:
            pkg.Scene(c, o, m)
        assert e.value.code == pkg.RTMI_ERR_BAD_ARG and "material" in str(e.value)
    badk = objs.copy()
    badk["kind"][0] = 7
    with pytest.raises(pkg.RtmiError) as e:
        pkg.Scene(cam, badk, mats)
    assert e.value.code == pkg.RTMI_ERR_BAD_ARG
    assert pkg.lib().rtmi_scene_create(None, None, 0, None, 0, None, None) == pkg.RTMI_ERR_BAD_ARG
    assert pkg.lib().rtmi_render_rows(None, 0, 1, 0, None, None) == pkg.RTMI_ERR_BAD_ARG
    assert pkg.lib().rtmi_render_rect(None, 0, 0, 8, 8, 0, None, None) == pkg.RTMI_ERR_BAD_ARG  # the tile-granular entries of 0.5
    assert pkg.lib().rtmi_render_rect_device(None, 0, 0, 8, 8, 0, None, None, None) == pkg.RTMI_ERR_BAD_ARG
    assert b"null scene" in pkg.lib().rtmi_last_error()
    info = pkg.LaunchInfo()
    info.struct_size = C.sizeof(pkg.LaunchInfo)
    assert pkg.lib().rtmi_scene_get_launch_info(None, C.byref(info)) == pkg.RTMI_ERR_BAD_ARG
    assert C.sizeof(pkg.LaunchInfo) == 88  # 40 in version 0.3, 52 in 0.4, 68 in 0.5: the struct grows at its end, struct_size tells the library what fits
    pkg.lib().rtmi_scene_destroy(None)  # no-op


@pytest.mark.parametrize("field,value", [("center", float("nan")), ("center", float("inf")), ("radius", float("nan")),
                                         ("radius", float("inf"))])
def test_non_finite_spheres_are_rejected(pkg, field, value):
    """A NaN centre breaks the ordering of the BVH builder's sorts, an infinite radius gives NaN boxes the walk can never
    enter while the linear scan still tests the sphere: both entry points refuse such objects, for either accel."""
    cam = pkg.camera_setup(pkg.camera_params(image_width=64))
    objs, mats = pkg.make_world_spheres(1)
    bad = objs.copy()
    if field == "center":
        bad["center"][17][1] = value
    else:
        bad["radius"][17] = value
    with pytest.raises(pkg.RtmiError) as e:
        pkg.bvh_build(bad)
    assert e.value.code == pkg.RTMI_ERR_BAD_ARG and "non-finite" in str(e.value)
    for accel in (pkg.ACCEL_BVH, pkg.ACCEL_BRUTE):
        with pytest.raises(pkg.RtmiError) as e:
            pkg.Scene(cam, bad, mats, accel=accel)
        assert e.value.code == pkg.RTMI_ERR_BAD_ARG and "non-finite" in str(e.value)


def test_frame_create_argument_errors(pkg):
    """rtmi_frame_create checks its device list before touching any device."""
    cam = pkg.camera_setup(pkg.camera_params(image_width=64))
    objs, mats = pkg.make_world_spheres(1)
    L = pkg.lib()
    h = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 0)
    assert L.rtmi_frame_create(C.byref(cam), None, 0, None, 0, None, None, 1, 8, C.byref(h)) == pkg.RTMI_ERR_BAD_ARG
    assert L.rtmi_frame_create(C.byref(cam), None, 0, None, 0, None, devs, 0, 8, C.byref(h)) == pkg.RTMI_ERR_BAD_ARG
    assert L.rtmi_frame_create(C.byref(cam), None, 0, None, 0, None, devs, 17, 8, C.byref(h)) == pkg.RTMI_ERR_BAD_ARG
    assert L.rtmi_frame_render(None, 0, None, None) == pkg.RTMI_ERR_BAD_ARG
    assert L.rtmi_frame_get_timing(None, None) == pkg.RTMI_ERR_BAD_ARG
    L.rtmi_frame_destroy(None)  # no-op
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pkg.RtmiError) as e:
            pkg.Frame(cam, objs, mats, devices=(0,))
        assert e.value.code == pkg.RTMI_ERR_HIP  # no device: fails loudly, no fallback


def test_tuning_struct_layout_and_unknown_knobs(pkg):
    assert C.sizeof(pkg.Tuning) == 92 and C.sizeof(pkg.SceneOptions) == 40  # (64 in 0.4, 76 in 0.5: the struct grows at its end)
    with pytest.raises(KeyError):
        pkg.make_tuning(no_such_knob=1)
    header = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    body = header[header.index("typedef struct rtmi_tuning {"):header.index("} rtmi_tuning;")]
    fields = re.findall(r"^\s+u?int32_t\s+([a-z_0-9]+)(?:\[\d+\])?;", body, flags=re.M)
    assert fields == [n for n, _ in pkg.Tuning._fields_]
    # the library reads no environment variables
    for src in sorted(os.listdir(os.path.join(ROOT, "raytracing.cpp_amd", "csrc"))):
        assert "getenv" not in open(os.path.join(ROOT, "raytracing.cpp_amd", "csrc", src)).read(), src


def _check_bvh(pkg, objs, leaf, passes=0):
    b = pkg.bvh_build(objs, leaf, passes)
    n = len(objs)
    assert sorted(b["slots"].tolist()) == list(range(n))  # every object in exactly one leaf slot
    nodes = b["nodes"]
    seen = np.zeros(n, bool)

    def box_of(ref):
        """returns (lo, hi) of everything below ref, checking containment on the way; also max depth"""
        if ref & 0x80000000:
            first, cnt = ref & 0xffffff, (ref >> 24) & 0x7f
            assert 1 <= cnt <= max(leaf, 1) or n == 0
            lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
            for s in range(first, first + cnt):
                o = objs[b["slots"][s]]
                assert not seen[b["slots"][s]]
                seen[b["slots"][s]] = True
                c, r = o["center"].astype(np.float64), abs(float(o["radius"]))
                lo, hi = np.minimum(lo, c - r), np.maximum(hi, c + r)
            return lo, hi, 0
        nd = nodes[ref]
        lo, hi, depth = np.full(3, np.inf), np.full(3, -np.inf), 0
        for k in range(2):
            clo, chi, d = box_of(int(nd["child"][k]))
            blo = nd["ctr"][k].astype(np.float64) - nd["half"][k].astype(np.float64)
            bhi = nd["ctr"][k].astype(np.float64) + nd["half"][k].astype(np.float64)
            assert np.all(blo <= clo) and np.all(bhi >= chi)  # stored box contains the subtree (outward rounding)
            lo, hi, depth = np.minimum(lo, clo), np.maximum(hi, chi), max(depth, d)
        return lo, hi, depth + 1

    if n:
        _, _, depth = box_of(b["root_ref"])
        assert seen.all() and depth == b["depth"]
    # (a scene whose spheres all sit in leaves peeled off the top of the tree has no box to pad: no class)
    assert b["pad_eps"] > 0 and b["pad_floor"] > 0 and 0 <= len(b["pad_classes"]) <= 4 or n == 0
    return b


def test_bvh_builder_invariants(pkg, rtow):
    from tests.scenes import cornell_like, random_spheres, three_spheres
    b = _check_bvh(pkg, rtow[0], 2)
    assert b["depth"] <= 16 and len(b["nodes"]) < 488
    for leaf in (1, 4):
        _check_bvh(pkg, rtow[0], leaf)
    _check_bvh(pkg, three_spheres()[0], 2)
    _check_bvh(pkg, cornell_like()[0], 2)
    _check_bvh(pkg, random_spheres(3000, seed=3)[0], 2)
    one = rtow[0][:1].copy()
    b = _check_bvh(pkg, one, 2)
    assert len(b["nodes"]) == 0 and b["root_ref"] == 0x80000000 | (1 << 24)
    dup = np.concatenate([rtow[0][4:5]] * 9)  # identical spheres: splits must still terminate
    dup["material"] = 0
    _check_bvh(pkg, dup, 2)
    # the reinsertion post-pass (rtmi_tuning::bvh_passes: n - 1 passes, 1 = the plain top-down tree) keeps every invariant,
    # never deepens the tree (the walk's stack is sized from its depth) and never grows the sum of the boxes' areas
    def area(b):
        h = b["nodes"]["half"].astype(np.float64)
        return float((h[..., 0] * h[..., 1] + h[..., 1] * h[..., 2] + h[..., 2] * h[..., 0]).sum())
    for objs in (rtow[0], random_spheres(3000, seed=3)[0], cornell_like()[0], dup):
        plain = _check_bvh(pkg, objs, 2, passes=1)
        for passes in (2, 9):
            opt = _check_bvh(pkg, objs, 2, passes=passes)
            assert opt["depth"] <= plain["depth"] and len(opt["nodes"]) == len(plain["nodes"])
            assert area(opt) <= area(plain) * (1 + 1e-6)


def test_camera_tile_entries_keep_the_walk_exact(pkg, ob, rtow):
    """rtmi_tuning::cam_entry (0.6): camera rays start their walk at their 8x8 tile's entry -- the lowest common ancestor of every
    sphere the tile's beam (lens disk x tile rectangle) can meet -- or do not walk at all.  The table is host geometry
    (rtmi_tile_entries_build, no device): the oracle's instrumented walk FOLLOWING it must give the frame of the oracle's linear
    scan, float for float: S-RTOW under several lenses and cameras (inside the glass sphere, grazing the ground, far away), random
    worlds of the differential fuzz (negative radii, huge and tiny scales, nested spheres), and it must skip work."""
    fuzz_world = pkg.workloads.fuzz_world
    cases = []
    for kw in (dict(image_width=160, samples_per_pixel=3), dict(image_width=160, samples_per_pixel=2, defocus_angle=0.0),
               dict(image_width=96, samples_per_pixel=2, defocus_angle=8.0, focus_distance=4.0),
               dict(image_width=96, samples_per_pixel=2, lookfrom=(0.0, 1.0, 0.3), lookat=(4.0, 1.0, 0.0), vertical_fov=70.0),
               dict(image_width=96, samples_per_pixel=2, lookfrom=(6.0, 0.21, 6.0), lookat=(0.0, 0.2, 0.0), defocus_angle=2.0),
               dict(image_width=64, samples_per_pixel=2, lookfrom=(900.0, 300.0, 500.0), vertical_fov=2.0, focus_distance=1000.0)):
        cases.append((*rtow, dict(max_depth=12, **kw), 2))
    rng = np.random.default_rng(66)
    for case in range(40):
        objs, mats, kw = fuzz_world(rng, case)
        cases.append((objs, mats, dict(kw, max_depth=min(kw["max_depth"], 8)), int(rng.choice([1, 2, 4]))))
    skipped = walked = 0
    for objs, mats, kw, leaf in cases:
        cam, ocam = pkg.camera_setup(pkg.camera_params(**kw)), ob.camera_setup(ob.camera_params(**kw))
        W, H = cam.img_width, cam.img_height
        want, _ = ob.render_rect_counter(ocam, objs, mats, 9, 0, 0, W, H, nthreads=8)
        bvh = pkg.bvh_build(objs, leaf)
        bvh = dict(bvh, nodes=bvh["nodes"].view(ob.BVH_NODE_DTYPE))
        _, _, c0 = ob.render_rect_counter(ocam, objs, mats, 9, 0, 0, W, H, nthreads=8, counters=True, bvh=bvh)
        ent = pkg.tile_entries_build(cam, objs, leaf)
        assert ent.shape == ((H + 7) // 8, (W + 7) // 8)
        got, _, c1 = ob.render_rect_counter(ocam, objs, mats, 9, 0, 0, W, H, nthreads=8, counters=True, bvh=dict(bvh, entries=ent))
        a, b = np.nan_to_num(got, nan=-1.0).view(np.uint32), np.nan_to_num(want, nan=-1.0).view(np.uint32)
        assert np.array_equal(a, b), (kw, int((a != b).any(axis=-1).sum()))
        assert c1["segments"] == c0["segments"] and c1["node_tests"] <= c0["node_tests"]
        skipped += c0["node_tests"] - c1["node_tests"]
        walked += c0["node_tests"]
    assert skipped > 0.05 * walked  # (S-RTOW at 1080p: camera rays 8.9 -> 2.0 node trips each, tools/wave_replay.py)


def test_walk_starts_keep_the_walk_exact(pkg, ob, rtow):
    """rtmi_tuning::walk_start (0.6, trees that stay in HBM): the walk of a ray scattered off a sphere starts in that sphere's own
    leaf, the siblings hanging off the path above it pre-loaded on the stack as way records (two levels a record, in the node
    format).  Host data (rtmi_walk_starts_build, no device): the oracle's walk FOLLOWING it must give the oracle's linear scan float
    for float -- S-RTOW, a jittered grid, the fuzz worlds (negative radii, nested spheres, one-sphere worlds, every leaf size) --
    with fewer box tests; every record is well formed (way records only behind the tree's nodes, every sibling of the path in
    exactly one of them)."""
    from tests.scenes import big_grid
    cases = [(*rtow, dict(image_width=120, samples_per_pixel=3, max_depth=16), 2)]
    o, m, kw = big_grid(40, seed=2)
    cases.append((o, m, dict(kw, image_width=96, samples_per_pixel=3, max_depth=16), 4))
    rng = np.random.default_rng(77)
    for case in range(40):
        o, m, kw = pkg.workloads.fuzz_world(rng, case)
        cases.append((o, m, dict(kw, max_depth=min(kw["max_depth"], 8)), int(rng.choice([1, 2, 3, 4]))))
    saved = total = 0
    for objs, mats, kw, leaf in cases:
        cam, ocam = pkg.camera_setup(pkg.camera_params(**kw)), ob.camera_setup(ob.camera_params(**kw))
        W, H = cam.img_width, cam.img_height
        want, _ = ob.render_rect_counter(ocam, objs, mats, 5, 0, 0, W, H, nthreads=8)
        bvh = pkg.bvh_build(objs, leaf)
        nodes, n_tree, rec = pkg.walk_starts_build(objs, leaf)
        assert n_tree == len(bvh["nodes"]) and nodes[:n_tree].tobytes() == bvh["nodes"].tobytes()
        assert rec.shape == (len(objs), 16) and (rec[:, 1] <= 12).all()
        ways = rec[:, 2:][np.arange(14)[None, :] < rec[:, 1:2]]
        assert ((ways >= n_tree) & (ways < len(nodes))).all()  # way records sit behind the tree's own nodes
        base = dict(bvh, nodes=bvh["nodes"].view(ob.BVH_NODE_DTYPE))
        _, _, c0 = ob.render_rect_counter(ocam, objs, mats, 5, 0, 0, W, H, nthreads=8, counters=True, bvh=base)
        got, _, c1 = ob.render_rect_counter(ocam, objs, mats, 5, 0, 0, W, H, nthreads=8, counters=True,
                                            bvh=dict(base, nodes=nodes.view(ob.BVH_NODE_DTYPE), walk_starts=rec))
        a, b = np.nan_to_num(got, nan=-1.0).view(np.uint32), np.nan_to_num(want, nan=-1.0).view(np.uint32)
        assert np.array_equal(a, b), (kw, int((a != b).any(axis=-1).sum()))
        assert c1["segments"] == c0["segments"]
        saved += c0["node_tests"] - c1["node_tests"]
        total += c0["node_tests"]
    assert saved > 0.1 * total


@pytest.mark.parametrize("height,block,world", [(1080, 8, 1), (1080, 8, 8), (675, 8, 2), (675, 8, 4), (225, 16, 8),
                                                (7, 8, 4), (54, 8, 3)])
def test_row_block_sharding_covers_every_row_once(pkg, height, block, world):
    plan = pkg.RowShardPlan(height, block, world)
    owner = np.full(height, -1)
    for r in range(world):
        y_first, n_blocks, rows = plan.shard(r)
        got = 0
        for k in range(n_blocks):
            y0 = y_first + k * world * block
            y1 = min(height, y0 + block)
            assert np.all(owner[y0:y1] == -1)
            owner[y0:y1] = r
            got += y1 - y0
        assert got == rows <= plan.max_rows
    assert np.all(owner >= 0)
    # load balance of interleaving: no rank has more than one block more than another
    rows = [s[2] for s in plan.shards]
    assert max(rows) - min(rows) <= block
    # de-interleave map: gathered (rank-major, padded) row -> scanline
    stacked = np.full((world * plan.max_rows,), -1)
    for r in range(world):
        y_first, n_blocks, _ = plan.shard(r)
        loc = 0
        for k in range(n_blocks):
            y0 = y_first + k * world * block
            for y in range(y0, min(height, y0 + block)):
                stacked[r * plan.max_rows + loc] = y
                loc += 1
    assert np.array_equal(stacked[plan.index], np.arange(height))


def test_header_is_plain_c_and_every_prototype_is_exported(pkg, tmp_path):
    """include/rtmi.h is the C-ABI: it compiles as C99 (-pedantic -Werror) with the record sizes the reference's layouts give,
    and every function it declares is a symbol of librtmi.so and in the binding's export list."""
    import subprocess
    src = tmp_path / "hc.c"
    src.write_text('#include "rtmi.h"\n'
                   'int main(void) { return (sizeof(rtmi_object) == 24 && sizeof(rtmi_material) == 20 && sizeof(rtmi_camera) == 100\n'
                   '                         && sizeof(rtmi_bvh_node) == 64 && sizeof(rtmi_launch_info) == 88 && sizeof(rtmi_tuning) == 92) ? 0 : 1; }\n')
    exe = tmp_path / "hc"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    header = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    protos = set(re.findall(r"^(?:int|void|const char\*)\s+(rtmi_[a-z0-9_]+)\(", header, flags=re.M))
    assert protos == set(pkg.EXPORTS), protos ^ set(pkg.EXPORTS)
    lib = pkg.lib()
    for name in protos:
        getattr(lib, name)
