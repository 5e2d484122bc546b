"""The C++ host mirror (raytracing.cpp_amd/host/rtmi_host.hpp): compiled with g++ against librtmi.so and driven the
way the reference's host would drive its render core (scene file -> default_setup -> tile/row rendering)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp", "host_mirror_check.cpp")


def _world_json(path, camera, **kw):
    """A scene file in the reference's schema (data/config/world.config.json)."""
    doc = {
        "camera": camera, "a_min": -11, "a_max": 11, "b_min": -11, "b_max": 11,
        "center": [0.2, 0.9, 0.2], "center_offset": [4.0, 0.2, 0.0], "center_dist_treshold": 0.9,
        "diffuse_material_treshold": 0.8, "metal_material_treshold": 0.95,
        "objects": [
            [{"center": [0.0, -1000.0, 0.0], "radius": 1000.0}, {"material_def": "AlbedoMatDef", "albedo": [0.5, 0.5, 0.5]}],
            [{"center": [0.0, 1.0, 0.0], "radius": 1.0}, {"material_def": "DielectricMatDef", "refindex": 1.5}],
            [{"center": [-4.0, 1.0, 0.0], "radius": 1.0},
             {"material_def": "AlbedoMatDef", "albedo": [0.4000000059604645, 0.20000000298023224, 0.10000000149011612]}],
            [{"center": [4.0, 1.0, 0.0], "radius": 1.0},
             {"material_def": "MetallicMatDef", "albedo": [0.699999988079071, 0.6000000238418579, 0.5], "fuzzines": 0.25}],
        ],
    }
    doc.update(kw)
    json.dump(doc, open(path, "w"), indent=2)


def _camera(width, spp, depth):
    return {"aspect_ratio": 16.0 / 9.0, "image_width": width, "samples_per_pixel": spp, "max_depth": depth,
            "vertical_fov": 20.0, "defocus_angle": 0.6, "focus_distance": 10.0, "lookfrom": [13.0, 2.0, 3.0],
            "lookat": [0.0, 0.0, 0.0], "world_up": [0.0, 1.0, 0.0]}


@pytest.fixture(scope="module")
def exe(pkg, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "host_mirror_check")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(libdir, "host"), CPP, "-o", out, "-L", libdir, "-lrtmi", "-lpthread",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return out


def _fixed_from_json():
    return [((0.0, -1000.0, 0.0), 1000.0, (0, (0.5, 0.5, 0.5, 0.0))), ((0.0, 1.0, 0.0), 1.0, (2, (1.5, 0.0, 0.0, 0.0))),
            ((-4.0, 1.0, 0.0), 1.0, (0, (0.4, 0.2, 0.1, 0.0))), ((4.0, 1.0, 0.0), 1.0, (1, (0.7, 0.6, 0.5, 0.25)))]


def test_scene_file_to_records_matches_oracle(exe, ob, tmp_path):
    cfg = str(tmp_path / "world.config.json")
    _world_json(cfg, _camera(1200, 100, 50))
    out = subprocess.run([exe, "setup", cfg, "12345"], check=True, capture_output=True, text=True).stdout.split("\n")
    cam_hex = out[0].split()[1]
    n_obj, obj_hex = out[1].split()[1:3]
    n_mat, mat_hex = out[2].split()[1:3]
    ocam = ob.camera_setup(ob.camera_params())
    assert bytes.fromhex(cam_hex) == bytes(ocam)
    objs, mats = ob.make_world_spheres(12345, ob.world_def(), _fixed_from_json())
    assert int(n_obj) == int(n_mat) == 488
    assert bytes.fromhex(obj_hex) == objs.tobytes() and bytes.fromhex(mat_hex) == mats.tobytes()
    assert out[3] == "mat3kind 1"


def test_scene_file_with_the_references_shipped_values(exe, tmp_path):
    """A scene file carrying the values of the reference's data/config/world.config.json (aspect 1.7, 8 spp, depth 8,
    fourth big sphere Lambertian): same schema, same tagged-union encoding."""
    cfg = str(tmp_path / "world.config.json")
    cam = _camera(1200, 8, 8)
    cam["aspect_ratio"] = 1.7
    _world_json(cfg, cam)
    doc = json.load(open(cfg))
    doc["objects"][3][1] = {"material_def": "AlbedoMatDef", "albedo": [0.699999988079071, 0.6000000238418579, 0.5]}
    json.dump(doc, open(cfg, "w"))
    out = subprocess.run([exe, "setup", cfg, "1"], check=True, capture_output=True, text=True).stdout.split("\n")
    cam = np.frombuffer(bytes.fromhex(out[0].split()[1]), dtype=np.uint32)
    assert cam[0] == 1200 and cam[1] == int(np.float32(1200.0) / np.float32(1.7))
    assert out[1].split()[1] == "488" and out[3] == "mat3kind 0"


def test_malformed_scene_file_is_an_error_not_a_crash(exe, tmp_path):
    bad = str(tmp_path / "bad.json")
    open(bad, "w").write('{"camera": {"aspect_ratio": 1.5}')
    r = subprocess.run([exe, "setup", bad, "1"], capture_output=True, text=True)
    assert r.returncode == 9 and "world config" in r.stderr
    r = subprocess.run([exe, "setup", str(tmp_path / "missing.json"), "1"], capture_output=True, text=True)
    assert r.returncode == 9 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_default_setup_and_render_through_the_mirror(exe, ob, tmp_path):
    cfg = str(tmp_path / "world.config.json")
    _world_json(cfg, _camera(96, 4, 20))
    out = str(tmp_path / "frame.bin")
    subprocess.run([exe, "render", cfg, "12345", "77", out], check=True)
    raw = open(out, "rb").read()
    w, h = np.frombuffer(raw[:8], np.uint32)
    rgb = np.frombuffer(raw[8:8 + w * h * 12], np.float32).reshape(h, w, 3)
    rgba = np.frombuffer(raw[8 + w * h * 12:], np.uint32).reshape(h, w)
    ocam = ob.camera_setup(ob.camera_params(image_width=96, samples_per_pixel=4, max_depth=20))
    objs, mats = ob.make_world_spheres(12345, ob.world_def(), _fixed_from_json())
    want, want8 = ob.render_rect_counter(ocam, objs, mats, 77, 0, 0, 96, ocam.img_height, nthreads=8)
    assert rgb.tobytes() == want.tobytes() and np.array_equal(rgba, want8)


@pytest.mark.gpu
def test_attached_devices_frame_and_multi_worker_tracer(exe, ob, tmp_path):
    """RayTracingCore::attach_devices + raytrace_frame (rtmi_frame_*: row-block shards, gather, scanline order) and the
    RayTracer with one worker per attached device, on the one GPU of the box (n = 1: no communicator); the program itself
    checks both against raytrace_rows of the single scene, this test checks the frame against the oracle."""
    cfg = str(tmp_path / "world.config.json")
    _world_json(cfg, _camera(104, 4, 20))
    out = str(tmp_path / "frame8.bin")
    r = subprocess.run([exe, "frame", cfg, "12345", "78", out, "1"], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    # three shards and three workers on the one device (rehearsal hook: copies instead of RCCL)
    r = subprocess.run([exe, "frame", cfg, "12345", "78", out, "-3"], capture_output=True, text=True)
    assert r.returncode == 0 and "devices 3" in r.stdout, (r.returncode, r.stdout, r.stderr)
    ocam = ob.camera_setup(ob.camera_params(image_width=104, samples_per_pixel=4, max_depth=20))
    objs, mats = ob.make_world_spheres(12345, ob.world_def(), _fixed_from_json())
    _, want8 = ob.render_rect_counter(ocam, objs, mats, 78, 0, 0, 104, ocam.img_height, nthreads=8)
    got = np.frombuffer(open(out, "rb").read(), np.uint32).reshape(ocam.img_height, 104)
    assert np.array_equal(got, want8)


def _ssbo(path):
    raw = open(path, "rb").read()
    sw, sh = np.frombuffer(raw[:8], np.uint32)
    # RayTracedImageSSBOData is alignas(16): {width, height, pad, pad?}: the pixel array starts at offset 8
    pix = np.frombuffer(raw[8:8 + int(sw) * int(sh) * 4], np.uint32).reshape(sh, sw)
    return int(sw), int(sh), pix


def test_display_contract_centres_and_flips(exe, tmp_path):
    """RayTracedImageDisplay::write_pixel (image.display.cc:108-117): image centred on the surface, GL lower-left origin."""
    out = str(tmp_path / "ssbo.bin")
    sw, sh, iw, ih = 40, 30, 24, 10
    subprocess.run([exe, "display", str(sw), str(sh), str(iw), str(ih), out], check=True)
    gw, gh, pix = _ssbo(out)
    assert (gw, gh) == (sw, sh)
    tx, ty = (sw - iw) // 2, (sh - ih) // 2
    want = np.zeros((sh, sw), np.uint32)
    for y in range(ih):
        for x in range(iw):
            want[sh - 1 - (y + ty), x + tx] = 0xff000000 | (y << 12) | x
    assert np.array_equal(pix, want)


@pytest.mark.gpu
def test_job_system_adapter_streams_row_blocks_into_the_display(exe, ob, tmp_path):
    cfg = str(tmp_path / "world.config.json")
    _world_json(cfg, _camera(160, 4, 20))
    out, ppm = str(tmp_path / "ssbo.bin"), str(tmp_path / "frame.ppm")
    r = subprocess.run([exe, "stream", cfg, "12345", "5", "200", "120", out, ppm], check=True, capture_output=True,
                       text=True)
    assert "progress_steps" in r.stdout
    ocam = ob.camera_setup(ob.camera_params(image_width=160, samples_per_pixel=4, max_depth=20))
    objs, mats = ob.make_world_spheres(12345, ob.world_def(), _fixed_from_json())
    _, want8 = ob.render_rect_counter(ocam, objs, mats, 5, 0, 0, 160, ocam.img_height, nthreads=8)
    sw, sh, pix = _ssbo(out)
    tx, ty = (200 - 160) // 2, (120 - ocam.img_height) // 2
    img = pix[::-1][ty:ty + ocam.img_height, tx:tx + 160]  # undo the y-flip and the centring
    assert np.array_equal(img, want8)
    hdr, body = open(ppm, "rb").read().split(b"\n255\n", 1)
    assert hdr == b"P6\n160 90"
    rgb = np.frombuffer(body, np.uint8).reshape(90, 160, 3)
    assert np.array_equal(rgb[..., 0], (want8 & 0xff).astype(np.uint8)) and np.array_equal(rgb[..., 2], ((want8 >> 16) & 0xff).astype(np.uint8))


def test_example_program_builds(pkg, tmp_path):
    """raytracing.cpp_amd/host/render_ppm.cpp (scene file -> frame -> PPM) compiles and links against librtmi.so."""
    libdir = os.path.dirname(pkg.LIB_PATH)
    out = str(tmp_path / "render_ppm")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(libdir, "host"),
                    os.path.join(libdir, "host", "render_ppm.cpp"), "-o", out, "-L", libdir, "-lrtmi", "-lpthread",
                    f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([out], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
