"""A/B of differently built libraries (diagnostic -D builds) on config 3: usage lib_ab.py name=flags:tuning ...
e.g. lib_ab.py base=: wpe7=-DRTMI_WPE=7:block_lanes=896   (--build compiles them first)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
specs = []
for a in sys.argv[1:]:
    if a.startswith("--"):
        continue
    name, rest = a.split("=", 1)
    flags, _, tun = rest.partition(":")
    tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in tun.split(",") if kv}
    specs.append((name, [f for f in flags.split(" ") if f], tuning))


def lib_path(name):
    return os.path.join(os.path.dirname(pkg.LIB_PATH), f"librtmi_ab_{name}.so")


if "--build" in sys.argv:
    for name, flags, _ in specs:
        cmd = ["/opt/rocm/bin/hipcc"] + pkg.HIPCC_FLAGS + flags + ["-I", "include", "-o", lib_path(name)] + pkg.CSRC + ["-ldl"]
        subprocess.run(cmd, check=True)
        print("built", lib_path(name))
    sys.exit(0)
w, spp = 1920, 512
GRID = "--grid" in sys.argv
CORNELL = "--cornell" in sys.argv
objs = mats = None
ref = None
rows = []
for rnd in range(2):
    for name, flags, tuning in specs:
        pkg._lib = None
        pkg.LIB_PATH = lib_path(name)
        if CORNELL:
            objs, mats, kw = pkg.workloads.cornell_like()
            kw.update(samples_per_pixel=256)
            cam = pkg.camera_setup(pkg.camera_params(**kw))
        elif GRID:
            objs, mats, kw = pkg.workloads.big_grid(316)
            kw.update(samples_per_pixel=32)
            cam = pkg.camera_setup(pkg.camera_params(**kw))
        else:
            if objs is None:
                objs, mats = pkg.make_world_spheres(12345)
            cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
        with pkg.Scene(cam, objs, mats, tuning=tuning or None) as sc:  # RTMI_ACCEL_AUTO
            if rnd == 0:
                print(f"      {name}: {sc.launch_info()}")
            ms = []
            for _ in range(2):
                rgb, _ = sc.render_rows(0, cam.img_height, 7, rgba=False)
                ms.append(sc.last_kernel_ms())
        if ref is None:
            ref = rgb
        d = int((np.nan_to_num(rgb).view(np.uint32) != np.nan_to_num(ref).view(np.uint32)).any(axis=-1).sum())
        print(f"round {rnd} {name:10s} {tuning}: {min(ms):8.2f} ms, pixels differing from the first variant {d}", flush=True)
