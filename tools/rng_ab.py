"""A/B of the counter RNG's block function (RTMI_RNG = 10, 7: Philox4x32 rounds; 0: pcg4d): one library per variant,
each checked against the oracle switched the same way, then timed on configs 2 and 3 (interleaved rounds)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
VARIANTS = (10, 7, 0)


def lib_path(v):
    return os.path.join(os.path.dirname(pkg.LIB_PATH), f"librtmi_rng{v}.so")


if "--build" in sys.argv:
    for v in VARIANTS:
        cmd = ["/opt/rocm/bin/hipcc"] + pkg.HIPCC_FLAGS + [f"-DRTMI_RNG={v}", "-I", "include", "-o", lib_path(v)] + pkg.CSRC + ["-ldl"]
        subprocess.run(cmd, check=True)
        print("built", lib_path(v))
    sys.exit(0)

import ctypes as C
from oracle import binding as ob
results = {}
for v in VARIANTS:
    # one process per variant would be cleaner; ctypes keeps libraries apart by path, so load each into a fresh module state
    pkg._lib = None
    pkg.LIB_PATH = lib_path(v)
    ob.set_counter_rng(v)
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=96, samples_per_pixel=8, max_depth=50)
    cam, ocam = pkg.camera_setup(pkg.camera_params(**kw)), ob.camera_setup(ob.camera_params(**kw))
    want, _ = ob.render_rect_counter(ocam, objs, mats, 5, 0, 0, ocam.img_width, ocam.img_height, nthreads=8)
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
        got, _ = sc.render_rows(0, cam.img_height, 5)
    diff = int((np.nan_to_num(got).view(np.uint32) != np.nan_to_num(want).view(np.uint32)).any(axis=-1).sum())
    times = {}
    for (w, spp) in ((1200, 100), (1920, 512)):
        cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
        with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
            ms = []
            for _ in range(3):
                sc.render_rows(0, cam.img_height, 7, rgba=False)
                ms.append(sc.last_kernel_ms())
        times[(w, spp)] = min(ms)
    mean = float(np.nanmean(got))
    print(f"RTMI_RNG={v}: pixels differing from the oracle {diff}; frame mean {mean:.5f}; "
          + "; ".join(f"{w}x{spp}spp {t:.1f} ms" for (w, spp), t in times.items()), flush=True)
