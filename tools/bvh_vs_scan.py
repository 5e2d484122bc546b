"""Stress of the BVH exactness argument: for many generated worlds the BVH walk and the linear scan (both on the GPU)
must produce bit-identical frames.  usage: bvh_vs_scan.py [n_worlds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
from tests.scenes import random_spheres, big_grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for i in range(n):
    if i % 3 == 0:
        objs, mats = pkg.make_world_spheres(1000 + i)
        kw = dict(image_width=800, samples_per_pixel=24, max_depth=50)
        name = f"rtow generator seed {1000 + i}"
    elif i % 3 == 1:
        objs, mats = random_spheres(300 + 40 * i, seed=i, extent=6.0 + i)
        kw = dict(image_width=640, samples_per_pixel=16, max_depth=30)
        name = f"random_spheres({300 + 40 * i})"
    elif i % 6 == 2:
        objs, mats, kw = big_grid(24 + 4 * (i % 40), seed=i)
        kw.update(image_width=480, samples_per_pixel=8)
        name = f"big_grid({24 + 4 * (i % 40)})"
    else:  # radii over seven decades, camera far away or inside the big spheres: the regime of the sqrt pad bound
        rng = np.random.default_rng(i)
        m = 200
        objs = np.zeros(m, pkg.OBJECT_DTYPE)
        mats = np.zeros(m, pkg.MATERIAL_DTYPE)
        objs["center"] = rng.normal(0.0, 30.0, (m, 3)).astype(np.float32)
        objs["radius"] = (10.0 ** rng.uniform(-3.0, 1.5, m)).astype(np.float32)
        objs["radius"][:3] = (1.0e3, 1.0e4, 3.0e2)
        objs["center"][:3] = ((0.0, -1.0e3, 0.0), (0.0, 0.0, -1.2e4), (350.0, 0.0, 0.0))
        objs["material"] = np.arange(m)
        mats["kind"] = rng.integers(0, 3, m)
        mats["p"] = rng.uniform(0.2, 1.0, (m, 4)).astype(np.float32)
        mats["p"][mats["kind"] == 2, 0] = 1.5
        far = float(rng.choice([20.0, 2.0e3, 3.0e4]))
        kw = dict(image_width=400, samples_per_pixel=8, max_depth=40, vertical_fov=50.0, defocus_angle=0.0,
                  focus_distance=10.0, lookfrom=(far, 0.3 * far + 1.0, 0.5 * far), lookat=(0.0, 0.0, 0.0), world_up=(0.0, 1.0, 0.0))
        name = f"radii 1e-3..1e4, camera at {far:g}"
    cam = pkg.camera_setup(pkg.camera_params(**kw))
    frames = []
    for accel in (pkg.ACCEL_BVH, pkg.ACCEL_BRUTE):
        with pkg.Scene(cam, objs, mats, accel=accel) as s:
            rgb, _ = s.render_rows(0, cam.img_height, 77 + i)
        frames.append(rgb)
    a, b = frames
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    d = int((~same).any(axis=-1).sum())
    bad += d
    print(f"{name:28s} {len(objs):6d} objects {cam.img_width}x{cam.img_height}x{kw['samples_per_pixel']}: pixels differing {d}", flush=True)
print("TOTAL differing pixels:", bad)
sys.exit(1 if bad else 0)
