"""Differential fuzz of the GPU path against the oracle: random worlds (1-90 spheres, negative radii, scales 1e-2 .. 1e3, all
three materials, bounce limits 1-120, defocus on and off; every fourth world wide -- up to 600 spheres over hundreds of radii, where
the reach-bounded box pad of round 4 is the library's choice) through the walk and the scan, each with the scene in LDS, forced into
HBM (top of the tree staged / not staged / 9 nodes staged), run-length encoded chains, whole-pixel work items, either pad rule forced,
the plain and the post-optimised tree; every float must match (NaN = NaN).
usage: fuzz_vs_oracle.py [seed] [cases]   (logs of the round-4 runs: profiles/r04_fuzz_vs_oracle.txt)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import rtmi_loader
pkg = rtmi_loader.load()
from oracle import binding as ob
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    wide = case % 4 == 3
    n = int(rng.integers(100, 600)) if wide else int(rng.integers(1, 90))
    n_mats = int(rng.integers(1, n + 1))
    objs = np.zeros(n, pkg.OBJECT_DTYPE); mats = np.zeros(n_mats, pkg.MATERIAL_DTYPE)
    scale = float(10.0 ** rng.uniform(-2, 3))
    for m in range(n_mats):
        k = int(rng.integers(0, 3))
        if k == 0: mats[m] = (0, (*rng.uniform(0.0, 1.2, 3), 0.0))
        elif k == 1: mats[m] = (1, (*rng.uniform(0.3, 1.0, 3), float(rng.uniform(0.0, 1.5))))
        else: mats[m] = (2, (float(rng.uniform(0.6, 2.2)), 0, 0, 0))
    objs["center"] = (rng.normal(0, 3.0, (n, 3)) * scale).astype(np.float32)
    objs["radius"] = (10.0 ** rng.uniform(-1.5, 0.8, n) * scale * rng.choice([1.0, 1.0, 1.0, -1.0], n)).astype(np.float32)
    if wide:  # a field hundreds of radii wide, flat or not
        objs["center"] = (rng.uniform(-1.0, 1.0, (n, 3)) * (float(rng.uniform(30, 300)), float(rng.choice([0.5, 30.0])), float(rng.uniform(30, 300))) * scale).astype(np.float32)
        objs["radius"] = (10.0 ** rng.uniform(-1.0, 0.0, n) * scale).astype(np.float32)
    objs["material"] = rng.integers(0, n_mats, n)
    if rng.random() < 0.5:
        objs["center"][0] = (0, -1000.0 * scale - scale, 0); objs["radius"][0] = 1000.0 * scale
    depth = int(rng.choice([1, 3, 8, 20, 50, 120]))
    lf = tuple(float(v) for v in rng.normal(0, 6.0, 3) * scale)
    kw = dict(aspect_ratio=1.0, image_width=int(rng.choice([17, 32, 40])), samples_per_pixel=int(rng.choice([1, 4, 9])), max_depth=depth,
              vertical_fov=float(rng.uniform(20, 90)), defocus_angle=float(rng.choice([0.0, 0.5, 3.0])), focus_distance=float(5 * scale),
              lookfrom=lf, lookat=(0.0, 0.0, 0.0), world_up=(0.0, 1.0, 0.0))
    cam, ocam = pkg.camera_setup(pkg.camera_params(**kw)), ob.camera_setup(ob.camera_params(**kw))
    want, want8 = ob.render_rect_counter(ocam, objs, mats, case, 0, 0, ocam.img_width, ocam.img_height, nthreads=8)
    for accel in (pkg.ACCEL_BVH, pkg.ACCEL_BRUTE):
        tunings = (None, dict(force_hbm_scene=1), dict(chain_mode=1), dict(chunk_samples=-1))
        if accel == pkg.ACCEL_BVH:
            tunings += (dict(pad_mode=2), dict(pad_mode=1, bvh_passes=1), dict(pad_mode=2, force_hbm_scene=1, lds_top_nodes=1),
                        dict(pad_mode=1, force_hbm_scene=1, lds_top_nodes=10, bvh_passes=9))
        for tun in tunings:
            with pkg.Scene(cam, objs, mats, accel=accel, tuning=tun) as s:
                rgb, rgba = s.render_rows(0, cam.img_height, case)
            same = (rgb.view(np.uint32) == want.view(np.uint32)) | (np.isnan(rgb) & np.isnan(want))
            if not same.all() or not np.array_equal(rgba, want8):
                bad += 1
                print("MISMATCH case", case, "accel", accel, tun, "pixels", int((~same).any(-1).sum()), "n", n, "scale", scale, "depth", depth, flush=True)
    if case % 100 == 99:
        print("..", case + 1, "worlds,", bad, "mismatches,", round(time.time() - t0, 1), "s", flush=True)
print("cases done, mismatches:", bad, "in", round(time.time() - t0, 1), "s", flush=True)
