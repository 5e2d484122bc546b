"""Differential fuzz of the GPU path against the oracle: random worlds (1-90 spheres, negative radii, scales 1e-2 .. 1e3, all
three materials, bounce limits 1-120, defocus on and off; every fourth world wide -- up to 600 spheres over hundreds of radii, where
the reach-bounded box pad of round 4 is the library's choice) through the walk and the scan, each with the scene in LDS, forced into
HBM (top of the tree staged / not staged / 9 nodes staged), run-length encoded chains, whole-pixel work items, either pad rule forced,
the plain and the post-optimised tree, cost-ordered tiles in three sequential bands, and (round 6) camera entries and walk starts on / off /
forced in both memory layouts; every float must match (NaN = NaN).
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
    objs, mats, kw = pkg.workloads.fuzz_world(rng, case)
    n, scale, depth = len(objs), float(kw['focus_distance']) / 5.0, kw['max_depth']
    cam, ocam = pkg.camera_setup(pkg.camera_params(**kw)), ob.camera_setup(ob.camera_params(**kw))
    want, want8 = ob.render_rect_counter(ocam, objs, mats, case, 0, 0, ocam.img_width, ocam.img_height, nthreads=8)
    for accel in (pkg.ACCEL_BVH, pkg.ACCEL_BRUTE):
        tunings = (None, dict(force_hbm_scene=1), dict(chain_mode=1), dict(chunk_samples=-1), dict(tile_order=2, bands=3, chunk_samples=2))
        if accel == pkg.ACCEL_BVH:
            tunings += (dict(pad_mode=2), dict(pad_mode=1, bvh_passes=1), dict(pad_mode=2, force_hbm_scene=1, lds_top_nodes=1),
                        dict(pad_mode=1, force_hbm_scene=1, lds_top_nodes=10, bvh_passes=9),
                        # round 6: camera entries (default on in LDS) off, walk starts (default on in HBM) off / without entries / with a
                        # staged block of one and of three levels of way records
                        dict(cam_entry=1), dict(force_hbm_scene=1, walk_start=1), dict(force_hbm_scene=1, cam_entry=1),
                        dict(force_hbm_scene=1, lds_top_nodes=40, wait_thresh=60), dict(force_hbm_scene=1, lds_top_nodes=120, pad_mode=2))
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
