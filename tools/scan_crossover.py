"""Where the BVH walk overtakes the reference's linear scan: trace-kernel time of both on random-sphere scenes of growing
size (plus the config-5 box), 800x450x32 spp x 50 bounces.  Decides the RTMI_ACCEL_AUTO threshold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
cases = [("cornell(7)", pkg.workloads.cornell_like)]
for n in (3, 6, 10, 14, 20, 28, 40, 60):
    cases.append((f"random({n + 1})", (lambda n=n: (*pkg.workloads.random_spheres(n, seed=5), dict(image_width=800, samples_per_pixel=32, max_depth=50)))))
for name, gen in cases:
    objs, mats, kw = gen()
    kw = dict(kw)
    if "cornell" in name:
        kw.update(image_width=400, samples_per_pixel=64)
    cam = pkg.camera_setup(pkg.camera_params(**kw))
    ms, frames = {}, {}
    for label, accel in (("scan", pkg.ACCEL_BRUTE), ("bvh", pkg.ACCEL_BVH)):
        with pkg.Scene(cam, objs, mats, accel=accel) as sc:
            best = 1e9
            for _ in range(3):
                rgb, _ = sc.render_rows(0, cam.img_height, 7, rgba=False)
                best = min(best, sc.last_kernel_ms())
        ms[label], frames[label] = best, rgb
    same = np.array_equal(np.nan_to_num(frames["scan"]).view(np.uint32), np.nan_to_num(frames["bvh"]).view(np.uint32))
    print(f"{name:14s} scan {ms['scan']:8.2f} ms   bvh {ms['bvh']:8.2f} ms   bvh/scan {ms['bvh'] / ms['scan']:.3f}   same frame {same}", flush=True)
