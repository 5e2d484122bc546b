"""Leaf size of the BVH against trace-kernel time.  usage: leaf_sweep.py [rtow|grid] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
which = sys.argv[1] if len(sys.argv) > 1 else "rtow"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if which == "grid":
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(samples_per_pixel=spp)
else:
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=1920, samples_per_pixel=spp, max_depth=50)
cam = pkg.camera_setup(pkg.camera_params(**kw))
for leaf in ((2, 3, 4) if which == "grid" else (1, 2, 3, 4)):  # (kMaxLeafSize = 4)
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, leaf_size=leaf) as sc:
        ms = []
        for _ in range(3):
            sc.render_rows(0, cam.img_height, 7, rgba=False)
            ms.append(sc.last_kernel_ms())
        print(f"{which} leaf {leaf}: {min(ms):8.2f} ms  {sc.launch_info()}", flush=True)
