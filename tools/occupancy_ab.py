import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
objs, mats = pkg.make_world_spheres(12345)
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=512, max_depth=50))
for rnd in range(2):
    for tun in ({}, dict(block_lanes=640), dict(block_lanes=704), dict(block_lanes=512), dict(block_lanes=576), dict(block_lanes=832)):
        with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, tuning=tun or None) as sc:
            li = sc.launch_info()
            ms = []
            for _ in range(2):
                sc.render_rows(0, cam.img_height, 7, rgba=False); ms.append(sc.last_kernel_ms())
        print(rnd, tun, li['blocks_per_cu'], li['lds_bytes'], f"{min(ms):.2f} ms", flush=True)
