"""Ad-hoc: launch tail of a 1080p x 512 spp frame vs the sample-chunk split (RTMI_CHUNK; 0 = off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
import time
objs, mats = pkg.make_world_spheres(12345)
w, spp = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 512)
cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
scs = {}
for chunk in (0, 16, 32, 64, 128):
    os.environ["RTMI_CHUNK"] = str(chunk)
    scs[chunk] = pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH)
res, ref = {}, None
for rnd in range(2):
    for chunk, sc in scs.items():
        t = time.time()
        rgb, _ = sc.render_rows(0, cam.img_height, 7, rgba=False)
        wall = (time.time() - t) * 1e3
        res.setdefault(chunk, []).append((sc.last_kernel_ms(), wall))
        if ref is None:
            ref = rgb
        elif rnd == 0:
            print(f"chunk {chunk}: pixels differing from unsplit: {int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())}", flush=True)
n = cam.img_width * cam.img_height * spp
for chunk in scs:
    k = min(v[0] for v in res[chunk]); wl = min(v[1] for v in res[chunk])
    print(f"chunk {chunk:4d}: trace kernel {k:8.1f} ms, call wall {wl:8.1f} ms -> {n/k/1e3:8.1f} / {n/wl/1e3:8.1f} Msamples/s", flush=True)
