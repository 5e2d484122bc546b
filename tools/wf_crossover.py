"""Where does the queue-scheduled kernel (kernel=2) beat the round-based one?  Random-sphere scenes of growing size and
two bounce limits, both kernels, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
for n, depth in ((4, 50), (12, 50), (40, 50), (120, 50), (487, 50), (12, 200), (120, 200)):
    objs, mats = pkg.workloads.random_spheres(n, seed=n, extent=4.0 + n ** 0.5)
    cam = pkg.camera_setup(pkg.camera_params(image_width=1280, samples_per_pixel=64, max_depth=depth))
    out = []
    for k in (1, 2):
        with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, tuning=dict(kernel=k, defer_mode=-1, wf_refill=40)) as sc:
            ms = []
            for _ in range(2):
                sc.render_rows(0, cam.img_height, 7, rgba=False)
                ms.append(sc.last_kernel_ms())
            b = sc.bvh()
        out.append(min(ms))
    print(f"{n + 1:4d} spheres ({len(b['nodes'])} nodes) depth {depth}: round-based {out[0]:7.2f} ms  queue-scheduled {out[1]:7.2f} ms  ratio {out[0] / out[1]:.2f}", flush=True)
