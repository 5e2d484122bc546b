"""Crossover of the two BVH kernels against the mean path length: the enclosed box of config 5 under growing bounce
limits (segments per sample from ~4 to ~80), and the RTOW scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
cases = [("cornell", d) for d in (4, 8, 16, 32, 64, 200)] + [("rtow", 50)]
for scene, depth in cases:
    if scene == "cornell":
        objs, mats, kw = pkg.workloads.cornell_like()
        kw.update(image_width=800, samples_per_pixel=64, max_depth=depth)
    else:
        objs, mats = pkg.make_world_spheres(12345)
        kw = dict(image_width=1200, samples_per_pixel=100, max_depth=depth)
    cam = pkg.camera_setup(pkg.camera_params(**kw))
    out = []
    for k in (1, 2):
        with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, tuning=dict(kernel=k, defer_mode=-1, wf_refill=40)) as sc:
            ms = []
            for _ in range(2):
                sc.render_rows(0, cam.img_height, 7, rgba=False)
                ms.append(sc.last_kernel_ms())
        out.append(min(ms))
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, collect_stats=True, tuning=dict(kernel=1)) as sc:
        sc.render_rows(0, min(cam.img_height, 64), 7, rgba=False)
        st = sc.stats()
    print(f"{scene} depth {depth:3d}: {st['segments'] / st['samples']:6.2f} segments/sample, {st['node_tests'] / st['segments']:5.1f} box tests/segment; "
          f"round-based {out[0]:8.2f} ms  queue-scheduled {out[1]:8.2f} ms  ratio {out[0] / out[1]:.2f}", flush=True)
