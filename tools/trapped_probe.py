"""What a segment of a path trapped inside the ground sphere costs when a wave has nothing else to do: the RTOW scene seen
from inside its ground sphere (every path is 50 such segments) against the normal view, trace-kernel time per segment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
objs, mats = pkg.make_world_spheres(12345)
for name, kw in (("normal view", dict(image_width=1920, samples_per_pixel=64, max_depth=50)),
                 ("from inside the ground sphere", dict(image_width=1920, samples_per_pixel=8, max_depth=50, lookfrom=(0.0, -500.0, 0.0),
                                                        lookat=(0.0, -1000.0, 30.0), defocus_angle=0.0))):
    cam = pkg.camera_setup(pkg.camera_params(**kw))
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, collect_stats=True) as sc:
        sc.render_rows(0, cam.img_height, 7, rgba=False)
        st = sc.stats(reset=True)
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
        ms = []
        for _ in range(3):
            sc.render_rows(0, cam.img_height, 7, rgba=False)
            ms.append(sc.last_kernel_ms())
    seg = st["segments"]
    print(f"{name}: {min(ms):.2f} ms, {seg / st['samples']:.2f} segments per sample, {st['node_tests'] / seg:.1f} box tests and "
          f"{st['sphere_tests'] / seg:.2f} sphere tests per segment, {seg / min(ms) / 1e6:.1f} G segments/s", flush=True)
