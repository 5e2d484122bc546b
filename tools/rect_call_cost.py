"""What one rtmi_render_rect call costs (the reference's queue unchanged: one call per 8x8 RayTracingWorkPackage) against
row blocks and the whole frame: S-RTOW 1920x1080 x 512 spp, host-pointer entries (launch + resolve + D2H + sync per call).
usage: rect_call_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
objs, mats = pkg.make_world_spheres(12345)
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=512, max_depth=50))
W, H = cam.img_width, cam.img_height
rng = np.random.default_rng(3)
with pkg.Scene(cam, objs, mats) as sc:
    sc.render_rect(0, 0, 8, 8, 7)  # first call: buffers
    for tw, th, n in ((8, 8, 400), (64, 16, 200), (W, 8, 60), (W, 64, 10)):
        xs = rng.integers(0, max(1, (W - tw) // 8 + 1), n) * 8
        ys = rng.integers(0, (H - th) // 8, n) * 8
        t0 = time.perf_counter()
        for x, y in zip(xs, ys):
            sc.render_rect(int(x), int(y), int(x) + tw, int(y) + th, 7, rgb=False)
        dt = (time.perf_counter() - t0) / n
        calls = -(-W // tw) * -(-H // th)
        print(f"{tw:4d} x {th:2d} rectangles: {dt * 1e3:8.3f} ms per call (random positions), {tw * th * 512 / dt / 1e6:8.1f} Msamples/s, "
              f"a 1080p frame = {calls} calls = {calls * dt:7.2f} s", flush=True)
    t0 = time.perf_counter()
    sc.render_rows(0, H, 7, rgb=False)
    dt = time.perf_counter() - t0
    print(f"whole frame in one call: {dt * 1e3:8.1f} ms, {W * H * 512 / dt / 1e6:8.1f} Msamples/s", flush=True)
