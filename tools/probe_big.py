"""Ad-hoc GPU probe of BASELINE configs 4 and 5 (not part of the test-suite)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
from tests.scenes import big_grid, cornell_like
from oracle import binding as ob

which = sys.argv[1]
if which == "grid":
    n_side = int(sys.argv[2]); width = int(sys.argv[3]); spp = int(sys.argv[4])
    objs, mats, kw = big_grid(n_side)
    kw.update(image_width=width, samples_per_pixel=spp)
else:
    width = int(sys.argv[2]); spp = int(sys.argv[3])
    objs, mats, kw = cornell_like()
    kw.update(image_width=width, samples_per_pixel=spp)
cam = pkg.camera_setup(pkg.camera_params(**kw))
ocam = ob.camera_setup(ob.camera_params(**kw))
t = time.time()
sc = pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, collect_stats=True)
print(f"{len(objs)} objects, scene_create {time.time()-t:.2f} s", flush=True)
t = time.time()
rgb, rgba = sc.render_rows(0, cam.img_height, 3)
dt = time.time() - t
n = cam.img_width * cam.img_height * spp
st = sc.stats()
print(f"{cam.img_width}x{cam.img_height}x{spp}: {dt*1e3:.1f} ms wall, kernel {sc.last_kernel_ms():.1f} ms, {n/sc.last_kernel_ms()/1e3:.1f} Msamples/s; "
      f"seg/sample {st['segments']/st['samples']:.2f} node tests/seg {st['node_tests']/st['segments']:.1f} sphere tests/seg {st['sphere_tests']/st['segments']:.2f}; mean {rgb.mean():.4f}", flush=True)
rng = np.random.default_rng(0)
bad = 0
t = time.time()
for x, y in zip(rng.integers(0, cam.img_width, 12), rng.integers(0, cam.img_height, 12)):
    want, _ = ob.render_rect_counter(ocam, objs, mats, 3, int(x), int(y), int(x) + 1, int(y) + 1)
    bad += int(want[0, 0].tobytes() != rgb[y, x].tobytes())
print(f"oracle spot check (12 pixels, linear scan): {bad} differ ({time.time()-t:.1f} s)", flush=True)
