"""Executed instructions per ray segment on views that isolate pieces of a round (rocprofv3's PC sampling is refused on this pool): the S-RTOW scene
  normal  the BASELINE camera                      -- everything, at its usual lane mix
  trapped from inside the ground sphere            -- every segment: set-up + one node trip + the ground hit + a Lambertian bounce + its draws, all lanes busy
  sky     the camera looking straight up           -- every sample: primary ray + set-up + a walk that misses + the sky colour + the record store
Run each view under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES` (tools/instr_budget.sh does, and prints
wave-instructions per 64 segments = per round of a full wave).  usage: instr_budget.py <normal|trapped|sky> [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
view = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
objs, mats = pkg.make_world_spheres(12345)
kw = {"normal": dict(image_width=1920, samples_per_pixel=spp or 64, max_depth=50),
      "trapped": dict(image_width=1920, samples_per_pixel=spp or 8, max_depth=50, lookfrom=(0.0, -500.0, 0.0), lookat=(0.0, -1000.0, 30.0), defocus_angle=0.0),
      "sky": dict(image_width=1920, samples_per_pixel=spp or 128, max_depth=50, lookfrom=(0.0, 3.0, 0.0), lookat=(0.0, 100.0, 1.0), world_up=(0.0, 0.0, 1.0))}[view]
cam = pkg.camera_setup(pkg.camera_params(**kw))
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, collect_stats=True, tuning=dict(tile_order=1)) as sc:
    sc.render_rows(0, cam.img_height, 7, rgba=False)
    st = sc.stats(reset=True)
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, tuning=dict(tile_order=1)) as sc:  # the shipped (non-counting) variant: the dispatch the counters are read from
    sc.render_rows(0, cam.img_height, 7, rgba=False)
    ms = sc.last_kernel_ms()
print(f"VIEW {view} segments {st['segments']} samples {st['samples']} node_tests {st['node_tests']} sphere_tests {st['sphere_tests']} kernel_ms {ms:.3f}", flush=True)
