"""One scene, two renders, for per-dispatch profiling (tools/pmc_per_dispatch.sh).
usage: one_render.py <width> <spp> [scene=rtow|grid|cornell] [lib=<suffix of librtmi_ab_*.so>] [tuning knob=value ...] [--stats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
w, spp = int(sys.argv[1]), int(sys.argv[2])
tun, scene = {}, "rtow"
for a in sys.argv[3:]:
    if a.startswith("scene="):
        scene = a.split("=")[1]
    elif a.startswith("lib="):
        pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), f"librtmi_ab_{a.split('=')[1]}.so")
    elif "=" in a:
        k, v = a.split("=")
        tun[k] = int(v)
if scene == "grid":
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(image_width=w, samples_per_pixel=spp)
elif scene == "cornell":
    objs, mats, kw = pkg.workloads.cornell_like()
    kw.update(image_width=w, samples_per_pixel=spp)
else:
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=w, samples_per_pixel=spp, max_depth=50)
cam = pkg.camera_setup(pkg.camera_params(**kw))
with pkg.Scene(cam, objs, mats, collect_stats="--stats" in sys.argv, tuning=tun) as sc:
    for _ in range(2):
        sc.render_rows(0, cam.img_height, 7, rgba=False)
    print("kernel ms", sc.last_kernel_ms(), sc.launch_info(), sc.stats() if "--stats" in sys.argv else "")
