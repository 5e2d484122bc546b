import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
w, spp = int(sys.argv[1]), int(sys.argv[2])
cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
objs, mats = pkg.make_world_spheres(12345)
tun = {}
for a in sys.argv[3:]:
    if "=" in a:
        k, v = a.split("=")
        tun[k] = int(v)
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, collect_stats="--stats" in sys.argv, tuning=tun) as sc:
    for _ in range(2):
        sc.render_rows(0, cam.img_height, 7, rgba=False)
    print("kernel ms", sc.last_kernel_ms(), sc.stats() if "--stats" in sys.argv else "")
