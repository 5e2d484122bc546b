import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import os, sys
sys.path.insert(0, %r)
import rtmi_loader
pkg = rtmi_loader.load()
pkg.LIB_PATH = sys.argv[1]
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=64, max_depth=50))
objs, mats = pkg.make_world_spheres(12345)
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BRUTE) as sc:
    ms = []
    for _ in range(3):
        sc.render_rows(0, cam.img_height, 7, rgba=False)
        ms.append(sc.last_kernel_ms())
print(min(ms))
''' % root
for rnd in range(2):
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "raytracing.cpp_amd", lib)], capture_output=True, text=True)
        print(rnd, lib, out.stdout.strip() or out.stderr[-300:], flush=True)
