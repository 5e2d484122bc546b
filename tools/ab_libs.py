"""A/B of differently compiled libraries in separate processes on the same box (kernel ms via the library's events).
usage: ab_libs.py <width> <spp> [rtow|grid|cornell] lib1.so lib2.so ...   (file names under raytracing.cpp_amd/)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
w, spp = args[0], args[1]
scene = "rtow"
libs = args[2:]
if libs and not libs[0].endswith(".so"):
    scene, libs = libs[0], libs[1:]
code = '''
import os, sys
sys.path.insert(0, %r)
import rtmi_loader
pkg = rtmi_loader.load()
pkg.LIB_PATH = sys.argv[1]
w, spp, scene = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
if scene == "rtow":
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=w, samples_per_pixel=spp, max_depth=50)
elif scene == "cornell":
    objs, mats, kw = pkg.workloads.cornell_like()
    kw.update(image_width=w, samples_per_pixel=spp)
else:
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(image_width=w, samples_per_pixel=spp)
cam = pkg.camera_setup(pkg.camera_params(**kw))
with pkg.Scene(cam, objs, mats) as sc:
    ms = []
    for _ in range(3):
        sc.render_rows(0, cam.img_height, 7, rgba=False)
        ms.append(sc.last_kernel_ms())
print(min(ms))
''' % root
for rnd in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "raytracing.cpp_amd", lib), w, spp, scene], capture_output=True, text=True)
        print(rnd, scene, lib, out.stdout.strip() or out.stderr[-300:], flush=True)
