"""A/B of (library, env) pairs in separate processes on the same box."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, spp = sys.argv[1], sys.argv[2]
pairs = [a.split(":") for a in sys.argv[3:]]  # lib:ENV=V,ENV2=V2
code = '''
import os, sys
sys.path.insert(0, %r)
import rtmi_loader
pkg = rtmi_loader.load()
pkg.LIB_PATH = sys.argv[1]
cam = pkg.camera_setup(pkg.camera_params(image_width=int(sys.argv[2]), samples_per_pixel=int(sys.argv[3]), max_depth=50))
objs, mats = pkg.make_world_spheres(12345)
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
    ms = []
    for _ in range(3):
        sc.render_rows(0, cam.img_height, 7, rgba=False)
        ms.append(sc.last_kernel_ms())
print(min(ms))
''' % root
for rnd in range(2):
    for lib, envs in pairs:
        env = dict(os.environ)
        for kv in envs.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "raytracing.cpp_amd", lib), w, spp], capture_output=True, text=True, env=env)
        print(rnd, lib, envs, out.stdout.strip() or out.stderr[-300:], flush=True)
