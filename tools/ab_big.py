import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import os, sys
sys.path.insert(0, %r)
import rtmi_loader
pkg = rtmi_loader.load()
from tests.scenes import big_grid
objs, mats, kw = big_grid(316)
kw.update(image_width=1920, samples_per_pixel=4)
cam = pkg.camera_setup(pkg.camera_params(**kw))
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
    ms = []
    for _ in range(2):
        sc.render_rows(0, cam.img_height, 3, rgba=False)
        ms.append(sc.last_kernel_ms())
print(min(ms))
''' % root
for top in sys.argv[1:]:
    env = dict(os.environ, RTMI_TOP_NODES=top)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    print("top nodes", top, out.stdout.strip() or out.stderr[-400:], flush=True)
