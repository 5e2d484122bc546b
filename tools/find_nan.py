import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
from oracle import binding as ob
objs, mats = pkg.make_world_spheres(12345)
kw = dict(image_width=1920, samples_per_pixel=512, max_depth=50)
cam = pkg.camera_setup(pkg.camera_params(**kw)); ocam = ob.camera_setup(ob.camera_params(**kw))
with pkg.Scene(cam, objs, mats) as sc:
    rgb, _ = sc.render_rows(0, cam.img_height, 2025, rgba=False)
bad = np.argwhere(~np.isfinite(rgb).all(axis=-1))
print("non-finite pixels:", len(bad), bad[:10].tolist())
for y, x in bad[:4]:
    want, _ = ob.render_rect_counter(ocam, objs, mats, 2025, int(x), int(y), int(x) + 1, int(y) + 1)
    print((x, y), "gpu", rgb[y, x], "oracle", want[0, 0], "same bits:", want[0, 0].tobytes() == rgb[y, x].tobytes())
    # which sample? bisect over spp with the oracle
    for spp in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        c2 = ob.camera_setup(ob.camera_params(image_width=1920, samples_per_pixel=spp, max_depth=50))
        w2, _ = ob.render_rect_counter(c2, objs, mats, 2025, int(x), int(y), int(x) + 1, int(y) + 1)
        if not np.isfinite(w2).all():
            print("   first non-finite within the first", spp, "samples (oracle)")
            break
