"""Trace-kernel time against samples per pixel for one or more library builds (fixed costs and tails show as an offset).
usage: spp_sweep.py [rtow|grid|cornell] name[=lib suffix] ...   e.g. spp_sweep.py grid new old   (librtmi_ab_<name>.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
which = sys.argv[1]
here = os.path.dirname(pkg.LIB_PATH)
for name in sys.argv[2:]:
    pkg._lib = None
    pkg.LIB_PATH = os.path.join(here, f"librtmi_ab_{name}.so") if name != "shipped" else os.path.join(here, "librtmi.so")
    out = []
    for spp in (4, 16, 64, 128):
        if which == "grid":
            objs, mats, kw = pkg.workloads.big_grid(316)
            kw.update(samples_per_pixel=spp)
        elif which == "cornell":
            objs, mats, kw = pkg.workloads.cornell_like()
            kw.update(samples_per_pixel=spp)
        else:
            objs, mats = pkg.make_world_spheres(12345)
            kw = dict(image_width=1920, samples_per_pixel=spp, max_depth=50)
        cam = pkg.camera_setup(pkg.camera_params(**kw))
        with pkg.Scene(cam, objs, mats) as sc:
            ms = []
            for _ in range(2):
                sc.render_rows(0, cam.img_height, 7, rgba=False)
                ms.append(sc.last_kernel_ms())
        out.append((spp, min(ms)))
    print(name, " ".join(f"{spp} spp: {ms:.2f} ms ({ms / spp:.3f}/spp)" for spp, ms in out), flush=True)
