"""A/B of scheduling knobs (rtmi_tuning) on one workload: interleaved rounds in one process, min and median of the
trace-kernel time.  usage: ab_tuning.py <width> <spp> [scene] -- knob=value[,knob=value] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()

args = sys.argv[1:]
w, spp = int(args[0]), int(args[1])
scene = args[2] if len(args) > 2 and "=" not in args[2] and args[2] != "--" else "rtow"
variants = [dict()]
for a in args:
    if "=" in a:
        variants.append({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")})
if scene == "rtow":
    objs, mats = pkg.make_world_spheres(12345)
    cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
elif scene == "cornell":
    objs, mats, kw = pkg.workloads.cornell_like()
    kw.update(image_width=w, samples_per_pixel=spp)
    cam = pkg.camera_setup(pkg.camera_params(**kw))
else:
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(image_width=w, samples_per_pixel=spp)
    cam = pkg.camera_setup(pkg.camera_params(**kw))
scenes = [pkg.Scene(cam, objs, mats, tuning=v or None) for v in variants]  # (RTMI_ACCEL_AUTO: the library's own choice)
res = [[] for _ in variants]
ref = None
for rnd in range(3):
    for i, sc in enumerate(scenes):
        rgb, _ = sc.render_rows(0, cam.img_height, 7, rgba=False)
        res[i].append(sc.last_kernel_ms())
        if rnd == 0:
            if ref is None:
                ref = rgb
            else:
                d = int((np.nan_to_num(rgb).view(np.uint32) != np.nan_to_num(ref).view(np.uint32)).any(axis=-1).sum())
                if d:
                    print(f"!! {variants[i]}: {d} pixels differ from the default build", flush=True)
n = cam.img_width * cam.img_height * spp
for v, r in zip(variants, res):
    print(f"{str(v or 'default'):60s} min {min(r):8.2f} ms  med {sorted(r)[1]:8.2f} ms  {n / min(r) / 1e3:8.1f} Msamples/s", flush=True)
