"""Ad-hoc A/B perf probe on the GPU: kernel versions / thresholds, interleaved rounds in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()

def make(width, spp, depth, env, accel=None, stats=False):
    for k, v in env.items():
        os.environ[k] = str(v)
    cam = pkg.camera_setup(pkg.camera_params(image_width=width, samples_per_pixel=spp, max_depth=depth))
    objs, mats = pkg.make_world_spheres(12345)
    sc = pkg.Scene(cam, objs, mats, accel=accel or pkg.ACCEL_BVH, collect_stats=stats)
    return sc, cam, env

if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    variants = {"w56": dict()}
    for wt in (44, 50, 60):
        variants[f"w{wt}"] = dict(RTMI_WAIT_THRESH=wt)
    for dw in (48, 62):
        variants[f"dw{dw}"] = dict(RTMI_DRAIN_WAIT=dw)
    scenes = {k: make(w, spp, 50, v) for k, v in variants.items()}
    ref = None
    res = {k: [] for k in scenes}
    for rnd in range(3):
        for k, (sc, cam, env) in scenes.items():
            for kk, vv in env.items():
                os.environ[kk] = str(vv)
            rgb, _ = sc.render_rows(0, cam.img_height, 7)
            for kk in env:
                os.environ.pop(kk, None)
            res[k].append(sc.last_kernel_ms())
            if ref is None:
                ref = rgb
            elif rnd == 0:
                d = int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
                print(f"{k}: pixels differing from v1: {d}", flush=True)
    n = scenes["w56"][1].img_width * scenes["w56"][1].img_height * spp
    for k, v in res.items():
        print(f"{k:10s} kernel ms min {min(v):8.2f} med {sorted(v)[len(v)//2]:8.2f}  -> {n/min(v)/1e3:8.1f} Msamples/s", flush=True)
