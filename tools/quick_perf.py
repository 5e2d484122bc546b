"""Ad-hoc GPU perf/parity probe (not part of the test-suite): BVH vs brute-force on full frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()

def run(width, spp, depth, accel, seed=7, reps=2, stats=False, leaf=0):
    cam = pkg.camera_setup(pkg.camera_params(image_width=width, samples_per_pixel=spp, max_depth=depth))
    objs, mats = pkg.make_world_spheres(12345)
    with pkg.Scene(cam, objs, mats, accel=accel, collect_stats=stats, leaf_size=leaf) as sc:
        best = 1e9
        for _ in range(reps):
            t = time.time()
            rgb, rgba = sc.render_rows(0, cam.img_height, seed)
            best = min(best, time.time() - t)
        kms = sc.last_kernel_ms()
        st = sc.stats() if stats else None
    n = cam.img_width * cam.img_height * spp
    print(f"accel={accel} leaf={leaf} {cam.img_width}x{cam.img_height}x{spp}spp d{depth}: wall {best*1e3:.1f} ms, kernel {kms:.1f} ms, "
          f"{n/kms/1e3:.1f} Msamples/s (kernel)", st if st else "", flush=True)
    return rgb, rgba

if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    a, _ = run(w, spp, 50, pkg.ACCEL_BVH)
    b, _ = run(w, spp, 50, pkg.ACCEL_BRUTE, reps=1)
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
    print("pixels where BVH != brute:", int(d.sum()), "of", d.size, " rmse", float(np.sqrt(np.mean((a.astype(np.float64)-b)**2))))
    run(w, spp, 50, pkg.ACCEL_BVH, stats=True, reps=1)
    for leaf in (1, 2, 3, 4):
        run(w, spp, 50, pkg.ACCEL_BVH, leaf=leaf, reps=1)
    for thr in (8, 16, 32, 48, 64):
        os.environ["RTMI_WAIT_THRESH"] = str(thr)
        print("wait_thresh", thr, end=": ")
        run(w, spp, 50, pkg.ACCEL_BVH, reps=1)
