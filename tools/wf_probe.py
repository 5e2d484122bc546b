"""Ad-hoc GPU probe: queue-scheduled kernel (tuning kernel=2) against the round-based one -- bits, then time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()


def render(width, spp, depth, tuning, reps=1, seed=7, stats=False, scene="rtow"):
    if scene == "rtow":
        objs, mats = pkg.make_world_spheres(12345)
        cam = pkg.camera_setup(pkg.camera_params(image_width=width, samples_per_pixel=spp, max_depth=depth))
    elif scene == "cornell":
        objs, mats, kw = pkg.workloads.cornell_like()
        kw.update(image_width=width, samples_per_pixel=spp, max_depth=depth)
        cam = pkg.camera_setup(pkg.camera_params(**kw))
    else:
        objs, mats, kw = pkg.workloads.big_grid(316)
        kw.update(image_width=width, samples_per_pixel=spp, max_depth=depth)
        cam = pkg.camera_setup(pkg.camera_params(**kw))
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, collect_stats=stats, tuning=tuning) as sc:
        ms = []
        for _ in range(reps):
            rgb, rgba = sc.render_rows(0, cam.img_height, seed)
            ms.append(sc.last_kernel_ms())
        st = sc.stats() if stats else None
    n = cam.img_width * cam.img_height * spp
    return rgb, rgba, min(ms), n, st


def compare(a, b):
    return int((np.nan_to_num(a).view(np.uint32) != np.nan_to_num(b).view(np.uint32)).any(axis=-1).sum())


if __name__ == "__main__":
    cases = sys.argv[1:] or ["small"]
    for case in cases:
        if case == "small":
            for (w, spp, d, scene) in ((64, 8, 50, "rtow"), (160, 16, 50, "rtow"), (48, 32, 200, "cornell")):
                a, a8, ms_a, n, sa = render(w, spp, d, dict(kernel=1), stats=True, scene=scene)
                b, b8, ms_b, _, sb = render(w, spp, d, dict(kernel=2), stats=True, scene=scene)
                print(f"{scene} {w}x{spp}spp d{d}: differing pixels {compare(a, b)} rgba {int((a8 != b8).sum())}  legacy {ms_a:.2f} ms  wf {ms_b:.2f} ms", flush=True)
                print("   stats legacy", sa, "\n   stats wf    ", sb, flush=True)
        elif case == "mid":
            a, _, ms_a, n, _ = render(1200, 100, 50, dict(kernel=1), reps=2)
            b, _, ms_b, _, _ = render(1200, 100, 50, dict(kernel=2), reps=2)
            print(f"config2: differing pixels {compare(a, b)}; legacy {ms_a:.1f} ms ({n/ms_a/1e3:.0f} Ms/s)  wf {ms_b:.1f} ms ({n/ms_b/1e3:.0f} Ms/s)", flush=True)
        elif case == "full":
            b, _, ms_b, n, _ = render(1920, 512, 50, dict(kernel=2), reps=2)
            print(f"config3 wf {ms_b:.1f} ms ({n/ms_b/1e3:.0f} Ms/s)", flush=True)
            a, _, ms_a, n, _ = render(1920, 512, 50, dict(kernel=1), reps=2)
            print(f"config3: differing pixels {compare(a, b)}; legacy {ms_a:.1f} ms ({n/ms_a/1e3:.0f} Ms/s)", flush=True)
        elif case == "others":
            for scene, w, spp, d in (("cornell", 800, 64, 200), ("grid", 1920, 16, 50)):
                a, _, ms_a, n, _ = render(w, spp, d, dict(kernel=1), reps=2, scene=scene)
                b, _, ms_b, _, _ = render(w, spp, d, dict(kernel=2), reps=2, scene=scene)
                c, _, ms_c, _, _ = render(w, spp, d, dict(kernel=2, wf_refill=40), reps=2, scene=scene)
                print(f"{scene} {w} x {spp} spp: differing pixels {compare(a, b)} {compare(a, c)}; legacy {ms_a:.1f} ms ({n/ms_a/1e3:.0f} Ms/s)  "
                      f"wf {ms_b:.1f} ms ({n/ms_b/1e3:.0f} Ms/s)  wf refill 40 {ms_c:.1f} ms ({n/ms_c/1e3:.0f} Ms/s)", flush=True)
        elif case.startswith("sweep"):
            w, spp = (1200, 100) if case == "sweep" else (1920, 256)
            for tun in (dict(kernel=1), dict(kernel=2), dict(kernel=2, wf_refill=16), dict(kernel=2, wf_refill=32), dict(kernel=2, wf_refill=40),
                        dict(kernel=2, wf_block_lanes=768), dict(kernel=2, wf_block_lanes=768, wf_refill=16), dict(kernel=2, wf_block_lanes=512),
                        dict(kernel=2, wf_slots=1024), dict(kernel=2, wf_slots=1280)):
                try:
                    _, _, ms, n, _ = render(w, spp, 50, tun, reps=2)
                    print(f"{tun}: {ms:.1f} ms ({n/ms/1e3:.0f} Ms/s)", flush=True)
                except Exception as e:
                    print(f"{tun}: {e}", flush=True)
