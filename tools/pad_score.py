"""Scores box-pad rules and trees on the CPU with the oracle's instrumented walk (no GPU): sphere and box tests per sample
on a strip of the S-GRID (config 4) and S-RTOW (config 3) frames, and whether the frame still equals the one rendered
with the round 1-3 class pad (oracle pad mode 1).  VERDICT r3 #1: "score candidates on the CPU first".

usage: python tools/pad_score.py [grid|rtow|both] [rows per strip] [spp]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader  # noqa: E402

pkg = rtmi_loader.load()
from oracle import binding as ob  # noqa: E402


def score(name, objs, mats, kw, rows, spp, leaf, y_list):
    kw = dict(kw, samples_per_pixel=spp)
    ocam = ob.camera_setup(ob.camera_params(**kw))
    bvh = pkg.bvh_build(objs, leaf)
    bvh = dict(bvh, nodes=bvh["nodes"].view(ob.BVH_NODE_DTYPE))
    W = ocam.img_width
    base = None
    for mode in (1, 2, 3, 0):  # class pad, refined, none (not exact), auto (the product's choice)
        ob.set_pad_mode(mode)
        t0 = time.time()
        c, frames = {}, []
        for y0 in y_list:  # strips spread over the frame
            strip, _, cs = ob.render_rect_counter(ocam, objs, mats, 404, 0, y0, W, y0 + rows, nthreads=8, counters=True, bvh=bvh)
            frames.append(strip)
            for k, v in cs.items():
                c[k] = c.get(k, 0) + v
        rgb = np.concatenate(frames)
        dt = time.time() - t0
        if base is None:
            base = rgb
        diff = int((np.nan_to_num(rgb).view(np.uint32) != np.nan_to_num(base).view(np.uint32)).any(axis=-1).sum())
        n = c["samples"]
        print(f"{name} leaf={leaf} pad_mode={mode}: {c['sphere_tests'] / n:8.2f} sphere + {c['node_tests'] / n:8.2f} box tests "
              f"/ sample, {c['segments'] / n:.3f} segments; pixels differing from mode 1: {diff} of {len(y_list) * rows * W}  ({dt:.1f} s)",
              flush=True)
    ob.set_pad_mode(0)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 18
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    if what in ("grid", "both"):
        objs, mats, kw = pkg.workloads.big_grid(316)
        score("S-GRID", objs, mats, kw, rows, spp, 4, [400, 600, 800, 1000])
    if what in ("rtow", "both"):
        objs, mats = pkg.make_world_spheres(12345)
        kw = dict(image_width=1920, max_depth=50)
        score("S-RTOW", objs, mats, kw, rows, spp, 2, [300, 500, 700, 900, 1060])


if __name__ == "__main__":
    main()
