"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference cannot be built in this image (missing glm / tl::optional / strong_type / reflect-cpp) and ships no
test vectors, so these files pin the ORACLE (regression), not the reference.  The only vectors that come from the
reference itself are the four pixel values in reference_pixels.json, recorded by the survey session from a run of the
reference's own translation units and reproduced bit-for-bit by the oracle's mt19937 path.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    objs, mats = ob.make_world_spheres(12345)
    np.savez_compressed(os.path.join(OUT, "rtow_scene_seed12345.npz"), objects=objs, materials=mats)

    cams = {}
    for w in (400, 1200, 1920):
        cam = ob.camera_setup(ob.camera_params(image_width=w))
        cams[str(w)] = {n: (list(getattr(cam, n)) if hasattr(getattr(cam, n), "__len__") else getattr(cam, n))
                        for n, _ in ob.Camera._fields_}
    json.dump(cams, open(os.path.join(OUT, "rtow_cameras.json"), "w"), indent=1)

    # counter-RNG frame, small enough for the oracle to finish in seconds, all three materials in view
    cp = dict(image_width=128, samples_per_pixel=16, max_depth=50)
    cam = ob.camera_setup(ob.camera_params(**cp))
    rgb, rgba, ctr = ob.render_rect_counter(cam, objs, mats, 2025, 0, 0, cam.img_width, cam.img_height, nthreads=8,
                                            counters=True)
    np.savez_compressed(os.path.join(OUT, "rtow_counter_128x72x16.npz"), rgb=rgb, rgba=rgba, seed=2025,
                        camera=json.dumps(cp), counters=json.dumps(ctr))

    # deep-bounce enclosed scene (config 5 shape, shrunk): box of six huge spheres with a sky opening
    from tests.scenes import cornell_like
    cobjs, cmats, ccp = cornell_like()
    ccp.update(image_width=48, samples_per_pixel=32, max_depth=200)
    ccam = ob.camera_setup(ob.camera_params(**ccp))
    rgb, rgba = ob.render_rect_counter(ccam, cobjs, cmats, 99, 0, 0, ccam.img_width, ccam.img_height, nthreads=8)
    np.savez_compressed(os.path.join(OUT, "cornell_counter_48x48x32.npz"), rgb=rgb, rgba=rgba, seed=99,
                        camera=json.dumps(ccp), objects=cobjs, materials=cmats)

    # mt19937 (reference RNG) pixels, sequential with one generator
    cam = ob.camera_setup(ob.camera_params())
    xy = [(x, y) for y in (10, 300, 420, 600) for x in (5, 333, 600, 1100)]
    rgb, rgba = ob.render_pixels_mt(cam, objs, mats, 777, xy)
    json.dump({"mt_seed": 777, "xy": xy, "rgba": [int(v) for v in rgba], "rgb": rgb.tolist()},
              open(os.path.join(OUT, "rtow_mt_pixels.json"), "w"))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
