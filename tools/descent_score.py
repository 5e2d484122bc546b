"""VERDICT r4 #4b, scored on the CPU before it costs GPU minutes: every secondary ray starts INSIDE the boxes of its own sphere's
ancestors, so the walk descends to that leaf whatever the tree.  With the oracle's instrumented walk of the scene's own tree: segments,
node trips and leaf trips by where a segment starts (camera / the peeled ground sphere / a sphere inside the tree), and for the last class
the levels of that forced descent and how many of their sibling boxes the ray hits (a per-leaf sibling list tested at segment set-up
would replace `levels` dependent two-box trips by `levels` single box tests with all reads in flight).
usage: descent_score.py [rtow|grid] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
from oracle import binding as ob
scene = sys.argv[1] if len(sys.argv) > 1 else "rtow"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if scene == "rtow":
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=1920, samples_per_pixel=spp, max_depth=50)
    leaf = 2
else:
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(image_width=1920, samples_per_pixel=spp, max_depth=50)
    leaf = 4
cam = ob.camera_setup(ob.camera_params(**kw))
bvh = pkg.bvh_build(objs, leaf)
bvh = dict(bvh, nodes=bvh["nodes"].view(ob.BVH_NODE_DTYPE))
tot = None
for y in range(4, cam.img_height, 24):  # strips over the whole frame
    _, _, c = ob.render_rect_counter(cam, objs, mats, 2025, 0, y, cam.img_width, y + 1, nthreads=8, counters=True, bvh=bvh)
    if tot is None:
        tot = {k: (list(v) if isinstance(v, list) else v) for k, v in c.items()}
    else:
        for k, v in c.items():
            if isinstance(v, list):
                tot[k] = [a + b for a, b in zip(tot[k], v)]
            else:
                tot[k] += v
n = tot["samples"]
print(f"# {scene}: {len(objs)} spheres, {len(bvh['nodes'])} nodes, depth {bvh['depth']}, leaves of {leaf}; {n} samples on strips of the 1080p frame")
print(f"segments per sample {tot['segments'] / n:.3f}, box tests per sample {tot['node_tests'] / n:.2f}, sphere tests per sample {tot['sphere_tests'] / n:.2f}")
names = ("camera", "peeled sphere (the ground)", "sphere inside the tree")
for k in range(3):
    s = max(1, tot["seg_class"][k])
    print(f"  segments from {names[k]:28s}: {100 * tot['seg_class'][k] / tot['segments']:5.1f} % of the segments, {tot['trips_class'][k] / s:6.2f} node trips and "
          f"{tot['leaf_trips_class'][k] / s:5.2f} leaf trips each ({100 * tot['trips_class'][k] / max(1, sum(tot['trips_class'])):5.1f} % of all node trips)")
s2 = max(1, tot["seg_class"][2])
lv, sh = tot["descent_levels"], tot["descent_sibling_hits"]
print(f"  of the {tot['trips_class'][2] / s2:.2f} node trips of a segment that starts on a tree sphere, {lv / s2:.2f} are the descent to its own leaf; "
      f"the sibling box is hit on {100 * sh / max(1, lv):.1f} % of those levels ({sh / s2:.2f} subtrees pushed per segment)")
