"""VERDICT r5 #1, scored on the CPU before any GPU minute: the trace kernel's 64-lane ROUNDS replayed on the oracle's arithmetic
(oracle/wave_replay.c) -- trips per round, not per lane -- under walk-start variants:
  way    segments that start on a tree sphere begin in its own leaf, the siblings hanging off the path above pre-loaded on the
         stack as way records (two levels = two boxes per record, the node format, so the unchanged node step tests them)
  grid   segments that start on the peeled ground sphere begin at the deepest node around their origin (3-d grid), way likewise
  entry  camera rays begin at their 8x8 tile's entry: the lowest common ancestor of every sphere the tile's beam can meet
usage: wave_replay.py [rtow|grid] [spp] [tile_stride]"""
import ctypes as C, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
from oracle import binding as ob
subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle")], check=True)
L = C.CDLL(os.path.join(root, "oracle", "_build", "libwave_replay.so"))
vp = C.c_void_p
L.orc_wave_replay.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_float, C.c_float,
                              C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
NAMES = ("rounds segments samples node_trips node_lanes leaf_trips leaf_lanes begin_lanes gen_rounds gen_lanes shade_hit shade_sky "
         "preload_rounds preload_levels_max preload_lanes way_records seg0 seg1 seg2 mismatch_way mismatch_cam trips0 trips1 trips2 "
         "tiles tiles_no_walk tile_cands entry_depth fetch_rounds carried_lanes").split()
scene = sys.argv[1] if len(sys.argv) > 1 else "rtow"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 23
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
nodes = np.ascontiguousarray(bvh["nodes"].view(ob.BVH_NODE_DTYPE))
slots = np.ascontiguousarray(bvh["slots"], dtype=np.uint32)
pc = np.ascontiguousarray(bvh["pad_classes"], dtype=np.float32).reshape(-1, 8)
objs = np.ascontiguousarray(objs); mats = np.ascontiguousarray(mats)
p = lambda a: a.ctypes.data_as(vp)


def run(variant, wait=52, chunk=32, waves=8):
    out = np.zeros(32, np.uint64)
    rc = L.orc_wave_replay(C.byref(cam), p(objs), len(objs), p(mats), len(mats), p(nodes), len(nodes), p(slots), len(slots), p(pc), len(pc),
                           bvh["pad_eps"], bvh["pad_floor"], 2025, stride, chunk, waves, wait, variant, p(out))
    assert rc == 0
    return dict(zip(NAMES, (int(v) for v in out)))


# marginal cost of the pieces of a round in VALU-equivalents (profiles/r03 census static counts; the node trip's 17 scalar instructions
# weighed 1.5x as DESIGN 5.1 measured them: ten s_add per trip +4.3 %, ten v_mov +2.8 %)
C_NODE, C_LEAF, C_FIXED, C_PRELOAD_LEVEL, C_PRELOAD_FIXED = 43 + 1.5 * 17 + 5, 130.0, 850.0, 8.0, 25.0
print(f"# {scene}: {len(objs)} spheres, {len(nodes)} nodes; 1080p, {spp} spp in one chunk, every {stride}th tile costliest first, 8 waves of 64 lanes, wait_thresh 52")
print("# variant                          rounds  seg/round  node trips/round (lanes)  leaf trips/round (lanes)  lane-trips/segment cam / ground / tree   way records  cost/segment  vs base")
base = None
for name, v in (("base (walks start at the root)", 16), ("way (tree-origin)", 17), ("grid+way (ground-origin)", 18), ("entry (camera)", 20),
                ("way + entry", 21), ("way + grid", 19), ("way + grid + entry", 23)):
    r = run(v)
    assert r["mismatch_way"] == 0 and r["mismatch_cam"] == 0, r
    R = r["rounds"]
    cost = (R * C_FIXED + r["node_trips"] * C_NODE + r["leaf_trips"] * C_LEAF + r["preload_rounds"] * C_PRELOAD_FIXED +
            r["preload_levels_max"] * C_PRELOAD_LEVEL) / r["segments"]
    base = base or cost
    lt = [r[f"trips{k}"] / max(1, r[f"seg{k}"]) for k in range(3)]
    print(f"{name:32s} {R:7d} {r['segments'] / R:9.2f} {r['node_trips'] / R:11.2f} ({r['node_lanes'] / max(1, r['node_trips']):4.1f}) "
          f"{r['leaf_trips'] / R:16.2f} ({r['leaf_lanes'] / max(1, r['leaf_trips']):4.1f}) {lt[0]:16.2f} / {lt[1]:5.2f} / {lt[2]:5.2f} "
          f"{r['way_records']:12d} {cost:12.2f} {100 * (cost / base - 1):+7.1f} %", flush=True)
    if v == 20:
        print(f"#   camera entries: {r['tiles']} tiles, {r['tiles_no_walk']} whose beam meets no tree sphere, {r['tile_cands'] / r['tiles']:.2f} candidate spheres "
              f"and entry depth {r['entry_depth'] / max(1, r['tiles'] - r['tiles_no_walk']):.2f} on average")
print("# wait_thresh sweep of the last variant:")
for w in (40, 46, 52, 58):
    r = run(23, wait=w)
    R = r["rounds"]
    cost = (R * C_FIXED + r["node_trips"] * C_NODE + r["leaf_trips"] * C_LEAF + r["preload_rounds"] * C_PRELOAD_FIXED + r["preload_levels_max"] * C_PRELOAD_LEVEL) / r["segments"]
    print(f"  wait_thresh {w}: rounds {R}, node trips/round {r['node_trips'] / R:.2f}, cost/segment {cost:.2f} ({100 * (cost / base - 1):+.1f} %)")
