"""Diagnostic build (-DRTMI_PROF) of the v1 kernel with s_memtime stamps: where a wave spends its cycles."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
prof_lib = os.path.join(os.path.dirname(pkg.LIB_PATH), "librtmi_prof.so")
if "--build" in sys.argv:
    cmd = ["/opt/rocm/bin/hipcc"] + pkg.HIPCC_FLAGS + ["-DRTMI_PROF", "-I", "include", "-o", prof_lib] + pkg.CSRC
    subprocess.run(cmd, check=True)
    print("built", prof_lib); sys.exit(0)
pkg.LIB_PATH = prof_lib

w, spp = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "grid":  # BASELINE config 4: 100k spheres, scene in HBM
    from tests.scenes import big_grid
    objs, mats, kw = big_grid(316)
    kw.update(image_width=w, samples_per_pixel=spp)
    cam = pkg.camera_setup(pkg.camera_params(**kw))
else:
    cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
    objs, mats = pkg.make_world_spheres(12345)
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
    sc.render_rows(0, cam.img_height, 7)
    ms = sc.last_kernel_ms()
    out = (C.c_ulonglong * 128)()
    pkg.lib().rtmi_prof_read(sc._h, out)
names = ["fetch", "gen", "traverse", "shade", "-", "coop-unit", "begin-seg", "-", "pre-coop", "loop-glue", "#rounds", "#trav-iters"]
for label, base in (("trace", 0),):
    v = [int(x) for x in out[base + 8:base + 20]]
    idle_leaf, idle_node = v[4], v[7]  # lanes waiting at a leaf during node trips / at a node during leaf trips
    v[4] = v[7] = 0
    tot = sum(v[:10])
    if tot == 0:
        continue
    print(f"[{label}] kernel(s) {ms:.1f} ms; stamped cycles (sum over waves) {tot:.3e}")
    for n, x in zip(names, v):
        print(f"  {n:12s} {x:16d}  {100.0*x/tot if n[0] != '#' else 0:6.2f} %")
    L = [int(x) for x in out[base + 20:base + 28]]
    rounds, iters = v[10], v[11]
    leaf_it, node_it = L[0], iters - L[0]
    print(f"  node iters/round {node_it/rounds:.2f} mean lanes {L[2]/max(1,node_it):.1f};  leaf iters/round {leaf_it/rounds:.2f} mean lanes {L[1]/max(1,leaf_it):.1f}")
    print(f"  lanes parked at a leaf per node trip {idle_leaf/max(1,node_it):.1f}; lanes parked at a node per leaf trip {idle_node/max(1,leaf_it):.1f}")
    print(f"  per round: shade lanes {L[3]/rounds:.1f} (miss {L[4]/rounds:.1f}), gen lanes {L[5]/rounds:.1f}, begin lanes {L[6]/rounds:.1f}, done lanes {L[7]/rounds:.1f}")
    print(f"  trav iters per round: {v[11]/max(1,v[10]):.2f}; cycles per trav iter: {v[2]/max(1,v[11]):.1f}; cycles per round: {tot/max(1,v[10]):.1f}")
