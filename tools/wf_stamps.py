"""Diagnostic build (-DRTMI_PROF) of the queue-scheduled kernel: where a wave's cycles go, how full its batches are."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
prof_lib = os.path.join(os.path.dirname(pkg.LIB_PATH), "librtmi_prof.so")
if "--build" in sys.argv:
    cmd = ["/opt/rocm/bin/hipcc"] + pkg.HIPCC_FLAGS + ["-DRTMI_PROF", "-I", "include", "-o", prof_lib] + pkg.CSRC + ["-ldl"]
    subprocess.run(cmd, check=True)
    print("built", prof_lib); sys.exit(0)
pkg.LIB_PATH = prof_lib
w, spp = int(sys.argv[1]), int(sys.argv[2])
tun = dict(kernel=2)
for a in sys.argv[3:]:
    k, v = a.split("=")
    tun[k] = int(v)
cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
objs, mats = pkg.make_world_spheres(12345)
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH, tuning=tun) as sc:
    sc.render_rows(0, cam.img_height, 7)
    ms = sc.last_kernel_ms()
    out = (C.c_ulonglong * 64)()
    pkg.lib().rtmi_prof_read(sc._h, out)
v = [int(x) for x in out[8:40]]
tot = sum(v[:8]) + sum(v[24:28])
print(f"{tun}: kernel {ms:.1f} ms; stamped cycles {tot:.3e}")
names = ["job_end tail", "job_hit: begin_ray+push", "walk", "handover+refill", "sched/idle + pops", "job_end: colour+store", "job_end: work fetch", "job_end: gen"]
for name, x in list(zip(names, v[:8])) + list(zip(["job_end: begin_ray+push", "job_hit: loads", "job_hit: coop draws", "job_hit: shade"], v[24:28])):
    print(f"  {name:26s} {x:16d} {100.0 * x / tot:6.2f} %")
je, jel, jh, jhl, wi, wl, rf, rfl = v[8:16]
print(f"  job_end  x{je}: {jel / max(1, je):.1f} lanes, {v[0] / max(1, je):.0f} cycles each")
print(f"  job_hit  x{jh}: {jhl / max(1, jh):.1f} lanes, {v[1] / max(1, jh):.0f} cycles each")
print(f"  walk iterations {wi}: {wl / max(1, wi):.1f} active lanes, {v[2] / max(1, wi):.0f} cycles each")
print(f"  refills {rf}: {rfl / max(1, rf):.1f} rays each, {v[3] / max(1, rf):.0f} cycles each")
