"""Same-box A/B of whole libraries ("spread or regression?"): every library renders configs 2, 3 and 5 in its own
process, the libraries alternate within a round, and every line carries the normaliser the driver's lines lacked -- the linear-scan
kernel (the reference's algorithm, whose code did not change between rounds 4 and 5) timed in the same process.
usage: ab_rounds.py [--rounds N] lib1.so lib2.so ...   (file names under raytracing.cpp_amd/)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 3
if args and args[0] == "--rounds":
    rounds, args = int(args[1]), args[2:]
code = '''
import os, sys, time
sys.path.insert(0, %r)
import rtmi_loader
pkg = rtmi_loader.load()
pkg.LIB_PATH = sys.argv[1]
objs, mats = pkg.make_world_spheres(12345)
out = []
def frame(sc, cam, n=3):
    k, c = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        sc.render_rows(0, cam.img_height, 7, rgba=False)
        c.append((time.perf_counter() - t0) * 1e3)
        k.append(sc.last_kernel_ms())
    return min(k), min(c)
cam = pkg.camera_setup(pkg.camera_params(image_width=1200, samples_per_pixel=100, max_depth=50))
with pkg.Scene(cam, objs, mats) as sc:
    out += list(frame(sc, cam, 5))
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=512, max_depth=50))
with pkg.Scene(cam, objs, mats) as sc:
    out += list(frame(sc, cam))
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=128, max_depth=50))
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BRUTE) as sc:
    out += [frame(sc, cam, 2)[0]]
o5, m5, kw = pkg.workloads.cornell_like()
kw.update(samples_per_pixel=1024)
cam = pkg.camera_setup(pkg.camera_params(**kw))
with pkg.Scene(cam, o5, m5) as sc:
    out += list(frame(sc, cam, 2))
print(" ".join("%%.3f" %% v for v in out))
''' % root
print("# kernel ms (library events) / call ms (host clock, D2H of the float frame included); min of the runs of one process")
print("# round lib | config2 kernel call | config3 kernel call | scan kernel (1080p x 128 spp) | walk/scan (x4 spp) | config 5 at 1024 spp kernel call")
for rnd in range(rounds):
    for lib in args:
        r = subprocess.run([sys.executable, "-c", code, os.path.join(root, "raytracing.cpp_amd", lib)], capture_output=True, text=True)
        v = r.stdout.strip().split()
        if len(v) == 7:
            f = [float(x) for x in v]
            print(f"{rnd} {lib:28s} | {f[0]:7.3f} {f[1]:7.3f} | {f[2]:8.3f} {f[3]:8.3f} | {f[4]:8.3f} | {f[2] / (4.0 * f[4]):.4f} | {f[5]:8.2f} {f[6]:8.2f}", flush=True)
        else:
            print(rnd, lib, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)
