"""Where the end of a launch goes (diagnostic build -DRTMI_TAILPROBE: every wave stamps its start, its first work-pool refill past
the end of the work, and its exit): for a shard of the S-RTOW frame, the kernel's ramp (first wave start -> last wave start), the
time from the first wave that finds the counter dry to the last exit, and how many lane-milliseconds the waves spend between
running dry and exiting.  usage: tail_profile.py [--build] <G> <spp> [knob=value ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
lib = os.path.join(os.path.dirname(pkg.LIB_PATH), "librtmi_tailprobe.so")
if "--build" in sys.argv:
    subprocess.run(["/opt/rocm/bin/hipcc"] + pkg.HIPCC_FLAGS + ["-DRTMI_TAILPROBE", "-I", "include", "-o", lib] + pkg.CSRC + ["-ldl"], check=True)
    print("built", lib)
    sys.exit(0)
import torch
pkg.LIB_PATH = lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
G, spp = int(args[0]), int(args[1])
tun = {a.split("=")[0]: int(a.split("=")[1]) for a in args[2:]}
objs, mats = pkg.make_world_spheres(12345)
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=spp, max_depth=50))
W, H = cam.img_width, cam.img_height
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
plan = pkg.RowShardPlan(H, 8, G)
y_first, n_blocks, rows = plan.shard(0)
buf = torch.zeros((plan.max_rows, W, 3), dtype=torch.float32, device=dev)
L = pkg.lib()
L.rtmi_prof_tail_read.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
with pkg.Scene(cam, objs, mats, tuning=tun or None) as sc:
    for _ in range(3):
        sc.render_row_blocks_device(y_first, 8 if G > 1 else H, G, n_blocks if G > 1 else 1, 7, buf.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
    li = sc.launch_info()
    n = C.c_uint32(0)
    out = np.zeros(li["grid_blocks"] * (li["block_lanes"] // 64) * 3, np.uint64)
    assert L.rtmi_prof_tail_read(sc._h, out.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
    t = out.reshape(-1, 3).astype(np.float64) / 100e3  # ms
    t0 = t[:, 0].min()
    start, dry, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
    total = end.max()
    print(f"G={G} {spp} spp {tun or 'default'}: kernel {sc.last_kernel_ms():.3f} ms (events), {total:.3f} ms (first wave start -> last wave exit), bands {li['bands']} order {li['tile_order']}")
    print(f"  ramp: last wave starts at {start.max():.3f} ms (median {np.median(start):.3f})")
    print(f"  first wave sees the counter dry at {dry.min():.3f} ms, the last one at {dry.max():.3f} ms; exits: first {end.min():.3f}, median {np.median(end):.3f}, 90 % {np.quantile(end, 0.9):.3f}, 99 % {np.quantile(end, 0.99):.3f}, last {end.max():.3f} ms")
    idle = (total - end).sum() / len(end)
    print(f"  mean time a wave slot is empty before the kernel ends: {idle:.3f} ms ({100 * idle / total:.1f} % of the kernel); mean dry -> exit of a wave: {(end - dry).mean():.3f} ms")
    hist, edges = np.histogram(total - end, bins=[0, 0.05, 0.1, 0.2, 0.4, 0.8, 1.6, 3.2, 6.4, 1e9])
    print("  waves by how long before the end they exited (ms): " + ", ".join(f"<{edges[i + 1]:g}: {hist[i]}" for i in range(len(hist))))
