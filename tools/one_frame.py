"""One BASELINE config, a few frames through the device-pointer entry with the given tuning (for rocprofv3 timelines).
usage: one_frame.py <2|3|4|5> <spp or 0> <frames> [knob=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtmi_loader
pkg = rtmi_loader.load()
config, spp, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
libs = [a.split("=", 1)[1] for a in sys.argv[4:] if a.startswith("lib=")]
if libs:  # another build of the library (tools/lib_ab.py --build names them librtmi_ab_<name>.so)
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), f"librtmi_ab_{libs[0]}.so")
tun = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:] if not a.startswith("lib=")}
cfg = {"2": ("rtow", 1200, 100), "3": ("rtow", 1920, 512), "4": ("grid", 1920, 256), "5": ("cornell", 800, 4096)}[config]
spp = spp or cfg[2]
if cfg[0] == "rtow":
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=cfg[1], samples_per_pixel=spp, max_depth=50)
elif cfg[0] == "grid":
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(image_width=cfg[1], samples_per_pixel=spp, max_depth=50)
else:
    objs, mats, kw = pkg.workloads.cornell_like()
    kw.update(image_width=cfg[1], samples_per_pixel=spp, max_depth=200)
cam = pkg.camera_setup(pkg.camera_params(**kw))
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
buf = torch.zeros((cam.img_height, cam.img_width, 3), dtype=torch.float32, device=dev)
with pkg.Scene(cam, objs, mats, tuning=tun or None) as sc:
    for f in range(frames):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sc.render_row_blocks_device(0, cam.img_height, 1, 1, 7, buf.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        print(f"frame {f}: {(time.perf_counter() - t0) * 1e3:.2f} ms, span {sc.last_kernel_ms():.2f} ms, {sc.launch_info()}", flush=True)
