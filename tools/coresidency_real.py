"""Does a small kernel on another stream become resident beside the library's trace kernel?  One single-band frame (~60 ms) on the
torch stream; 10 ms in, small torch kernels of several shapes on a second stream, timed from the host.
usage: coresidency_real.py [knob=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtmi_loader
pkg = rtmi_loader.load()
tun = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:]}
tun.setdefault("bands", 1)
tun.setdefault("tile_order", 1)
objs, mats = pkg.make_world_spheres(12345)
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=128, max_depth=50))
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
buf = torch.zeros((cam.img_height, cam.img_width, 3), dtype=torch.float32, device=dev)
small = torch.zeros(1 << 16, device=dev)
big = torch.zeros(1 << 26, device=dev)  # 256 MB: a memory-bound kernel of ~0.1 ms
with pkg.Scene(cam, objs, mats, tuning=tun) as sc:
    for what in ("warm", "small add (256 KB)", "large add (256 MB)", "small add again"):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sc.render_row_blocks_device(0, cam.img_height, 1, 1, 7, buf.data_ptr(), 0, s1.cuda_stream)
        time.sleep(0.010)
        t1 = time.perf_counter()
        with torch.cuda.stream(s2):
            (big if "large" in what else small).add_(1.0)
        s2.synchronize()
        t2 = time.perf_counter()
        s1.synchronize()
        t3 = time.perf_counter()
        print(f"{what:22s}: launched {1e3 * (t1 - t0):6.2f} ms into the frame, done {1e3 * (t2 - t1):7.3f} ms later; frame done at {1e3 * (t3 - t0):6.2f} ms "
              f"(trace span {sc.last_kernel_ms():.2f} ms, {sc.launch_info()['bands']} band)", flush=True)
