"""Fixed cost of a launch: trace-kernel time of the S-RTOW 1080p frame (and of the eighth of it one of 8 ranks renders)
against samples per pixel, least-squares fit t = a + b * spp.  What does not scale with the work -- ramp, tail, staging --
is `a`: it is what strong scaling over N GPUs loses.  usage: launch_fixed_cost.py [knob=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtmi_loader
pkg = rtmi_loader.load()
tun = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
objs, mats = pkg.make_world_spheres(12345)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
for G in (1, 8):
    pts = []
    for spp in (32, 64, 128, 256, 512):
        cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=spp, max_depth=50))
        W, H = cam.img_width, cam.img_height
        plan = pkg.RowShardPlan(H, 8, G)
        y_first, n_blocks, rows = plan.shard(0)
        buf = torch.zeros((plan.max_rows, W, 3), dtype=torch.float32, device=dev)
        with pkg.Scene(cam, objs, mats, tuning=tun or None) as sc:
            ms = []
            for _ in range(3):
                sc.render_row_blocks_device(y_first, 8, G, n_blocks, 7, buf.data_ptr(), 0, stream)
                torch.cuda.synchronize(dev)
                ms.append(sc.last_kernel_ms())
        pts.append((spp, min(ms)))
    x = np.array([p[0] for p in pts], float)
    y = np.array([p[1] for p in pts], float)
    b, a = np.polyfit(x, y, 1)
    print(f"G={G} {tun or 'default'}: " + " ".join(f"{s}: {m:.2f}" for s, m in pts) + f"  ->  fixed {a:.2f} ms + {b:.4f} ms/spp "
          f"(fixed = {100 * a / y[-1]:.1f} % of the 512 spp launch)", flush=True)
