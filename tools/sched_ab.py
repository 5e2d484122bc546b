"""A/B of the round-5 scheduling knobs (rtmi_tuning::tile_order / bands / chunk_samples ...) on a BASELINE config or on the shard
rank 0 of a G-GPU job renders: wall time of the device-pointer call (trace + resolve, frame left in HBM) and the span of its trace
kernels, min / median over interleaved rounds; frames compared bit for bit with the first variant's.
usage: sched_ab.py <2|3|4|5> [G] [spp] [rounds] -- knob=value[,knob=value] ...   (the library's defaults are always the first variant)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtmi_loader
pkg = rtmi_loader.load()

args = sys.argv[1:]
head = args[:args.index("--")] if "--" in args else args
tail = args[args.index("--") + 1:] if "--" in args else []
config = head[0] if head else "3"
G = int(head[1]) if len(head) > 1 else 1
cfg = {"2": ("rtow", 1200, 100), "3": ("rtow", 1920, 512), "4": ("grid", 1920, 256), "5": ("cornell", 800, 4096)}[config]
spp = int(head[2]) if len(head) > 2 and int(head[2]) > 0 else cfg[2]
rounds = int(head[3]) if len(head) > 3 else 3
variants = [dict()] + [{kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")} for a in tail]
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
W, H = cam.img_width, cam.img_height
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
plan = pkg.RowShardPlan(H, 8, G)
y_first, n_blocks, rows = plan.shard(0)
scenes = [pkg.Scene(cam, objs, mats, tuning=v or None) for v in variants]
bufs = torch.zeros((plan.max_rows, W, 3), dtype=torch.float32, device=dev)
ref = None
wall = [[] for _ in variants]
span = [[] for _ in variants]
first = []
for rnd in range(rounds + 1):  # round 0: first frames (allocations, the cost probe)
    for i, sc in enumerate(scenes):
        bufs.zero_()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sc.render_row_blocks_device(y_first, 8 if G > 1 else H, G, n_blocks if G > 1 else 1, 7, bufs.data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3
        if rnd == 0:
            first.append(ms)
            got = bufs[:rows].cpu().numpy()
            if ref is None:
                ref = got
            else:
                d = int((np.nan_to_num(got).view(np.uint32) != np.nan_to_num(ref).view(np.uint32)).any(axis=-1).sum())
                if d:
                    print(f"!! {variants[i]}: {d} pixels differ from the default", flush=True)
        else:
            wall[i].append(ms)
            span[i].append(sc.last_kernel_ms())
n = rows * W * spp
print(f"# config {config} ({cfg[0]} {W}x{H} x {spp} spp), shard of rank 0 of {G}: {rows} rows; wall = call + sync, span = trace kernels", flush=True)
for i, v in enumerate(variants):
    li = scenes[i].launch_info()
    w, k = sorted(wall[i]), sorted(span[i])
    print(f"{str(v or 'default'):58s} bands {li['bands']:2d} order {li['tile_order']} probe {li['probe_us']:5d} us  first {first[i]:8.1f}  "
          f"wall min {w[0]:8.3f} med {w[len(w) // 2]:8.3f}  span min {k[0]:8.3f}  {n / w[0] / 1e3:8.1f} Msamples/s", flush=True)
