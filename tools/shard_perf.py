"""One-GPU rehearsal of the row-block sharding: render the shard that rank r of a G-GPU job would own (8-row blocks, dealt out
block b -> rank b mod G or by the scene's cost map; same C-ABI call as bench.py) and time it.  Strong scaling is bounded by G * t_shard(G) / t_shard(1):
what the slowest rank spends, before the 3 MB gather.  usage: shard_perf.py [width] [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtmi_loader

pkg = rtmi_loader.load()
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
cam = pkg.camera_setup(pkg.camera_params(image_width=w, samples_per_pixel=spp, max_depth=50))
objs, mats = pkg.make_world_spheres(12345)
W, H = cam.img_width, cam.img_height
stream = torch.cuda.current_stream(dev).cuda_stream
with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BVH) as sc:
    base = None
    gs = [int(g) for g in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    BR = 8
    bc = pkg.block_costs(sc.tile_costs(), H, BR) if sc.tile_costs() is not None else None
    for G in gs:
        for name, plan in (("block b -> rank b mod N", pkg.CostShardPlan(H, BR, G)), ("dealt out by cost (rtmi_shard_plan)", pkg.CostShardPlan(H, BR, G, bc))):
            if G == 1 and name.startswith("dealt"):
                continue
            per_rank = []
            for r in range(G):
                rgb = torch.zeros((plan.max_rows, W, 3), dtype=torch.float32, device=dev)
                ts = []
                for it in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    if G == 1:
                        sc.render_row_blocks_device(0, BR, 1, len(plan.blocks(0)), 7, rgb.data_ptr(), 0, stream)
                    else:
                        sc.render_block_list_device(BR, plan.blocks(r), 7, rgb.data_ptr(), 0, stream)
                    torch.cuda.synchronize(dev)
                    ts.append((time.perf_counter() - t0) * 1e3)
                per_rank.append(min(ts[1:]))
            worst = max(per_rank)
            if base is None:
                base = worst
            print(f"G={G} {name:38s}: slowest rank {worst:8.2f} ms, fastest {min(per_rank):8.2f} ms, speed-up bound {base / worst:5.2f}x "
                  f"({100.0 * base / worst / G:5.1f} % of ideal)", flush=True)
